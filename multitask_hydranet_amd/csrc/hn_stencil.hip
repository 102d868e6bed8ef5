// Im2col-free NHWC stencil kernels (VALU; channel counts per group are too thin for MFMA):
//   stem 3x3/s2 dense conv from the NCHW fp32 frame, grouped 3x3 (group width 8) fwd/dgrad/wgrad, depthwise 3x3
//   fwd/dgrad/wgrad, the two 3x3/s2 max-pool flavours fwd/bwd, nearest x2 up-sampling, the BiFPN weighted-fusion
//   node fwd/bwd, and the fold that turns the segmentation decoder's padded-domain dgrad back into input gradients.
// Reference ops covered: Stem (net/anynet.py:8-20), XBlock conv_block_2 (net/anynet.py:34-38), SeparableConvBlock's
// depthwise conv (net/common.py:91-92,104), MaxPool2dStaticSamePadding (net/common.py:117-152; ZERO pad participates),
// nn.MaxPool2d(3,2,1) (head_lane/lanedetect.py:40), F.interpolate nearest (net/bifpn.py:43-46),
// BiFPN._forward_fast_attention fusion nodes (net/bifpn.py:177-231), ReflectionPad2d backward
// (head_seg/segmentation.py:40).
#include "hn_common.h"

// ---------------------------------------------------------------------------------------------------------
// stem: x NCHW fp32 [N,3,H,W] -> z NHWC bf16 [N,H/2,W/2,32]; weights fp32 [32][3][3][3]
// ---------------------------------------------------------------------------------------------------------
// One thread = TWO neighbouring output pixels (2p, 2p + 1) of a row: every weight read from LDS (a broadcast ds_read, 216 of them per
// thread against 864 FMAs) feeds two FMAs, and the two 3 x 3 stride-2 windows share one of their six input columns.  With one pixel per
// thread the kernel was bound by those LDS reads (186 us for 352 MB; 2 M pixels x 216 wave-wide reads).
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* x, const float* w, bf16* z, bf16* patches, int N, int H, int W) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    __shared__ float sw[27][32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) {
        const int co = i & 31, t = i >> 5;        // t = ci*9 + ky*3 + kx
        sw[t][co] = w[co * 27 + t];
    }
    __syncthreads();
    const int Ho = H >> 1, Wo = W >> 1, Wp = (Wo + 1) >> 1;
    const long total = (long)N * Ho * Wp;
    const long idx = (long)bidx * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Wp) * 2;
    const long t1 = idx / Wp;
    const int oy = (int)(t1 % Ho);
    const long n = t1 / Ho;
    const bool two = ox + 1 < Wo;
    float acc[2][32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { acc[0][c] = 0.f; acc[1][c] = 0.f; }
    // im2col rows (27 taps + 5 zeros) kept in bf16 for the MFMA wgrad: staged in LDS (2-byte writes are cheap there) and written out as
    // 16-byte stores instead of 27 two-byte global stores per pixel
    // (rows of 36 bf16 = 18 dwords per thread, pixel-major planes: the 2-byte LDS writes of a wave fall on 16 distinct banks -- a
    // 40-element row put them on 4, i.e. 16-way conflicts on each of the 54 writes: that, not HBM, was most of the kernel's time)
    __shared__ __attribute__((aligned(16))) bf16 sp[2][256][36];
    bf16* prow = patches ? sp[0][threadIdx.x] : nullptr;
    constexpr int PH = 256 * 36;                   // elements between the two pixels' rows of a thread
    auto load_row = [&](int t, float* v) {         // input columns 2 ox - 1 ... 2 ox + 3 of row (ci, ky) = (t / 3, t % 3)
        const int ci = t / 3, ky = t - ci * 3;     // (unconditional loads from clamped coordinates + selects measured slower: 179 vs 156 us)
        const int iy = 2 * oy + ky - 1;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int ix = 2 * ox + k - 1;
            v[k] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((n * 3 + ci) * H + iy) * (long)W + ix] : 0.f;
        }
    };
    float vn[5];
    load_row(0, vn);
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {                  // t = ci*3 + ky ; kept rolled so the 864 weights are not hoisted
        float v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = vn[k];
        if (t + 1 < 9) load_row(t + 1, vn);        // the next row's loads fly under this row's 192 FMAs (the rolled loop exposed ~2 us per row)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if (prow) { prow[t * 3 + kx] = f2bf(v[kx]); prow[PH + t * 3 + kx] = f2bf(v[kx + 2]); }
            const float* wr = sw[t * 3 + kx];
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float wc = wr[c];
                acc[0][c] = fmaf(v[kx], wc, acc[0][c]);
                acc[1][c] = fmaf(v[kx + 2], wc, acc[1][c]);
            }
        }
    }
    const long pix = (n * Ho + oy) * (long)Wo + ox;
    // Whole blocks of an even-width grid own 512 consecutive pixels (pix = 2 idx): rows leave through LDS so that every store instruction
    // of a wave writes 1 KB of consecutive addresses.  (Thread-owned rows put the 64 lanes of a 16-byte store on 64 different 128-byte
    // lines: 16 such instructions per thread = 1 024 partial-line writes per wave where 128 full ones do.)
    if (!(Wo & 1) && ((long)bidx + 1) * 256 <= total) {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const long wbase = (((long)bidx * 256 + wv * 64) * 2) * 32;      // the wave's first element of z / patches
        if (prow) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 27; t < 32; ++t) prow[h * PH + t] = f2bf(0.f);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int P = j * 64 + ln, t = P >> 3, h = (P >> 2) & 1, q = P & 3;
                const bf16* src = sp[h][wv * 64 + t] + q * 8;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(src), hi = *reinterpret_cast<const bf16x4*>(src + 4);
                bf16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                st8(patches + wbase + (long)P * 8, v8);
            }
            __syncthreads();
        }
        // z rows: 128 bytes per thread, the eight 16-byte pieces XOR-swizzled by (thread >> 1) & 7 (writes and reads conflict free)
        char* sz = reinterpret_cast<char*>(&sp[0][0][0]);
        const int sw8 = (threadIdx.x >> 1) & 7;
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
            bf16x8 v8;
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = f2bf(acc[q8 >> 2][(q8 & 3) * 8 + k]);
            *reinterpret_cast<bf16x8*>(sz + threadIdx.x * 128 + ((q8 ^ sw8) << 4)) = v8;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int P = j * 64 + ln, T = wv * 64 + (P >> 3), q8 = P & 7;
            const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(sz + T * 128 + ((q8 ^ ((T >> 1) & 7)) << 4));
            st8(z + wbase + (long)P * 8, v8);
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && !two) break;
        if (prow) {
#pragma unroll
            for (int t = 27; t < 32; ++t) prow[h * PH + t] = f2bf(0.f);
            bf16* pg = patches + (pix + h) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                          // own rows: no barrier needed; 8-byte aligned LDS rows
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(prow + h * PH + q * 8);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(prow + h * PH + q * 8 + 4);
                bf16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                st8(pg + q * 8, v8);
            }
        }
        bf16* o = z + (pix + h) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 v8;
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = f2bf(acc[h][q * 8 + k]);
            st8(o + q * 8, v8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// grouped 3x3 conv, group width 8, pad 1.  Packed weights wk[tap][i][G][8 o] (bf16): 16 B per (tap, i, group).
//   out[n, oy, ox, g*8 + o] = sum_{tap, i} in[n, oy*s + ky - 1, ox*s + kx - 1, g*8 + i] * wk[tap][i][g][o]
// One thread = one group x a strip of 4 output pixels along x.
// ---------------------------------------------------------------------------------------------------------
// LDSW: the packed weights (G x 1152 B) are copied to LDS first and read from there: every thread reads 72 16-byte weight pieces against
// 36 pixel pieces, and through the vector-memory path (64 B/clk per CU) those re-reads, not HBM, set the time; LDS serves them at 256 B/clk
// (stage 0, 16 x 256 x 512 x 24: 71 -> 40 us cold).
__device__ __forceinline__ const bf16* stage_gconv_weights(const bf16* wk, int G, bool ldsw) {
    extern __shared__ __attribute__((aligned(16))) char gconv_smem[];
    if (!ldsw) return wk;
    const int pieces = 72 * G;                                 // 16-byte pieces
    for (int i = threadIdx.x; i < pieces; i += 256) *reinterpret_cast<bf16x8*>(gconv_smem + i * 16) = ld8(wk + (long)i * 8);
    __syncthreads();
    return reinterpret_cast<const bf16*>(gconv_smem);
}
template <int S, bool LDSW = false>
__global__ __launch_bounds__(256) void gconv_fwd_kernel(const bf16* in, int ldi, const bf16* wk_g, bf16* out, int ldo, int N, int Hi,
                                                        int Wi, int Ho, int Wo, int G) {
    const bf16* wk = stage_gconv_weights(wk_g, G, LDSW);
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int strips = (Wo + 3) >> 2;
    const long total = (long)N * Ho * strips * G;
    const long idx = (long)bidx * 256 + threadIdx.x;
    if (idx >= total) return;
    // (32-bit decomposition: the host checks total < 2^32)
    const unsigned ui = (unsigned)idx;
    const int g = (int)(ui % (unsigned)G);
    unsigned t = ui / (unsigned)G;
    const int sx = (int)(t % (unsigned)strips);
    t /= (unsigned)strips;
    const int oy = (int)(t % (unsigned)Ho);
    const long n = t / (unsigned)Ho;
    const int ox0 = sx * 4;
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[p][o] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {          // rolled: 4 pixel pieces + 8 weight pieces live per tap (L1-resident re-reads)
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy * S + ky - 1;
        if (iy < 0 || iy >= Hi) continue;
        float xf[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ix = (ox0 + p) * S + kx - 1;
            const bf16x8 v = (ix >= 0 && ix < Wi) ? ld8(in + ((n * Hi + iy) * (long)Wi + ix) * ldi + g * 8) : zero8();
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[p][i] = bf2f(v[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 wv = ld8(wk + (((long)tap * 8 + i) * G + g) * 8);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const float wf = bf2f(wv[o]);
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p][o] = fmaf(xf[p][i], wf, acc[p][o]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (ox0 + p >= Wo) break;
        bf16x8 v;
#pragma unroll
        for (int o = 0; o < 8; ++o) v[o] = f2bf(acc[p][o]);
        st8(out + ((n * Ho + oy) * (long)Wo + ox0 + p) * ldo + g * 8, v);
    }
}


// Packed-bf16 dot products for the stride-2 grouped convs (v_dot2c_f32_bf16: d += a.lo * b.lo + a.hi * b.hi, fp32 accumulate).  With the
// contraction index contiguous in BOTH operands (8 input channels of a pixel piece / of a weight piece) a 3x3 tap of one group is 8 outputs
// x 4 instructions instead of 64 FMAs + 72 bf16 -> f32 conversions: the kernels above were VALU-bound, not bandwidth-bound.
__device__ __forceinline__ float dot2c_bf16(unsigned a, unsigned b, float c) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dot8_bf16(const u32x4 a, const u32x4 b, float c) {
    c = dot2c_bf16(a[0], b[0], c);
    c = dot2c_bf16(a[1], b[1], c);
    c = dot2c_bf16(a[2], b[2], c);
    return dot2c_bf16(a[3], b[3], c);
}
// forward, stride 2: wo = weights packed [tap][o][G][i] (i contiguous: the `wd` array of hn_gconv_pack(flip = 0)).  Same thread mapping as
// gconv_fwd_kernel (one group x a strip of 4 output pixels).
template <bool LDSW>
__global__ __launch_bounds__(256) void gconv_fwd_s2_kernel(const bf16* in, int ldi, const bf16* wo_g, bf16* out, int ldo, int N, int Hi,
                                                           int Wi, int Ho, int Wo, int G) {
    const bf16* wo = stage_gconv_weights(wo_g, G, LDSW);
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);
    const int strips = (Wo + 3) >> 2;
    const long total = (long)N * Ho * strips * G;
    const long idx = (long)bidx * 256 + threadIdx.x;
    if (idx >= total) return;
    const unsigned ui = (unsigned)idx;
    const int g = (int)(ui % (unsigned)G);
    unsigned t = ui / (unsigned)G;
    const int sx = (int)(t % (unsigned)strips);
    t /= (unsigned)strips;
    const int oy = (int)(t % (unsigned)Ho);
    const long n = t / (unsigned)Ho;
    const int ox0 = sx * 4;
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[p][o] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy * 2 + ky - 1;
        if (iy < 0 || iy >= Hi) continue;
        u32x4 xv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ix = (ox0 + p) * 2 + kx - 1;
            xv[p] = (ix >= 0 && ix < Wi) ? *reinterpret_cast<const u32x4*>(in + ((n * Hi + iy) * (long)Wi + ix) * ldi + g * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
        u32x4 wv[8];                                          // all eight weight pieces requested before the first dot (one round trip)
#pragma unroll
        for (int o = 0; o < 8; ++o) wv[o] = *reinterpret_cast<const u32x4*>(wo + (((long)tap * 8 + o) * G + g) * 8);
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p][o] = dot8_bf16(xv[p], wv[o], acc[p][o]);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (ox0 + p >= Wo) break;
        bf16x8 v;
#pragma unroll
        for (int o = 0; o < 8; ++o) v[o] = f2bf(acc[p][o]);
        st8(out + ((n * Ho + oy) * (long)Wo + ox0 + p) * ldo + g * 8, v);
    }
}

// stride-2 dgrad: dx[n, iy, ix, g*8+i] = sum_{ky,kx,o : (iy+1-ky, ix+1-kx) even} dz[n, (iy+1-ky)/2, (ix+1-kx)/2, g*8+o] * w[o][i][tap]
// with the weights packed [tap][i][G][o] (o, the contraction index, contiguous; tap NOT flipped) = hn_gconv_pack's `wk` array.
// Thread = one 2x2 quad of input pixels (2yh+py, 2xh+px) x one group: the four parity classes use 1 + 2 + 2 + 4 of the nine taps and all of
// them read the same four dz pixels (yh..yh+1, xh..xh+1), so the quad costs exactly the nine tap products with no divergence (a thread
// per input pixel runs all nine tap bodies under lane masks in a mixed-parity wave: measured 157 us at stage 0 against a 22 us HBM floor).
template <bool LDSW>
__global__ __launch_bounds__(256) void gconv_dgrad_s2_kernel(const bf16* dz, int ldz, const bf16* wd_g, bf16* dx, int ldx, int N, int Hi,
                                                             int Wi, int Ho, int Wo, int G) {
    const bf16* wd = stage_gconv_weights(wd_g, G, LDSW);
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const long total = (long)N * Ho * Wo * G;
    const long idx = (long)bidx * 256 + threadIdx.x;
    if (idx >= total) return;
    const unsigned ui = (unsigned)idx;                       // (the host checks total < 2^32)
    const int g = (int)(ui % (unsigned)G);
    unsigned t = ui / (unsigned)G;
    const int xh = (int)(t % (unsigned)Wo);
    t /= (unsigned)Wo;
    const int yh = (int)(t % (unsigned)Ho);
    const long n = t / (unsigned)Ho;
    u32x4 zv[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool ok = yh + a < Ho && xh + c < Wo;
            zv[a][c] = ok ? *reinterpret_cast<const u32x4*>(dz + ((n * Ho + yh + a) * (long)Wo + xh + c) * ldz + g * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
    float acc[2][2][8];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[py][px][i] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        // input row parity py receives tap ky when iy + 1 - ky is even: py = 0 <-> ky = 1 (dz row yh); py = 1 <-> ky = 0 (yh + 1), ky = 2 (yh)
        const int py = ky == 1 ? 0 : 1, a = ky == 0 ? 1 : 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int px = kx == 1 ? 0 : 1, c = kx == 0 ? 1 : 0;
            // wd here = weights packed [tap][i][G][o] (o contiguous: the `wk` array of hn_gconv_pack): one packed dot over the 8 couts
            u32x4 wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = *reinterpret_cast<const u32x4*>(wd + (((long)(ky * 3 + kx) * 8 + i) * G + g) * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[py][px][i] = dot8_bf16(zv[a][c], wv[i], acc[py][px][i]);
        }
    }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f2bf(acc[py][px][i]);
            st8(dx + ((n * Hi + 2 * yh + py) * (long)Wi + 2 * xh + px) * ldx + g * 8, v);
        }
}

// wgrad partials: part[chunk][((g*8+o)*8 + i)*9 + tap] = sum over the chunk's output pixels of dz[pix][g*8+o] * x[pix(tap)][g*8+i]
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const bf16* x, int ldx, const bf16* dz, int ldz, float* part, int N, int Hi,
                                                          int Wi, int Ho, int Wo, int G, int S, long ppc, long nchunks) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int items = G * 9;
    const long tid = (long)bidx * 256 + threadIdx.x;       // (chunk, item) flattened: neighbouring lanes = neighbouring groups
    const long chunk = tid / items;
    const int item = (int)(tid - chunk * items);
    if (chunk >= nchunks) return;
    const int g = item % G, tap = item / G;
    const int ky = tap / 3, kx = tap - ky * 3;
    const long total = (long)N * Ho * Wo;
    const long p0 = chunk * ppc;
    long p1 = p0 + ppc;
    if (p1 > total) p1 = total;
    float acc[8][8];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[o][i] = 0.f;
    // eight pixels in flight per thread (the loads of a batch are all issued before the first FMA); out-of-image taps load nothing
    for (long pb = p0; pb < p1; pb += 8) {
        bf16x8 zv[8], xv[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long pix = pb + u;
            const int ox = (int)(pix % Wo);
            const long t1 = pix / Wo;
            const int oy = (int)(t1 % Ho);
            const long n = t1 / Ho;
            const int iy = oy * S + ky - 1, ix = ox * S + kx - 1;
            ok[u] = pix < p1 && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
            if (ok[u]) {
                zv[u] = ld8(dz + pix * ldz + g * 8);
                xv[u] = ld8(x + ((n * Hi + iy) * (long)Wi + ix) * ldx + g * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!ok[u]) continue;
            float xf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[i] = bf2f(xv[u][i]);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const float zf = bf2f(zv[u][o]);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[o][i] = fmaf(zf, xf[i], acc[o][i]);
            }
        }
    }
    float* dst = part + chunk * G * 576;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[((long)(g * 8 + o) * 8 + i) * 9 + tap] = acc[o][i];
}


// The same partial sums with SUB = 8 sub-chunks per pixel chunk folded inside the workgroup: workgroup = (chunk, block of 32 items) x 8
// sub-chunks, so the launch has 8x the threads for the same number of partial rows (the thread count above is tied to chunks x items, and
// the partial rows [chunks][C * 72] are what the fold pass pays for: stage 0 ran as 432 workgroups, 1.7 per CU, 16 dependent load rounds
// each).  The eight sub-chunk sums are added in fixed order through LDS: deterministic.
__global__ __launch_bounds__(256) void gconv_wgrad_sub_kernel(const bf16* x, int ldx, const bf16* dz, int ldz, float* part, int N, int Hi,
                                                              int Wi, int Ho, int Wo, int G, int S, long ppc, long nchunks) {
    __shared__ float red[256][9];
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);
    const int items = G * 9, iblocks = (items + 31) >> 5;
    const long chunk = bidx / iblocks;
    const int ib = bidx - (int)(chunk * iblocks);
    const int il = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const int item = ib * 32 + il;
    const bool live = item < items;
    const int g = live ? item % G : 0, tap = live ? item / G : 0;
    const int ky = tap / 3, kx = tap - ky * 3;
    const long total = (long)N * Ho * Wo;
    const long c0 = chunk * ppc;
    long c1 = c0 + ppc;
    if (c1 > total) c1 = total;
    const long q = (ppc + 7) >> 3;
    const long p0 = c0 + sub * q;
    long p1 = p0 + q;
    if (p1 > c1) p1 = c1;
    float acc[8][8];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[o][i] = 0.f;
    if (live && p0 < p1) {
        // pixel coordinates by one 32-bit decomposition + increments (the 64-bit divisions of the kernel above cost more than the FMAs)
        unsigned ox = (unsigned)p0 % (unsigned)Wo, t1 = (unsigned)p0 / (unsigned)Wo;
        unsigned oy = t1 % (unsigned)Ho, n = t1 / (unsigned)Ho;
        for (long pb = p0; pb < p1; pb += 4) {
            u32x4 zv[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long pix = pb + u;
                const int iy = (int)oy * S + ky - 1, ix = (int)ox * S + kx - 1;
                const bool ok = pix < p1 && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
                zv[u] = (u32x4){0u, 0u, 0u, 0u};
                xv[u] = (u32x4){0u, 0u, 0u, 0u};
                if (ok) {
                    zv[u] = *reinterpret_cast<const u32x4*>(dz + pix * ldz + g * 8);
                    xv[u] = *reinterpret_cast<const u32x4*>(x + (((long)n * Hi + iy) * (long)Wi + ix) * ldx + g * 8);
                }
                if (++ox == (unsigned)Wo) { ox = 0; if (++oy == (unsigned)Ho) { oy = 0; ++n; } }
            }
            // two pixels per packed dot: (dz[p][o], dz[p+1][o]) . (x[p][i], x[p+1][i]); the pairs are built with byte permutes
            // (16 per pixel pair) and feed 64 v_dot2c_f32_bf16 -- instead of 128 FMAs + 32 conversions (a missing pixel is a zero vector)
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                unsigned zp[8], xp[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    zp[2 * k] = __builtin_amdgcn_perm(zv[u + 1][k], zv[u][k], 0x05040100u);
                    zp[2 * k + 1] = __builtin_amdgcn_perm(zv[u + 1][k], zv[u][k], 0x07060302u);
                    xp[2 * k] = __builtin_amdgcn_perm(xv[u + 1][k], xv[u][k], 0x05040100u);
                    xp[2 * k + 1] = __builtin_amdgcn_perm(xv[u + 1][k], xv[u][k], 0x07060302u);
                }
#pragma unroll
                for (int o = 0; o < 8; ++o)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[o][i] = dot2c_bf16(zp[o], xp[i], acc[o][i]);
            }
        }
    }
    float* dst = part + chunk * G * 576;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[o][i];
        __syncthreads();
        if (sub == 0 && live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += red[k * 32 + il][i];
                dst[((long)(g * 8 + o) * 8 + i) * 9 + tap] = t;
            }
        }
    }
}

// pack fp32 grouped weights [C][8][3][3] into wk[tap][i][G][o] (forward) and wd[tap'][o][G][i]:
//   flip = 1: tap' = 8 - tap (stride-1 dgrad runs the forward kernel on dz);  flip = 0: tap' = tap (stride-2 dgrad kernel)
__global__ void gconv_pack_kernel(const float* w, bf16* wk, bf16* wd, int G, int flip) {
    const long total = (long)G * 576;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // idx enumerates the destination layout [tap][a][G][b]
    const int b = (int)(idx & 7);
    long t = idx >> 3;
    const int g = (int)(t % G);
    t /= G;
    const int a = (int)(t & 7);
    const int tap = (int)(t >> 3);
    wk[idx] = f2bf(w[((long)(g * 8 + b) * 8 + a) * 9 + tap]);                       // a = i, b = o
    if (wd) wd[idx] = f2bf(w[((long)(g * 8 + a) * 8 + b) * 9 + (flip ? 8 - tap : tap)]);   // a = o, b = i
}

// ---------------------------------------------------------------------------------------------------------
// depthwise 3x3, stride 1, zero pad 1.  Packed weights wk[tap][C] bf16.  Thread = 8 channels x strip of 4 pixels.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const bf16* in, int ldi, const bf16* wk, bf16* out, int ldo, int N, int C,
                                                         const Levels L, int accumulate) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3;
    long idx = (long)bidx * 256 + threadIdx.x;
    if (idx >= L.work_off[L.n]) {
        // ragged level packing: every level is padded to a multiple of the row alignment; the work items behind the real ones ZERO those
        // alignment rows, so that downstream GEMMs see exact zeros there (their BatchNorm statistics are corrected analytically)
        idx -= L.work_off[L.n];
        if (accumulate) return;                                        // the first writer already zeroed the alignment rows
        for (int l = 0; l < L.n; ++l) {
            const long real = (long)N * L.H[l] * L.W[l];
            const long pad = L.row_off[l + 1] - L.row_off[l] - real;
            if (idx < pad * C8) {
                st8(out + (L.row_off[l] + real + idx / C8) * ldo + (idx % C8) * 8, zero8());
                return;
            }
            idx -= pad * C8;
        }
        return;
    }
    int lv = 0;
    while (lv + 1 < L.n && idx >= L.work_off[lv + 1]) ++lv;
    idx -= L.work_off[lv];
    const int H = L.H[lv], W = L.W[lv], strips = (W + 3) >> 2;
    in += L.row_off[lv] * ldi;
    out += L.row_off[lv] * ldo;
    const int cg = (int)(idx % C8);
    long t = idx / C8;
    const int sx = (int)(t % strips);
    t /= strips;
    const int oy = (int)(t % H);
    const long n = t / H;
    const int ox0 = sx * 4;
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[p][k] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if (iy < 0 || iy >= H) continue;
        bf16x8 row[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = ox0 - 1 + j;
            row[j] = (ix >= 0 && ix < W) ? ld8(in + ((n * H + iy) * (long)W + ix) * ldi + cg * 8) : zero8();
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 wv = ld8(wk + (long)(ky * 3 + kx) * C + cg * 8);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[p][k] = fmaf(bf2f(row[p + kx][k]), bf2f(wv[k]), acc[p][k]);
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (ox0 + p >= W) break;
        bf16* dst = out + ((n * H + oy) * (long)W + ox0 + p) * ldo + cg * 8;
        bf16x8 v;
        if (accumulate) {                                              // a second consumer's gradient of the same tensor (ops.GradSlot)
            const bf16x8 prev = ld8(dst);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = f2bf(acc[p][k] + bf2f(prev[k]));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = f2bf(acc[p][k]);
        }
        st8(dst, v);
    }
}

// wgrad partials: part[block][c*9 + tap] = sum over the block's pixels of dz[pix][c] * x[pix + tap - 1][c].
// A block = (C/8 channel groups) x (256 / (C/8) lanes); every lane walks `spl` consecutive 4-pixel strips (global strip index over all
// packed levels): per strip the 3x6 input window (18 loads) and 4 gradients serve 36 tap products -- 5.5 loads per pixel instead of 10.
// Window elements outside the image read a clamped address and are masked to zero.  The lanes are then summed through LDS (three rounds
// of 24 accumulators) so that one partial row per block leaves the chip.
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const bf16* x, int ldx, const bf16* dz, int ldz, float* part, int N, int C,
                                                           int spl, const Levels L) {
    __shared__ float red[256][25];
    const int C8 = C >> 3;
    const int lanes = 256 / C8;
    const int tid = threadIdx.x;
    const int cg = tid % C8, lane = tid / C8;
    const bool active = lane < lanes;
    const long total = L.work_off[L.n];                               // strips
    const long s0 = ((long)blockIdx.x * lanes + lane) * spl;
    long s1 = s0 + spl;
    if (s1 > total) s1 = total;
    float acc[9][8];
#pragma unroll
    for (int tq = 0; tq < 9; ++tq)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[tq][k] = 0.f;
    if (active) {
        int lv = 0;
        for (long sidx = s0; sidx < s1; ++sidx) {
            while (lv + 1 < L.n && sidx >= L.work_off[lv + 1]) ++lv;
            const int H = L.H[lv], W = L.W[lv], strips = (W + 3) >> 2;
            const long ls = sidx - L.work_off[lv];
            const int sx = (int)(ls % strips);
            const long t1 = ls / strips;
            const int oy = (int)(t1 % H);
            const long n = t1 / H;
            const int ox0 = sx * 4;
            const bf16* xl = x + L.row_off[lv] * ldx;
            const bf16* zl = dz + L.row_off[lv] * ldz;
            bf16x8 xw[3][6], zv[4];
            float xm[3][6], zm[4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy + r - 1;
                const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int ix = ox0 + c - 1;
                    const int ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                    xm[r][c] = (iy == iyc && ix == ixc) ? 1.f : 0.f;
                    xw[r][c] = ld8(xl + ((n * H + iyc) * (long)W + ixc) * ldx + cg * 8);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int ox = ox0 + p;
                zm[p] = ox < W ? 1.f : 0.f;
                zv[p] = ld8(zl + ((n * H + oy) * (long)W + (ox < W ? ox : W - 1)) * ldz + cg * 8);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float zf[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) zf[k] = bf2f(zv[p][k]) * zm[p];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float m = xm[ky][p + kx];
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[ky * 3 + kx][k] = fmaf(zf[k] * m, bf2f(xw[ky][p + kx][k]), acc[ky * 3 + kx][k]);
                    }
            }
        }
    }
    float* dst = part + (long)blockIdx.x * C * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        __syncthreads();
#pragma unroll
        for (int tq = 0; tq < 3; ++tq)
#pragma unroll
            for (int k = 0; k < 8; ++k) red[tid][tq * 8 + k] = active ? acc[r * 3 + tq][k] : 0.f;
        __syncthreads();
        for (int o = tid; o < C8 * 24; o += 256) {
            const int g = o / 24, v = o - g * 24;
            float sum = 0.f;
            for (int l = 0; l < lanes; ++l) sum += red[l * C8 + g][v];
            const int tq = r * 3 + v / 8, k = v & 7;
            dst[(long)(g * 8 + k) * 9 + tq] = sum;
        }
    }
}

// Backward of the depthwise conv in ONE pass over (dz, x): the data gradient dx[q] = sum_t wf[t] * dz[q + t - 1] (wf = flipped weights)
// and the weight gradient dw[ky][kx] = sum_q x[q] * dz[q - (ky-1, kx-1)] read the SAME 3x3 neighbourhood of dz around q, so the 3x6
// window of a 4-pixel strip serves 36 products for dx and 36 for dw.  Thread = 4 channels (8-byte loads: with 8 channels the window, the
// 72 accumulators and 22 addresses do not fit 256 registers) x `spl` consecutive strips; dx leaves per strip, the 36 weight-gradient
// accumulators are summed over the block's lanes through LDS once at the end (one partial row per block).  Separate launches read dz
// twice and pay the lane reduction for two strips of work: 43 + 26 us on the packed det map.  Blocks behind the real ones zero the
// alignment rows of dx (ragged level packing), as the forward kernel does.
__device__ __forceinline__ bf16x4 ld4(const bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ bf16x4 zero4() { bf16x4 z; z[0] = z[1] = z[2] = z[3] = (bf16)0.0f; return z; }

__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const bf16* dz, int ldz, const bf16* x, int ldx, const bf16* wf, bf16* dx, int lddx,
                                                         float* part, int N, int C, int spl, long blocks, const Levels L, int accumulate) {
    extern __shared__ float red[];                                    // [256][37] lane sums, then [9][C] fp32 flipped weights
    float* wl = red + 256 * 37;
    const int C4 = C >> 2;
    const int tid = threadIdx.x;
    const long bidx = (long)blockIdx.x < blocks ? xcd_remap(blockIdx.x, (int)blocks) : blockIdx.x;    // row-order placement (hn_common.h)
    if ((long)blockIdx.x >= blocks) {
        const int C8 = C >> 3;
        long idx = ((long)blockIdx.x - blocks) * 256 + tid;
        for (int l = 0; l < L.n; ++l) {
            const long real = (long)N * L.H[l] * L.W[l];
            const long pad = L.row_off[l + 1] - L.row_off[l] - real;
            if (idx < pad * C8) {
                st8(dx + (L.row_off[l] + real + idx / C8) * lddx + (idx % C8) * 8, zero8());
                return;
            }
            idx -= pad * C8;
        }
        return;
    }
    const int lanes = 256 / C4;
    const int cg = tid % C4, lane = tid / C4;
    const bool active = lane < lanes;
    const long total = L.work_off[L.n];                               // strips
    const long s0 = (bidx * lanes + lane) * spl;
    long s1 = s0 + spl;
    if (s1 > total) s1 = total;
    float acc[9][4];
#pragma unroll
    for (int tq = 0; tq < 9; ++tq)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[tq][k] = 0.f;
    if (dx) {
        for (int i = tid; i < 9 * C; i += 256) wl[i] = bf2f(wf[i]);
        __syncthreads();
    }
    if (active) {
        int lv = 0;
        for (long sidx = s0; sidx < s1; ++sidx) {
            while (lv + 1 < L.n && sidx >= L.work_off[lv + 1]) ++lv;
            const int H = L.H[lv], W = L.W[lv], strips = (W + 3) >> 2;
            const long ls = sidx - L.work_off[lv];
            const int sx = (int)(ls % strips);
            const long t1 = ls / strips;
            const int oy = (int)(t1 % H);
            const long n = t1 / H;
            const int ox0 = sx * 4;
            const bf16* zl = dz + L.row_off[lv] * ldz + cg * 4;
            const bf16* xl = x + L.row_off[lv] * ldx + cg * 4;
            bf16x4 zw[3][6], xv[4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy + r - 1;
                const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int ix = ox0 + c - 1;
                    const int ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                    zw[r][c] = ld4(zl + ((n * H + iyc) * (long)W + ixc) * ldz);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int ox = ox0 + p;
                xv[p] = ld4(xl + ((n * H + oy) * (long)W + (ox < W ? ox : W - 1)) * ldx);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (ox0 + p >= W) xv[p] = zero4();
            float da[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) da[p][k] = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy + r - 1;
                const bool rok = iy >= 0 && iy < H;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int ix = ox0 + c - 1;
                    if (!(rok && ix >= 0 && ix < W)) zw[r][c] = zero4();
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (dx) {
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (r * 3 + j) * C + cg * 4);
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int k = 0; k < 4; ++k) da[p][k] = fmaf(bf2f(zw[r][p + j][k]), w0[k], da[p][k]);
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            acc[8 - (r * 3 + j)][k] = fmaf(bf2f(xv[p][k]), bf2f(zw[r][p + j][k]), acc[8 - (r * 3 + j)][k]);
                }
            }
            if (dx) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (ox0 + p >= W) break;
                    bf16* dst = dx + (L.row_off[lv] + (n * H + oy) * (long)W + ox0 + p) * lddx + cg * 4;
                    bf16x4 v;
                    if (accumulate) {
                        const bf16x4 prev = ld4(dst);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = f2bf(da[p][k] + bf2f(prev[k]));
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = f2bf(da[p][k]);
                    }
                    *reinterpret_cast<bf16x4*>(dst) = v;
                }
            }
        }
    }
#pragma unroll
    for (int tq = 0; tq < 9; ++tq)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[tid * 37 + tq * 4 + k] = active ? acc[tq][k] : 0.f;
    __syncthreads();
    float* dst = part + bidx * C * 9;
    for (int o = tid; o < C4 * 36; o += 256) {
        const int g = o / 36, v = o - g * 36;
        float sum = 0.f;
        for (int l = 0; l < lanes; ++l) sum += red[(l * C4 + g) * 37 + v];
        const int tq = v >> 2, k = v & 3;
        dst[(long)(g * 4 + k) * 9 + tq] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same one-pass backward, LDS-tiled (round 6; C <= 120, maps of >= 512 tiles).  The strip kernel above asks the cache hierarchy for every
// dz element 5.5 times (18 eight-byte loads per 4-pixel strip) and its threads walk their strips one after the other -- loads, wait,
// ~450 VALU instructions, stores -- so the memory system idles while the lanes compute: 2.0-2.5 TB/s of dz + x + dx on the level-packed
// tower tensor and the P3 maps.  Here a workgroup owns 8 x 16-pixel output tiles with ALL channels: the 10 x 18 dz halo tile and the x tile
// arrive by LDS-DMA in whole 224-byte pixel rows (every byte of dz crosses L2 -> LDS 1.4 times), two workgroups per CU cover each other's
// DMA waits, a thread = (4 channels, two tile columns) slides a 3 x 3 window down the tile's 8 rows (3 + 1 ds_read_b64 per output pixel)
// with its 36 weights in registers, and its 36 weight-gradient accumulators live across the workgroup's `tpw` tiles.  dx: the same
// products in the same order as dwconv_fwd_kernel with the flipped weights (bit-identical); dweight: one partial row per workgroup
// (fixed-order LDS fold).  Measured and not kept: 8 channels per thread with the weights re-read from LDS (latency of 18 reads per row:
// 50 us on the tower tensor), 4 x 16 tiles double-buffered (more halo, more requests: 47), one 512-thread workgroup per CU with two
// 8 x 16 buffers and the per-piece coordinates precomputed (45); this form: 42.5 (strip form 56.7).
// ---------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) bf16 st_zero_piece[8];
struct DwTiles { long tile_off[HN_MAX_LEVELS + 1]; int ty[HN_MAX_LEVELS], tx[HN_MAX_LEVELS]; };
__device__ __forceinline__ void st_glds16(const bf16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define DWT_TH 8                                                     // tile rows (x 16 columns)
template <bool HAS_DX>
__global__ __launch_bounds__(256, 2) void dwconv_bwd_tiled_kernel(const bf16* dz, int ldz, const bf16* x, int ldx, const bf16* wf, bf16* dx, int lddx,
                                                                  float* part, int N, int C, int tpw, int blocks, const Levels L, const DwTiles T,
                                                                  int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TH = DWT_TH, ZPX = (TH + 2) * 18, XPX = TH * 16;
    const int C8 = C >> 3, C4 = C >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= blocks) {                                   // ragged level packing: the alignment rows of dx are zeroed
        long idx = ((long)blockIdx.x - blocks) * 256 + tid;
        for (int l = 0; l < L.n; ++l) {
            const long real = (long)N * L.H[l] * L.W[l];
            const long pad = L.row_off[l + 1] - L.row_off[l] - real;
            if (idx < pad * C8) {
                st8(dx + (L.row_off[l] + real + idx / C8) * lddx + (idx % C8) * 8, zero8());
                return;
            }
            idx -= pad * C8;
        }
        return;
    }
    const int ZCH = (ZPX * C8 + 63) >> 6, XCH = (XPX * C8 + 63) >> 6;   // 1 KB DMA chunks of the dz halo tile / the x tile
    char* sZ = smem;
    char* sX = smem + ZCH * 1024;
    const int bidx = xcd_remap(blockIdx.x, blocks);
    const long total = T.tile_off[L.n];
    const long t0 = (long)bidx * tpw;
    long t1 = t0 + tpw;
    if (t1 > total) t1 = total;
    // thread = (4 channels, tile columns colh and colh + 8): 9 x 4 weights, 9 x 4 weight-gradient accumulators and the 3 x 3 window in
    // registers (108 of the 256 a wave has with two 4-wave workgroups per CU; 8 channels per thread would be 216 + temporaries)
    const int cq = tid % C4, colh = tid / C4;
    const bool active = colh < 8;
    float wr[9][4], acc[9][4];
#pragma unroll
    for (int tq = 0; tq < 9; ++tq) {
        const bf16x4 wv = (HAS_DX && active) ? ld4(wf + tq * C + cq * 4) : zero4();
#pragma unroll
        for (int k = 0; k < 4; ++k) { wr[tq][k] = bf2f(wv[k]); acc[tq][k] = 0.f; }
    }
    const unsigned magic = (65536u + C8 - 1) / C8;                      // e / C8 = (e * magic) >> 16 for e < 4096 (C8 <= 16)
    int lv = 0;
    for (long t = t0; t < t1; ++t) {
        while (lv + 1 < L.n && t >= T.tile_off[lv + 1]) ++lv;           // (tiles are visited in increasing order: lv only grows)
        const int H = L.H[lv], W = L.W[lv];
        long lt = t - T.tile_off[lv];
        const int tx = (int)(lt % T.tx[lv]);
        lt /= T.tx[lv];
        const int ty = (int)(lt % T.ty[lv]);
        const long n = lt / T.ty[lv];
        const int oy0 = ty * TH, ox0 = tx * 16;
        const long row0 = L.row_off[lv] + n * H * (long)W;
        __syncthreads();                                               // the previous tile's readers are done
        for (int ch = wave; ch < ZCH; ch += 4) {
            const int e = ch * 64 + lane;
            const int px = (int)(((unsigned)e * magic) >> 16), pc = e - px * C8;
            const int py = px / 18, pxx = px - py * 18;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
            const bf16* src = (px < ZPX && iy >= 0 && iy < H && ix >= 0 && ix < W) ? dz + (row0 + (long)iy * W + ix) * ldz + pc * 8 : st_zero_piece;
            st_glds16(src, sZ + ch * 1024);
        }
        for (int ch = wave; ch < XCH; ch += 4) {
            const int e = ch * 64 + lane;
            const int px = (int)(((unsigned)e * magic) >> 16), pc = e - px * C8;
            const int iy = oy0 + (px >> 4), ix = ox0 + (px & 15);
            const bf16* src = (px < XPX && iy < H && ix < W) ? x + (row0 + (long)iy * W + ix) * ldx + pc * 8 : st_zero_piece;
            st_glds16(src, sX + ch * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (active)
#pragma unroll 1
        for (int col = colh; col < 16; col += 8) {
            bf16x4 win[3][3];                                          // dz rows r-1, r, r+1 (halo rows r, r+1, r+2) x halo columns col, col+1, col+2
            auto load_row = [&](int py, bf16x4 (&w3)[3]) {
#pragma unroll
                for (int j = 0; j < 3; ++j) w3[j] = *reinterpret_cast<const bf16x4*>(sZ + ((py * 18 + col + j) * C4 + cq) * 8);
            };
            load_row(0, win[0]);
            load_row(1, win[1]);
            // rows in groups of three (the window's slot of a row is then a compile-time index) inside a ROLLED loop: fully unrolled, the
            // compiler requests all ten halo rows up front
#pragma unroll 1
            for (int r3 = 0; r3 < TH; r3 += 3)
#pragma unroll
            for (int rq = 0; rq < 3; ++rq) {
                const int r = r3 + rq;
                if (r >= TH) break;
                load_row(r + 2, win[(rq + 2) % 3]);
                const bf16x4 xb = *reinterpret_cast<const bf16x4*>(sX + ((r * 16 + col) * C4 + cq) * 8);
                float xv[4], da[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { xv[k] = bf2f(xb[k]); da[k] = 0.f; }
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const bf16x4 zq = win[(rq + rr) % 3][j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float zf = bf2f(zq[k]);
                            if (HAS_DX) da[k] = fmaf(zf, wr[rr * 3 + j][k], da[k]);
                            acc[8 - (rr * 3 + j)][k] = fmaf(xv[k], zf, acc[8 - (rr * 3 + j)][k]);
                        }
                    }
                if (HAS_DX && oy0 + r < H && ox0 + col < W) {
                    bf16* dst = dx + (row0 + (long)(oy0 + r) * W + ox0 + col) * lddx + cq * 4;
                    bf16x4 v;
                    if (accumulate) {
                        const bf16x4 prev = ld4(dst);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = f2bf(da[k] + bf2f(prev[k]));
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = f2bf(da[k]);
                    }
                    *reinterpret_cast<bf16x4*>(dst) = v;
                }
            }
        }
    }
    // one partial row per workgroup: the 8 column lanes of a channel group are folded through LDS in lane order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                      // [8 * C4][37]
    if (active) {
#pragma unroll
        for (int tq = 0; tq < 9; ++tq)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[(colh * C4 + cq) * 37 + tq * 4 + k] = acc[tq][k];
    }
    __syncthreads();
    float* dst = part + (long)bidx * C * 9;
    for (int o = tid; o < C4 * 36; o += 256) {
        const int g = o / 36, v = o - g * 36;
        float sum = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) sum += red[(c2 * C4 + g) * 37 + v];
        dst[(long)(g * 4 + (v & 3)) * 9 + (v >> 2)] = sum;
    }
}

// fp32 [C][1][3][3] -> wk[tap][C] and flipped wkf[8 - tap][C]
__global__ void dw_pack_kernel(const float* w, bf16* wk, bf16* wkf, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 9 * C) return;
    const int c = idx % C, tap = idx / C;
    const bf16 v = f2bf(w[c * 9 + tap]);
    wk[idx] = v;
    if (wkf) wkf[(8 - tap) * C + c] = v;
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 stride-2 max pools.  mode 0: ZERO pad right/bottom (window rows 2oy..2oy+2, out-of-range taps contribute 0.0),
// mode 1: -inf pad 1 (window rows 2oy-1..2oy+1, out-of-range taps ignored).  Ho = H/2, Wo = W/2 (H, W even) -- also
// valid for odd-free shapes down to 2x2.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pool_window(const bf16* in, int ldi, long n, int H, int W, int oy, int ox, int c, int mode, float* best,
                                            int* arg) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; arg[k] = -1; }
    const int off = mode ? -1 : 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = 2 * oy + ky + off, ix = 2 * ox + kx + off;
            const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
            if (!inside && mode) continue;
            bf16x8 v = inside ? ld8(in + ((n * H + iy) * (long)W + ix) * ldi + c) : zero8();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = bf2f(v[k]);
                if (f > best[k]) { best[k] = f; arg[k] = inside ? ky * 3 + kx : 9; }   // first maximum wins (strict >)
            }
        }
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16* in, int ldi, bf16* out, int ldo, int N, int H, int W, int C,
                                                          int mode) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const long n = t / Ho;
        float best[8];
        int arg[8];
        pool_window(in, ldi, n, H, W, oy, ox, cg * 8, mode, best, arg);
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = f2bf(best[k]);
        st8(out + ((n * Ho + oy) * (long)Wo + ox) * ldo + cg * 8, v);
    }
}

// dx[n, iy, ix, c] = wscale * sum over windows whose (recomputed) arg-max is (iy, ix) of dout[window]
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16* in, int ldi, const bf16* dout, int ldd, bf16* dx, int ldx,
                                                          const float* wscale, int N, int H, int W, int C, int mode) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * H * W * C8;
    const float ws = wscale ? *wscale : 1.0f;
    const int off = mode ? -1 : 0;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        // windows oy with 2*oy + off <= iy <= 2*oy + off + 2
        for (int oy = (iy - off - 2 + 1) >> 1; 2 * oy + off <= iy; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            for (int ox = (ix - off - 2 + 1) >> 1; 2 * ox + off <= ix; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                float best[8];
                int arg[8];
                pool_window(in, ldi, n, H, W, oy, ox, cg * 8, mode, best, arg);
                const int mine = (iy - 2 * oy - off) * 3 + (ix - 2 * ox - off);
                const bf16x8 g = ld8(dout + ((n * Ho + oy) * (long)Wo + ox) * ldd + cg * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (arg[k] == mine) acc[k] += bf2f(g[k]);
            }
        }
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = f2bf(acc[k] * ws);
        st8(dx + ((n * H + iy) * (long)W + ix) * ldx + cg * 8, v);
    }
}

// two-pass backward: (1) one thread per output window recomputes its arg-max (9 loads) and stores it as a byte per channel; (2) one thread
// per input pixel visits its <= 4 windows and reads only their arg bytes and gradients (the one-pass kernel above re-reads 9 inputs per
// window per pixel: 40 loads per pixel instead of ~8)
__global__ __launch_bounds__(256) void maxpool_arg_kernel(const bf16* in, int ldi, unsigned char* arg, int N, int H, int W, int C, int mode) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const long n = t / Ho;
        float best[8];
        int a[8];
        pool_window(in, ldi, n, H, W, oy, ox, cg * 8, mode, best, a);
        unsigned long long packed = 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) packed |= (unsigned long long)(a[k] & 0xff) << (8 * k);
        *reinterpret_cast<unsigned long long*>(arg + ((n * Ho + oy) * (long)Wo + ox) * C + cg * 8) = packed;
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd_arg_kernel(const unsigned char* arg, const bf16* dout, int ldd, bf16* dx, int ldx,
                                                              const float* wscale, int N, int H, int W, int C, int mode, int accumulate) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * H * W * C8;
    const float ws = wscale ? *wscale : 1.0f;
    const int off = mode ? -1 : 0;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int oy = (iy - off - 2 + 1) >> 1; 2 * oy + off <= iy; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            for (int ox = (ix - off - 2 + 1) >> 1; 2 * ox + off <= ix; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                const long wpix = (n * Ho + oy) * (long)Wo + ox;
                const unsigned long long a = *reinterpret_cast<const unsigned long long*>(arg + wpix * C + cg * 8);
                const int mine = (iy - 2 * oy - off) * 3 + (ix - 2 * ox - off);
                const bf16x8 g = ld8(dout + wpix * ldd + cg * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((int)((a >> (8 * k)) & 0xff) == mine) acc[k] += bf2f(g[k]);
            }
        }
        bf16* dst = dx + ((n * H + iy) * (long)W + ix) * ldx + cg * 8;
        bf16x8 v;
        if (accumulate) {
            const bf16x8 o = ld8(dst);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = f2bf(fmaf(acc[k], ws, bf2f(o[k])));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = f2bf(acc[k] * ws);
        }
        st8(dst, v);
    }
}

// nearest x2 up-sampling (forward) and its backward (2x2 sum, optional device-side scale)
__global__ __launch_bounds__(256) void up2_fwd_kernel(const bf16* in, int ldi, bf16* out, int ldo, int N, int H, int W, int C) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3, Ho = 2 * H, Wo = 2 * W;
    const long total = (long)N * Ho * Wo * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const long n = t / Ho;
        st8(out + ((n * Ho + oy) * (long)Wo + ox) * ldo + cg * 8, ld8(in + ((n * H + (oy >> 1)) * (long)W + (ox >> 1)) * ldi + cg * 8));
    }
}
__global__ __launch_bounds__(256) void sum2x2_kernel(const bf16* g, int ldg, bf16* out, int ldo, const float* wscale, int N, int H, int W,
                                                     int C, int accumulate) {   // H, W = LOW resolution
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3;
    const long total = (long)N * H * W * C8;
    const float ws = wscale ? *wscale : 1.0f;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const long n = t / H;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const bf16x8 v = ld8(g + ((n * 2 * H + 2 * y + dy) * (long)(2 * W) + 2 * x + dx) * ldg + cg * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
            }
        bf16* dst = out + ((n * H + y) * (long)W + x) * ldo + cg * 8;
        bf16x8 o;
        if (accumulate) {
            const bf16x8 prev = ld8(dst);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(fmaf(acc[k], ws, bf2f(prev[k])));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(acc[k] * ws);
        }
        st8(dst, o);
    }
}

// ---------------------------------------------------------------------------------------------------------
// BiFPN fusion node: out = swish(sum_i w[i] * T_i(in_i)),  T in {1: identity, 2: nearest x2 of a half-res map,
// 3: zero-pad-same 3x3/s2 max-pool of a double-res map}; mode 0 = absent.  w lives in device memory (normalised by the host
// graph from the learnable fusion parameters, net/bifpn.py:179-180).
// ---------------------------------------------------------------------------------------------------------
struct Fuse {
    const bf16* in[3]; int ld[3]; int mode[3];
    const float* praw; int nw; float eps; float* wn;   // praw != null: the kernel normalises the raw fusion parameters itself (and stores wn)
    const float* w;
    bf16* out; int ldo;
    int N, H, W, C;       // output resolution
};
__device__ __forceinline__ void fuse_gather(const Fuse& p, int i, long n, int y, int x, int c, float* v) {
    if (p.mode[i] == 1) {
        const bf16x8 t = ld8(p.in[i] + ((n * p.H + y) * (long)p.W + x) * p.ld[i] + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf2f(t[k]);
    } else if (p.mode[i] == 2) {
        const bf16x8 t = ld8(p.in[i] + ((n * (p.H >> 1) + (y >> 1)) * (long)(p.W >> 1) + (x >> 1)) * p.ld[i] + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf2f(t[k]);
    } else {
        int arg[8];
        pool_window(p.in[i], p.ld[i], n, 2 * p.H, 2 * p.W, y, x, c, 0, v, arg);
    }
}
__global__ __launch_bounds__(256) void fuse_fwd_kernel(const Fuse p) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = p.C >> 3;
    const long total = (long)p.N * p.H * p.W * C8;
    float wv[3];
    if (p.praw) {                      // w = relu(p) / (sum relu(p) + eps), net/bifpn.py:179-180 (no separate one-thread launch)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { wv[i] = i < p.nw ? fmaxf(p.praw[i], 0.f) : 0.f; sum += wv[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) wv[i] = wv[i] / (sum + p.eps);
        if (bidx == 0 && threadIdx.x < 3) p.wn[threadIdx.x] = wv[threadIdx.x];
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) wv[i] = p.w[i];
    }
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int x = (int)(t % p.W);
        t /= p.W;
        const int y = (int)(t % p.H);
        const long n = t / p.H;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!p.mode[i]) continue;
            float v[8];
            fuse_gather(p, i, n, y, x, cg * 8, v);
            const float wi = wv[i];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = fmaf(wi, v[k], acc[k]);
        }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(act_fwd(acc[k], HN_ACT_SWISH));
        st8(p.out + ((n * p.H + y) * (long)p.W + x) * p.ldo + cg * 8, o);
    }
}

// dp_i = [p_i > 0] * (dw_i - sum_j w_j dw_j) / (sum relu(p) + eps), dw = column sums of the per-block partials pw[blocks][3]
__global__ __launch_bounds__(256) void fuse_dweights_kernel(const float* pw, int blocks, const float* praw, int nw, float eps, float* dp) {
    __shared__ float red[4][3];
    float a[3] = {0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < blocks; b += 256)
        for (int i = 0; i < 3; ++i) a[i] += pw[b * 3 + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < 3; ++i) {
        const float t = wave_sum(a[i]);
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float dw[3], r[3], sum = 0.f, dot = 0.f;
        for (int i = 0; i < 3; ++i) {
            dw[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
            r[i] = i < nw ? fmaxf(praw[i], 0.f) : 0.f;
            sum += r[i];
        }
        for (int i = 0; i < 3; ++i) dot += dw[i] * r[i] / (sum + eps);
        for (int i = 0; i < nw; ++i) dp[i] = praw[i] > 0.f ? (dw[i] - dot) / (sum + eps) : 0.f;
    }
}

// backward part 1: g = dout * swish'(pre) (bf16, output resolution), d_in for identity inputs (= w[i] * g), and per-block partial
// sums of dw[i] = sum g * T_i(in_i)  -> pw[block][3]
struct FuseBwd {
    Fuse f;
    const bf16* dout; int ldd;
    bf16* g; int ldg;
    bf16* din[3]; int ldin[3];      // only for mode 1 inputs (else null)
    int acc[3];                     // 1: din[i] += (the tensor already holds the gradient of another consumer)
    float* pw;
    unsigned char* arg[3];          // mode 3 inputs (optional): the arg-max byte of every pooling window / channel, the layout of maxpool_arg_kernel --
};                                  // the kernel recomputes the windows anyway; hn_maxpool_bwd_from_arg then needs no arg pass of its own
// Inputs of mode 2 (nearest x2 of a half-resolution map) that come with a destination (din[i] at the LOW resolution): the kernel walks
// the output in 2 x 2 quads and writes w_i * (sum of the quad's g) itself -- the separate hn_sum2x2 pass over g (one launch per top-down
// fusion node, 12 per step) is not needed; the sum is taken over the bf16-rounded g, as that pass did.
__global__ __launch_bounds__(256) void fuse_bwd_kernel(const FuseBwd q) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const Fuse& p = q.f;
    const int C8 = p.C >> 3;
    int up_i = -1;                                         // the (at most one) up-sampled input whose gradient is folded here
#pragma unroll
    for (int i = 0; i < 3; ++i) if (p.mode[i] == 2 && q.din[i]) up_i = i;
    const bool quads = up_i >= 0;
    const int Wq = quads ? p.W >> 1 : p.W, Hq = quads ? p.H >> 1 : p.H;
    const long total = (long)p.N * Hq * Wq * C8;
    float dw[3] = {0.f, 0.f, 0.f};
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int xq = (int)(t % Wq);
        t /= Wq;
        const int yq = (int)(t % Hq);
        const long n = t / Hq;
        float qsum[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qsum[k] = 0.f;
        for (int sub = 0; sub < (quads ? 4 : 1); ++sub) {
        const int x = quads ? 2 * xq + (sub & 1) : xq, y = quads ? 2 * yq + (sub >> 1) : yq;
        float v[3][8], pre[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pre[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!p.mode[i]) continue;
            if (p.mode[i] == 3 && q.arg[i]) {
                int a[8];
                pool_window(p.in[i], p.ld[i], n, 2 * p.H, 2 * p.W, y, x, cg * 8, 0, v[i], a);
                unsigned long long packed = 0ull;
#pragma unroll
                for (int k = 0; k < 8; ++k) packed |= (unsigned long long)(a[k] & 0xff) << (8 * k);
                *reinterpret_cast<unsigned long long*>(q.arg[i] + ((n * p.H + y) * (long)p.W + x) * p.C + cg * 8) = packed;
            } else {
                fuse_gather(p, i, n, y, x, cg * 8, v[i]);
            }
            const float wi = p.w[i];
#pragma unroll
            for (int k = 0; k < 8; ++k) pre[k] = fmaf(wi, v[i][k], pre[k]);
        }
        const long orow = (n * p.H + y) * (long)p.W + x;
        const bf16x8 d = ld8(q.dout + orow * q.ldd + cg * 8);
        float gg[8];
        bf16x8 go;
#pragma unroll
        for (int k = 0; k < 8; ++k) { gg[k] = bf2f(d[k]) * act_bwd(pre[k], HN_ACT_SWISH); go[k] = f2bf(gg[k]); qsum[k] += bf2f(go[k]); }
        st8(q.g + orow * q.ldg + cg * 8, go);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!p.mode[i]) continue;
#pragma unroll
            for (int k = 0; k < 8; ++k) dw[i] = fmaf(gg[k], v[i][k], dw[i]);
            if (p.mode[i] == 1 && q.din[i]) {
                const float wi = p.w[i];
                bf16* dst = q.din[i] + orow * q.ldin[i] + cg * 8;
                bf16x8 o;
                if (q.acc[i]) {
                    const bf16x8 prev = ld8(dst);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = f2bf(fmaf(wi, gg[k], bf2f(prev[k])));
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = f2bf(wi * gg[k]);
                }
                st8(dst, o);
            }
        }
        }
        if (quads) {                                       // = sum2x2_kernel on this quad
            const float ws = p.w[up_i];
            bf16* dst = q.din[up_i] + ((n * Hq + yq) * (long)Wq + xq) * q.ldin[up_i] + cg * 8;
            bf16x8 o;
            if (q.acc[up_i]) {
                const bf16x8 prev = ld8(dst);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = f2bf(fmaf(qsum[k], ws, bf2f(prev[k])));
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = f2bf(qsum[k] * ws);
            }
            st8(dst, o);
        }
    }
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float s = wave_sum(dw[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) q.pw[bidx * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// The top-down nodes (ONE identity input + ONE nearest-up input with a low-resolution destination: every top-down node of net/bifpn.py:
// 185-199) in their own kernel: the generic loop above walks the four pixels of a quad one after the other, each a load -> swish' -> store
// chain of its own, i.e. three or four 16-byte loads in flight per thread.  Here every load of the quad (4 x dout, 4 x the identity input,
// 4 x its accumulated destination, the low-resolution pixel and its destination) is issued before the first use: up to 14 loads in
// flight per thread.  Same arithmetic in the same order (inputs are summed in index order).
// 136 VGPRs = three workgroups per CU; held to 128 (four) it spills six registers and measures the same.  With it the 2 x 2 sums are
// worth folding: 863 -> 869 img/s against the separate hn_sum2x2 launches (the generic quad walk: 861).
__global__ __launch_bounds__(256) void fuse_bwd_quads_kernel(const FuseBwd q) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const Fuse& p = q.f;
    const int C8 = p.C >> 3;
    int ia = 0, ib = 0;                                    // the identity and the up-sampled input
#pragma unroll
    for (int i = 0; i < 3; ++i) { if (p.mode[i] == 1) ia = i; if (p.mode[i] == 2) ib = i; }
    const bool a_first = ia < ib;
    const float wa = p.w[ia], wb = p.w[ib];
    const bf16* ina = p.in[ia]; const int lda = p.ld[ia];
    const bf16* inb = p.in[ib]; const int ldb = p.ld[ib];
    bf16* da = q.din[ia]; const int ldda = q.ldin[ia]; const bool acca = da && q.acc[ia];
    bf16* db = q.din[ib]; const int lddb = q.ldin[ib]; const bool accb = q.acc[ib];
    const int Wq = p.W >> 1, Hq = p.H >> 1;
    const long total = (long)p.N * Hq * Wq * C8;
    float dwa = 0.f, dwb = 0.f;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int xq = (int)(t % Wq);
        t /= Wq;
        const int yq = (int)(t % Hq);
        const long n = t / Hq;
        const long orow0 = (n * p.H + 2 * yq) * (long)p.W + 2 * xq, lrow = (n * Hq + yq) * (long)Wq + xq;
        bf16x8 rd[4], ra[4], rprev[4], rup, rprev_up;
#pragma unroll
        for (int s = 0; s < 4; ++s) rd[s] = ld8(q.dout + (orow0 + (s >> 1) * p.W + (s & 1)) * q.ldd + cg * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) ra[s] = ld8(ina + (orow0 + (s >> 1) * p.W + (s & 1)) * lda + cg * 8);
        rup = ld8(inb + lrow * ldb + cg * 8);
        if (acca) {
#pragma unroll
            for (int s = 0; s < 4; ++s) rprev[s] = ld8(da + (orow0 + (s >> 1) * p.W + (s & 1)) * ldda + cg * 8);
        }
        bf16* dup = db + lrow * lddb + cg * 8;
        if (accb) rprev_up = ld8(dup);
        float qsum[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { qsum[k] = 0.f; vb[k] = bf2f(rup[k]); }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long orow = orow0 + (s >> 1) * p.W + (s & 1);
            float gg[8];
            bf16x8 go;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float va = bf2f(ra[s][k]);
                const float pre = a_first ? fmaf(wb, vb[k], fmaf(wa, va, 0.f)) : fmaf(wa, va, fmaf(wb, vb[k], 0.f));
                gg[k] = bf2f(rd[s][k]) * act_bwd(pre, HN_ACT_SWISH);
                go[k] = f2bf(gg[k]);
                qsum[k] += bf2f(go[k]);
                dwa = fmaf(gg[k], va, dwa);
                dwb = fmaf(gg[k], vb[k], dwb);
            }
            st8(q.g + orow * q.ldg + cg * 8, go);
            if (da) {
                bf16x8 o;
                if (acca) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = f2bf(fmaf(wa, gg[k], bf2f(rprev[s][k])));
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = f2bf(wa * gg[k]);
                }
                st8(da + orow * ldda + cg * 8, o);
            }
        }
        bf16x8 o;
        if (accb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(fmaf(qsum[k], wb, bf2f(rprev_up[k])));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(qsum[k] * wb);
        }
        st8(dup, o);
    }
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dw[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) dw[i] = i == ia ? dwa : (i == ib ? dwb : 0.f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float sm = wave_sum(dw[i]);
        if (lane == 0) red[wave][i] = sm;
    }
    __syncthreads();
    if (threadIdx.x < 3) q.pw[bidx * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------
// fold of the seg decoder's padded-domain dgrad: dvp [N][H+2][W+2][ldv] (channels c0..c0+C) ->
//   up = 0: out[n, y, x, :]  = sum over padded positions reflecting onto (y, x)
//   up = 1: out[n, y, x, :]  = the same summed over the 2x2 block (2y..2y+1, 2x..2x+1)      (out is H/2 x W/2)
//   optionally multiplied by ELU'(yprev) given the post-ELU activation of the producing ConvBlock (y > 0 ? 1 : y + 1)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int refl_pre(int q, int L, int* pos, int clamp) {   // padded coordinates (q+1) that reflect / clamp onto q
    int n = 0;
    pos[n++] = q + 1;
    if (q == (clamp ? 0 : 1)) pos[n++] = 0;
    if (q == (clamp ? L - 1 : L - 2)) pos[n++] = L + 1;
    return n;
}
__global__ __launch_bounds__(256) void seg_fold_kernel(const bf16* dvp, int ldv, int c0, bf16* out, int ldo, const bf16* yprev, int ldy,
                                                       int N, int H, int W, int C, int up, int clamp) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3, Ho = H >> up, Wo = W >> up;
    const long total = (long)N * Ho * Wo * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % C8);
        long t = idx / C8;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const long n = t / Ho;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int dy = 0; dy <= up; ++dy)
            for (int dx = 0; dx <= up; ++dx) {
                int py[3], px[3];
                const int ny = refl_pre((y << up) + dy, H, py, clamp), nx = refl_pre((x << up) + dx, W, px, clamp);
                for (int a = 0; a < ny; ++a)
                    for (int b = 0; b < nx; ++b) {
                        const bf16x8 v = ld8(dvp + ((n * (H + 2) + py[a]) * (long)(W + 2) + px[b]) * ldv + c0 + cg * 8);
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
                    }
            }
        const long orow = (n * Ho + y) * (long)Wo + x;
        bf16x8 o;
        if (yprev) {
            const bf16x8 yv = ld8(yprev + orow * ldy + cg * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float yy = bf2f(yv[k]); o[k] = f2bf(yy > 0.f ? acc[k] : acc[k] * (yy + 1.0f)); }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(acc[k]);
        }
        st8(out + orow * ldo + cg * 8, o);
    }
}

// ---------------------------------------------------------------------------------------------------------
// pixel shuffles for the phase-decomposed final seg conv (3x3 over a nearest-x2 up-sampled map == four 2x2-phase 3x3 convs on the
// low-resolution map): in [N][h][w][(py*2+px)*k + o] <-> out [N][2h][2w][k]
// ---------------------------------------------------------------------------------------------------------
__global__ void space_to_depth_kernel(const float* dy, bf16* out, int ldo, int N, int h, int w, int k) {
    const long total = (long)N * h * w * ldo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % ldo);
        long t = idx / ldo;
        float v = 0.f;
        if (c < 4 * k) {
            const int x = (int)(t % w);
            const long t2 = t / w;
            const int y = (int)(t2 % h);
            const long n = t2 / h;
            const int ph = c / k, o = c - ph * k;
            v = dy[((n * 2 * h + 2 * y + (ph >> 1)) * (long)(2 * w) + 2 * x + (ph & 1)) * k + o];
        }
        out[idx] = f2bf(v);
    }
}

// bf16 [N][2h][2w][k] (row stride ldi) -> bf16 [N][h][w][4k]: channel (py*2+px)*k + o of low-res pixel (y, x) = in(2y+py, 2x+px, o); 16-byte pieces
// psum (optional, [gridDim.x][k] fp32): per-block channel sums of the tensor that passes through -- the bias gradient of the conv whose dz
// this is, for free (a separate column reduction re-read the 268 MB full-resolution dz of decoder.7).  Needs 256 % (4 * k/8) == 0: then a
// thread keeps one channel group for the whole grid-stride loop.
__global__ __launch_bounds__(256) void space_to_depth_bf16_kernel(const bf16* in, int ldi, bf16* out, int N, int h, int w, int k, float* psum) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    __shared__ float red[256][9];
    const int k8 = k >> 3;
    const long total = (long)N * h * w * 4 * k8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (long idx = (long)bidx * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % (4 * k8));
        long t = idx / (4 * k8);
        const int x = (int)(t % w);
        t /= w;
        const int y = (int)(t % h);
        const long n = t / h;
        const int ph = c8 / k8, o = (c8 - ph * k8) * 8;
        const bf16x8 v = ld8(in + ((n * 2 * h + 2 * y + (ph >> 1)) * (long)(2 * w) + 2 * x + (ph & 1)) * ldi + o);
        st8(out + idx * 8, v);
        if (psum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
    }
    if (psum) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
        __syncthreads();
        for (int c = threadIdx.x; c < k; c += 256) {                  // channel c: threads t with (t % (4*k8)) % k8 == c / 8 hold its group
            const int g = c >> 3, j = c & 7;
            float s = 0.f;
            for (int t = g; t < 256; t += k8) s += red[t][j];
            psum[(long)bidx * k + c] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// head-gradient gather: fp32 gradient of a head output laid out [N][rows_total][Nout] (per-image stride img_stride, row stride
// lds) -> zero-padded bf16 dz [M = N*rpi][ldz]; optional sigmoid' from the saved fp32 output.
// ---------------------------------------------------------------------------------------------------------
__global__ void head_grad_kernel(const float* dy, const float* y, long rpi, long img_stride, int lds_, int Nout, bf16* dz, int ldz, long M,
                                 int sigmoid) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const long total = M * ldz;
    for (long idx = (long)bidx * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / ldz;
        const int c = (int)(idx - m * ldz);
        float v = 0.f;
        if (c < Nout) {
            const long o = (m / rpi) * img_stride + (m % rpi) * lds_ + c;
            v = dy[o];
            if (sigmoid) { const float s = y[o]; v *= s * (1.f - s); }
        }
        dz[idx] = f2bf(v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static inline int ew_grid(long items) {
    long b = (items + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int hn_stem_fwd(const float* x, const float* w, void* z, void* patches, int N, int H, int W, hipStream_t st) {
    HN_CHECK_ARG(x && w && z && N > 0 && H > 1 && W > 1 && !(H & 1) && !(W & 1));
    const long total = (long)N * (H >> 1) * (((W >> 1) + 1) >> 1);          // one thread per pair of output pixels
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, x, w, (bf16*)z, (bf16*)patches, N, H, W);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_gconv_pack(const float* w, void* wk, void* wd, int C, int flip, hipStream_t st) {
    HN_CHECK_ARG(w && wk && C > 0 && (C & 7) == 0);
    const int G = C >> 3;
    hipLaunchKernelGGL(gconv_pack_kernel, dim3(cdiv((long)G * 576, 256)), dim3(256), 0, st, w, (bf16*)wk, (bf16*)wd, G, flip);
    HN_LAUNCH_CHECK();
}
// forward (and stride-1 dgrad when called on dz with the flipped/transposed pack)
extern "C" int hn_gconv_fwd(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int Hi, int Wi, int C, int stride,
                            hipStream_t st) {
    HN_CHECK_ARG(in && wk && out && (C & 7) == 0 && ((ldi | ldo) & 7) == 0 && (stride == 1 || stride == 2));
    const int G = C >> 3, Ho = stride == 1 ? Hi : Hi >> 1, Wo = stride == 1 ? Wi : Wi >> 1;
    const long total = (long)N * Ho * ((Wo + 3) >> 2) * G;
    HN_CHECK_ARG(total < (1L << 32));
    if (stride == 1)
        hipLaunchKernelGGL(gconv_fwd_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16*)in, ldi, (const bf16*)wk, (bf16*)out,
                           ldo, N, Hi, Wi, Ho, Wo, G);
    else if (G * 1152 <= 24 * 1024)
        hipLaunchKernelGGL(gconv_fwd_s2_kernel<true>, dim3(cdiv(total, 256)), dim3(256), (size_t)G * 1152, st, (const bf16*)in, ldi,
                           (const bf16*)wk, (bf16*)out, ldo, N, Hi, Wi, Ho, Wo, G);
    else
        hipLaunchKernelGGL(gconv_fwd_s2_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16*)in, ldi, (const bf16*)wk,
                           (bf16*)out, ldo, N, Hi, Wi, Ho, Wo, G);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_gconv_dgrad_s2(const void* dz, int ldz, const void* wd, void* dx, int ldx, int N, int Hi, int Wi, int C, hipStream_t st) {
    HN_CHECK_ARG(dz && wd && dx && (C & 7) == 0 && ((ldz | ldx) & 7) == 0);
    const int G = C >> 3;
    HN_CHECK_ARG(!(Hi & 1) && !(Wi & 1));
    const long total = (long)N * (Hi >> 1) * (Wi >> 1) * G;       // one thread per 2x2 input quad and group
    HN_CHECK_ARG(total < (1L << 32));
    if (G * 1152 <= 24 * 1024)
        hipLaunchKernelGGL(gconv_dgrad_s2_kernel<true>, dim3(cdiv(total, 256)), dim3(256), (size_t)G * 1152, st, (const bf16*)dz, ldz,
                           (const bf16*)wd, (bf16*)dx, ldx, N, Hi, Wi, Hi >> 1, Wi >> 1, G);
    else
        hipLaunchKernelGGL(gconv_dgrad_s2_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16*)dz, ldz, (const bf16*)wd,
                           (bf16*)dx, ldx, N, Hi, Wi, Hi >> 1, Wi >> 1, G);
    HN_LAUNCH_CHECK();
}
extern "C" long hn_wgrad_chunks(long pixels, long items) {          // pixel chunks so that chunks*items ~ 256k threads
    long chunks = (262144 + items - 1) / items;
    const long maxc = (pixels + 63) / 64;
    if (chunks > maxc) chunks = maxc;
    if (chunks > 4096) chunks = 4096;
    if (chunks < 1) chunks = 1;
    return chunks;
}
// part: fp32 [hn_wgrad_chunks(N*Ho*Wo, G*9)][C*72]; reduce with hn_rows_reduce(part, dw, 1, chunks, C*72, 1)
extern "C" int hn_gconv_wgrad(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int Hi, int Wi, int C, int stride,
                              hipStream_t st) {
    HN_CHECK_ARG(x && dz && part && (C & 7) == 0 && ((ldx | ldz) & 7) == 0 && (stride == 1 || stride == 2));
    const int G = C >> 3, Ho = stride == 1 ? Hi : Hi >> 1, Wo = stride == 1 ? Wi : Wi >> 1;
    const long pixels = (long)N * Ho * Wo;
    const long chunks = hn_wgrad_chunks(pixels, G * 9);
    const long ppc = (pixels + chunks - 1) / chunks;
    if (ppc >= 32 && pixels < (1L << 32))
        hipLaunchKernelGGL(gconv_wgrad_sub_kernel, dim3((unsigned)(chunks * cdiv(G * 9, 32))), dim3(256), 0, st, (const bf16*)x, ldx,
                           (const bf16*)dz, ldz, part, N, Hi, Wi, Ho, Wo, G, stride, ppc, chunks);
    else
        hipLaunchKernelGGL(gconv_wgrad_kernel, dim3(cdiv(chunks * G * 9, 256)), dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)dz, ldz,
                           part, N, Hi, Wi, Ho, Wo, G, stride, ppc, chunks);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_dw_pack(const float* w, void* wk, void* wkf, int C, hipStream_t st) {
    HN_CHECK_ARG(w && wk && C > 0);
    hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv(9L * C, 256)), dim3(256), 0, st, w, (bf16*)wk, (bf16*)wkf, C);
    HN_LAUNCH_CHECK();
}
static int fill_levels(Levels& L, int N, int nlev, const int* H, const int* W, int row_align = 1) {
    HN_CHECK_ARG(nlev >= 1 && nlev <= HN_MAX_LEVELS && H && W && row_align >= 1);
    L.n = nlev;
    L.row_off[0] = 0;
    for (int l = 0; l < nlev; ++l) {
        HN_CHECK_ARG(H[l] > 0 && W[l] > 0);
        L.H[l] = H[l]; L.W[l] = W[l];
        const long real = (long)N * H[l] * W[l];
        L.row_off[l + 1] = L.row_off[l] + (real + row_align - 1) / row_align * row_align;     // ragged packing: levels start on aligned rows
    }
    return HN_OK;
}
static int dwconv_fwd_launch(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int C, Levels& L, hipStream_t st,
                             int accumulate = 0) {
    HN_CHECK_ARG(in && wk && out && (C & 7) == 0 && ((ldi | ldo) & 7) == 0);
    L.work_off[0] = 0;
    for (int l = 0; l < L.n; ++l) L.work_off[l + 1] = L.work_off[l] + (long)N * L.H[l] * ((L.W[l] + 3) >> 2) * (C >> 3);
    long pad_items = 0;
    for (int l = 0; l < L.n; ++l) pad_items += (L.row_off[l + 1] - L.row_off[l] - (long)N * L.H[l] * L.W[l]) * (C >> 3);
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(cdiv(L.work_off[L.n] + pad_items, 256)), dim3(256), 0, st, (const bf16*)in, ldi, (const bf16*)wk, (bf16*)out,
                       ldo, N, C, L, accumulate);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_dwconv_fwd(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int H, int W, int C, hipStream_t st) {
    Levels L;
    const int rc = fill_levels(L, N, 1, &H, &W);
    return rc != HN_OK ? rc : dwconv_fwd_launch(in, ldi, wk, out, ldo, N, C, L, st);
}
extern "C" int hn_dwconv_fwd_levels(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int C, int nlev, const int* H,
                                    const int* W, int row_align, int accumulate, hipStream_t st) {
    Levels L;
    const int rc = fill_levels(L, N, nlev, H, W, row_align);
    return rc != HN_OK ? rc : dwconv_fwd_launch(in, ldi, wk, out, ldo, N, C, L, st, accumulate);
}
// number of partial rows (= blocks) of hn_dwconv_wgrad; part is fp32 [blocks][C*9], reduce with hn_rows_reduce(part, dw, 1, blocks, C*9, 1)
// partial rows (= blocks) of hn_dwconv_wgrad* for `strips` = sum over levels of N * H * ceil(W / 4) four-pixel strips; part is fp32
// [blocks][C*9], reduce with hn_rows_reduce(part, dw, 1, blocks, C*9, 1)
extern "C" long hn_dwconv_wgrad_blocks(long strips, int C) {
    const int lanes = 256 / (C >> 3);
    long spl = (strips + 2047L * lanes) / (2048L * lanes);       // aim for ~2048 blocks
    if (spl < 2) spl = 2;
    return (strips + spl * lanes - 1) / (spl * lanes);
}
static int dwconv_wgrad_launch(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int C, Levels& L, hipStream_t st) {
    HN_CHECK_ARG(x && dz && part && (C & 7) == 0 && C <= 2048 && ((ldx | ldz) & 7) == 0);
    L.work_off[0] = 0;
    for (int l = 0; l < L.n; ++l) L.work_off[l + 1] = L.work_off[l] + (long)N * L.H[l] * ((L.W[l] + 3) >> 2);
    const long strips = L.work_off[L.n];
    const int lanes = 256 / (C >> 3);
    const long blocks = hn_dwconv_wgrad_blocks(strips, C);
    const int spl = (int)((strips + blocks * lanes - 1) / (blocks * lanes));
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)dz, ldz, part, N, C, spl, L);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_dwconv_wgrad(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int H, int W, int C, hipStream_t st) {
    Levels L;
    const int rc = fill_levels(L, N, 1, &H, &W);
    return rc != HN_OK ? rc : dwconv_wgrad_launch(x, ldx, dz, ldz, part, N, C, L, st);
}
extern "C" int hn_dwconv_wgrad_levels(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int C, int nlev, const int* H,
                                      const int* W, int row_align, hipStream_t st) {
    Levels L;
    const int rc = fill_levels(L, N, nlev, H, W, row_align);
    return rc != HN_OK ? rc : dwconv_wgrad_launch(x, ldx, dz, ldz, part, N, C, L, st);
}

// one-pass backward of the depthwise conv (data gradient + weight-gradient partial rows): blocks / launch
static long dwconv_bwd_strip_blocks(long strips, int C) {
    const int lanes = 256 / (C >> 2);
    const long tb = g_hn_knob[7] > 0 ? g_hn_knob[7] : 768;        // ~768 blocks: three co-resident per CU, the lane reduction amortised (knob 7: sweeps)
    long spl = (strips + (tb - 1) * lanes) / (tb * lanes);
    if (spl < 2) spl = 2;
    return (strips + spl * lanes - 1) / (spl * lanes);
}
extern "C" long hn_dwconv_bwd_blocks(long strips, int C) { return dwconv_bwd_strip_blocks(strips, C); }
// The LDS-tiled form takes maps with enough 8 x 16 tiles to fill the chip and channel counts whose tiles leave room for two workgroups per
// CU; tiles per workgroup so that ~512 workgroups (two per CU) share the launch.  (knob 7 < 0: the strip form always -- tools/ A-B)
static size_t dwconv_bwd_tiled_lds(int C8) {      // (dz halo tile | x tile) in 1 KB DMA chunks (>= the final fold's 8 * C/4 * 37 floats)
    return (size_t)((((DWT_TH + 2) * 18 * C8 + 63) >> 6) + ((DWT_TH * 16 * C8 + 63) >> 6)) * 1024;
}
static bool dwconv_bwd_tiled_plan(int N, int C, const Levels& L, DwTiles& T, int& tpw, long& blocks) {
    T.tile_off[0] = 0;
    for (int l = 0; l < L.n; ++l) {
        T.ty[l] = (L.H[l] + DWT_TH - 1) / DWT_TH;
        T.tx[l] = (L.W[l] + 15) >> 4;
        T.tile_off[l + 1] = T.tile_off[l] + (long)N * T.ty[l] * T.tx[l];
    }
    const long total = T.tile_off[L.n];
    const int C8 = C >> 3;
    const size_t lds = dwconv_bwd_tiled_lds(C8);
    // (measured, tools/bench_dwconv_bwd.py: towers 42.5 vs 56.7 us, P3 map 33.2 vs 41.1; the 256-tile P4 map 18.4 vs 17.1 -- one tile per
    // workgroup on half the slots -- stays with the strip form)
#ifdef HN_NO_DW_TILED      // tools/ab_tree.sh: the strip form always (same-box A-B of the product build)
    return false;
#endif
    if (g_hn_knob[7] < 0 || C8 > 16 || lds > 80 * 1024 || total < 512) return false;
    tpw = (int)((total + 511) / 512);
    blocks = (total + tpw - 1) / tpw;
    return true;
}
/* partial rows of hn_dwconv_bwd_levels for these maps (either form of the kernel) */
extern "C" long hn_dwconv_bwd_blocks_levels(int N, int C, int nlev, const int* H, const int* W) {
    Levels L;
    if (fill_levels(L, N, nlev, H, W) != HN_OK || (C & 7) || C < 8) return -1;
    DwTiles T;
    int tpw = 0;
    long blocks = 0;
    if (dwconv_bwd_tiled_plan(N, C, L, T, tpw, blocks)) return blocks;
    long strips = 0;
    for (int l = 0; l < nlev; ++l) strips += (long)N * H[l] * ((W[l] + 3) >> 2);
    return dwconv_bwd_strip_blocks(strips, C);
}
extern "C" int hn_dwconv_bwd_levels(const void* dz, int ldz, const void* x, int ldx, const void* wf, void* dx, int lddx, float* part, int N, int C,
                                    int nlev, const int* H, const int* W, int row_align, int accumulate, hipStream_t st) {
    Levels L;
    const int rc = fill_levels(L, N, nlev, H, W, row_align);
    if (rc != HN_OK) return rc;
    HN_CHECK_ARG(dz && x && part && (!dx || wf) && (C & 7) == 0 && C >= 8 && C <= 1024 && ((ldz | ldx) & 7) == 0 && (!dx || (lddx & 7) == 0));
    long pad_items = 0;
    if (dx && !accumulate)
        for (int l = 0; l < L.n; ++l) pad_items += (L.row_off[l + 1] - L.row_off[l] - (long)N * L.H[l] * L.W[l]) * (C >> 3);
    DwTiles T;
    int tpw = 0;
    long tblocks = 0;
    if (dwconv_bwd_tiled_plan(N, C, L, T, tpw, tblocks)) {
        const int C8 = C >> 3;
        const size_t lds = dwconv_bwd_tiled_lds(C8);
        if (lds > 64 * 1024) {
            static std::atomic<unsigned long long> optin_t{0};
            if (!lds_optin(optin_t, {(const void*)dwconv_bwd_tiled_kernel<true>, (const void*)dwconv_bwd_tiled_kernel<false>})) return HN_ERR_LAUNCH;
        }
        L.work_off[0] = 0;
        if (dx)
            hipLaunchKernelGGL(dwconv_bwd_tiled_kernel<true>, dim3((unsigned)(tblocks + cdiv(pad_items, 256))), dim3(256), lds, st, (const bf16*)dz,
                               ldz, (const bf16*)x, ldx, (const bf16*)wf, (bf16*)dx, lddx, part, N, C, tpw, (int)tblocks, L, T, accumulate);
        else
            hipLaunchKernelGGL(dwconv_bwd_tiled_kernel<false>, dim3((unsigned)tblocks), dim3(256), lds, st, (const bf16*)dz, ldz, (const bf16*)x, ldx,
                               (const bf16*)wf, (bf16*)dx, lddx, part, N, C, tpw, (int)tblocks, L, T, accumulate);
        HN_LAUNCH_CHECK();
    }
    L.work_off[0] = 0;
    for (int l = 0; l < L.n; ++l) L.work_off[l + 1] = L.work_off[l] + (long)N * L.H[l] * ((L.W[l] + 3) >> 2);
    const long strips = L.work_off[L.n];
    const int lanes = 256 / (C >> 2);
    const long blocks = dwconv_bwd_strip_blocks(strips, C);
    const int spl = (int)((strips + blocks * lanes - 1) / (blocks * lanes));
    const size_t lds = (256 * 37 + 9 * (size_t)C) * sizeof(float);
    if (lds > 64 * 1024) {                                             // > 64 KiB of dynamic LDS: opt-in once per device (C > 727)
        static std::atomic<unsigned long long> optin{0};
        if (!lds_optin(optin, {(const void*)dwconv_bwd_kernel})) return HN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(dwconv_bwd_kernel, dim3((unsigned)(blocks + cdiv(pad_items, 256))), dim3(256), lds, st, (const bf16*)dz, ldz, (const bf16*)x,
                       ldx, (const bf16*)wf, (bf16*)dx, lddx, part, N, C, spl, blocks, L, accumulate);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_maxpool_fwd(const void* in, int ldi, void* out, int ldo, int N, int H, int W, int C, int mode, hipStream_t st) {
    HN_CHECK_ARG(in && out && (C & 7) == 0 && ((ldi | ldo) & 7) == 0 && !(H & 1) && !(W & 1) && (mode == 0 || mode == 1));
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid((long)N * (H >> 1) * (W >> 1) * (C >> 3))), dim3(256), 0, st, (const bf16*)in, ldi,
                       (bf16*)out, ldo, N, H, W, C, mode);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_maxpool_bwd(const void* in, int ldi, const void* dout, int ldd, void* dx, int ldx, const float* wscale, int N, int H,
                              int W, int C, int mode, hipStream_t st) {
    HN_CHECK_ARG(in && dout && dx && (C & 7) == 0 && ((ldi | ldd | ldx) & 7) == 0 && !(H & 1) && !(W & 1));
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, st, (const bf16*)in, ldi,
                       (const bf16*)dout, ldd, (bf16*)dx, ldx, wscale, N, H, W, C, mode);
    HN_LAUNCH_CHECK();
}
/* two-pass form: arg_ws = N*(H/2)*(W/2)*C bytes of scratch (arg-max of every window); same results as hn_maxpool_bwd */
extern "C" int hn_maxpool_bwd2(const void* in, int ldi, const void* dout, int ldd, void* dx, int ldx, const float* wscale, void* arg_ws,
                               int N, int H, int W, int C, int mode, int accumulate, hipStream_t st) {
    HN_CHECK_ARG(in && dout && dx && arg_ws && (C & 7) == 0 && ((ldi | ldd | ldx) & 7) == 0 && !(H & 1) && !(W & 1));
    hipLaunchKernelGGL(maxpool_arg_kernel, dim3(ew_grid((long)N * (H >> 1) * (W >> 1) * (C >> 3))), dim3(256), 0, st, (const bf16*)in, ldi,
                       (unsigned char*)arg_ws, N, H, W, C, mode);
    hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, st, (const unsigned char*)arg_ws,
                       (const bf16*)dout, ldd, (bf16*)dx, ldx, wscale, N, H, W, C, mode, accumulate);
    HN_LAUNCH_CHECK();
}
/* second pass of hn_maxpool_bwd2 alone: the arg-max bytes come from elsewhere (hn_fuse_bwd_arg); H, W = the INPUT resolution of the pool */
extern "C" int hn_maxpool_bwd_from_arg(const void* arg, const void* dout, int ldd, void* dx, int ldx, const float* wscale, int N, int H, int W,
                                       int C, int mode, int accumulate, hipStream_t st) {
    HN_CHECK_ARG(arg && dout && dx && (C & 7) == 0 && ((ldd | ldx) & 7) == 0 && !(H & 1) && !(W & 1));
    hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, st, (const unsigned char*)arg,
                       (const bf16*)dout, ldd, (bf16*)dx, ldx, wscale, N, H, W, C, mode, accumulate);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_up2_fwd(const void* in, int ldi, void* out, int ldo, int N, int H, int W, int C, hipStream_t st) {
    HN_CHECK_ARG(in && out && (C & 7) == 0 && ((ldi | ldo) & 7) == 0);
    hipLaunchKernelGGL(up2_fwd_kernel, dim3(ew_grid((long)N * 4 * H * W * (C >> 3))), dim3(256), 0, st, (const bf16*)in, ldi, (bf16*)out, ldo,
                       N, H, W, C);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_sum2x2(const void* g, int ldg, void* out, int ldo, const float* wscale, int N, int H, int W, int C, int accumulate,
                         hipStream_t st) {
    HN_CHECK_ARG(g && out && (C & 7) == 0 && ((ldg | ldo) & 7) == 0);
    hipLaunchKernelGGL(sum2x2_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, st, (const bf16*)g, ldg, (bf16*)out, ldo, wscale,
                       N, H, W, C, accumulate);
    HN_LAUNCH_CHECK();
}

static int fill_fuse(Fuse& f, const void* const* in, const int* ld, const int* mode, const float* w, void* out, int ldo, int N, int H,
                     int W, int C) {
    HN_CHECK_ARG(in && ld && mode && (C & 7) == 0 && (ldo & 7) == 0);
    for (int i = 0; i < 3; ++i) {
        f.in[i] = (const bf16*)in[i]; f.ld[i] = ld[i]; f.mode[i] = mode[i];
        HN_CHECK_ARG(mode[i] >= 0 && mode[i] <= 3 && (mode[i] == 0 || (in[i] && (ld[i] & 7) == 0)));
        HN_CHECK_ARG(mode[i] != 2 || (!(H & 1) && !(W & 1)));
    }
    f.w = w; f.out = (bf16*)out; f.ldo = ldo; f.N = N; f.H = H; f.W = W; f.C = C;
    f.praw = nullptr; f.nw = 0; f.eps = 0.f; f.wn = nullptr;
    return HN_OK;
}
extern "C" int hn_fuse_fwd(const void* const* in, const int* ld, const int* mode, const float* w, void* out, int ldo, int N, int H, int W,
                           int C, hipStream_t st) {
    Fuse f;
    HN_CHECK_ARG(out);
    const int rc = fill_fuse(f, in, ld, mode, w, out, ldo, N, H, W, C);
    if (rc) return rc;
    hipLaunchKernelGGL(fuse_fwd_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, st, f);
    HN_LAUNCH_CHECK();
}
/* hn_fuse_fwd with the weight normalisation inside: praw = the nw (2 or 3) raw fusion parameters, wn [3] receives relu(p)/(sum relu(p)+eps)
 * (kept for the backward pass) */
extern "C" int hn_fuse_fwd_raw(const void* const* in, const int* ld, const int* mode, const float* praw, int nw, float eps, float* wn,
                               void* out, int ldo, int N, int H, int W, int C, hipStream_t st) {
    Fuse f;
    HN_CHECK_ARG(out && praw && wn && nw >= 1 && nw <= 3);
    const int rc = fill_fuse(f, in, ld, mode, nullptr, out, ldo, N, H, W, C);
    if (rc) return rc;
    f.praw = praw; f.nw = nw; f.eps = eps; f.wn = wn;
    hipLaunchKernelGGL(fuse_fwd_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, st, f);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_fuse_bwd_blocks(int N, int H, int W, int C) {
    long b = ((long)N * H * W * (C >> 3) + 255) / 256;
    if (b > 1024) b = 1024;                  // four workgroups per CU (92 VGPRs: five fit); swept 768 .. uncapped on the step: 1024 +0.6 %, 1280 / 1152 / uncapped +-0
    return (int)(b < 1 ? 1 : b);
}
// pw: fp32 [hn_fuse_bwd_blocks][3]; reduce with hn_rows_reduce(pw, dw, 1, blocks, 3, 1)
extern "C" int hn_fuse_bwd(const void* const* in, const int* ld, const int* mode, const float* w, const void* dout, int ldd, void* g,
                           int ldg, void* const* din, const int* ldin, const int* acc, float* pw, int N, int H, int W, int C,
                           hipStream_t st) {
    FuseBwd q;
    HN_CHECK_ARG(dout && g && pw && din && ldin && ((ldd | ldg) & 7) == 0);
    const int rc = fill_fuse(q.f, in, ld, mode, w, nullptr, 8, N, H, W, C);
    if (rc) return rc;
    q.dout = (const bf16*)dout; q.ldd = ldd; q.g = (bf16*)g; q.ldg = ldg; q.pw = pw;
    for (int i = 0; i < 3; ++i) { q.din[i] = (bf16*)din[i]; q.ldin[i] = ldin[i]; q.acc[i] = acc ? acc[i] : 0; q.arg[i] = nullptr; }
    int n_id = 0, n_up = 0, n_pool = 0, up_dst = 0;
    for (int i = 0; i < 3; ++i) { n_id += mode[i] == 1; n_up += mode[i] == 2; n_pool += mode[i] == 3; up_dst += mode[i] == 2 && q.din[i]; }
    if (n_id == 1 && n_up == 1 && up_dst == 1 && n_pool == 0 && !(H & 1) && !(W & 1))        // a top-down node: every load of a quad in flight at once
        hipLaunchKernelGGL(fuse_bwd_quads_kernel, dim3(hn_fuse_bwd_blocks(N, H, W, C)), dim3(256), 0, st, q);
    else
        hipLaunchKernelGGL(fuse_bwd_kernel, dim3(hn_fuse_bwd_blocks(N, H, W, C)), dim3(256), 0, st, q);
    HN_LAUNCH_CHECK();
}
/* hn_fuse_bwd that also writes, for every mode-3 (max-pooled) input i with arg_out[i] != NULL, the arg-max bytes of its pooling windows
 * ([N][H][W][C] uint8, the layout hn_maxpool_bwd_from_arg reads) */
extern "C" int hn_fuse_bwd_arg(const void* const* in, const int* ld, const int* mode, const float* w, const void* dout, int ldd, void* g,
                               int ldg, void* const* din, const int* ldin, const int* acc, float* pw, void* const* arg_out, int N, int H,
                               int W, int C, hipStream_t st) {
    FuseBwd q;
    HN_CHECK_ARG(dout && g && pw && din && ldin && arg_out && ((ldd | ldg) & 7) == 0);
    const int rc = fill_fuse(q.f, in, ld, mode, w, nullptr, 8, N, H, W, C);
    if (rc) return rc;
    q.dout = (const bf16*)dout; q.ldd = ldd; q.g = (bf16*)g; q.ldg = ldg; q.pw = pw;
    for (int i = 0; i < 3; ++i) {
        q.din[i] = (bf16*)din[i]; q.ldin[i] = ldin[i]; q.acc[i] = acc ? acc[i] : 0;
        q.arg[i] = mode[i] == 3 ? (unsigned char*)arg_out[i] : nullptr;
    }
    int n_id = 0, n_up = 0, n_pool = 0, up_dst = 0;
    for (int i = 0; i < 3; ++i) { n_id += mode[i] == 1; n_up += mode[i] == 2; n_pool += mode[i] == 3; up_dst += mode[i] == 2 && q.din[i]; }
    if (n_id == 1 && n_up == 1 && up_dst == 1 && n_pool == 0 && !(H & 1) && !(W & 1))        // a top-down node: every load of a quad in flight at once
        hipLaunchKernelGGL(fuse_bwd_quads_kernel, dim3(hn_fuse_bwd_blocks(N, H, W, C)), dim3(256), 0, st, q);
    else
        hipLaunchKernelGGL(fuse_bwd_kernel, dim3(hn_fuse_bwd_blocks(N, H, W, C)), dim3(256), 0, st, q);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_seg_fold(const void* dvp, int ldv, int c0, void* out, int ldo, const void* yprev, int ldy, int N, int H, int W, int C,
                           int up, hipStream_t st) {
    HN_CHECK_ARG(dvp && out && (C & 7) == 0 && (c0 & 7) == 0 && ((ldv | ldo) & 7) == 0 && H >= 4 && W >= 4 && up >= 0 && up <= 2);
    HN_CHECK_ARG(!yprev || (ldy & 7) == 0);
    const int clamp = up == 2;                                  // up = 2: replicate-padding fold, no up-sampling
    if (clamp) up = 0;
    hipLaunchKernelGGL(seg_fold_kernel, dim3(ew_grid((long)N * (H >> up) * (W >> up) * (C >> 3))), dim3(256), 0, st, (const bf16*)dvp, ldv, c0,
                       (bf16*)out, ldo, (const bf16*)yprev, ldy, N, H, W, C, up, clamp);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_space_to_depth(const float* dy, void* out, int ldo, int N, int h, int w, int k, hipStream_t st) {
    HN_CHECK_ARG(dy && out && N > 0 && h > 0 && w > 0 && k > 0 && ldo >= 4 * k && (ldo & 7) == 0);
    hipLaunchKernelGGL(space_to_depth_kernel, dim3(ew_grid((long)N * h * w * ldo)), dim3(256), 0, st, dy, (bf16*)out, ldo, N, h, w, k);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_space_to_depth_blocks(int N, int h, int w, int k) { return ew_grid((long)N * h * w * 4 * (k >> 3)); }
/* psum (optional): fp32 [hn_space_to_depth_blocks(N,h,w,k)][k] per-block channel sums of the tensor (needs 256 % (k/2) == 0: k = 64 ... 512) */
extern "C" int hn_space_to_depth_bf16(const void* in, int ldi, void* out, int N, int h, int w, int k, float* psum, hipStream_t st) {
    HN_CHECK_ARG(in && out && N > 0 && h > 0 && w > 0 && k > 0 && (k & 7) == 0 && (ldi & 7) == 0);
    HN_CHECK_ARG(!psum || 256 % (4 * (k >> 3)) == 0);
    hipLaunchKernelGGL(space_to_depth_bf16_kernel, dim3(hn_space_to_depth_blocks(N, h, w, k)), dim3(256), 0, st, (const bf16*)in, ldi,
                       (bf16*)out, N, h, w, k, psum);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_head_grad(const float* dy, const float* y, long rpi, long img_stride, int lds_, int Nout, void* dz, int ldz, long M,
                            int sigmoid, hipStream_t st) {
    HN_CHECK_ARG(dy && dz && rpi > 0 && Nout > 0 && ldz >= Nout && (ldz & 7) == 0 && M > 0 && (!sigmoid || y));
    hipLaunchKernelGGL(head_grad_kernel, dim3(ew_grid(M * ldz)), dim3(256), 0, st, dy, y, rpi, img_stride, lds_, Nout, (bf16*)dz, ldz, M,
                       sigmoid);
    HN_LAUNCH_CHECK();
}

// the same for ALL pyramid levels of a level-packed head output in one launch: dy / y rows of image n are the levels' pixels one after the
// other ([N][sum H W][lds]); dz is the packed operand ([level][n][pixel], every level on an aligned row; the alignment rows are the caller's)
__global__ __launch_bounds__(256) void head_grad_levels_kernel(const float* dy, const float* y, long img_stride, int lds_, int Nout, bf16* dz,
                                                               int ldz, int N, int sigmoid, const Levels L) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const long total = L.work_off[L.n];                    // real rows x (ldz / 8): a thread writes eight channels of a row (16 bytes)
    const unsigned l8 = (unsigned)(ldz >> 3);
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int lv = 0;
        unsigned pix0 = 0;                                 // pixels of an image before this level
        while (lv + 1 < L.n && idx >= L.work_off[lv + 1]) { pix0 += (unsigned)(L.H[lv] * L.W[lv]); ++lv; }
        const unsigned li = (unsigned)(idx - L.work_off[lv]);          // (< 2^32: host check)
        const unsigned r = li / l8, c0 = (li - r * l8) * 8;
        const unsigned hw = (unsigned)(L.H[lv] * L.W[lv]);
        const unsigned n = r / hw, p = r - n * hw;
        const long o = (long)n * img_stride + (long)(pix0 + p) * lds_ + c0;
        bf16x8 v8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = 0.f;
            if ((int)c0 + k < Nout) {
                v = dy[o + k];
                if (sigmoid) { const float sg = y[o + k]; v *= sg * (1.f - sg); }
            }
            v8[k] = f2bf(v);
        }
        st8(dz + (L.row_off[lv] + r) * ldz + c0, v8);
    }
}
/* hn_head_grad for every level of a level-packed head at once (head_detect/detection.py:36-60: the towers run on the five pyramid levels
 * with shared weights; their outputs are concatenated per image).  dy / y: fp32 [N][sum_l H_l W_l][lds]; dz: bf16, level l's N H_l W_l
 * rows start at the row_align-aligned offset of the packing (hn_dwconv_fwd_levels); alignment rows are not written. */
extern "C" int hn_head_grad_levels(const float* dy, const float* y, long img_stride, int lds_, int Nout, void* dz, int ldz, int N, int nlev,
                                   const int* H, const int* W, int row_align, int sigmoid, hipStream_t st) {
    HN_CHECK_ARG(dy && dz && Nout > 0 && ldz >= Nout && (ldz & 7) == 0 && N > 0 && (!sigmoid || y));
    Levels L;
    const int rc = fill_levels(L, N, nlev, H, W, row_align);
    if (rc != HN_OK) return rc;
    L.work_off[0] = 0;
    for (int l = 0; l < L.n; ++l) L.work_off[l + 1] = L.work_off[l] + (long)N * L.H[l] * L.W[l] * (ldz >> 3);
    HN_CHECK_ARG(L.work_off[L.n] < (1L << 32) && (reinterpret_cast<uintptr_t>(dz) & 15) == 0);
    hipLaunchKernelGGL(head_grad_levels_kernel, dim3(ew_grid(L.work_off[L.n])), dim3(256), 0, st, dy, y, img_stride, lds_, Nout, (bf16*)dz, ldz, N,
                       sigmoid, L);
    HN_LAUNCH_CHECK();
}

// pyramid levels [N][H_l][W_l][C] (row strides ld[l]) -> one level-packed tensor, every level on an aligned row: one launch for all levels
struct PackSrc { const bf16* p[HN_MAX_LEVELS]; int ld[HN_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void pack_levels_kernel(const PackSrc src, bf16* dst, int ldd, int C8, const Levels L) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const long total = L.work_off[L.n];                    // real rows x C / 8
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int lv = 0;
        while (lv + 1 < L.n && idx >= L.work_off[lv + 1]) ++lv;
        const unsigned li = (unsigned)(idx - L.work_off[lv]);
        const unsigned r = li / (unsigned)C8, c = (li - r * (unsigned)C8) * 8;
        st8(dst + (L.row_off[lv] + r) * ldd + c, ld8(src.p[lv] + (long)r * src.ld[lv] + c));
    }
}
/* Stacks the pyramid levels a shared-weight head runs on (head_detect/detection.py:36-60 loops over them) into the level-packed operand
 * of the *_levels entry points: src[l] bf16 [N][H_l][W_l][C] with row stride ld[l]; dst rows of level l start at the row_align-aligned
 * offset; alignment rows are not written. */
extern "C" int hn_pack_levels(const void* const* src, const int* ld, void* dst, int ldd, int N, int C, int nlev, const int* H, const int* W,
                              int row_align, hipStream_t st) {
    HN_CHECK_ARG(src && ld && dst && (C & 7) == 0 && (ldd & 7) == 0 && N > 0);
    Levels L;
    const int rc = fill_levels(L, N, nlev, H, W, row_align);
    if (rc != HN_OK) return rc;
    PackSrc ps;
    L.work_off[0] = 0;
    for (int l = 0; l < L.n; ++l) {
        HN_CHECK_ARG(src[l] && (ld[l] & 7) == 0);
        ps.p[l] = (const bf16*)src[l]; ps.ld[l] = ld[l];
        L.work_off[l + 1] = L.work_off[l] + (long)N * L.H[l] * L.W[l] * (C >> 3);
    }
    HN_CHECK_ARG(L.work_off[L.n] < (1L << 32));
    hipLaunchKernelGGL(pack_levels_kernel, dim3(ew_grid(L.work_off[L.n])), dim3(256), 0, st, ps, (bf16*)dst, ldd, C >> 3, L);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_fuse_dweights(const float* pw, int blocks, const float* praw, int nw, float eps, float* dp, hipStream_t st) {
    HN_CHECK_ARG(pw && praw && dp && blocks > 0 && nw >= 1 && nw <= 3);
    hipLaunchKernelGGL(fuse_dweights_kernel, dim3(1), dim3(256), 0, st, pw, blocks, praw, nw, eps, dp);
    HN_LAUNCH_CHECK();
}

