// Implicit-GEMM MFMA kernels (bf16 in, fp32 accumulate) for every dense contraction on the HydraNet hot path:
//   * hn_conv_gemm_nt : out[pixel][cout] = sum_k X[pixel][k] * Wp[cout][k]      (fwd and dgrad)
//   * hn_conv_gemm_tn : dW[cout][k]      = sum_pixel dZ[pixel][cout] * X[pixel][k] (wgrad, split over pixels)
// X is never materialised as an im2col matrix: the tile loader gathers NHWC pixel rows on the fly
// (1x1, 1x1 stride 2, 3x3 reflect-pad with nearest-x2 upsample + channel concat folded in, 3x3 full correlation
// for dgrad).  LDS tiles are [row][64 k] bf16 with the (row&7)<<4 XOR swizzle (conflict-free ds_read_b128 for the
// 16x16x32 operand maps); wgrad stages pixel-major tiles and reads fragments with ds_read_b64_tr_b16.
// Reference ops covered: nn.Conv2d 1x1 (net/anynet.py:29-33,52-60; net/bifpn.py:58-102; net/common.py:95;
// head_lane/lanedetect.py:45-64) and the segmentation decoder's ReflectionPad2d(1)+Conv2d(3)+upsample+cat
// (head_seg/segmentation.py:32-48,84-105).
#include "hn_common.h"

struct XSrc {
    const bf16* x0;
    const bf16* x1;
    int mode;      // 0 plain rows, 1 1x1 stride-2 gather, 2 3x3 reflect (+up2 of x0, +concat x1), 3 3x3 full corr. (zero fill)
    int H, W;      // output grid (row m -> n, oy, ox); unused for mode 0
    int Hi, Wi;    // full-resolution input grid
    int C0, C1;    // channels taken from x0 / x1
    int ld0, ld1;  // row strides (elements)
    int up;        // mode 2: x0 lives at (Hi>>1, Wi>>1)
    long M;        // number of output rows
};

__device__ __forceinline__ void decomp_row(const XSrc& s, long m, int& n, int& oy, int& ox) {
    if (s.mode == 0) { n = 0; oy = 0; ox = 0; return; }
    const int hw = s.H * s.W;
    n = (int)(m / hw);
    const int r = (int)(m - (long)n * hw);
    oy = r / s.W;
    ox = r - oy * s.W;
}

// one 16-byte piece (8 channels starting at c) of the gathered activation row (n, oy, ox) for filter tap `tap`
__device__ __forceinline__ bf16x8 load_x_piece(const XSrc& s, long m, int n, int oy, int ox, int tap, int c) {
    if (m >= s.M || c >= s.C0 + s.C1) return zero8();
    if (s.mode == 0) return ld8(s.x0 + m * s.ld0 + c);
    if (s.mode == 1) return ld8(s.x0 + (((long)n * s.Hi + 2 * oy) * s.Wi + 2 * ox) * s.ld0 + c);
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    if (s.mode == 2) {
        int iy = oy + ky - 1, ix = ox + kx - 1;
        iy = iy < 0 ? -iy : (iy >= s.Hi ? 2 * s.Hi - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= s.Wi ? 2 * s.Wi - 2 - ix : ix);
        if (c < s.C0) {
            const int hh = s.Hi >> s.up, ww = s.Wi >> s.up;
            return ld8(s.x0 + (((long)n * hh + (iy >> s.up)) * ww + (ix >> s.up)) * s.ld0 + c);
        }
        return ld8(s.x1 + (((long)n * s.Hi + iy) * s.Wi + ix) * s.ld1 + (c - s.C0));
    }
    const int iy = oy - ky, ix = ox - kx;                                    // mode 3
    if (iy < 0 || iy >= s.Hi || ix < 0 || ix >= s.Wi) return zero8();
    return ld8(s.x0 + (((long)n * s.Hi + iy) * s.Wi + ix) * s.ld0 + c);
}

__device__ __forceinline__ int swz(int row, int piece) { return row * 128 + ((piece ^ (row & 7)) << 4); }

struct GemmNT {
    XSrc x;
    const bf16* w;   // [Nout][taps*KP]
    int Nout, KP, taps;
    const float* bias;
    int act;
    void* out;
    int ldc;
    float* psum;     // [gridDim.x * WGP][Nout] or null
    float* psq;
    long rpi;        // rows per image for the per-image output mapping below (0 = plain pix*ldc)
    long img_stride; // out offset(pix) = (pix / rpi) * img_stride + (pix % rpi) * ldc  (det-head level concat)
};

template <int BC, int BP, int WGC, int WGP, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmNT p) {
    constexpr int WC = BC / WGC, WP = BP / WGP, TC = WC / 16, TP = WP / 16;
    constexpr int XR = BP / 32, WR = (BC + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sX = smem + BC * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WGP, wp = wave % WGP;
    const int c_blk = blockIdx.y * BC;
    const long p_blk = (long)blockIdx.x * BP;
    const int pj = tid & 7, r0 = tid >> 3, half = pj >> 2, sub = (pj & 3) * 8;
    const int kc = p.KP >> 5, Q = p.taps * kc, S = (Q + 1) >> 1;
    const int Ktot = p.taps * p.KP;

    int xn[XR], xy[XR], xx[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) decomp_row(p.x, p_blk + r0 + 32 * i, xn[i], xy[i], xx[i]);

    int tap = half / kc, cidx = half - tap * kc;   // chunk q = 2*stage + half -> (tap, cidx)
    bf16x8 xr[XR], wr[WR];

    auto fetch = [&](int stage) {
        const int q = 2 * stage + half;
        const bool qv = q < Q;
        const int c = cidx * 32 + sub;
#pragma unroll
        for (int i = 0; i < XR; ++i)
            xr[i] = qv ? load_x_piece(p.x, p_blk + r0 + 32 * i, xn[i], xy[i], xx[i], tap, c) : zero8();
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int row = r0 + 32 * i;
            const int co = c_blk + row;
            wr[i] = (qv && row < BC && co < p.Nout) ? ld8(p.w + (long)co * Ktot + q * 32 + sub) : zero8();
        }
        cidx += 2;
        while (cidx >= kc) { cidx -= kc; ++tap; }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<bf16x8*>(sX + swz(r0 + 32 * i, pj)) = xr[i];
#pragma unroll
        for (int i = 0; i < WR; ++i)
            if (r0 + 32 * i < BC) *reinterpret_cast<bf16x8*>(sW + swz(r0 + 32 * i, pj)) = wr[i];
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash();
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) fetch(s + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TC], b[TP];
            const int piece = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < TC; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sW + swz(wc * WC + i * 16 + (lane & 15), piece));
#pragma unroll
            for (int j = 0; j < TP; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sX + swz(wp * WP + j * 16 + (lane & 15), piece));
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (s + 1 < S) {
            stash();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, activation, store (4 consecutive couts per lane), optional BN partial statistics
    const bool want_stats = p.psum != nullptr;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
        float bsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bsv[r] = (p.bias && co0 + r < p.Nout) ? p.bias[co0 + r] : 0.f;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const long pix = p_blk + wp * WP + j * 16 + (lane & 15);
            const bool pv = pix < p.x.M;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] + bsv[r];
                if (want_stats) {
                    const float q = OUT_F32 ? v[r] : bfround(v[r]);
                    if (pv) { s1[r] += q; s2[r] += q * q; }
                }
                v[r] = act_fwd(v[r], p.act);
            }
            if (pv) {
                const long orow = p.rpi ? (pix / p.rpi) * p.img_stride + (pix % p.rpi) * p.ldc : pix * p.ldc;
                if (OUT_F32) {
                    float* o = reinterpret_cast<float*>(p.out) + orow + co0;
                    if (co0 + 3 < p.Nout && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
                        *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (co0 + r < p.Nout) o[r] = v[r];
                    }
                } else {
                    bf16* o = reinterpret_cast<bf16*>(p.out) + orow + co0;
                    if (co0 + 3 < p.Nout && (reinterpret_cast<uintptr_t>(o) & 7) == 0) {
                        bf16x4 t = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                        *reinterpret_cast<bf16x4*>(o) = t;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (co0 + r < p.Nout) o[r] = f2bf(v[r]);
                    }
                }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[r] += __shfl_xor(s1[r], o);
                    s2[r] += __shfl_xor(s2[r], o);
                }
            }
            if ((lane & 15) == 0) {
                const long prow = (long)blockIdx.x * WGP + wp;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < p.Nout) {
                        p.psum[prow * p.Nout + co0 + r] = s1[r];
                        p.psq[prow * p.Nout + co0 + r] = s2[r];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad: part[split][cout][tap*KP + ci] = sum over this split's pixel rows of dZ[pixel][cout] * X[pixel(tap)][ci]
// ---------------------------------------------------------------------------------------------------------
struct GemmTN {
    XSrc x;
    const bf16* dz;   // [M][Nout] (row stride ldz)
    int ldz, Nout, KP, taps;
    float* part;      // [splits][Nout][taps*KP]
    long rows_per_split;   // multiple of 64
};

template <int BC, int BN, int WGC, int WGN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTN p) {
    constexpr int WC = BC / WGC, WN = BN / WGN, TC = WC / 16, TN = WN / 16;
    constexpr int PZ = BC * 2 + 32, PX = BN * 2 + 32;            // LDS row pitches (bytes): cols/2 + 8 dwords
    constexpr int ZPR = BC / 8, XPR = BN / 8;                     // 16-byte pieces per row
    constexpr int ZL = (64 * ZPR + 255) / 256, XL = (64 * XPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) char sZ[64 * PZ];
    __shared__ __attribute__((aligned(16))) char sXm[64 * PX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WGN, wn = wave % WGN;
    const int ntile = (p.KP + BN - 1) / BN;
    const int tap = blockIdx.x / ntile;
    const int ci_blk = (blockIdx.x - tap * ntile) * BN;
    const int c_blk = blockIdx.y * BC;
    const long m_begin = (long)blockIdx.z * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.x.M) m_end = p.x.M;

    f32x4 acc[TC][TN];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 zr[ZL], xr[XL];
    auto fetch = [&](long m0) {
#pragma unroll
        for (int i = 0; i < ZL; ++i) {
            const int e = tid + 256 * i;
            const int row = e / ZPR, cp = e - row * ZPR;
            const long m = m0 + row;
            const int co = c_blk + cp * 8;
            // dZ rows are zero padded up to ldz (>= Nout rounded up to 8), so a piece that starts below Nout is readable
            zr[i] = (row < 64 && m < m_end && co < p.Nout) ? ld8(p.dz + m * p.ldz + co) : zero8();
        }
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int e = tid + 256 * i;
            const int row = e / XPR, cp = e - row * XPR;
            const long m = m0 + row;
            int n, oy, ox;
            decomp_row(p.x, m, n, oy, ox);
            xr[i] = (row < 64 && m < m_end) ? load_x_piece(p.x, m, n, oy, ox, tap, ci_blk + cp * 8) : zero8();
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < ZL; ++i) {
            const int e = tid + 256 * i;
            const int row = e / ZPR, cp = e - row * ZPR;
            if (row < 64) *reinterpret_cast<bf16x8*>(sZ + row * PZ + cp * 16) = zr[i];
        }
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int e = tid + 256 * i;
            const int row = e / XPR, cp = e - row * XPR;
            if (row < 64) *reinterpret_cast<bf16x8*>(sXm + row * PX + cp * 16) = xr[i];
        }
    };

    // transposed-read lane addressing: group g = lane>>4 owns k rows {s*16 + g*4 + q}; lane 4q+pp supplies row q, cols 4pp..4pp+3
    const int g = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
    typedef __bf16 trv4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
    typedef __attribute__((address_space(3))) trv4* lds_b4;

    if (m_begin < m_end) {
        fetch(m_begin);
        stash();
        __syncthreads();
        for (long m0 = m_begin; m0 < m_end; m0 += 64) {
            const bool more = m0 + 64 < m_end;
            if (more) fetch(m0 + 64);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[TC], b[TN];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int col = wc * WC + i * 16 + pp * 4;
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + (ks * 32 + g * 4 + q) * PZ + col * 2));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + (ks * 32 + 16 + g * 4 + q) * PZ + col * 2));
                    a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wn * WN + j * 16 + pp * 4;
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sXm + (ks * 32 + g * 4 + q) * PX + col * 2));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sXm + (ks * 32 + 16 + g * 4 + q) * PX + col * 2));
                    b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                stash();
                __syncthreads();
            }
        }
    }
    const int Ktot = p.taps * p.KP;
    float* part = p.part + (long)blockIdx.z * p.Nout * Ktot;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci_blk + wn * WN + j * 16 + (lane & 15);
            if (ci >= p.KP) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c_blk + wc * WC + i * 16 + (lane >> 4) * 4 + r;
                if (co < p.Nout) part[(long)co * Ktot + tap * p.KP + ci] = acc[i][j][r];
            }
        }
}

// dW[co][ci][tap] (PyTorch [Cout][Cin][kh][kw] order) = sum_split part[split][co][tap*KP + ci].
// block = 32 consecutive partial columns x 16 split lanes: coalesced rows, LDS tree over the lanes.
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(const float* part, float* dw, int splits, int Nout, int Cin, int KP, int taps) {
    __shared__ float red[16][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long Ktot = (long)taps * KP;
    const long cols = (long)Nout * Ktot;
    const long col = (long)blockIdx.x * 32 + tx;
    float s = 0.f;
    if (col < cols)
        for (int k = ty; k < splits; k += 16) s += part[(long)k * cols + col];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && col < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        const int co = (int)(col / Ktot);
        const int r = (int)(col - (long)co * Ktot);
        const int tap = r / KP, ci = r - tap * KP;
        if (ci < Cin) dw[((long)co * Cin + ci) * taps + tap] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// weight packing: fp32 master weights [Cout][Cin][taps] -> bf16 forward operand Wp[Cout][taps][KP(Cin)] and
// dgrad operand Wt[Cin][taps][KP(Cout)]   (KP = channel count rounded up to 32, zero filled)
// ---------------------------------------------------------------------------------------------------------
__global__ void pack_w_kernel(const float* w, bf16* wp, bf16* wt, int Cout, int Cin, int taps, int KPi, int KPo) {
    const long nf = (long)Cout * taps * KPi;
    const long nt = wt ? (long)Cin * taps * KPo : 0;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nf) {
        const int k = (int)(idx % KPi);
        const long t = idx / KPi;
        const int tap = (int)(t % taps);
        const int co = (int)(t / taps);
        wp[idx] = f2bf(k < Cin ? w[((long)co * Cin + k) * taps + tap] : 0.f);
    } else if (idx < nf + nt) {
        const long j = idx - nf;
        const int k = (int)(j % KPo);
        const long t = j / KPo;
        const int tap = (int)(t % taps);
        const int ci = (int)(t / taps);
        wt[j] = f2bf(k < Cout ? w[((long)k * Cin + ci) * taps + tap] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static XSrc make_xsrc(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                      int up, long M) {
    XSrc s;
    s.x0 = (const bf16*)x0;
    s.x1 = (const bf16*)x1;
    s.mode = mode;
    s.H = H; s.W = W;
    s.Hi = H; s.Wi = W;
    if (mode == 1) { s.Hi = 2 * H; s.Wi = 2 * W; }
    if (mode == 3) { s.Hi = H - 2; s.Wi = W - 2; }
    s.C0 = C0; s.C1 = C1; s.ld0 = ld0; s.ld1 = ld1; s.up = up;
    s.M = M;
    (void)n_img;
    return s;
}

template <int BC, int BP, int WGC, int WGP>
static int launch_nt(const GemmNT& p, int out_f32, hipStream_t st) {
    dim3 grid(cdiv(p.x.M, BP), cdiv(p.Nout, BC));
    const size_t lds = (size_t)(BC + BP) * 128;
    if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, false>), grid, dim3(256), lds, st, p);
    HN_LAUNCH_CHECK();
}

static int pick_bc(int Nout) {
    if (Nout <= 16) return 16;
    if (Nout <= 32) return 32;
    if (Nout <= 64) return 64;
    // prefer the tile with the least padding; ties go to the larger tile
    int best = 128, pad = cdiv(Nout, 128) * 128;
    const int p64 = cdiv(Nout, 64) * 64;
    if (p64 < pad) { best = 64; pad = p64; }
    return best;
}

extern "C" int hn_nt_stat_rows(long M, int Nout) {
    const int bc = pick_bc(Nout);
    if (bc == 16) return cdiv(M, 256) * 4;
    if (bc == 32) return cdiv(M, 128) * 4;
    return cdiv(M, 128) * 2;
}

extern "C" int hn_conv_gemm_nt(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                               int up, long M, const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out,
                               int out_f32, int ldc, long rpi, long img_stride, float* psum, float* psq, hipStream_t st) {
    HN_CHECK_ARG(x0 && w && out && M > 0 && Nout > 0 && KP > 0 && (KP & 31) == 0 && taps >= 1 && taps <= 9);
    HN_CHECK_ARG((C0 & 7) == 0 && (C1 & 7) == 0 && (ld0 & 7) == 0 && (C1 == 0 || (x1 && (ld1 & 7) == 0)));
    HN_CHECK_ARG(C0 + C1 <= KP && mode >= 0 && mode <= 3);
    HN_CHECK_ARG(mode == 0 || (long)n_img * H * W == M);
    HN_CHECK_ARG(mode < 2 ? taps == 1 : taps == 9);
    HN_CHECK_ARG(mode != 2 || (H >= 2 && W >= 2));
    GemmNT p;
    p.x = make_xsrc(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M);
    p.w = (const bf16*)w; p.Nout = Nout; p.KP = KP; p.taps = taps;
    p.bias = bias; p.act = act; p.out = out; p.ldc = ldc; p.psum = psum; p.psq = psq;
    p.rpi = rpi; p.img_stride = img_stride;
    switch (pick_bc(Nout)) {
        case 16: return launch_nt<16, 256, 1, 4>(p, out_f32, st);
        case 32: return launch_nt<32, 128, 1, 4>(p, out_f32, st);
        case 64: return launch_nt<64, 128, 2, 2>(p, out_f32, st);
        default: return launch_nt<128, 128, 2, 2>(p, out_f32, st);
    }
}

template <int BC, int BN, int WGC, int WGN>
static int launch_tn(const GemmTN& p, int splits, hipStream_t st) {
    dim3 grid(cdiv(p.KP, BN) * p.taps, cdiv(p.Nout, BC), splits);
    hipLaunchKernelGGL((gemm_tn_kernel<BC, BN, WGC, WGN>), grid, dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

static void tn_tiles(int Nout, int KP, int& bc, int& bn) {
    bc = Nout <= 16 ? 16 : (Nout <= 32 ? 32 : (Nout <= 64 ? 64 : 128));
    bn = KP <= 32 ? 32 : (KP <= 64 ? 64 : 128);
    if (bc == 16 && bn < 64) bn = 64;                       // 4 waves need >= 16 columns each
}

// plan the pixel split for wgrad: returns splits, rows per split (multiple of 64) and the fp32 workspace size in bytes
extern "C" int hn_wgrad_plan(long M, int Nout, int KP, int taps, int* splits, long* rows_per_split, long* ws_bytes) {
    HN_CHECK_ARG(M > 0 && Nout > 0 && KP > 0 && taps > 0 && splits && rows_per_split && ws_bytes);
    int bc, bn;
    tn_tiles(Nout, KP, bc, bn);
    const long tiles = (long)cdiv(Nout, bc) * cdiv(KP, bn) * taps;
    long want = (1024 + tiles - 1) / tiles;                 // ~4 workgroups per CU in total
    const long max_splits = (M + 255) / 256;                // at least 256 rows per split
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    long rps = ((M + want - 1) / want + 63) / 64 * 64;
    *splits = (int)((M + rps - 1) / rps);
    *rows_per_split = rps;
    *ws_bytes = (long)(*splits) * Nout * taps * KP * 4;
    return HN_OK;
}

extern "C" int hn_conv_gemm_tn(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                               int up, long M, const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw,
                               hipStream_t st) {
    HN_CHECK_ARG(x0 && dz && workspace && dw && M > 0 && (KP & 31) == 0 && (ldz & 7) == 0 && ldz >= ((Nout + 7) & ~7));
    HN_CHECK_ARG((C0 & 7) == 0 && (C1 & 7) == 0 && (ld0 & 7) == 0 && mode >= 0 && mode <= 2);
    HN_CHECK_ARG(mode == 0 || (long)n_img * H * W == M);
    int splits; long rps, wsb;
    hn_wgrad_plan(M, Nout, KP, taps, &splits, &rps, &wsb);
    GemmTN p;
    p.x = make_xsrc(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M);
    p.dz = (const bf16*)dz; p.ldz = ldz; p.Nout = Nout; p.KP = KP; p.taps = taps;
    p.part = workspace; p.rows_per_split = rps;
    int bc, bn, rc;
    tn_tiles(Nout, KP, bc, bn);
#define TN_CASE(BC_, BN_, A_, B_) if (bc == BC_ && bn == BN_) rc = launch_tn<BC_, BN_, A_, B_>(p, splits, st); else
    TN_CASE(128, 128, 2, 2) TN_CASE(128, 64, 2, 2) TN_CASE(128, 32, 4, 1)
    TN_CASE(64, 128, 2, 2) TN_CASE(64, 64, 2, 2) TN_CASE(64, 32, 4, 1)
    TN_CASE(32, 128, 1, 4) TN_CASE(32, 64, 1, 4) TN_CASE(32, 32, 2, 2)
    TN_CASE(16, 128, 1, 4) TN_CASE(16, 64, 1, 4)
    rc = HN_ERR_UNSUPPORTED;
#undef TN_CASE
    if (rc != HN_OK) return rc;
    const long cols = (long)Nout * taps * KP;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(cols, 32)), dim3(512), 0, st, workspace, dw, splits, Nout, C0 + C1, KP, taps);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_pack_weight(const float* w, void* wp, void* wt, int Cout, int Cin, int taps, hipStream_t st) {
    HN_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && taps > 0);
    const int KPi = (Cin + 31) / 32 * 32, KPo = (Cout + 31) / 32 * 32;
    const long total = (long)Cout * taps * KPi + (wt ? (long)Cin * taps * KPo : 0);
    hipLaunchKernelGGL(pack_w_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, (bf16*)wp, (bf16*)wt, Cout, Cin, taps, KPi, KPo);
    HN_LAUNCH_CHECK();
}
