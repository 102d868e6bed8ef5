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
#include <climits>

// Ablation bits / stamps of the tools/ scripts (hn_debug_knob 9 / 14 / 15): compiled into the kernels only with -DHN_TUNING
// (HN_TUNING=1 builds libhydranet_hip_tuning.so: multitask_hydranet_amd/_lib.py); the shipped library's kernels carry none of it.
#ifdef HN_TUNING
#define HN_DBG(p) ((p).dbg)
#else
#define HN_DBG(p) 0
#endif

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
    int clamp;     // mode 2: replicate (clamp) padding instead of reflection (API mode 4)
    int diag;      // API mode 5 (grouped conv, group width 8): zero "same" padding, and cout tile t (64 couts = 8 groups) contracts only
                   // over input channels [64t, 64t+64) with block-diagonal packed weights [C][9][64]
};

// padded-border source index of a 3x3 tap: ReflectionPad2d(1) or replicate padding
__device__ __forceinline__ int border_idx(int v, int L, int clamp) {
    if (clamp == 2) return (v < 0 || v >= L) ? -1 : v;              // zero padding: outside = no source
    if (clamp) return v < 0 ? 0 : (v >= L ? L - 1 : v);
    return v < 0 ? -v : (v >= L ? 2 * L - 2 - v : v);
}

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

// 16 zero bytes: the source of every out-of-range / padded piece of an LDS-DMA (global_load_lds cannot zero-fill)
__device__ __attribute__((aligned(16))) bf16 g_zero_piece[8];

// address of the 16-byte piece (8 channels starting at c) of the gathered activation row, or the zero piece
__device__ __forceinline__ const bf16* x_piece_ptr(const XSrc& s, long m, int n, int oy, int ox, int tap, int c) {
    if (m >= s.M || c >= s.C0 + s.C1) return g_zero_piece;
    if (s.mode == 0) return s.x0 + m * s.ld0 + c;
    if (s.mode == 1) return s.x0 + (((long)n * s.Hi + 2 * oy) * s.Wi + 2 * ox) * s.ld0 + c;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    if (s.mode == 2) {
        int iy = oy + ky - 1, ix = ox + kx - 1;
        iy = iy < 0 ? -iy : (iy >= s.Hi ? 2 * s.Hi - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= s.Wi ? 2 * s.Wi - 2 - ix : ix);
        if (c < s.C0) {
            const int hh = s.Hi >> s.up, ww = s.Wi >> s.up;
            return s.x0 + (((long)n * hh + (iy >> s.up)) * ww + (ix >> s.up)) * s.ld0 + c;
        }
        return s.x1 + (((long)n * s.Hi + iy) * s.Wi + ix) * s.ld1 + (c - s.C0);
    }
    const int iy = oy - ky, ix = ox - kx;                                    // mode 3
    if (iy < 0 || iy >= s.Hi || ix < 0 || ix >= s.Wi) return g_zero_piece;
    return s.x0 + (((long)n * s.Hi + iy) * s.Wi + ix) * s.ld0 + c;
}

// element offset (channel 0) into x0 (which = 0) or x1 (which = 1) of output row (n, oy, ox) under filter tap `tap`; -1 = reads as
// zeros.  Evaluated once per (row, tap): the per-stage address is then just base + offset + channel.
__device__ __forceinline__ long pixel_off(const XSrc& s, long m, int n, int oy, int ox, int tap, int which) {
    if (m >= s.M) return -1;
    if (s.mode == 0) return which ? -1 : m * s.ld0;
    if (s.mode == 1) return which ? -1 : (((long)n * s.Hi + 2 * oy) * s.Wi + 2 * ox) * s.ld0;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    if (s.mode == 2) {
        int iy = oy + ky - 1, ix = ox + kx - 1;
        iy = iy < 0 ? -iy : (iy >= s.Hi ? 2 * s.Hi - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= s.Wi ? 2 * s.Wi - 2 - ix : ix);
        if (which) return s.C1 ? (((long)n * s.Hi + iy) * s.Wi + ix) * s.ld1 : -1;
        const int hh = s.Hi >> s.up, ww = s.Wi >> s.up;
        return (((long)n * hh + (iy >> s.up)) * ww + (ix >> s.up)) * s.ld0;
    }
    const int iy = oy - ky, ix = ox - kx;                                    // mode 3
    if (which || iy < 0 || iy >= s.Hi || ix < 0 || ix >= s.Wi) return -1;
    return (((long)n * s.Hi + iy) * s.Wi + ix) * s.ld0;
}

__device__ __forceinline__ void glds16(const bf16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row, int piece) { return row * 128 + ((piece ^ (row & 7)) << 4); }
// 64-byte LDS rows (32 bf16) read by ds_read_b128 with lane & 15 = 16 consecutive rows (offset 0..2) and lane >> 4 = the 16-byte piece:
// physical piece = piece ^ pswz32(row).  The four lane groups of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) then touch
// 16 distinct 16-byte bank slots each (checked exhaustively for row offsets 0..2: tools/lds_swizzle_check.py).
__device__ __forceinline__ int pswz32(int row) { return (row >> 1) & 2; }

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
    int d2s;         // direct 3x3 kernel, fp32 out: cout c = phase*d2s + o is stored depth-to-space, out[n][2y+phase/2][2x+phase%2][o]
                     // with d2s channels per output pixel (the 4-phase final seg conv writes the logits in place, no shuffle pass)
    // operand transform (XF kernels, mode 0/1): the pixel operand is act(xscale[c]*x + xshift[c]) (rounded to bf16) [* xgate[row / xhw][c]]
    // applied between the global load and the LDS store -- a BatchNorm apply (+ ReLU, + SE gate) that is never materialised
    const float* xscale; const float* xshift; const float* xgate; long xhw; int xact;
    const bf16* addend; int ld_add;   // staged bf16 epilogue: out = bf16(bf16(acc) + addend[pix][co])  (residual-gradient add of a dgrad)
    // Phase form of a 3x3 conv over a nearest-x2 up-sampled map (direct kernel): the conv runs on the LOW-resolution grid with 4 * k
    // outputs (one k-vector per output phase (py, px)) and summed-tap effective weights; phase (py, px) only has non-zero weights on the
    // taps ky in {py, py+1}, kx in {px, px+1}, so 4 of the 9 taps are visited (2.25x fewer MACs than convolving the up-sampled map):
    //   phase_mode 1 (forward): the cout tile's phase = cout / phase_span;  2 (data gradient, mode 3): the K chunk's phase = channel / phase_span
    int phase_mode, phase_span;
    int add_s2;                       // 1: the addend lives on the stride-2 sub-grid [N][H/2][W/2] and is added at even (y, x) only (the data
                                      //    gradient of a stride-2 1x1 conv joining the gradient of a full-resolution 1x1 conv of the same input)
    long* amax;                       // fp32 depth-to-space output mode only: write arg-max over the d2s classes (int64 per output pixel, first
                                      // maximum wins) instead of the logits (deploy forward: model/model.py:197 only needs the mask)
    int add_pre;                      // 1: the addend goes in BEFORE the activation: out = act(acc + bias + addend) (inference: folded
                                      // BatchNorm + identity branch + ReLU of an XBlock in conv_block_3's epilogue)
    // 1x1 GEMMs, one weight matrix PER IMAGE (w_rpi > 0): rows [n * w_rpi, (n + 1) * w_rpi) use w + n * w_img_stride.  Inference: the SE
    // gate g[n][ci] of an XBlock folded into conv_block_3's weights (W_n = W diag(g_n), hn_scale_weight_gate) instead of a pass over the
    // activation; a pixel tile must lie inside one image (w_rpi % 128 == 0)
    long w_img_stride; unsigned w_rpi;
    // level-packed rows (det towers), inference: per-LEVEL eval-mode BatchNorm of the output in the epilogue, out = act(lcoef[l][0][c] *
    // (acc + bias) + lcoef[l][1][c]) with l = the level of the pixel tile (levels start on 128-row boundaries: tile-uniform);
    // lcoef [ln][4][Nout] (scale, shift, -, -), lrow = cumulative rows per level
    const float* lcoef; int ln; long lrow[HN_MAX_LEVELS + 1];
    // level-packed rows -> per-image concatenated output (generic epilogue only; lo_n = images, 0 = off): row r of level l (rows lrow[l]...,
    // [image][pixel], alignment rows behind the real ones) is stored at image * img_stride + (lo_pix[l] + pixel) * ldc
    int lo_n; unsigned lo_hw[HN_MAX_LEVELS]; unsigned lo_pix[HN_MAX_LEVELS];
    // Direct 3x3 kernel, mode 3 (data gradient on the padded (H+2) x (W+2) grid), staged bf16 epilogue: fold = 1 writes the INTERIOR of
    // the padded grid straight to the unpadded gradient out [N][H][W] (row stride ldc), multiplied by ELU'(fold_y) when the producer's
    // ELU output is given, and the one-pixel RING to ring [N][2 (W+2) + 2 H][Nout] (top row, bottom row, left column, right column);
    // seg_ring_fix_kernel then adds the ring to the border pixels it reflects / clamps onto.  No padded tensor, no full fold pass.
    // fold = 2: the same values in SPACE-TO-DEPTH order, out [N][H/2][W/2][4 Nout] (pixel (y, x), channel c at row (y/2, x/2), channel
    // ((y&1) 2 + (x&1)) Nout + c; row stride ldc): the operand form of the phase-form block that consumes this gradient
    // fold = 3: both -- out plain, fold_out2 (row stride ld_fo2) in space-to-depth order
    int fold;
    bf16* fold_out2; int ld_fo2;
    bf16* ring;
    const bf16* fold_y; int ld_fy;
    // Statistics epilogue operand (psum / psq rows): what is summed per channel over the tile's pixels, q = the bf16-rounded output
    //   emode 0: s1 = sum q, s2 = sum q^2                                                  (BatchNorm forward statistics)
    //   emode 1: s1 = sum q * bf16(relu(sc z + sh))                                        (SE gate-gradient partials of the XBlock's
    //            dbg = dz3 W3 GEMM: the pass hn_se_bwd_reduce_fused made over (dbg, z2); psq is not written)
    //   emode 2: g = q [sc z + sh > 0]; s1 = sum g, s2 = sum g (z - mu) rs               (BatchNorm-backward partial sums of a data-gradient
    //            producer: the reduce pass of hn_bn_bwd_fused over (da, z1))
    // ez: the forward pre-BatchNorm tensor at the output's rows / channels (row stride ld_ez); ecoef: [4][Nout] = sc, sh, mu, rs
    int emode; const bf16* ez; int ld_ez; const float* ecoef;
    //   emode 3 (1x1 GEMMs, staged bf16 output with a post-activation addend): g = out [ey > 0] with out = the FINAL stored value
    //            (accumulator + addend, rounded); s1 = sum g, s2 = sum g (z - mu) rs: the reduce pass of the masked BatchNorm backward of the
    //            PREVIOUS XBlock (hn_bn_bwd_fused over (dout, z3, y)) out of the epilogue of the launch that produces that dout = this
    //            block's data gradient dz1 W1 + g; computed in the write-out phase, where the addend joins
    const bf16* ey; int ld_ey;
    int tile_major;                   // direct kernel: block id order (see there)
    int wpre;                         // direct kernel: all weight tiles of the (single) chunk preloaded, one LDS slot per tap step
    unsigned long long* dbg_buf;      // tools/ only: stamp buffer (hn_debug_knob 15)
    int dbg;                          // tools/ only (hn_debug_knob 14; HN_TUNING builds): direct kernel ablation bits: 1 = no epilogue, 2 = no MFMAs,
                                      // 4 = no operand DMA after the first tile, 8 = no LDS fragment reads
};

// one pixel x 4 consecutive channels of the statistics epilogue (GemmNT::emode); cf = (sc, sh, mu, rs) of the 4 channels
struct StatCoef { f32x4 sc, sh, mu, rs; };
__device__ __forceinline__ StatCoef stat_coef(const float* ecoef, int Nout, int emode, int co0, bool ev) {
    StatCoef c;
    c.sc = c.sh = c.mu = c.rs = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ev) {
        c.sc = *reinterpret_cast<const f32x4*>(ecoef + co0);
        c.sh = *reinterpret_cast<const f32x4*>(ecoef + Nout + co0);
        if (emode == 2) {
            c.mu = *reinterpret_cast<const f32x4*>(ecoef + 2 * Nout + co0);
            c.rs = *reinterpret_cast<const f32x4*>(ecoef + 3 * Nout + co0);
        }
    }
    return c;
}
// (branch-free: the three forms differ by selects, the accumulators stay in registers)
__device__ __forceinline__ void stat_terms4(int emode, const f32x4 q, const f32x4 z, const StatCoef& cf, float& a0, float& a1, float& a2, float& a3,
                                            float& b0, float& b1, float& b2, float& b3) {
    float t1[4], t2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float pre = cf.sc[r] * z[r] + cf.sh[r];
        const float b = bfround(pre > 0.f ? pre : 0.f);
        const float g = pre > 0.f ? q[r] : 0.f;
        t1[r] = emode == 0 ? q[r] : (emode == 1 ? q[r] * b : g);
        t2[r] = emode == 0 ? q[r] * q[r] : g * (z[r] - cf.mu[r]) * cf.rs[r];
    }
    a0 += t1[0]; a1 += t1[1]; a2 += t1[2]; a3 += t1[3];
    b0 += t2[0]; b1 += t2[1]; b2 += t2[2]; b3 += t2[3];
}

// KG = 2: 512 threads = two independent 4-wave groups that walk alternate K stages (their own LDS stages, common barriers) and meet in LDS
// before the epilogue: half the barrier-separated K steps per workgroup for the small-M GEMMs of the deep stages, whose 15-step K loop is
// pure latency (one or two workgroups per CU).
// PLAIN: plain pixel rows of ONE tensor, one tap (x.mode 0: every 1x1 conv and its data gradient) -- no coordinate tables, no tap
// bookkeeping, per-row source offsets computed once: the generic form carries ~1 000 instructions of prologue and a branchy loop body for
// the 3x3 / stride-2 / concat modes that these launches (the latency-bound majority of the step's GEMMs) never use.
template <int BC, int BP, int WGC, int WGP, bool OUT_F32, int R, bool XF = false, int KG = 1, bool PLAIN = false, bool EPRE = false>
__global__ __launch_bounds__(256 * KG) void gemm_nt_kernel(const GemmNT p) {
    static_assert(!XF || R == 2, "the register-staged operand transform is written for the double buffer");
    static_assert(!PLAIN || !XF, "the plain-rows form has no operand transform");
    static_assert(KG == 1 || (!XF && R == 2), "the K-group form is written for the plain double buffer");
    constexpr int WC = BC / WGC, WP = BP / WGP, TC = WC / 16, TP = WP / 16;
    constexpr int XR = BP / 32, WR = (BC + 31) / 32;
    constexpr int STAGE = (BC + BP) * 128;                        // one K stage (64 k) of both operands
    extern __shared__ __attribute__((aligned(16))) char smem[];   // R stages, ONE array (keeps the compiler's LDS-DMA waits minimal)
    const int tid = threadIdx.x & 255, kg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (thread / wave index inside the K group)
    const int wc = wave / WGP, wp = wave % WGP;
    const int ncy = (p.Nout + BC - 1) / BC;                       // cout tiles: fastest logical index => they share the pixel tile in L2
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int c_tile = lid % ncy, p_tile = lid / ncy;
    const int c_blk = c_tile * BC;
    const long p_blk = (long)p_tile * BP;
    // LDS-DMA staging: a wave instruction writes 64 x 16 B = 8 consecutive tile rows (lane-linear).  Thread t owns PHYSICAL piece t&7 of
    // rows (t>>3) + 32 i; the XOR swizzle is applied on the SOURCE side: it fetches logical piece (t&7) ^ (row&7).
    const int r0 = tid >> 3, lp = (tid & 7) ^ (r0 & 7), half = lp >> 2, sub = (lp & 3) * 8;
    const int kc = p.KP >> 5, Q = p.taps * kc, S = (((Q + 1) >> 1) + KG - 1) / KG;       // S = K steps of this workgroup (KG stages each)
    const int Ktot = p.taps * p.KP;

    int xn[XR], xy[XR], xx[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        if constexpr (PLAIN) { xn[i] = 0; xy[i] = 0; xx[i] = 0; }
        else decomp_row(p.x, p_blk + r0 + 32 * i, xn[i], xy[i], xx[i]);
    }

    // 3x3 modes: the source coordinates of a tap are separable in (oy, ky) and (ox, kx), so they are tabulated once per block in LDS:
    //   ty0[ky][oy], tx0[kx][ox] in x0's grid (reflected, >> up for mode 2; -1 = outside for mode 3); ty1/tx1 in x1's full-res grid
    int* ty0 = reinterpret_cast<int*>(smem + R * KG * STAGE);
    int* tx0 = ty0 + 3 * p.x.H;
    int* ty1 = tx0 + 3 * p.x.W;
    int* tx1 = ty1 + 3 * p.x.H;
    if (!PLAIN && p.x.mode >= 2) {
        for (int i = tid; i < 3 * p.x.H; i += 256) {
            const int k = i / p.x.H, o = i - k * p.x.H;
            if (p.x.mode == 2) {
                int v = o + k - 1;
                v = border_idx(v, p.x.Hi, p.x.clamp);
                ty0[i] = v >> p.x.up;
                if (p.x.C1) ty1[i] = v;
            } else {
                const int v = o - k;
                ty0[i] = (v < 0 || v >= p.x.Hi) ? -1 : v;
            }
        }
        for (int i = tid; i < 3 * p.x.W; i += 256) {
            const int k = i / p.x.W, o = i - k * p.x.W;
            if (p.x.mode == 2) {
                int v = o + k - 1;
                v = border_idx(v, p.x.Wi, p.x.clamp);
                tx0[i] = v >> p.x.up;
                if (p.x.C1) tx1[i] = v;
            } else {
                const int v = o - k;
                tx0[i] = (v < 0 || v >= p.x.Wi) ? -1 : v;
            }
        }
        __syncthreads();
    }
    const int hh0 = p.x.mode == 2 ? p.x.Hi >> p.x.up : p.x.Hi, ww0 = p.x.mode == 2 ? p.x.Wi >> p.x.up : p.x.Wi;

    int tap = PLAIN ? 0 : (2 * kg + half) / kc, cidx = (2 * kg + half) - tap * kc;   // chunk q = 2*stage + half -> (tap, cidx); group kg owns stages it*KG + kg
    const int Ctot = PLAIN ? p.x.C0 : p.x.C0 + p.x.C1;
    int pix0[XR], pix1[XR];                        // per-row source PIXEL index for the current tap (-1 = zeros); pixels fit int32
    long xoff[XR];                                 // PLAIN: element offset of the row in x0 (-1 = zeros)
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const long m = p_blk + r0 + 32 * i;
        xoff[i] = m < p.x.M ? m * p.x.ld0 : -1;
    }
    auto retap = [&]() {
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const long m = p_blk + r0 + 32 * i;
            int a0 = -1, a1 = -1;
            if (m < p.x.M) {
                if (p.x.mode == 0) a0 = (int)m;
                else if (p.x.mode == 1) a0 = (xn[i] * p.x.Hi + 2 * xy[i]) * p.x.Wi + 2 * xx[i];
                else {
                    const int y0 = ty0[ky * p.x.H + xy[i]], x0c = tx0[kx * p.x.W + xx[i]];
                    if ((y0 | x0c) >= 0) a0 = (xn[i] * hh0 + y0) * ww0 + x0c;
                    if (p.x.C1) a1 = (xn[i] * p.x.Hi + ty1[ky * p.x.H + xy[i]]) * p.x.Wi + tx1[kx * p.x.W + xx[i]];
                }
            }
            pix0[i] = a0;
            pix1[i] = a1;
        }
    };
    if (!PLAIN && tap < p.taps) retap();
    else {
#pragma unroll
        for (int i = 0; i < XR; ++i) { pix0[i] = -1; pix1[i] = -1; }
    }
    const bf16* wimg = p.w + (p.w_rpi ? (long)((unsigned)p_blk / p.w_rpi) * p.w_img_stride : 0);    // (per-image weights: GemmNT::w_rpi)
    long wo[WR];                                   // weight row offset + this thread's in-chunk offset (-1 = zero row)
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int co = c_blk + r0 + 32 * i;
        wo[i] = (r0 + 32 * i < BC && co < p.Nout) ? (long)co * Ktot + sub : -1;
    }

    // the tile's bias values are requested here and arrive under the K loop (behind it they were a dependent trip in front of every
    // epilogue: all GEMMs of the folded inference path and the heads' biased convs)
    float pbias[TC][4];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) pbias[i][r] = (p.bias && co0 + r < p.Nout) ? p.bias[co0 + r] : 0.f;
    }
    // EPRE (the instance the small-tile launches with a statistics operand or an addend take): epilogue operands whose addresses do not
    // depend on the product are requested HERE and arrive under the K loop (older than every stage load, so the loop's counted vmcnt
    // waits still hold): the forward tensor of the statistics epilogue (emode 1 / 2), the addend and the emode-3 operands of the write-out
    // phase (<= 2 iterations per thread).  Behind the loop each of them was a dependent trip to memory -- one per write-out iteration --
    // at the end of a 14-16 us launch.  (Their ~40 registers stay out of the plain instance: 76 VGPRs, six workgroups per CU.)
    constexpr int NPC_ = BC / 8, NIT = (BP * NPC_ + 255) / 256, PRE = (EPRE && NIT <= 2) ? NIT : 0;
    static_assert(!EPRE || NIT <= 2, "the prefetching instance is for tiles of <= 512 output pieces");
    const bool staged_ = !OUT_F32 && (p.Nout & 7) == 0 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 &&
                         (p.rpi == 0 || (p.img_stride & 7) == 0);
    const bool pre_stat = EPRE && p.psum != nullptr && (p.emode == 1 || p.emode == 2);
    bf16x4 pez[EPRE ? TC : 1][EPRE ? TP : 1];
    if constexpr (EPRE) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
            const bool ev = co0 + 3 < p.Nout;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const long pix = p_blk + wp * WP + j * 16 + (lane & 15);
                pez[i][j] = (bf16x4){};
                if (pre_stat && ev && pix < p.x.M) pez[i][j] = *reinterpret_cast<const bf16x4*>(p.ez + pix * p.ld_ez + co0);
            }
        }
    }
    const bool pre_add = staged_ && p.addend && !p.add_pre, pre_e3 = staged_ && p.psum != nullptr && p.emode == 3;
    bf16x8 padd[PRE ? PRE : 1], pey[PRE ? PRE : 1], pzz[PRE ? PRE : 1];
    bool phas[PRE ? PRE : 1];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int idx = tid + 256 * u, row = idx / NPC_, pc = idx % NPC_;
        const long pix = p_blk + row;
        const int co = c_blk + pc * 8;
        const bool ok = kg == 0 && idx < BP * NPC_ && pix < p.x.M && co < p.Nout;
        padd[u] = pey[u] = pzz[u] = zero8();
        phas[u] = false;
        if (ok && pre_add) {
            long arow = pix;
            bool has = true;
            if (p.add_s2) {
                const unsigned W = (unsigned)p.x.W, H = (unsigned)p.x.H;
                const unsigned x = (unsigned)pix % W, t = (unsigned)pix / W;
                const unsigned y = t % H, n = t / H;
                has = !((x | y) & 1u);
                arow = ((long)n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
            }
            phas[u] = has;
            if (has) padd[u] = ld8(p.addend + arow * p.ld_add + co);
        }
        if (ok && pre_e3) { pey[u] = ld8(p.ey + pix * p.ld_ey + co); pzz[u] = ld8(p.ez + pix * p.ld_ez + co); }
    }

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // software pipeline over a ring of R LDS stages.  Iteration `it` issues the LDS-DMA of stage `it` (while it < S) and multiplies stage
    // c = it-(R-1).  LDS-DMA completion is only ordered by the issuing wave's own counted vmcnt wait followed by a barrier: every thread
    // issues exactly G loads per stage, so "stage c has landed" is vmcnt(newer * G) with newer = stages issued after c.  The first R-1
    // stages go out back to back (one memory latency for short K instead of one per stage); nothing is issued past the last stage.
    constexpr int G = XR + WR;
    static_assert(R == 2 || BC >= 32, "deeper rings need every wave to issue the same number of loads");
    // XF: the pixel operand of the stage in flight lives in registers (raw rows + the per-channel coefficients of its 8 channels + the
    // per-image gate) and is transformed / stored to LDS after the MFMAs of the previous stage
    bf16x8 xraw[XF ? XR : 1];
    bool xval[XF ? XR : 1];
    int xch = 0;
    // per-channel coefficients (scale, shift, gate of the tile's image: xhw is a multiple of BP) staged in LDS once, behind the ring
    float* xcoef = reinterpret_cast<float*>(smem + R * KG * STAGE);
    if (XF) {
        const long img = p.xgate ? p_blk / p.xhw : 0;
        for (int i = tid; i < p.KP; i += 256) {
            const bool v = i < Ctot;
            xcoef[i] = v ? p.xscale[i] : 0.f;
            xcoef[p.KP + i] = v ? p.xshift[i] : 0.f;
            xcoef[2 * p.KP + i] = (v && p.xgate) ? p.xgate[img * Ctot + i] : 1.f;
        }
        __syncthreads();                                          // the first transform (end of iteration 0) reads other threads' entries
    }
    for (int it = 0; it < S + R - 1; ++it) {
        if (it >= R - 1) {
            const int newer = (it < S ? it : S) - 1 - (it - (R - 1));
            if (XF) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (R >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");
            else if (R >= 3 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        bool xcv = false;
        if (it < S) {
            char* sW = smem + ((it % R) * KG + kg) * STAGE;
            char* sX = sW + BC * 128;
            const int q = 2 * (it * KG + kg) + half;
            const bool qv = q < Q;
            const int c = cidx * 32 + sub;
            const bool from0 = c < p.x.C0;
            const bool cv = qv && c < Ctot;
            const bf16* xbase = from0 ? p.x.x0 + c : p.x.x1 + (c - p.x.C0);
            const long ldx = from0 ? p.x.ld0 : p.x.ld1;
            if (XF) {
                xcv = cv;
                xch = c;
#pragma unroll
                for (int i = 0; i < XR; ++i) {
                    const int pix = from0 ? pix0[i] : pix1[i];
                    xval[i] = cv && pix >= 0;
                    const bf16* src = xval[i] ? xbase + (long)pix * ldx : g_zero_piece;
                    xraw[i] = ld8(src);
                }
            } else if constexpr (PLAIN) {
#pragma unroll
                for (int i = 0; i < XR; ++i) {
                    const bf16* src = (cv && xoff[i] >= 0) ? p.x.x0 + xoff[i] + c : g_zero_piece;
                    glds16(src, sX + (wave * 8 + 32 * i) * 128);
                }
            } else {
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int pix = from0 ? pix0[i] : pix1[i];
                const bf16* src = (cv && pix >= 0) ? xbase + (long)pix * ldx : g_zero_piece;
                glds16(src, sX + (wave * 8 + 32 * i) * 128);
            }
            }
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                if (wave * 8 + 32 * i < BC) {                          // wave-uniform (always true for BC >= 32)
                    const bf16* src = (qv && wo[i] >= 0) ? wimg + wo[i] + q * 32 : g_zero_piece;
                    glds16(src, sW + (wave * 8 + 32 * i) * 128);
                }
            }
            if constexpr (PLAIN) {
                cidx += 2 * KG;                                        // (one tap: cidx is the chunk index q itself, q < Q ends the loads)
            } else if (qv) {
                cidx += 2 * KG;
                if (cidx >= kc) {
                    while (cidx >= kc) { cidx -= kc; ++tap; }
                    if (tap < p.taps) retap();
                }
            }
        }
        if (it >= R - 1) {
            const char* sW = smem + (((it - (R - 1)) % R) * KG + kg) * STAGE;
            const char* sX = sW + BC * 128;
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
        }
        if (XF && it < S) {                                       // transform the stage in flight and store it where the DMA would have
            char* sX = smem + (it % R) * STAGE + BC * 128;
            float sc[8], sh[8], gt[8];
            if (xcv) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(xcoef + xch + 4 * h);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(xcoef + p.KP + xch + 4 * h);
                    const f32x4 g = *reinterpret_cast<const f32x4*>(xcoef + 2 * p.KP + xch + 4 * h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { sc[4 * h + k] = a[k]; sh[4 * h + k] = b[k]; gt[4 * h + k] = g[k]; }
                }
            }
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                bf16x8 o = zero8();
                if (xval[i]) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = bf2f(xraw[i][k]) * sc[k] + sh[k];
                    act_fwd_n(v, p.xact);
                    if (p.xgate) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = bfround(v[k]) * gt[k];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k]);
                }
                *reinterpret_cast<bf16x8*>(sX + (wave * 8 + 32 * i) * 128 + lane * 16) = o;
            }
        }
    }

    if (KG == 2) {                                                // the two K groups meet: group 1 hands its partial tile over and retires
        __syncthreads();
        float* hand = reinterpret_cast<float*>(smem);              // [256 threads][TC*TP*4], lane-contiguous
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hand[((i * TP + j) * 4 + r) * 256 + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += hand[((i * TP + j) * 4 + r) * 256 + tid];
        // (the epilogue's own leading barrier, when it stages through LDS, now only involves the surviving group)
    }
    // ---- epilogue: bias, activation, optional BN partial statistics; store.  bf16 outputs whose rows are 16-B aligned go through an
    // LDS tile ([BP][BC], 16-B pieces XOR-swizzled by the pixel row) so that every wave writes whole contiguous row segments; the
    // remaining cases (fp32 head outputs, ragged Nout) store 4 consecutive couts per lane directly.
    const bool want_stats = p.psum != nullptr;
    constexpr int NPC = BC / 8;                                       // 16-B pieces per staged row
    const bool staged = !OUT_F32 && (p.Nout & 7) == 0 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 &&
                        (p.rpi == 0 || (p.img_stride & 7) == 0);
    if (staged) __syncthreads();                                      // every wave is done reading the last K stage
    float vv[TC * TP * 4];                                            // the wave tile, flat: one uniform activation branch for all of it
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[(i * TP + j) * 4 + r] = acc[i][j][r] + pbias[i][r];
    if (p.lcoef) {                                                    // per-level BatchNorm (running statistics) of the tile's level
        int lv = 0;
        while (lv + 1 < p.ln && p_blk >= p.lrow[lv + 1]) ++lv;
        const float* cf = p.lcoef + (long)lv * 4 * p.Nout;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = co0 + r < p.Nout ? cf[co0 + r] : 0.f, sh = co0 + r < p.Nout ? cf[p.Nout + co0 + r] : 0.f;
#pragma unroll
                for (int j = 0; j < TP; ++j) vv[(i * TP + j) * 4 + r] = vv[(i * TP + j) * 4 + r] * sc + sh;
            }
        }
    }
    if (p.addend && p.add_pre) {
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const long pix = p_blk + wp * WP + j * 16 + (lane & 15);
            if (pix >= p.x.M) continue;
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
                if (co0 + 3 < p.Nout) {
                    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.addend + pix * p.ld_add + co0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[(i * TP + j) * 4 + r] += bf2f(a[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (co0 + r < p.Nout) vv[(i * TP + j) * 4 + r] += bf2f(p.addend[pix * p.ld_add + co0 + r]);
                }
            }
        }
    }
    const bool e3 = want_stats && p.emode == 3;                       // statistics of the final values: in the write-out phase below
    if (want_stats && !e3) {
        if (!staged) __syncthreads();                                 // the operand ring is free: [WGP][BC][2] floats of it hold the wave sums
        // (BEHIND the staged output tile: the wave sums and the tile are written in one phase and read after ONE common barrier --
        // the statistics used to cost two barriers of their own, +2.2 us on the 11 us stage-4 GEMM)
        static_assert(BP * BC * 2 + WGP * BC * 8 <= R * KG * STAGE, "staged tile + wave sums fit the operand ring");
        float* red = reinterpret_cast<float*>(smem + BP * BC * 2);
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.emode == 0) {                                       // BatchNorm forward statistics: sums and sums of squares, nothing to load
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const bool pv = p_blk + wp * WP + j * 16 + (lane & 15) < p.x.M;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float q = OUT_F32 ? vv[(i * TP + j) * 4 + r] : bfround(vv[(i * TP + j) * 4 + r]);
                        q = pv ? q : 0.f;
                        s1[r] += q;
                        s2[r] += q * q;
                    }
                }
            } else {
            const bool ev = co0 + 3 < p.Nout;
            const StatCoef cf = stat_coef(p.ecoef, p.Nout, p.emode, co0, ev);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const long pix = p_blk + wp * WP + j * 16 + (lane & 15);
                const bool pv = pix < p.x.M;
                f32x4 q, z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[r] = OUT_F32 ? vv[(i * TP + j) * 4 + r] : bfround(vv[(i * TP + j) * 4 + r]);
                    q[r] = pv ? q[r] : 0.f;
                }
                if constexpr (EPRE) {                                 // (requested before the K loop)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = bf2f(pez[i][j][r]);
                } else if (ev && pv) {
                    const bf16x4 zv = *reinterpret_cast<const bf16x4*>(p.ez + pix * p.ld_ez + co0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = bf2f(zv[r]);
                }
                stat_terms4(p.emode, q, z, cf, s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]);
            }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum(s1[r]);
                s2[r] = row16_sum(s2[r]);
            }
            if ((lane & 15) == 0) {
                const int cl = wc * WC + i * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) { red[(wp * BC + cl + r) * 2] = s1[r]; red[(wp * BC + cl + r) * 2 + 1] = s2[r]; }
            }
        }
    }
    act_fwd_n(vv, p.act);
    if (staged) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int col = wc * WC + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int prow_l = wp * WP + j * 16 + (lane & 15);
                const float* v = vv + (i * TP + j) * 4;
                bf16x4 t = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *reinterpret_cast<bf16x4*>(smem + prow_l * (BC * 2) + ((((col >> 3) ^ prow_l) & (NPC - 1)) << 4) + ((col >> 2) & 1) * 8) = t;
            }
        }
    } else {
        int olv = 0;                                                  // (pixel tiles never straddle two levels: rows are multiples of 128)
        if (p.lo_n) while (olv + 1 < p.ln && p_blk >= p.lrow[olv + 1]) ++olv;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const long pix = p_blk + wp * WP + j * 16 + (lane & 15);
            if (pix >= p.x.M) continue;
            long orow = pix * p.ldc;
            if (p.rpi) {
                const unsigned im = (unsigned)pix / (unsigned)p.rpi;
                orow = (long)im * p.img_stride + (long)((unsigned)pix - im * (unsigned)p.rpi) * p.ldc;
            }
            if (p.lo_n) {
                const unsigned rel = (unsigned)(pix - p.lrow[olv]), hw = p.lo_hw[olv];
                const unsigned im = rel / hw;
                if (im >= (unsigned)p.lo_n) continue;                 // an alignment row of the packing
                orow = (long)im * p.img_stride + (long)(p.lo_pix[olv] + rel - im * hw) * p.ldc;
            }
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int co0 = c_blk + wc * WC + i * 16 + (lane >> 4) * 4;
                const float* v = vv + (i * TP + j) * 4;
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
    }
    if (staged || want_stats) __syncthreads();
    if (want_stats && !e3 && tid < BC && c_blk + tid < p.Nout) {      // one partial row per pixel tile: the WGP wave sums in fixed order
        const float* red = reinterpret_cast<const float*>(smem + BP * BC * 2);
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < WGP; ++k) { t1 += red[(k * BC + tid) * 2]; t2 += red[(k * BC + tid) * 2 + 1]; }
        p.psum[(long)p_tile * p.Nout + c_blk + tid] = t1;
        if (p.psq) p.psq[(long)p_tile * p.Nout + c_blk + tid] = t2;
    }
    if (staged) {
        bf16* outp = reinterpret_cast<bf16*>(p.out);
        float e1[8], e2[8], emu[8], ers[8];                           // emode 3: this thread's channel piece is tid % NPC in every iteration
#pragma unroll
        for (int k = 0; k < 8; ++k) { e1[k] = 0.f; e2[k] = 0.f; emu[k] = 0.f; ers[k] = 0.f; }
        if (e3) {
            const int co = c_blk + (tid % NPC) * 8;
            if (co < p.Nout) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { emu[k] = p.ecoef[2 * p.Nout + co + k]; ers[k] = p.ecoef[3 * p.Nout + co + k]; }
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = tid + 256 * u;
            if (idx >= BP * NPC) break;
            const int row = idx / NPC, pc = idx % NPC;
            const long pix = p_blk + row;
            const int co = c_blk + pc * 8;
            if (pix < p.x.M && co < p.Nout) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * (BC * 2) + (((pc ^ row) & (NPC - 1)) << 4));
                long orow = pix * p.ldc;
                if (p.rpi) {
                    const unsigned im = (unsigned)pix / (unsigned)p.rpi;
                    orow = (long)im * p.img_stride + (long)((unsigned)pix - im * (unsigned)p.rpi) * p.ldc;
                }
                if (p.addend && !p.add_pre) {
                    bool has = true;
                    bf16x8 a;
                    if (PRE) {                                         // small tiles: requested before the K loop
                        has = phas[u < PRE ? u : 0];
                        a = padd[u < PRE ? u : 0];
                    } else {
                        long arow = pix;
                        if (p.add_s2) {
                            const unsigned W = (unsigned)p.x.W, H = (unsigned)p.x.H;
                            const unsigned x = (unsigned)pix % W, t = (unsigned)pix / W;
                            const unsigned y = t % H, n = t / H;
                            has = !((x | y) & 1u);
                            arow = ((long)n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
                        }
                        a = has ? ld8(p.addend + arow * p.ld_add + co) : zero8();
                    }
                    if (has) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = f2bf(bf2f(v[k]) + bf2f(a[k]));
                    }
                }
                if (e3) {
                    const bf16x8 yv = PRE ? pey[u < PRE ? u : 0] : ld8(p.ey + pix * p.ld_ey + co);
                    const bf16x8 zv = PRE ? pzz[u < PRE ? u : 0] : ld8(p.ez + pix * p.ld_ez + co);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float g = bf2f(yv[k]) > 0.f ? bf2f(v[k]) : 0.f;
                        e1[k] += g;
                        e2[k] += g * (bf2f(zv[k]) - emu[k]) * ers[k];
                    }
                }
                *reinterpret_cast<bf16x8*>(outp + orow + co) = v;
            }
        }
        if (e3) {
            // the 256 / NPC threads that share a channel piece are folded through LDS (behind the staged tile and the wave sums), in
            // thread order: deterministic
            float* r3 = reinterpret_cast<float*>(smem + BP * BC * 2 + WGP * BC * 8);
            static_assert(BP * BC * 2 + WGP * BC * 8 + 256 * 16 * 4 <= R * KG * STAGE || BC < 64, "emode 3 scratch fits the operand ring");
#pragma unroll
            for (int k = 0; k < 8; ++k) { r3[tid * 16 + k] = e1[k]; r3[tid * 16 + 8 + k] = e2[k]; }
            __syncthreads();
            if (tid < BC && c_blk + tid < p.Nout) {
                const int pc = tid >> 3, k = tid & 7;
                float t1 = 0.f, t2 = 0.f;
                for (int j = 0; j < 256 / NPC; ++j) { t1 += r3[(j * NPC + pc) * 16 + k]; t2 += r3[(j * NPC + pc) * 16 + 8 + k]; }
                p.psum[(long)p_tile * p.Nout + c_blk + tid] = t1;
                p.psq[(long)p_tile * p.Nout + c_blk + tid] = t2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Direct 3x3 convolution (im2col-free): one workgroup = a 16x16 output-pixel patch x BC couts.  Per 64-channel chunk the 18x18 input
// patch (with halo; reflection / nearest-up / concat or zero padding resolved while loading) is DMA'd into LDS ONCE and reused by all
// nine taps; only the [BC][64] weight slice of a tap is streamed per stage.  L2 traffic per FLOP is ~3x lower than the row-gather
// GEMM above (which re-reads the pixel rows for every tap), which is what bounds that kernel on MI355X.
//   mode 2: out(y,x) = sum_tap V(refl(y+ky-1), refl(x+kx-1)) W[tap]        (forward of the seg decoder convs)
//   mode 3: out(y,x) = sum_tap Z0(y-ky, x-kx) W[tap], Z0 zero outside       (their dgrad on the padded (H+2)x(W+2) grid)
// 512 threads = 8 waves: WGC = BC/64 cout groups x (8/WGC) pixel-row groups; wave tile = 64 couts x (16/WGP rows x 16 px).
// ---------------------------------------------------------------------------------------------------------
// PIPE = false: one patch buffer + two weight buffers (73 KB LDS, 128 VGPRs): two workgroups per CU cover each other's DMA waits; every
//   tap step still exposes most of the weight-tile DMA latency (measured 2.2 us per tap step against 0.43 us of MFMA time).
// PIPE = true: ONE workgroup per CU with a software pipeline inside it: two patch buffers (the next chunk's patch lands during the
//   current chunk's taps) and a ring of three weight tiles with counted s_waitcnt vmcnt (the tile of tap step i+2 is issued before the
//   MFMAs of step i, so two tile loads are always in flight); up to 256 VGPRs (no scratch spills).
template <int BC, bool OUT_F32, bool PIPE, int XB = (PIPE ? 2 : 1)>   // XB: patch buffers (PIPE with a single K chunk needs one)
__global__ __launch_bounds__(512, 4) void conv3x3_direct_kernel(const GemmNT p) {
    constexpr int KC = PIPE ? 32 : 64;                                // channels per K chunk
    constexpr int NPP = KC / 8, PB = KC * 2;                          // 16-byte pieces / bytes per pixel (or weight) row of a chunk in LDS
    constexpr int XBUFS = XB, WBUFS = PIPE ? 4 : 2;
    constexpr int WCO = BC >= 64 ? 64 : BC;                           // couts per wave
    constexpr int WGC = BC / WCO, WGP = 8 / WGC, ROWS = 16 / WGP;     // rows of the patch per wave
    constexpr int TC = WCO / 16, TP = ROWS;
    // X buffer padded to whole 1 KiB DMA runs (PIPE: X and W buffers padded to whole 512-thread rounds, so that every wave issues the
    // same number of loads and the counted waits hold for all of them)
    constexpr int PPIX = 18 * 18, XL = (PPIX * NPP + 511) / 512, WL = (BC * NPP + 511) / 512;
    // (PRE32: the 32-channel-chunk layout with ONE patch buffer and ALL tap tiles of its single chunk preloaded, unpadded: 24 + 36 KB)
    constexpr bool PRE32 = PIPE && XB == 1;
    constexpr int XBYTES = PIPE ? XL * 512 * 16 : (PPIX * 128 + 1023) / 1024 * 1024, WBYTES = PRE32 ? BC * PB : (PIPE ? WL * 512 * 16 : BC * 128);
    extern __shared__ __attribute__((aligned(16))) char smem[];       // X patch x2 | W tile x2
    char* sXb = smem;
    char* sWb = smem + XBUFS * XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: M0 values, LDS bases and wave-uniform branches off the vector unit)
    const int wc = wave / WGP, wp = wave % WGP;
    const XSrc& xs = p.x;
    const int ncy = (p.Nout + BC - 1) / BC;
    const int tx_n = (xs.W + 15) >> 4, ty_n = (xs.H + 15) >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // tools/ only (knob 14 bit 16 + knob 15 = buffer): s_memtime stamps of every 64th workgroup's wave 0: [0] start, [1] loop entry,
    // [2 + it] after the barrier of iteration it, then loop end and kernel end
    unsigned long long* stampb = (HN_DBG(p) & 16) && (blockIdx.x & 63) == 0 && tid == 0
                                     ? reinterpret_cast<unsigned long long*>(p.dbg_buf) + (blockIdx.x >> 6) * 128 : nullptr;
    int stampi = 0;
    auto stamp = [&]() {
#ifdef HN_TUNING
        __builtin_amdgcn_sched_barrier(0);
        if (stampb && stampi < 128) stampb[stampi++] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    stamp();
    // block id -> (cout tile, patch).  p.tile_major = 0: the cout tiles of a patch are neighbours (an XCD works on a contiguous range of
    // patches with ALL cout tiles: the patch is read once into its L2, the whole weight tensor must stay there); 1: a cout tile's patches
    // are neighbours (an XCD works on few cout tiles -- their weights stay L2-hot for the per-tap tile streams -- and reads every patch)
    const int npatch = gridDim.x / ncy;
    const int c_tile = p.tile_major ? lid / npatch : lid % ncy;
    int t = p.tile_major ? lid % npatch : lid / ncy;
    const int patch_id = t;
    const int tx = t % tx_n;
    t /= tx_n;
    const int ty = t % ty_n;
    const int n = t / ty_n;
    const int c_blk = c_tile * BC, oy0 = ty * 16, ox0 = tx * 16;
    const int org = xs.mode == 2 ? -1 : -2;                           // patch origin relative to the output tile
    const int Ctot = xs.C0 + xs.C1;
    const int NT = p.phase_mode ? 4 : 9;                              // taps visited per 64-channel chunk
    const int nchunk = xs.diag ? 1 : (p.KP + KC - 1) / KC, S = nchunk * NT;
    const int Ktot = 9 * p.KP;
    const int tile_phase = p.phase_mode == 1 ? c_blk / p.phase_span : 0;
    // Loop bookkeeping without integer divisions: a cursor (chunk, tap index, phase) advanced by compare-and-wrap.  (Stamps of the round-3
    // loop -- tools/stamp_seg.py -- showed ~2500 cycles per tap step with the DMA, the LDS reads and the MFMAs all ablated: st / NT,
    // st % wslots, (chunk * 64) / phase_span and the per-read swizzle arithmetic, replicated in 16 waves per CU, cost more VALU issue
    // time than the step's 1024 MFMA cycles.)
    const int cpp = p.phase_mode == 2 ? p.phase_span / KC : 0;        // chunks per phase (data gradient: the K chunk's phase)
    struct Cur { int chunk, ti, ph, cnt; };
    auto cur0 = [&]() { Cur c; c.chunk = 0; c.ti = 0; c.ph = tile_phase; c.cnt = 0; return c; };
    auto adv = [&](Cur& c) {
        if (++c.ti == NT) {
            c.ti = 0;
            ++c.chunk;
            if (p.phase_mode == 2 && ++c.cnt == cpp) { c.cnt = 0; ++c.ph; }
        }
    };
    auto tap_at = [&](const Cur& c) { return p.phase_mode ? ((c.ph >> 1) + (c.ti >> 1)) * 3 + (c.ph & 1) + (c.ti & 1) : c.ti; };
    const int xc0 = xs.diag ? c_blk : 0;                              // first input channel of chunk 0
    // the tile's bias values wait in LDS behind the operand buffers (the epilogue's per-sub-tile global loads were a dependent round trip each)
    float* sbias = reinterpret_cast<float*>(smem + XBUFS * XBYTES + ((!PIPE || PRE32) && p.wpre ? S : WBUFS) * WBYTES);
    if (tid < BC) sbias[tid] = (p.bias && c_blk + tid < p.Nout) ? p.bias[c_blk + tid] : 0.f;

    // patch pieces owned by this thread: e = tid + 512 i -> patch pixel e>>3, PHYSICAL piece e&7 (logical = physical ^ (patch column & 7)).
    // One register per piece: the source pixel as (gy << 16 | gx) in full-resolution coordinates (-1: outside / zero); the row index in
    // either operand and the channel sub-offset are recomputed when the load is issued (this kernel sits at the 128-VGPR limit of two
    // co-resident workgroups: three registers per piece cost 12 spilled VGPRs = 52 B/lane of scratch traffic).
    // (bits 13-15: the LOGICAL 16-byte piece of the chunk this thread's physical LDS piece holds -- the swizzle resolved once; gx < 8192)
    if (BC == 64 && !PIPE && xs.diag) {
        // grouped conv: the 9 KB of diagonal weight blocks do not depend on anything computed below -- requested first, their latency runs
        // under the prologue's index arithmetic
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (512 * i + 64 * wave < 9 * 64) {                         // wave-uniform
                const int e = 512 * i + tid, tap = e >> 6, col = e & 63, co = c_blk + col;
                const bf16* src = co < p.Nout ? p.w + co * (9 * p.KP) + tap * 64 + (col >> 3) * 8 : g_zero_piece;
                glds16(src, smem + XBUFS * XBYTES + (512 * i + 64 * wave) * 16);
            }
        }
    }
    // (the 18 source rows and 18 source columns of the patch are resolved ONCE, by 36 threads, into an LDS table: done per piece, the
    // reflect / clamp / zero border arithmetic was ~150 of the prologue's ~380 VALU instructions, each costing 16 cycles of workgroup
    // lifetime with four waves per SIMD starting at the same time)
    // [0..17]: source row of patch row py (-1: none), [18..35]: columns; behind the operand buffers and the bias values (dynamic LDS: a
    // static array on top of the 160 KB dynamic opt-in is refused by the runtime)
    int* patch_src = reinterpret_cast<int*>(sbias + BC);
    if (tid < 36) {
        const bool isy = tid < 18;
        const int k = isy ? tid : tid - 18, L = isy ? xs.Hi : xs.Wi;
        int g = (isy ? oy0 : ox0) + org + k;
        if (xs.mode == 2) g = border_idx(g, L, xs.clamp);             // (negative only for pixels that feed no in-image output)
        else g = (g >= 0 && g < L) ? g : -1;
        patch_src[tid] = g;
    }
    __syncthreads();
    // one operand only (no concat: every phase-form conv and data gradient) and < 2^28 source pixels: the piece holds its source PIXEL index
    // ((pixel << 3) | logical piece), the per-chunk address is one multiply-add instead of the two-operand coordinate arithmetic
    const bool single = xs.C1 == 0 && (long)p.x.M < (1L << 27);
    int spack[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int e = tid + 512 * i;
        const int pp = e / NPP;
        spack[i] = -1;
        if (pp < PPIX) {
            const int py = pp / 18, px = pp - py * 18;
            const int sub = PIPE ? ((e & 3) ^ pswz32(px)) : ((e & 7) ^ (px & 7));
            const int gy = patch_src[py], gx = patch_src[18 + px];
            if ((gy | gx) >= 0)
                spack[i] = single ? ((((n * (xs.Hi >> xs.up) + (gy >> xs.up)) * (xs.Wi >> xs.up) + (gx >> xs.up)) << 3) | sub)
                                  : ((gy << 16) | (sub << 13) | gx);
        }
    }
    auto issue_x = [&](int chunk) {
        if ((HN_DBG(p) & 4) && chunk >= XBUFS) return;
        const int k0 = chunk * KC;
        char* sX = sXb + (XBUFS == 2 ? (chunk & 1) : 0) * XBYTES;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            if (PIPE || 512 * i + 64 * wave < PPIX * 8) {              // PIPE: every wave issues all XL rounds (padded buffer)
                const bf16* src = g_zero_piece;
                int pk = spack[i];
                asm volatile("" : "+v"(pk));                          // keep the 64-bit row pointers out of loop-invariant registers
                if (single) {
                    const int c = xc0 + k0 + ((pk & 7) << 3);
                    if (c < Ctot && pk >= 0) src = xs.x0 + c + (long)(pk >> 3) * xs.ld0;
                } else {
                    const int c = xc0 + k0 + (((pk >> 13) & 7) << 3);
                    if (c < Ctot && pk >= 0) {
                        const int gy = pk >> 16, gx = pk & 0x1fff;
                        if (c < xs.C0) src = xs.x0 + c + (long)((n * (xs.Hi >> xs.up) + (gy >> xs.up)) * (xs.Wi >> xs.up) + (gx >> xs.up)) * xs.ld0;
                        else src = xs.x1 + (c - xs.C0) + (long)((n * xs.Hi + gy) * xs.Wi + gx) * xs.ld1;
                    }
                }
                glds16(src, sX + (512 * i + 64 * wave) * 16);
            }
        }
    };
    // The barrier-free forms (all tap tiles preloaded / grouped conv) start with the patch of chunk 0: requested HERE, as soon as its source
    // table exists -- the weight-row offsets, accumulators and fragment offsets below are computed under the DMA instead of in front of it
    // (stamps of the stage-4 grouped conv: 4 400 cycles of prologue, then 7 700 of waiting for the patch, of a 19 800-cycle workgroup)
    const bool early_x = ((!PIPE || PRE32) && p.wpre) || (BC == 64 && !PIPE && xs.diag);
    if (early_x) issue_x(0);
    // weight pieces: row = (tid>>3) + 64 i, physical piece tid&7 (32-channel chunks: row = (tid>>2) + 128 i, physical piece tid&3; the
    // swizzle of 64-byte rows is pswz32 below)
    const int wsub = PIPE ? (((tid & 3) ^ pswz32(tid >> 2)) << 3) : (((tid & 7) ^ ((tid >> 3) & 7)) << 3);
    int wrow[WL];                                                      // element offset of the thread's weight row (32-bit: Nout * 9 * KP < 2^31)
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int r = tid / NPP + (512 / NPP) * i;                    // (512 / NPP is a multiple of 8: the row's swizzle key is that of tid)
        const int co = c_blk + r;
        wrow[i] = (r < BC && co < p.Nout) ? co * Ktot + wsub : -1;
    }

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane LDS read offsets, hoisted out of the tap loop: A rows i * 16 + (lane & 15) of the weight tile, B patch columns
    // (lane & 15) + dx for the three tap columns; the second K half of a 128-byte row is the first one's address ^ 64
    const int aofs = PIPE ? (wc * WCO + (lane & 15)) * PB + (((lane >> 4) ^ pswz32(lane & 15)) << 4) : swz(wc * WCO + (lane & 15), lane >> 4);
    int bofs[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int pc = (lane & 15) + d;
        bofs[d] = (wp * ROWS * 18 + pc) * PB + (PIPE ? (((lane >> 4) ^ pswz32(pc)) << 4) : (((lane >> 4) ^ (pc & 7)) << 4));
    }

    // stage = (chunk, tap).  Iteration `it` issues the DMA of stage `it` (+ the X patch of its chunk when tap == 0) and multiplies
    // stage `it - 1`.  Weight-tile slots: two (four: PIPE) recycled ones, or -- p.wpre, single-chunk convs whose S tiles all fit -- one
    // slot per tap step
    // a wave whose output rows all lie below the image (ragged bottom patches: the 8-row maps of stage 4, the (H + 2)-row padded grids of
    // the data gradients) multiplies nothing: it still loads and meets every barrier, but leaves the MFMA pipe to the live waves
    const bool rows_live = oy0 + wp * ROWS < xs.H;
    auto compute = [&](const Cur& c, int slot) {
        if (!rows_live) return;
        const int tap = tap_at(c);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int dy = xs.mode == 2 ? ky : 2 - ky, dx = xs.mode == 2 ? kx : 2 - kx;
        // (integer offsets into the LDS arrays, never pointer <-> integer casts: those make the reads flat_load instead of ds_read.
        // The second K half of a 128-byte row = the first one's offset ^ 64, and adding multiples of 128 commutes with that.)
        const int wo = slot * WBYTES + aofs;
        const int xo = (XBUFS == 2 ? (c.chunk & 1) : 0) * XBYTES + dy * (18 * PB) + (dx == 0 ? bofs[0] : (dx == 1 ? bofs[1] : bofs[2]));
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            bf16x8 a[TC], b[TP];
            const int wk = ks ? (wo ^ 64) : wo, xk = ks ? (xo ^ 64) : xo;
#pragma unroll
            for (int i = 0; i < TC; ++i) a[i] = (HN_DBG(p) & 8) ? zero8() : *reinterpret_cast<const bf16x8*>(sWb + wk + i * 16 * PB);
#pragma unroll
            for (int j = 0; j < TP; ++j) b[j] = (HN_DBG(p) & 8) ? zero8() : *reinterpret_cast<const bf16x8*>(sXb + xk + j * 18 * PB);
            if (HN_DBG(p) & 2) {
#pragma unroll
                for (int i = 0; i < (TC < TP ? TC : TP); ++i) asm volatile("" :: "v"(a[i]), "v"(b[i]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    auto issue_w = [&](const Cur& c, int slot, int seq) {
        if ((HN_DBG(p) & 4) && seq >= WBUFS) return;
        const int k0 = c.chunk * KC;
        const int koff = tap_at(c) * p.KP + k0;                        // (uniform)
        char* sW = sWb + slot * WBYTES;
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            if ((PIPE && !PRE32) || wave * (64 / NPP) + (512 / NPP) * i < BC) {   // wave-uniform (PIPE ring: every wave issues every round, padded tile)
                const bf16* src = (wrow[i] >= 0 && k0 + wsub < p.KP) ? p.w + wrow[i] + koff : g_zero_piece;
                glds16(src, sW + (512 * i + 64 * wave) * 16);
            }
        }
    };
    // phase form with a pre-activation addend (the skip operand's partial result): all TC x TP loads of a lane are requested up front.
    // Issued one by one inside the loop below each was a dependent round trip: the stamps show 26 000 cycles for this epilogue, a
    // quarter of the workgroup's lifetime, against 16 x 3 400 cycles for all of its tap steps (tools/stamp_seg.py).
    const int ox = ox0 + (lane & 15);
    const bool pre_add = !OUT_F32 && p.d2s && p.addend;
    constexpr int AQ = 2;                                             // ring: cout sub-tile i (in use) and i + 1 (in flight); a third slot
    bf16x4 addq[AQ][TP];                                              // spilled 13 VGPRs = +24 MB of scratch traffic per launch
    auto load_add = [&](int i, bf16x4 (&dst)[TP]) {
        const int co0 = c_blk + wc * WCO + i * 16 + (lane >> 4) * 4;
        const int ph = co0 / p.d2s, oc = co0 - ph * p.d2s;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int oy = oy0 + wp * ROWS + j;
            const long opix = ((long)(n * 2 * xs.H + 2 * oy + (ph >> 1)) * (2 * xs.W) + 2 * ox + (ph & 1));
            dst[j] = (oy < xs.H && ox < xs.W && co0 < p.Nout) ? *reinterpret_cast<const bf16x4*>(p.addend + opix * p.ld_add + oc)
                                                              : (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        }
    };
    bool add_issued = false;
    stamp();
    if ((!PIPE || PRE32) && p.wpre) {
        // Single 64-channel chunk and few tap steps (the phase-form output convs: 4 steps): the patch and ALL weight tiles are requested
        // together and the tap loop runs without barriers or DMA waits -- such a workgroup lived for ~14 us of which the four
        // barrier-separated weight-tile round trips were a third.
        Cur ci = cur0();                                                // (the patch was requested in the prologue)
        for (int st = 0; st < S; ++st) { issue_w(ci, st, 0); adv(ci); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();
        Cur cc = cur0();
        if constexpr (PRE32) {                                        // (this instance only: in the others the extra code costs registers --
            if (S == 9 && p.phase_mode == 0) {                        // 22 spilled VGPRs in the 128-cout instance)
#pragma unroll                                                        // nine plain taps, unrolled: a tap's fragment reads under the previous tap's MFMAs
                for (int st = 0; st < 9; ++st) { compute(cc, st); adv(cc); }
            } else {
                for (int st = 0; st < S; ++st) { compute(cc, st); adv(cc); }
            }
        } else {
            for (int st = 0; st < S; ++st) { compute(cc, st); adv(cc); }
        }
    } else if (BC == 64 && !PIPE && xs.diag) {
        // Grouped conv (group width 8) as block-diagonal 64 x 64 tiles: of the 8 KB weight tile of a tap only the eight 8 x 8 diagonal
        // blocks (1 KB) are non-zero.  All nine taps' blocks (9 KB: [tap][cout 64][8 ci]) are fetched ONCE next to the patch, and the A
        // fragments are built from them by a lane select -- the tap loop has no barrier and no DMA wait.  The tile-streaming loop below
        // exposed one weight-DMA round trip per tap: 9 x 2.2 us = 19.8 us per launch on the deep stages, for 1 us of MFMA work.
        // (the nine taps' diagonal blocks were requested at kernel start, the patch in the prologue)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kq = lane >> 4;
        for (int tap = 0; tap < (rows_live ? 9 : 0); ++tap) {
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int dy = xs.mode == 2 ? ky : 2 - ky, dx = xs.mode == 2 ? kx : 2 - kx;
            const int pxk = ((lane & 15) + dx) & 7;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[TC], b[TP];
                const int piece = ks * 4 + kq;
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int col = i * 16 + (lane & 15);              // (BC == 64: one cout group of waves, wc == 0)
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(sWb + (tap * 64 + col) * 16);
                    a[i] = piece == (col >> 3) ? v : zero8();
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int pidx = (wp * ROWS + j + dy) * 18 + (lane & 15) + dx;
                    b[j] = *reinterpret_cast<const bf16x8*>(sXb + pidx * 128 + ((piece ^ pxk) << 4));
                }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    } else if (PIPE) {
        // 32-channel chunks: two patch buffers + a ring of four 8 KB weight tiles = 80 KB: still two workgroups per CU, and now with the
        // loads of the next three tap steps (and of the next chunk's patch) in flight under COUNTED waits -- the two-buffer loop below
        // waits for a DMA round trip (vmcnt(0)) at every tap step and for the patch at every chunk boundary.
        //   iteration i: everything up to W(i) has landed (in-order completion: the loads issued after W(i) may stay in flight);
        //   barrier: every wave is done with step i - 1, so ring slot (i - 1) % R and (at a chunk's first step) the other patch buffer
        //   are free; issue X(chunk + 1) / W(i + R - 1); multiply step i.
        constexpr int R = WBUFS;
        issue_x(0);
        Cur ci = cur0(), cc = cur0();
        int si = 0;                                                    // ring slot of the next weight tile to issue (= its step % R)
        for (int s0 = 0; s0 < R - 1 && s0 < S; ++s0) { issue_w(ci, si, s0); adv(ci); si = si + 1 == R ? 0 : si + 1; }
        unsigned xhist = 0;                                            // bit k: iteration i - 1 - k issued a patch
        int sc = 0;                                                    // ring slot of step i
        for (int i = 0; i < S; ++i) {
            const int nw = min(R - 2, S - 1 - i);                      // weight tiles issued after W(i)
            const int nx = __builtin_popcount(xhist & ((1u << (R - 2)) - 1u));   // patches issued after W(i) (iterations i-R+2 .. i-1)
            switch (nw * WL + nx * XL) {
                case 0: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case WL: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WL) : "memory"); break;
                case 2 * WL: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * WL) : "memory"); break;
                case XL: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(XL) : "memory"); break;
                case WL + XL: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WL + XL) : "memory"); break;
                case 2 * WL + XL: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * WL + XL) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
            }
            stamp();
            const bool xi = cc.ti == 0 && cc.chunk + 1 < nchunk;
            if (xi) issue_x(cc.chunk + 1);                             // its buffer was last read by the previous chunk's last tap
            if (i + R - 1 < S) { issue_w(ci, si, i + R - 1); adv(ci); si = si + 1 == R ? 0 : si + 1; }   // slot last read by step i - 1
            xhist = (xhist << 1) | (xi ? 1u : 0u);
            compute(cc, sc);
            adv(cc);
            sc = sc + 1 == R ? 0 : sc + 1;
        }
    } else {
        Cur ci = cur0(), cc = cur0();
        for (int it = 0; it < S; ++it) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stamp();
            // single patch buffer: at a chunk boundary the last tap of the old chunk is multiplied BEFORE the new patch may overwrite it
            const bool boundary = XBUFS == 1 && it > 0 && ci.ti == 0;
            if (boundary) {
                compute(cc, (it - 1) & 1);
                adv(cc);
                __syncthreads();
            }
            issue_w(ci, it & 1, it);
            if (ci.ti == 0) issue_x(ci.chunk);
            adv(ci);
            if (it > 0 && !boundary) {
                compute(cc, (it - 1) & 1);
                adv(cc);
            }
        }
        // last tap step, peeled: the loaders' registers (patch / weight row offsets) are dead here, which makes room for the addend of the
        // first cout sub-tile -- requested now (HBM: 2-3 us under load), it flies under this step's MFMAs instead of heading the epilogue;
        // the second one follows behind the MFMAs (both at once: 4 spilled VGPRs at BC = 128)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();
        if (pre_add) {
            load_add(0, addq[0]);
            add_issued = true;
        }
        compute(cc, (S - 1) & 1);
        if (pre_add && TC > 1) load_add(1, addq[1]);
    }

    stamp();
    // optional BatchNorm partial statistics of the bf16-rounded outputs: one row per workgroup (patch), psum/psq [gridDim.x / ncy][Nout];
    // wave sums by DPP row rotations, the WGP pixel-row groups are folded through LDS (the operand buffers are free now)
    // (only the grouped convs -- block-diagonal 64-cout tiles -- are launched with statistics rows)
    if (BC == 64 && p.psum) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);                  // [WGP][BC][2]
        const int oxs = ox0 + (lane & 15);
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int cl = wc * WCO + i * 16 + (lane >> 4) * 4;
            const int co0 = c_blk + cl;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            if (rows_live) {                                          // (a row group below the image contributes zeros: no arithmetic)
                const f32x4 bsr = *reinterpret_cast<const f32x4*>(sbias + cl);
                if (p.emode == 0) {                                   // BatchNorm forward statistics: sums and sums of squares, nothing to load
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const bool pv = oy0 + wp * ROWS + j < xs.H && oxs < xs.W;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float q = OUT_F32 ? acc[i][j][r] + bsr[r] : bfround(acc[i][j][r] + bsr[r]);
                            q = pv ? q : 0.f;
                            s1[r] += q;
                            s2[r] += q * q;
                        }
                    }
                } else {
                    const bool ev = co0 + 3 < p.Nout;
                    const StatCoef cf = stat_coef(p.ecoef, p.Nout, p.emode, co0, ev);
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const int oy = oy0 + wp * ROWS + j;
                        const bool pv = oy < xs.H && oxs < xs.W;
                        f32x4 q, z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            q[r] = OUT_F32 ? acc[i][j][r] + bsr[r] : bfround(acc[i][j][r] + bsr[r]);
                            q[r] = pv ? q[r] : 0.f;
                        }
                        if (ev && pv) {
                            const bf16x4 zv = *reinterpret_cast<const bf16x4*>(p.ez + ((long)(n * xs.H + oy) * xs.W + oxs) * p.ld_ez + co0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) z[r] = bf2f(zv[r]);
                        }
                        stat_terms4(p.emode, q, z, cf, s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
            }
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { red[(wp * BC + cl + r) * 2] = s1[r]; red[(wp * BC + cl + r) * 2 + 1] = s2[r]; }
            }
        }
        __syncthreads();
        if (tid < BC && c_blk + tid < p.Nout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 0; k < WGP; ++k) { t1 += red[(k * BC + tid) * 2]; t2 += red[(k * BC + tid) * 2 + 1]; }
            const long prow = patch_id;
            p.psum[prow * p.Nout + c_blk + tid] = t1;
            if (p.psq) p.psq[prow * p.Nout + c_blk + tid] = t2;
        }
    }

    if (HN_DBG(p) & 1) {                                                   // ablation: one store keeps the accumulators alive
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) reinterpret_cast<float*>(p.out)[0] = t;
        return;
    }
    // epilogue: bias, activation (one uniform branch per 4 values, in place on the accumulators: no second copy of the wave tile in
    // registers), store 4 consecutive couts per lane
    // phase form, bf16: the depth-to-space scatter writes 8 bytes per lane 1 KiB apart; partial-line writes make L2 fetch every output
    // line before merging (measured: FETCH_SIZE +124 MB, WRITE_SIZE +28 MB on a 67 MB output).  The finished bf16 tile is staged in LDS
    // ([256 px][BC], 16-byte pieces XOR-swizzled by the pixel) and leaves as whole BC*2-byte runs per output pixel.
    // The same holds for every bf16 output of this kernel (8 bytes per lane, one row stride apart): all of them are staged.
    const bool stage_d2s = !OUT_F32 && (p.ldc & 7) == 0 && (p.Nout & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && BC >= 64 &&
                           (!p.d2s || ((p.d2s & 7) == 0 && c_blk + BC <= p.Nout));
    // fp32 depth-to-space output (the 5-class logits): 4-byte stores scattered over four output pixels per lane; the 32 x 32-pixel x k
    // output tile is assembled in LDS and written as contiguous rows of 32*k floats
    const bool stage_f32 = OUT_F32 && p.d2s && c_blk == 0 && p.Nout <= BC && 32 * 32 * p.d2s * 4 <= 32768 && ((2 * xs.W * p.d2s) & 3) == 0 &&
                           ((32 * p.d2s) & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
    char* stage = smem;
    if (stage_d2s || stage_f32) __syncthreads();                      // every wave is done with the operand buffers
    if (pre_add && !add_issued) {                                      // (loop forms without the early request)
#pragma unroll
        for (int q = 0; q < AQ && q < TC; ++q) load_add(q, addq[q]);
    }
    if constexpr (BC >= 64 && !OUT_F32) {
        // (the host entry point only launches these instantiations when the staged form applies: stage_d2s is true)
        // Compact form of the staged bf16 epilogue (every seg-decoder launch with >= 64 couts per tile): per cout sub-tile, bias and addend
        // are added in place on the accumulators, the activation is one uniform branch around its 16 values, then the sub-tile is rounded
        // and staged.  The generic epilogue below re-decides activation and store form inside each of its TC x TP unrolled bodies: 10 000
        // instructions (80 KB: more than the instruction cache two CUs share), and the stamps showed 17 000 cycles for an epilogue
        // without a single global load (tools/stamp_seg.py) -- instruction fetch, not arithmetic.
        const int act = p.act;
#pragma unroll
        for (int i = 0; i < (rows_live ? TC : 0); ++i) {              // (a row group below the image stages nothing: the write-out masks its pixels)
            // one cout sub-tile at a time, start to finish: its 16 accumulator registers are dead once it is staged (all 64 values
            // through bias / addend, then all through the activation, then all staged kept everything live at once: 13 spilled VGPRs)
            const f32x4 bs = *reinterpret_cast<const f32x4*>(sbias + wc * WCO + i * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] += bs;
            if (pre_add) {
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += bf2f(addq[i % AQ][j][r]);
                if (i + AQ < TC) load_add(i + AQ, addq[i % AQ]);
            }
            if (act == HN_ACT_ELU) {
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float x = acc[i][j][r]; acc[i][j][r] = x > 0.f ? x : (__expf(x) - 1.0f); }
            } else if (act != HN_ACT_NONE) {
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {                          // (rare: any other activation; the lane's 4 values rotate through element 0)
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const float x = act_fwd(acc[i][j][0], act);
                        acc[i][j] = (f32x4){acc[i][j][1], acc[i][j][2], acc[i][j][3], x};
                    }
                }
            }
            const int cl = wc * WCO + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int pl = (wp * ROWS + j) * 16 + (lane & 15);
                const bf16x4 tv = {f2bf(acc[i][j][0]), f2bf(acc[i][j][1]), f2bf(acc[i][j][2]), f2bf(acc[i][j][3])};
                *reinterpret_cast<bf16x4*>(stage + pl * (BC * 2) + ((((cl >> 3) ^ pl) & (BC / 8 - 1)) << 4) + (cl & 7) * 2) = tv;
            }
        }
        stamp();
    } else if (OUT_F32 && stage_f32) {
        // Compact form of the staged fp32 depth-to-space epilogue (the seg output conv: logits or their arg-max).  The generic loop below
        // decides the store form per value, divides by d2s per value and loads the bias from global memory: the stamps showed 10 800 of a
        // workgroup's 32 500 cycles in it (tools/stamp_seg.py LAYER=out).  Here: bias from LDS, ONE division per cout sub-tile (the
        // lane's four couts advance through (phase, channel) by increment), scalar LDS stores into the 32 x 32-pixel x d2s tile.
        const int d2s = p.d2s, rowf = 32 * d2s;
        float* stf = reinterpret_cast<float*>(stage);
#pragma unroll
        for (int i = 0; i < (rows_live ? TC : 0); ++i) {
            const int cl = wc * WCO + i * 16 + (lane >> 4) * 4, co0 = c_blk + cl;
            const f32x4 bs = *reinterpret_cast<const f32x4*>(sbias + cl);
            int ph = co0 / d2s, oc = co0 - ph * d2s;
            int off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                off[r] = co0 + r < p.Nout ? (ph >> 1) * rowf + (ph & 1) * d2s + oc : -1;
                if (++oc == d2s) { oc = 0; ++ph; }
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int base = 2 * (wp * ROWS + j) * rowf + 2 * (lane & 15) * d2s;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bs[r];
                act_fwd_n(v, p.act);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (off[r] >= 0) stf[base + off[r]] = v[r];
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int co0 = c_blk + wc * WCO + i * 16 + (lane >> 4) * 4;
        float bsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bsv[r] = (p.bias && co0 + r < p.Nout) ? p.bias[co0 + r] : 0.f;
        bf16x4 addv[TP];
        if (pre_add) {
#pragma unroll
            for (int j = 0; j < TP; ++j) addv[j] = addq[i % AQ][j];
            if (i + AQ < TC) load_add(i + AQ, addq[i % AQ]);
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int oy = oy0 + wp * ROWS + j;
            if (oy >= xs.H || ox >= xs.W) continue;
            const long orow = ((long)(n * xs.H + oy) * xs.W + ox) * p.ldc;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bsv[r];
            if (!OUT_F32 && p.d2s) {
                // phase form, bf16: the 4 couts of a lane belong to one phase; output pixel (2y+py, 2x+px), d2s channels per pixel;
                // the partial result of the full-resolution (skip connection) operand arrives as a pre-activation addend
                if (co0 < p.Nout) {
                    const int ph = co0 / p.d2s, oc = co0 - ph * p.d2s;
                    const long opix = ((long)(n * 2 * xs.H + 2 * oy + (ph >> 1)) * (2 * xs.W) + 2 * ox + (ph & 1));
                    if (p.addend) {
                        // (plain loads: the four cout sub-tiles of a lane touch the same 128-byte lines one after the other -- a
                        // non-temporal load re-fetched them from memory every time: 2x FETCH_SIZE, 128 -> 178 us)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += bf2f(addv[j][r]);
                    }
                    act_fwd_n(v, p.act);
                    bf16x4 tv = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    if (stage_d2s) {
                        const int pl = (wp * ROWS + j) * 16 + (lane & 15), cl = wc * WCO + i * 16 + (lane >> 4) * 4;
                        *reinterpret_cast<bf16x4*>(stage + pl * (BC * 2) + ((((cl >> 3) ^ pl) & (BC / 8 - 1)) << 4) + (cl & 7) * 2) = tv;
                    } else {
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.out) + opix * p.ldc + oc) = tv;
                    }
                }
                continue;
            }
            act_fwd_n(v, p.act);
            if (OUT_F32 && p.d2s) {
                float* ob = reinterpret_cast<float*>(p.out);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = co0 + r;
                    if (c < p.Nout) {
                        const int ph = c / p.d2s, oc = c - ph * p.d2s;
                        if (stage_f32) {                               // tile-local: row 2*ly + py, column (2*lx + px) * k + oc
                            const int ly = wp * ROWS + j, lx = lane & 15;
                            reinterpret_cast<float*>(stage)[(2 * ly + (ph >> 1)) * (32 * p.d2s) + (2 * lx + (ph & 1)) * p.d2s + oc] = v[r];
                        } else {
                            ob[((long)(n * 2 * xs.H + 2 * oy + (ph >> 1)) * (2 * xs.W) + 2 * ox + (ph & 1)) * p.d2s + oc] = v[r];
                        }
                    }
                }
            } else if (OUT_F32) {
                float* o = reinterpret_cast<float*>(p.out) + orow + co0;
                if (co0 + 3 < p.Nout && (reinterpret_cast<uintptr_t>(o) & 15) == 0) *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (co0 + r < p.Nout) o[r] = v[r];
                }
            } else {
                bf16* o = reinterpret_cast<bf16*>(p.out) + orow + co0;
                if (stage_d2s) {
                    bf16x4 tv = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    const int pl = (wp * ROWS + j) * 16 + (lane & 15), cl = wc * WCO + i * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<bf16x4*>(stage + pl * (BC * 2) + ((((cl >> 3) ^ pl) & (BC / 8 - 1)) << 4) + (cl & 7) * 2) = tv;
                } else if (co0 + 3 < p.Nout && (reinterpret_cast<uintptr_t>(o) & 7) == 0) {
                    bf16x4 tv = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    *reinterpret_cast<bf16x4*>(o) = tv;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (co0 + r < p.Nout) o[r] = f2bf(v[r]);
                }
            }
        }
    }
    if (p.amax && !stage_f32) return;                                // (the host entry point guarantees the staged path)
    if (stage_f32 && p.amax) {
        __syncthreads();
        const int rowf = 32 * p.d2s;
        for (int idx = tid; idx < 32 * 32; idx += 512) {              // one output pixel per thread: arg-max of its d2s logits in the tile
            const int Y = idx >> 5, X = idx & 31;
            const int gy = 2 * oy0 + Y, gx = 2 * ox0 + X;
            if (gy < 2 * xs.H && gx < 2 * xs.W) {
                const float* r = reinterpret_cast<const float*>(stage) + Y * rowf + X * p.d2s;
                float best = r[0];
                int arg = 0;
                for (int c = 1; c < p.d2s; ++c)
                    if (r[c] > best) { best = r[c]; arg = c; }
                p.amax[(long)(n * 2 * xs.H + gy) * (2 * xs.W) + gx] = arg;
            }
        }
    } else if (stage_f32) {
        __syncthreads();
        const int rowf = 32 * p.d2s, row4 = rowf >> 2;                // floats / float4 per tile row
        float* ob = reinterpret_cast<float*>(p.out);
        for (int idx = tid; idx < 32 * row4; idx += 512) {
            const int Y = idx / row4, q4 = idx - Y * row4;
            const int gy = 2 * oy0 + Y;
            const int valid = (2 * xs.W - 2 * ox0) * p.d2s;           // floats of this row that lie inside the image
            if (gy < 2 * xs.H && q4 * 4 < valid) {
                float* dst = ob + ((long)(n * 2 * xs.H + gy) * (2 * xs.W) + 2 * ox0) * p.d2s + q4 * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(stage) + Y * rowf + q4 * 4);
                if (q4 * 4 + 4 <= valid) *reinterpret_cast<f32x4*>(dst) = v;
                else for (int k = 0; q4 * 4 + k < valid; ++k) dst[k] = v[k];
            }
        }
    }
    stamp();
    if (stage_d2s) {
        __syncthreads();
        constexpr int NPC = BC / 8;                                   // 16-byte pieces per output pixel
        const int ph = p.d2s ? c_blk / p.d2s : 0, oc0 = p.d2s ? c_blk - ph * p.d2s : c_blk;    // (d2s: the whole cout tile lies in one phase)
        bf16* outp = reinterpret_cast<bf16*>(p.out);
        if (p.fold) {
            constexpr int ITER = 256 * NPC / 512;
            const int H = xs.H - 2, W = xs.W - 2;
            const int pc = tid % NPC, c = c_blk + pc * 8;             // (512 % NPC == 0: a thread keeps its 8-channel piece)
            long dst[ITER];                                           // element offset in out (>= 0) or -(ring offset) - 1; LONG_MIN: nothing
            bf16x8 yv[ITER];
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int pl = (tid + 512 * i) / NPC;
                const int oy = oy0 + (pl >> 4), oxx = ox0 + (pl & 15);
                const int iy = oy - 1, ix = oxx - 1;
                dst[i] = LONG_MIN;
                yv[i] = zero8();
                if (oy < xs.H && oxx < xs.W && c < p.Nout) {
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                        const long pix = (long)(n * H + iy) * W + ix;
                        const long cell = (long)(n * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1);
                        const int pch = ((iy & 1) * 2 + (ix & 1)) * p.Nout + c;
                        dst[i] = p.fold == 2 ? cell * p.ldc + pch : pix * p.ldc + c;
                        if (p.fold_y) yv[i] = ld8(p.fold_y + pix * p.ld_fy + c);
                    } else {
                        const int r = oy == 0 ? oxx : (oy == H + 1 ? xs.W + oxx : (oxx == 0 ? 2 * xs.W + iy : 2 * xs.W + H + iy));
                        dst[i] = -(((long)n * (2 * xs.W + 2 * H) + r) * p.Nout + c) - 1;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                if (dst[i] == LONG_MIN) continue;
                const int pl = (tid + 512 * i) / NPC;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + pl * (BC * 2) + (((pc ^ pl) & (NPC - 1)) << 4));
                if (dst[i] >= 0) {
                    if (p.fold_y) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { const float yy = bf2f(yv[i][k]), f = bf2f(v[k]); v[k] = f2bf(yy > 0.f ? f : f * (yy + 1.0f)); }
                    }
                    *reinterpret_cast<bf16x8*>(outp + dst[i]) = v;
                    if (p.fold == 3) {                                 // (offset recomputed: eight more 64-bit registers would spill)
                        const int iy = oy0 + (pl >> 4) - 1, ix = ox0 + (pl & 15) - 1;
                        const long cell = (long)(n * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1);
                        *reinterpret_cast<bf16x8*>(p.fold_out2 + cell * p.ld_fo2 + ((iy & 1) * 2 + (ix & 1)) * p.Nout + c) = v;
                    }
                } else {
                    *reinterpret_cast<bf16x8*>(p.ring - dst[i] - 1) = v;
                }
            }
            return;
        }
#pragma unroll 2
        for (int idx = tid; idx < 256 * NPC; idx += 512) {
            const int pl = idx / NPC, pc = idx - pl * NPC;
            const int oy = oy0 + (pl >> 4), oxx = ox0 + (pl & 15);
            if (oy < xs.H && oxx < xs.W && c_blk + pc * 8 < p.Nout) {
                const long opix = p.d2s ? ((long)(n * 2 * xs.H + 2 * oy + (ph >> 1)) * (2 * xs.W) + 2 * oxx + (ph & 1))
                                        : ((long)(n * xs.H + oy) * xs.W + oxx);
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + pl * (BC * 2) + (((pc ^ pl) & (NPC - 1)) << 4));
                *reinterpret_cast<bf16x8*>(outp + opix * p.ldc + oc0 + pc * 8) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();
    }
}

// ---------------------------------------------------------------------------------------------------------
// The seg head's output conv (64 channels -> 4 phases x k <= 8 classes on the low-resolution grid, replicate padding; fp32 logits in
// depth-to-space order, or their arg-max) as a PERSISTENT launch with the weights in registers (round 6).  Through conv3x3_direct_kernel<32>
// the layer is 8 192 workgroups of ~30 us whose life is 25 % prologue, 50 % nine tap steps that each wait for a 4 KB weight tile, and
// 25 % write-out, for 72 MFMAs per wave.  Here: one 512-thread workgroup per CU walks its share of the 16 x 16-pixel patches; the 36 weight
// fragments of a wave (2 cout tiles x 9 taps x 2 K halves = 144 registers) are loaded ONCE; the 18 x 18 x 64-channel patch of the NEXT
// iteration lands by LDS-DMA in the second buffer while this one is multiplied (no tap-step barriers at all); the 32 x 32 x k fp32 tile
// is staged and leaves as whole rows.  Same products in the same order as the direct kernel (bit-identical logits).
// ---------------------------------------------------------------------------------------------------------
struct SegOut { const bf16* x; int ldx, N, H, W; const bf16* w; int Nout, k; const float* bias; float* out; long* amax; int npatch, ppw; };
template <bool AMAX>
__global__ __launch_bounds__(512, 2) void seg_out_conv_kernel(const SegOut p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPIX = 18 * 18, XCH = (PPIX * 8 + 63) / 64, XBYTES = XCH * 1024;     // 41 one-KB DMA chunks per patch buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* stf = reinterpret_cast<float*>(smem + 2 * XBYTES);          // [32][32 * k] fp32 output tile
    const int k = p.k, rowf = 32 * k;
    const int tx_n = (p.W + 15) >> 4, ty_n = (p.H + 15) >> 4;
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);
    const int t0 = bidx * p.ppw;
    int t1 = t0 + p.ppw;
    if (t1 > p.npatch) t1 = p.npatch;
    if (t0 >= t1) return;
    // weight fragments: A operand of v_mfma_f32_16x16x32_bf16 = 16 couts x 32 channels, lane = (cout l & 15, 8 channels (l >> 4) * 8)
    bf16x8 wf[2][9][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int co = i * 16 + (lane & 15);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[i][tap][ks] = co < p.Nout ? *reinterpret_cast<const bf16x8*>(p.w + ((long)co * 9 + tap) * 64 + ks * 32 + (lane >> 4) * 8) : zero8();
    }
    float bs[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = i * 16 + (lane >> 4) * 4 + r;
            bs[i][r] = (p.bias && co < p.Nout) ? p.bias[co] : 0.f;
        }
    // the thread's DMA requests, resolved once: request i covers piece e = (wave + 8 i) 64 + lane of the patch = (pixel e >> 3, PHYSICAL
    // piece e & 7); the logical piece is physical ^ (patch column & 7) -- the swizzle the fragment reads undo
    int req[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = (wave + 8 * i) * 64 + lane, pp = e >> 3;
        const int py = pp / 18, px = pp - py * 18;
        req[i] = (wave + 8 * i < XCH && pp < PPIX) ? ((py << 16) | (px << 8) | ((e & 7) ^ (px & 7))) : -1;
    }
    auto issue = [&](int t, char* sX) {
        const int tx = t % tx_n, t2 = t / tx_n, ty = t2 % ty_n, n = t2 / ty_n;
        const int oy0 = ty * 16, ox0 = tx * 16;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (wave + 8 * i < XCH) {                                  // (wave-uniform)
                const int rq = req[i];
                int cy = oy0 - 1 + (rq >> 16), cx = ox0 - 1 + ((rq >> 8) & 255);
                cy = cy < 0 ? 0 : (cy >= p.H ? p.H - 1 : cy);             // replicate padding = clamped source coordinates
                cx = cx < 0 ? 0 : (cx >= p.W ? p.W - 1 : cx);
                const bf16* src = rq >= 0 ? p.x + ((long)(n * p.H + cy) * p.W + cx) * p.ldx + (rq & 255) * 8 : g_zero_piece;
                glds16(src, sX + (wave + 8 * i) * 1024);
            }
        }
    };
    // B fragment offsets: pixel (patch row 2 wave + j + ky, column (l & 15) + kx), 8 channels (l >> 4) + 4 ks
    int bofs[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int pc = (lane & 15) + d;
        bofs[d] = (wave * 2 * 18 + pc) * 128 + (((lane >> 4) ^ (pc & 7)) << 4);
    }
    issue(t0, smem);
    for (int t = t0; t < t1; ++t) {
        const int slot = (t - t0) & 1;
        // (vmcnt(0) also waits for the previous tile's stores; counted waits that leave them in flight measured the same: the launch moves
        // 1.27 x the input + the fp32 logits at ~4.1 TB/s)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // patch t landed; the other buffer and the output tile are free
        if (t + 1 < t1) issue(t + 1, smem + (slot ^ 1) * XBYTES);
        const char* sX = smem + slot * XBYTES;
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int xo = ky * (18 * 128) + bofs[kx];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int xk = ks ? (xo ^ 64) : xo;
                bf16x8 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sX + xk + j * 18 * 128);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][tap][ks], b[j], acc[i][j], 0, 0, 0);
            }
        }
        // depth-to-space staging: cout c = (phase, class) -> output pixel (2 ly + py, 2 lx + px), class
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co0 = i * 16 + (lane >> 4) * 4;
            int ph = co0 / k, oc = co0 - ph * k;
            int off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                off[r] = co0 + r < p.Nout ? (ph >> 1) * rowf + (ph & 1) * k + oc : -1;
                if (++oc == k) { oc = 0; ++ph; }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int base = 2 * (wave * 2 + j) * rowf + 2 * (lane & 15) * k;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (off[r] >= 0) stf[base + off[r]] = acc[i][j][r] + bs[i][r];
            }
        }
        __syncthreads();
        const int tx = t % tx_n, t2 = t / tx_n, ty = t2 % ty_n, n = t2 / ty_n;
        const int oy0 = ty * 16, ox0 = tx * 16;
        if (AMAX) {
            for (int idx = tid; idx < 32 * 32; idx += 512) {           // one output pixel per thread: arg-max of its k logits (first maximum wins)
                const int Y = idx >> 5, X = idx & 31;
                const int gy = 2 * oy0 + Y, gx = 2 * ox0 + X;
                if (gy < 2 * p.H && gx < 2 * p.W) {
                    const float* r = stf + Y * rowf + X * k;
                    float best = r[0];
                    int arg = 0;
                    for (int c = 1; c < k; ++c)
                        if (r[c] > best) { best = r[c]; arg = c; }
                    p.amax[(long)(n * 2 * p.H + gy) * (2 * p.W) + gx] = arg;
                }
            }
        } else {
            const int row4 = rowf >> 2;                                 // float4 per tile row (32 k % 4 == 0)
            for (int idx = tid; idx < 32 * row4; idx += 512) {
                const int Y = idx / row4, q4 = idx - Y * row4;
                const int gy = 2 * oy0 + Y;
                const int valid = (2 * p.W - 2 * ox0) * k;             // floats of this row that lie inside the image
                if (gy < 2 * p.H && q4 * 4 < valid) {
                    float* dst = p.out + ((long)(n * 2 * p.H + gy) * (2 * p.W) + 2 * ox0) * k + q4 * 4;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stf + Y * rowf + q4 * 4);
                    if (q4 * 4 + 4 <= valid) *reinterpret_cast<f32x4*>(dst) = v;
                    else for (int c = 0; q4 * 4 + c < valid; ++c) dst[c] = v[c];
                }
            }
        }
    }
}

// The same persistent form for the last decoder block's phase-form conv (64 channels -> 4 phases x 64 couts on the low-resolution grid,
// 4 of the 9 effective taps per phase, bias + ELU, bf16 depth-to-space output): through conv3x3_direct_kernel<64> it is 8 192 workgroups
// of 64 MFMAs per wave each (180 us for 335 MB and 69 GFLOP).  A workgroup owns ONE phase (its 4 x 4 x 2 weight fragments = 128
// registers) and walks the patches; the four phase workgroups of a patch range are neighbours on one XCD (the patch comes from HBM once).
struct SegPhase { const bf16* x; int ldx, N, H, W; const bf16* w; const float* bias; bf16* out; int ldc, act, npatch, ppw; };
__global__ __launch_bounds__(512, 2) void seg_phase64_conv_kernel(const SegPhase p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPIX = 18 * 18, XCH = (PPIX * 8 + 63) / 64, XBYTES = XCH * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* stage = smem + 2 * XBYTES;                                   // [256 px][64 couts] bf16, 16-byte pieces XOR-swizzled by the pixel
    const int tx_n = (p.W + 15) >> 4, ty_n = (p.H + 15) >> 4;
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);
    const int ph = bidx & 3, py = ph >> 1, px = ph & 1;
    const int t0 = (bidx >> 2) * p.ppw;
    int t1 = t0 + p.ppw;
    if (t1 > p.npatch) t1 = p.npatch;
    if (t0 >= t1) return;
    bf16x8 wf[4][4][2];                                                // [cout tile][tap of the phase][K half]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = ph * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tap = (py + (t >> 1)) * 3 + px + (t & 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[i][t][ks] = *reinterpret_cast<const bf16x8*>(p.w + ((long)co * 9 + tap) * 64 + ks * 32 + (lane >> 4) * 8);
        }
    }
    f32x4 bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[i][r] = p.bias ? p.bias[ph * 64 + i * 16 + (lane >> 4) * 4 + r] : 0.f;
    int req[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = (wave + 8 * i) * 64 + lane, pp = e >> 3;
        const int qy = pp / 18, qx = pp - qy * 18;
        req[i] = (wave + 8 * i < XCH && pp < PPIX) ? ((qy << 16) | (qx << 8) | ((e & 7) ^ (qx & 7))) : -1;
    }
    auto issue = [&](int t, char* sX) {
        const int tx = t % tx_n, t2 = t / tx_n, ty = t2 % ty_n, n = t2 / ty_n;
        const int oy0 = ty * 16, ox0 = tx * 16;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (wave + 8 * i < XCH) {
                const int rq = req[i];
                int cy = oy0 - 1 + (rq >> 16), cx = ox0 - 1 + ((rq >> 8) & 255);
                cy = cy < 0 ? 0 : (cy >= p.H ? p.H - 1 : cy);
                cx = cx < 0 ? 0 : (cx >= p.W ? p.W - 1 : cx);
                const bf16* src = rq >= 0 ? p.x + ((long)(n * p.H + cy) * p.W + cx) * p.ldx + (rq & 255) * 8 : g_zero_piece;
                glds16(src, sX + (wave + 8 * i) * 1024);
            }
        }
    };
    int bofs[2];                                                       // the phase's two tap columns
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int pc = (lane & 15) + px + d;
        bofs[d] = ((wave * 2 + py) * 18 + pc) * 128 + (((lane >> 4) ^ (pc & 7)) << 4);
    }
    issue(t0, smem);
    for (int t = t0; t < t1; ++t) {
        const int slot = (t - t0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t1) issue(t + 1, smem + (slot ^ 1) * XBYTES);
        const char* sX = smem + slot * XBYTES;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const int xo = (tq >> 1) * (18 * 128) + bofs[tq & 1];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int xk = ks ? (xo ^ 64) : xo;
                bf16x8 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sX + xk + j * 18 * 128);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][tq][ks], b[j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] += bs[i];
            if (p.act == HN_ACT_ELU) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; acc[i][j][r] = v > 0.f ? v : (__expf(v) - 1.0f); }
            }
            const int cl = i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pl = (wave * 2 + j) * 16 + (lane & 15);
                const bf16x4 tv = {f2bf(acc[i][j][0]), f2bf(acc[i][j][1]), f2bf(acc[i][j][2]), f2bf(acc[i][j][3])};
                *reinterpret_cast<bf16x4*>(stage + pl * 128 + ((((cl >> 3) ^ pl) & 7) << 4) + (cl & 7) * 2) = tv;
            }
        }
        __syncthreads();
        const int tx = t % tx_n, t2 = t / tx_n, ty = t2 % ty_n, n = t2 / ty_n;
        const int oy0 = ty * 16, ox0 = tx * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 512 * u;
            const int pl = idx >> 3, pc = idx & 7;
            const int oy = oy0 + (pl >> 4), oxx = ox0 + (pl & 15);
            if (oy < p.H && oxx < p.W) {
                const long opix = (long)(n * 2 * p.H + 2 * oy + py) * (2 * p.W) + 2 * oxx + px;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + pl * 128 + (((pc ^ pl) & 7) << 4));
                *reinterpret_cast<bf16x8*>(p.out + opix * p.ldc + pc * 8) = v;
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
    int gy;           // number of cout tiles
    int phase_span;   // patch wgrad of a phase-form conv (see GemmNT::phase_mode): couts per phase; the cout tile's phase only has
                      // non-zero effective weights on 4 of the 9 taps, the other five are skipped (their slab entries stay zero)
    float* bias_part; // patch wgrad only (optional): [slabs][Nout] per-slab column sums of dZ (= the conv's bias gradient), accumulated by
                      // one extra MFMA per k-step against an all-ones operand while the dZ fragments are in registers anyway
    int dbg;          // tools/ only (hn_debug_knob 9): 1 = skip the epilogue stores, 2 = skip the MFMAs, 4 = skip the loads
    int out_ld;       // gemm_tn: 0 = partial slabs [split][Nout][taps*KP]; > 0 = ONE split writing the gradient itself, row stride out_ld
    int cin_lim;      //          (= Cin of a 1x1 conv: dw[co][ci]), columns >= cin_lim (the K padding) are dropped
};

// LDS image of a pixel-major tile: rows of COLS bf16, UNPADDED (LDS-DMA writes lane-linear 1 KiB runs), 16-byte pieces XOR-swizzled so
// that the 8 rows x 32 B a half-wave touches in one ds_read_b64_tr_b16 cover all 64 banks exactly once:
//   physical piece = piece ^ ((((row & 7) / (16 / NP)) << 1) & (NP - 1)),  NP = COLS / 8 pieces per row.
template <int COLS>
__device__ __forceinline__ int tn_swz(int row, int piece) {
    constexpr int NP = COLS / 8, RPL = NP >= 16 ? 1 : 16 / NP;      // (rows of >= 256 B alias the same banks: one 32-byte shift per row)
    return piece ^ ((((row & 7) / RPL) << 1) & (NP - 1));
}

template <int BC, int BN, int WGC, int WGN>
__device__ __forceinline__ void gemm_tn_body(const GemmTN& p, const int lid) {   // lid: logical block id inside this GEMM (see below)
    constexpr int WC = BC / WGC, WN = BN / WGN, TC = WC / 16, TN = WN / 16;
    constexpr int ZPR = BC / 8, XPR = BN / 8;                     // 16-byte pieces per row
    constexpr int NT = 64 * WGC * WGN;                             // 4 waves, or 8 for the largest tile
    constexpr int ZL = (64 * ZPR + NT - 1) / NT, XL = (64 * XPR + NT - 1) / NT;
    constexpr int ZB = 64 * BC * 2, XB = 64 * BN * 2, STAGE = ZB + XB;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two stages of [dZ tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: M0 values, LDS bases and wave-uniform branches off the vector unit)
    const int wc = wave / WGN, wn = wave % WGN;
    const int ntile = (p.KP + BN - 1) / BN;
    // logical block id: (tap, ci tile) fastest, then cout tile, then pixel split -- the blocks of one split share dZ / X rows in one L2
    const int gx = ntile * p.taps;
    const int bx = lid % gx, by = (lid / gx) % p.gy, bz = lid / (gx * p.gy);
    const int tap = bx / ntile;
    const int ci_blk = (bx - tap * ntile) * BN;
    const int c_blk = by * BC;
    const long m_begin = (long)bz * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.x.M) m_end = p.x.M;
    const int S = m_end > m_begin ? (int)((m_end - m_begin + 63) >> 6) : 0;

    f32x4 acc[TC][TN];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient (column sums of dZ): the (tap 0, first ci tile) workgroup's wn == 0 waves multiply their dZ fragments with ones
    const bool do_bias = p.bias_part && bx == 0 && wn == 0;
    f32x4 accb[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (bf16)1.0f;

    // per-piece pixel coordinates, advanced by 64 rows per stage without divisions
    int pn[XL], py[XL], px[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) decomp_row(p.x, m_begin + (tid + NT * i) / XPR, pn[i], py[i], px[i]);
    const int adv_q = p.x.mode ? 64 / p.x.W : 0, adv_r = p.x.mode ? 64 % p.x.W : 0;
    const int Ctot = p.x.C0 + p.x.C1;
    // mode 2 (3x3 reflect + up + concat), this block's tap: source coordinates are separable, so they are tabulated once per block:
    //   ty0[oy], tx0[ox] = coordinates in x0's grid (after reflection and >> up), ty1/tx1 = coordinates in x1's (full-res) grid
    int* ty0 = reinterpret_cast<int*>(smem + 2 * STAGE);
    int* tx0 = ty0 + p.x.H;
    int* ty1 = tx0 + p.x.W;
    int* tx1 = ty1 + p.x.H;
    if (p.x.mode == 2) {
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        for (int i = tid; i < p.x.H; i += NT) {
            int iy = i + ky - 1;
            iy = border_idx(iy, p.x.Hi, p.x.clamp);
            ty1[i] = iy;
            ty0[i] = iy >> p.x.up;
        }
        for (int i = tid; i < p.x.W; i += NT) {
            int ix = i + kx - 1;
            ix = border_idx(ix, p.x.Wi, p.x.clamp);
            tx1[i] = ix;
            tx0[i] = ix >> p.x.up;
        }
    }
    const int hh0 = p.x.Hi >> p.x.up, ww0 = p.x.Wi >> p.x.up;

    // transposed-read lane addressing: group g = lane>>4 owns k rows {s*16 + g*4 + q}; lane 4q+pp supplies row q, cols 4pp..4pp+3
    const int g = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
    typedef __bf16 trv4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
    typedef __attribute__((address_space(3))) trv4* lds_b4;

    // per-piece source offsets, advanced by 64 rows per stage (the 1x1 wgrads are most of the launches: no index arithmetic in the loop)
    long zoff[ZL], xoff[XL];
    int zrow[ZL], xrow[XL];
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
        const int e = tid + NT * i;
        const int row = e / ZPR, cp = tn_swz<BC>(row, e - row * ZPR);
        const int co = c_blk + cp * 8;
        zrow[i] = row;
        zoff[i] = co < p.Nout ? (m_begin + row) * (long)p.ldz + co : -1;
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int e = tid + NT * i;
        const int row = e / XPR, cp = tn_swz<BN>(row, e - row * XPR);
        const int c = ci_blk + cp * 8;
        xrow[i] = row;
        xoff[i] = (p.x.mode == 0 && c < Ctot) ? (m_begin + row) * (long)p.x.ld0 + c : -1;
    }
    // per-lane LDS offsets of the transposed fragment reads (stage 0, k rows 0..15 of the 32-row half; +16 rows / +32 rows are constants:
    // the swizzle key row & 7 does not change with them)
    int zofs[TC], xofs[TN];
    {
        const int r0_ = g * 4 + q;
#pragma unroll
        for (int i = 0; i < TC; ++i) zofs[i] = r0_ * (BC * 2) + tn_swz<BC>(r0_, (wc * WC + i * 16) / 8 + (pp >> 1)) * 16 + (pp & 1) * 8;
#pragma unroll
        for (int j = 0; j < TN; ++j) xofs[j] = r0_ * (BN * 2) + tn_swz<BN>(r0_, (wn * WN + j * 16) / 8 + (pp >> 1)) * 16 + (pp & 1) * 8;
    }
    for (int it = 0; it <= S; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's LDS-DMA of stage it-1 has landed ...
        __syncthreads();                                           // ... and so has everybody else's; buffer it&1 is free again
        if (it < S && !(HN_DBG(p) & 4)) {
            char* sZ = smem + (it & 1) * STAGE;
            char* sX = sZ + ZB;
            const long m0 = m_begin + (long)it * 64;
#pragma unroll
            for (int i = 0; i < ZL; ++i) {
                if (NT * i + 64 * wave < 64 * ZPR) {              // wave-uniform: this 1 KiB run lies inside the tile
                    // dZ rows are zero padded up to ldz (>= Nout rounded up to 8), so a piece that starts below Nout is readable
                    const bf16* src = (m0 + zrow[i] < m_end && zoff[i] >= 0) ? p.dz + zoff[i] : g_zero_piece;
                    glds16(src, sZ + (NT * i + 64 * wave) * 16);
                    zoff[i] += zoff[i] >= 0 ? 64L * p.ldz : 0;
                }
            }
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                if (NT * i + 64 * wave < 64 * XPR) {
                    if (p.x.mode == 0) {                           // plain rows: pointer walk
                        const bf16* src0 = (m0 + xrow[i] < m_end && xoff[i] >= 0) ? p.x.x0 + xoff[i] : g_zero_piece;
                        glds16(src0, sX + (NT * i + 64 * wave) * 16);
                        xoff[i] += xoff[i] >= 0 ? 64L * p.x.ld0 : 0;
                        continue;
                    }
                    const int e = tid + NT * i;
                    const int row = e / XPR, cp = tn_swz<BN>(row, e - row * XPR);
                    const long m = m0 + row;
                    const int c = ci_blk + cp * 8;
                    const bf16* src = g_zero_piece;
                    if (m < m_end && c < Ctot) {
                        if (p.x.mode == 2) {
                            if (c < p.x.C0) src = p.x.x0 + c + (long)((pn[i] * hh0 + ty0[py[i]]) * ww0 + tx0[px[i]]) * p.x.ld0;
                            else src = p.x.x1 + (c - p.x.C0) + (long)((pn[i] * p.x.Hi + ty1[py[i]]) * p.x.Wi + tx1[px[i]]) * p.x.ld1;
                        } else {
                            const long off = pixel_off(p.x, m, pn[i], py[i], px[i], tap, 0);
                            if (off >= 0) src = p.x.x0 + c + off;
                        }
                    }
                    glds16(src, sX + (NT * i + 64 * wave) * 16);
                    if (p.x.mode) {                                // advance this piece's pixel by 64 rows
                        px[i] += adv_r;
                        py[i] += adv_q;
                        if (px[i] >= p.x.W) { px[i] -= p.x.W; ++py[i]; }
                        while (py[i] >= p.x.H) { py[i] -= p.x.H; ++pn[i]; }
                    }
                }
            }
        }
        if (it > 0 && !(HN_DBG(p) & 2)) {
            // (all LDS read offsets are loop invariants hoisted above: the per-read swizzle arithmetic -- 32 transposed reads per step at ~8
            // VALU each against 32 MFMAs -- was a third of this kernel's time: the loop skeleton alone, loads / MFMAs / stores ablated, took
            // 115 of the stage-4 group's 370 us, tools/bench_wgrad_group.py)
            const int sbase = ((it - 1) & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[TC], b[TN];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const char* pa = smem + sbase + zofs[i] + ks * 32 * (BC * 2);
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pa));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pa + 16 * (BC * 2)));
                    a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const char* pb = smem + sbase + ZB + xofs[j] + ks * 32 * (BN * 2);
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pb));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pb + 16 * (BN * 2)));
                    b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < TC; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
                }
            }
        }
    }
    if (do_bias && (lane & 15) == 0) {
        float* bp = p.bias_part + (long)bz * p.Nout;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c_blk + wc * WC + i * 16 + (lane >> 4) * 4 + r;
                if (co < p.Nout) bp[co] = accb[i][r];
            }
    }
    const int Ktot = p.taps * p.KP;
    if (HN_DBG(p) & 1) return;
    const int row_ld = p.out_ld ? p.out_ld : Ktot, ci_lim = p.out_ld ? p.cin_lim : p.KP;
    float* part = p.part + (long)bz * p.Nout * Ktot;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci_blk + wn * WN + j * 16 + (lane & 15);
            if (ci >= ci_lim) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c_blk + wc * WC + i * 16 + (lane >> 4) * 4 + r;
                if (co < p.Nout) part[(long)co * row_ld + tap * p.KP + ci] = acc[i][j][r];
            }
        }
}

template <int BC, int BN, int WGC, int WGN>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmTN p) {   // <= 256 VGPRs: two workgroups per CU overlap issue / wait / MFMA
    gemm_tn_body<BC, BN, WGC, WGN>(p, xcd_remap(blockIdx.x, gridDim.x));
}

// The same contraction for the plain 1x1 cases (mode 0 rows / mode 1 stride-2 gather, one tap) with a ring of R LDS stages of BK pixel
// rows and COUNTED vmcnt waits: stages it+1 .. it+R-2 stay in flight while stage it is multiplied.  The two-stage loop above exposes one
// whole LDS-DMA round trip per 64-row step (measured on the grouped stage-4 launch: the loads alone take 186 us, the MFMAs alone 90 us,
// together 304 us -- nothing overlaps once a workgroup's step is a 2-3 us round trip); the ring keeps R-1 round trips in flight.
template <int BC, int BN, int WGC, int WGN, int BK, int R>
__device__ __forceinline__ void gemm_tn_ring_body(const GemmTN& p, const int lid) {
    constexpr int WC = BC / WGC, WN = BN / WGN, TC = WC / 16, TN = WN / 16;
    constexpr int ZPR = BC / 8, XPR = BN / 8, NT = 64 * WGC * WGN;
    static_assert((BK * ZPR) % NT == 0 && (BK * XPR) % NT == 0 && BK % 32 == 0 && R >= 2 && R <= 4, "every wave issues the same number of loads per stage");
    constexpr int ZL = BK * ZPR / NT, XL = BK * XPR / NT, G = ZL + XL;
    constexpr int ZB = BK * BC * 2, XB = BK * BN * 2, STAGE = ZB + XB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: M0 values, LDS bases and wave-uniform branches off the vector unit)
    const int wc = wave / WGN, wn = wave % WGN;
    const int ntile = (p.KP + BN - 1) / BN;
    const int bx = lid % ntile, by = (lid / ntile) % p.gy, bz = lid / (ntile * p.gy);
    const int ci_blk = bx * BN, c_blk = by * BC;
    const long m_begin = (long)bz * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.x.M) m_end = p.x.M;
    const int S = m_end > m_begin ? (int)((m_end - m_begin + BK - 1) / BK) : 0;

    f32x4 acc[TC][TN];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-piece source offsets, advanced by BK rows per stage
    long zoff[ZL], xoff[XL];
    int zrow[ZL], xrow[XL], pn[XL], py[XL], px[XL];
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
        const int e = tid + NT * i;
        const int row = e / ZPR, cp = tn_swz<BC>(row, e - row * ZPR);
        const int co = c_blk + cp * 8;
        zrow[i] = row;
        zoff[i] = co < p.Nout ? (m_begin + row) * (long)p.ldz + co : -1;
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int e = tid + NT * i;
        const int row = e / XPR, cp = tn_swz<BN>(row, e - row * XPR);
        const int c = ci_blk + cp * 8;
        xrow[i] = row;
        xoff[i] = c < p.x.C0 ? (p.x.mode == 0 ? (m_begin + row) * (long)p.x.ld0 + c : (long)c) : -1;
        decomp_row(p.x, m_begin + row, pn[i], py[i], px[i]);
    }
    const int adv_q = p.x.mode ? BK / p.x.W : 0, adv_r = p.x.mode ? BK % p.x.W : 0;
    const int g = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
    typedef __bf16 trv4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
    typedef __attribute__((address_space(3))) trv4* lds_b4;

    for (int it = 0; it < S + R - 1; ++it) {
        if (it >= R - 1) {
            const int newer = (it < S ? it : S) - 1 - (it - (R - 1));     // stages issued after the one multiplied now
            if (R >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");
            else if (R >= 3 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (it < S && !(HN_DBG(p) & 4)) {
            char* sZ = smem + (it % R) * STAGE;
            char* sX = sZ + ZB;
            const long m0 = m_begin + (long)it * BK;
#pragma unroll
            for (int i = 0; i < ZL; ++i) {
                const bf16* src = (m0 + zrow[i] < m_end && zoff[i] >= 0) ? p.dz + zoff[i] : g_zero_piece;
                glds16(src, sZ + (NT * i + 64 * wave) * 16);
                zoff[i] += zoff[i] >= 0 ? (long)BK * p.ldz : 0;
            }
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const bf16* src = g_zero_piece;
                if (m0 + xrow[i] < m_end && xoff[i] >= 0) {
                    if (p.x.mode == 0) src = p.x.x0 + xoff[i];
                    else src = p.x.x0 + xoff[i] + (((long)pn[i] * p.x.Hi + 2 * py[i]) * p.x.Wi + 2 * px[i]) * p.x.ld0;
                }
                glds16(src, sX + (NT * i + 64 * wave) * 16);
                if (p.x.mode == 0) xoff[i] += xoff[i] >= 0 ? (long)BK * p.x.ld0 : 0;
                else {
                    px[i] += adv_r;
                    py[i] += adv_q;
                    if (px[i] >= p.x.W) { px[i] -= p.x.W; ++py[i]; }
                    while (py[i] >= p.x.H) { py[i] -= p.x.H; ++pn[i]; }
                }
            }
        }
        if (it >= R - 1 && !(HN_DBG(p) & 2)) {
            const char* sZ = smem + ((it - (R - 1)) % R) * STAGE;
            const char* sX = sZ + ZB;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                bf16x8 a[TC], b[TN];
                const int rlo = ks * 32 + g * 4 + q, rhi = rlo + 16;
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int piece = (wc * WC + i * 16) / 8 + (pp >> 1);
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + rlo * (BC * 2) + tn_swz<BC>(rlo, piece) * 16 + (pp & 1) * 8));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + rhi * (BC * 2) + tn_swz<BC>(rhi, piece) * 16 + (pp & 1) * 8));
                    a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int piece = (wn * WN + j * 16) / 8 + (pp >> 1);
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sX + rlo * (BN * 2) + tn_swz<BN>(rlo, piece) * 16 + (pp & 1) * 8));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sX + rhi * (BN * 2) + tn_swz<BN>(rhi, piece) * 16 + (pp & 1) * 8));
                    b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    if (HN_DBG(p) & 1) return;
    const int row_ld = p.out_ld ? p.out_ld : p.KP, ci_lim = p.out_ld ? p.cin_lim : p.KP;
    float* part = p.part + (long)bz * p.Nout * p.KP;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci_blk + wn * WN + j * 16 + (lane & 15);
            if (ci >= ci_lim) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c_blk + wc * WC + i * 16 + (lane >> 4) * 4 + r;
                if (co < p.Nout) part[(long)co * row_ld + ci] = acc[i][j][r];
            }
        }
}

// The same contraction with the operand stages prefetched into REGISTERS (T14: issue early, write late): stage it + RS is requested with
// global_load_dwordx4 while stage it is multiplied; the loaded registers are written to the two LDS buffers one step before their turn.
// Why: this kernel is bound by bytes in flight.  Two workgroups per CU x one 32 KB LDS-DMA stage = 64 KB per CU against a 2.5 us
// (Infinity-cache / HBM) round trip = 26 GB/s per CU (stage-4 group: loads 154 of the launch's 287 us, tools/trace_wgrad_group.sh); LDS
// cannot hold a deeper ring at two workgroups per CU, but the register file can: a 256-thread workgroup at two per CU may use 256 VGPRs per
// lane and the tile needs 64 accumulators -- RS = 2 stages are 64 more registers and triple the bytes in flight.  Plain rows (mode 0) and the
// stride-2 gather (mode 1), one tap: the grouped 1x1 weight gradients.
// PLAIN (x.mode 0, the stride-1 1x1 convs: most jobs): branch-free fetch -- every lane loads (the zero piece where its row or channel is
// out of range), no stride-2 coordinate bookkeeping.  The generic fetch wraps each of its 8 loads in exec-mask branches and carries the
// gather's (n, y, x) walk with a run-time loop: ~400 instructions per 64-row stage against 32 MFMAs per wave -- the stage-4 group ran
// 182 of its 288 us with loads AND MFMAs ablated (tools/bench_wgrad_group.py, HN_DBG=6).
template <int BC, int BN, int WGC, int WGN, int RS, bool PLAIN = false>
__device__ __forceinline__ void gemm_tn_regs_body(const GemmTN& p, const int lid) {
    constexpr int WC = BC / WGC, WN = BN / WGN, TC = WC / 16, TN = WN / 16;
    constexpr int ZPR = BC / 8, XPR = BN / 8, NT = 64 * WGC * WGN;
    static_assert((64 * ZPR) % NT == 0 && (64 * XPR) % NT == 0 && RS == 2, "whole 16-byte pieces per thread and stage; two register stages");
    constexpr int ZL = 64 * ZPR / NT, XL = 64 * XPR / NT;
    constexpr int ZB = 64 * BC * 2, XB = 64 * BN * 2, STAGE = ZB + XB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: M0 values, LDS bases and wave-uniform branches off the vector unit)
    const int wc = wave / WGN, wn = wave % WGN;
    const int ntile = (p.KP + BN - 1) / BN;
    const int bx = lid % ntile, by = (lid / ntile) % p.gy, bz = lid / (ntile * p.gy);
    const int ci_blk = bx * BN, c_blk = by * BC;
    const long m_begin = (long)bz * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.x.M) m_end = p.x.M;
    const int S = m_end > m_begin ? (int)((m_end - m_begin + 63) >> 6) : 0;

    f32x4 acc[TC][TN];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-piece source offsets, advanced by 64 rows per stage; LDS image as in gemm_tn_body (lane-linear 16-byte pieces, source-side swizzle)
    long zoff[ZL], xoff[XL];
    int zrow[ZL], xrow[XL], pn[XL], py[XL], px[XL];
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
        const int e = tid + NT * i;
        const int row = e / ZPR, cp = tn_swz<BC>(row, e - row * ZPR);
        const int co = c_blk + cp * 8;
        zrow[i] = row;
        zoff[i] = co < p.Nout ? (m_begin + row) * (long)p.ldz + co : -1;
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int e = tid + NT * i;
        const int row = e / XPR, cp = tn_swz<BN>(row, e - row * XPR);
        const int c = ci_blk + cp * 8;
        xrow[i] = row;
        xoff[i] = c < p.x.C0 ? ((PLAIN || p.x.mode == 0) ? (m_begin + row) * (long)p.x.ld0 + c : (long)c) : -1;
        if constexpr (PLAIN) { pn[i] = 0; py[i] = 0; px[i] = 0; }
        else decomp_row(p.x, m_begin + row, pn[i], py[i], px[i]);
    }
    const int adv_q = (!PLAIN && p.x.mode) ? 64 / p.x.W : 0, adv_r = (!PLAIN && p.x.mode) ? 64 % p.x.W : 0;
    const int g = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
    typedef __bf16 trv4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
    typedef __attribute__((address_space(3))) trv4* lds_b4;
    int zofs[TC], xofs[TN];
    {
        const int r0_ = g * 4 + q;
#pragma unroll
        for (int i = 0; i < TC; ++i) zofs[i] = r0_ * (BC * 2) + tn_swz<BC>(r0_, (wc * WC + i * 16) / 8 + (pp >> 1)) * 16 + (pp & 1) * 8;
#pragma unroll
        for (int j = 0; j < TN; ++j) xofs[j] = r0_ * (BN * 2) + tn_swz<BN>(r0_, (wn * WN + j * 16) / 8 + (pp >> 1)) * 16 + (pp & 1) * 8;
    }

    bf16x8 rz[RS][ZL], rx[RS][XL];
    auto fetch = [&](int st, bf16x8 (&dz_)[ZL], bf16x8 (&dx_)[XL]) {         // stage st -> registers (zeros past the split / the tensor)
        const long m0 = m_begin + (long)st * 64;
        const bool live = st < S && !(HN_DBG(p) & 4);
        if constexpr (PLAIN) {
#pragma unroll
            for (int i = 0; i < ZL; ++i) {
                const bf16* src = (live && m0 + zrow[i] < m_end && zoff[i] >= 0) ? p.dz + zoff[i] : g_zero_piece;
                dz_[i] = ld8(src);
                zoff[i] += zoff[i] >= 0 ? 64L * p.ldz : 0;
            }
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const bf16* src = (live && m0 + xrow[i] < m_end && xoff[i] >= 0) ? p.x.x0 + xoff[i] : g_zero_piece;
                dx_[i] = ld8(src);
                xoff[i] += xoff[i] >= 0 ? 64L * p.x.ld0 : 0;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < ZL; ++i) {
            dz_[i] = (live && m0 + zrow[i] < m_end && zoff[i] >= 0) ? ld8(p.dz + zoff[i]) : zero8();
            zoff[i] += zoff[i] >= 0 ? 64L * p.ldz : 0;
        }
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const bf16* src = nullptr;
            if (live && m0 + xrow[i] < m_end && xoff[i] >= 0)
                src = p.x.mode == 0 ? p.x.x0 + xoff[i] : p.x.x0 + xoff[i] + (((long)pn[i] * p.x.Hi + 2 * py[i]) * p.x.Wi + 2 * px[i]) * p.x.ld0;
            dx_[i] = src ? ld8(src) : zero8();
            if (p.x.mode == 0) xoff[i] += xoff[i] >= 0 ? 64L * p.x.ld0 : 0;
            else {
                px[i] += adv_r;
                py[i] += adv_q;
                if (px[i] >= p.x.W) { px[i] -= p.x.W; ++py[i]; }
                while (py[i] >= p.x.H) { py[i] -= p.x.H; ++pn[i]; }
            }
        }
    };
    auto stash = [&](int buf, const bf16x8 (&dz_)[ZL], const bf16x8 (&dx_)[XL]) {     // registers -> LDS buffer (the DMA path's lane-linear image)
        char* sZ = smem + buf * STAGE;
        char* sX = sZ + ZB;
#pragma unroll
        for (int i = 0; i < ZL; ++i) *reinterpret_cast<bf16x8*>(sZ + (NT * i + tid) * 16) = dz_[i];
#pragma unroll
        for (int i = 0; i < XL; ++i) *reinterpret_cast<bf16x8*>(sX + (NT * i + tid) * 16) = dx_[i];
    };
    auto multiply = [&](int buf) {
        if (HN_DBG(p) & 2) return;
        const int sbase = buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TC], b[TN];
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const char* pa = smem + sbase + zofs[i] + ks * 32 * (BC * 2);
                const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pa));
                const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pa + 16 * (BC * 2)));
                a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const char* pb = smem + sbase + ZB + xofs[j] + ks * 32 * (BN * 2);
                const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pb));
                const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(pb + 16 * (BN * 2)));
                b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // step it: the registers of slot it & 1 hold stage it (requested two steps ago) -> LDS buffer it & 1 (last read by step it - 2, which
    // every wave finished before the barrier of step it - 1); request stage it + 2 into the same registers; barrier; multiply stage it.
    fetch(0, rz[0], rx[0]);
    fetch(1, rz[1], rx[1]);
    for (int it = 0; it < S; it += 2) {
        stash(0, rz[0], rx[0]);
        fetch(it + 2, rz[0], rx[0]);
        __syncthreads();
        multiply(0);
        if (it + 1 < S) {                                              // (workgroup-uniform)
            stash(1, rz[1], rx[1]);
            fetch(it + 3, rz[1], rx[1]);
            __syncthreads();
            multiply(1);
        }
    }
    if (HN_DBG(p) & 1) return;
    const int row_ld = p.out_ld ? p.out_ld : p.KP, ci_lim = p.out_ld ? p.cin_lim : p.KP;
    float* part = p.part + (long)bz * p.Nout * p.KP;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci_blk + wn * WN + j * 16 + (lane & 15);
            if (ci >= ci_lim) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c_blk + wc * WC + i * 16 + (lane >> 4) * 4 + r;
                if (co < p.Nout) part[(long)co * row_ld + ci] = acc[i][j][r];
            }
        }
}

// Several independent 1x1 weight gradients in ONE launch (hn_wgrad_group): the weight gradients of a whole backbone stage are not on the
// backward pass's critical path, so they are deferred to the stage boundary and run together -- one launch that fills the chip (a
// stage-4 gradient alone is 64 tiles of 128 x 128 over 2048 rows) instead of ~3 launches + 1 slab reduce per XBlock, mostly without a
// pixel split (so without slabs and without a reduce pass: the tiles write the fp32 gradient itself).  Jobs travel by value in the
// kernel arguments (nothing to upload inside a captured hipGraph).
#define HN_TN_GROUP_MAX 32
struct TNJob {
    const bf16* x0; const bf16* dz; float* part;
    long M, rows_per_split;
    int mode, H, W, Hi, Wi, C0, ld0, ldz, Nout, KP, gy, out_ld;
    int unit0, splits, tiles;      // placement: this job's units (one unit = one pixel split = `tiles` workgroups that share dZ / X rows)
};                                 // are the global units [unit0, unit0 + splits)
struct TNJobs { TNJob j[HN_TN_GROUP_MAX]; int n; int dbg; };

// Placement of a grouped launch.  A 128 x 128 tile over 2048 rows reads 1 MB of operands for 67 MFLOP: the launch lives on L2 hits, i.e. on
// the tiles that share operand rows running on ONE XCD at the same time (measured with consecutive logical ids spread by xcd_remap over
// whole-launch ranges: 1.9 GB of L2 misses for 220 MB of operands, 378 us for the 104 GFLOP of stage 4).  So the unit of placement is
// (job, pixel split) = the `tiles` workgroups that walk the same rows: unit u runs on XCD u % 8 (workgroups are dealt round-robin over the
// XCDs: hardware block b is on XCD b % 8 and is that XCD's (b / 8)-th workgroup), units of one XCD follow each other, so its 64 resident
// workgroups are whole units in lockstep and every operand slab is fetched into that L2 once.  Speed only: any placement is correct.
template <int BC, int BN, int WGC, int WGN, int BK = 64, int R = 0>
__global__ __launch_bounds__(64 * WGC * WGN, 2) void gemm_tn_group_kernel(const TNJobs jobs) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int acc = 0, ji = -1, lid = 0;
    for (int k = 0; k < jobs.n; ++k) {
        const int u0 = jobs.j[k].unit0, S = jobs.j[k].splits, T = jobs.j[k].tiles;
        const int first = u0 + ((xcd - u0) & 7);                    // first unit of this job on this XCD
        const int cnt = first < u0 + S ? (u0 + S - 1 - first) / 8 + 1 : 0;
        if (slot < acc + cnt * T) {
            const int r = slot - acc;
            lid = (first + 8 * (r / T) - u0) * T + r % T;            // split * tiles + tile: the body's (tile fastest, split slowest) order
            ji = k;
            break;
        }
        acc += cnt * T;
    }
    if (ji < 0) return;                                              // this XCD has fewer workgroups than the busiest one
    const TNJob& jb = jobs.j[ji];
    GemmTN p;
    p.x.x0 = jb.x0; p.x.x1 = nullptr; p.x.mode = jb.mode; p.x.H = jb.H; p.x.W = jb.W; p.x.Hi = jb.Hi; p.x.Wi = jb.Wi;
    p.x.C0 = jb.C0; p.x.C1 = 0; p.x.ld0 = jb.ld0; p.x.ld1 = 0; p.x.up = 0; p.x.M = jb.M; p.x.clamp = 0; p.x.diag = 0;
    p.dz = jb.dz; p.ldz = jb.ldz; p.Nout = jb.Nout; p.KP = jb.KP; p.taps = 1; p.part = jb.part; p.rows_per_split = jb.rows_per_split;
    p.gy = jb.gy; p.phase_span = 0; p.bias_part = nullptr; p.out_ld = jb.out_ld; p.cin_lim = jb.C0; p.dbg = jobs.dbg;
    if constexpr (R == 0) gemm_tn_body<BC, BN, WGC, WGN>(p, lid);
    else if constexpr (R < 0) {
        if (jb.mode == 0) gemm_tn_regs_body<BC, BN, WGC, WGN, -R, true>(p, lid);
        else gemm_tn_regs_body<BC, BN, WGC, WGN, -R, false>(p, lid);
    }
    else gemm_tn_ring_body<BC, BN, WGC, WGN, BK, R>(p, lid);
}

// 56-entry table of a patch's source rows / columns (wgrad3x3_patch_body): r0[10] | c0[18] in x0's grid (up-sampling applied), r1[10] | c1[18]
// in x1's full-resolution grid; -1 = no source (pixels that feed no in-image output)
__device__ __forceinline__ void fill_patch_table(int* tab, int tid, const XSrc& xs, int n, int pty, int ptx) {
    if (tid < 56) {
        const bool op1 = tid >= 28;
        const int k = op1 ? tid - 28 : tid;
        const bool isy = k < 10;
        const int j = isy ? k : k - 10;
        int g = (isy ? pty * 8 : ptx * 16) - 1 + j;
        g = border_idx(g, isy ? xs.Hi : xs.Wi, xs.clamp);
        int v = -1;
        if (g >= 0) {
            if (!op1) v = isy ? (n * (xs.Hi >> xs.up) + (g >> xs.up)) * (xs.Wi >> xs.up) : (g >> xs.up);
            else v = isy ? (n * xs.Hi + g) * xs.Wi : g;
        }
        tab[tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 weight gradient with patch reuse: one workgroup owns dW[BC couts][9 taps][CI ci] and walks 8x16-pixel output patches.  Per
// patch the dZ tile [128 px][BC] and the 10x18 input patch [180 px][CI] are DMA'd into LDS once and serve all nine taps (the tap
// only shifts which patch pixels the transposed reads pick up), so the bytes per FLOP drop ~5x against the row-gather wgrad above.
// 512 threads = 8 waves = (BC/64 cout groups) x (CI/16 ci groups); wave tile = 64 couts x 16 ci x 9 taps (144 accumulator VGPRs).
// ---------------------------------------------------------------------------------------------------------
// PH = 1: phase-form conv (GemmTN::phase_span): the cout tile's phase (py, px) only has the taps ky in {py, py+1}, kx in {px, px+1} --
// four accumulators per cout sub-tile instead of nine, and no per-tap branch: with the tap loop fully unrolled and branch-free the
// compiler requests the B fragments of every tap ahead of the MFMAs (the masked nine-tap loop exposed one LDS round trip per tap:
// decoder.3's weight gradient ran 8 700 cycles per patch against 2 048 cycles of MFMA work, 24 % MFMA busy in profiles/r03).
template <int BC, int CI, int PH>
__device__ __forceinline__ void wgrad3x3_patch_body(const GemmTN& p, const int patches_per_split, const int n_patches, const int lid) {
    constexpr int WCO = BC >= 64 ? 64 : BC, TC = WCO / 16;
    constexpr int WGC = BC / WCO, WGN = CI / 16, KSPLIT = 8 / (WGC * WGN);   // KSPLIT > 1: waves also split the patch's k-steps
    static_assert(WGC * WGN * KSPLIT == 8, "8 waves must tile BC x CI x k-split");
    constexpr int ZB = (128 * BC * 2 + 1023) / 1024 * 1024;           // dZ tile bytes
    constexpr int XPIX = 10 * 18, XROW = CI * 2, XNP = CI / 8;
    constexpr int XB = ((XPIX * XNP + 511) / 512) * 512 * 16;          // X patch bytes, padded to whole 512-thread DMA rounds
    constexpr int ZL = (128 * (BC / 8) + 511) / 512, XL = (XPIX * XNP + 511) / 512;
    constexpr int STAGE = ZB + XB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: M0 values, LDS bases and wave-uniform branches off the vector unit)
    const int wn = wave % WGN, wc = (wave / WGN) % WGC, wk = wave / (WGN * WGC);
    const XSrc& xs = p.x;
    const int ntile = (p.KP + CI - 1) / CI;                            // (diag: KP == CI == 64 -> 1)
    const int bx = lid % ntile, by = (lid / ntile) % p.gy, bz = lid / (ntile * p.gy);
    const int c_blk = by * BC;
    const int ci_blk = p.x.diag ? c_blk : bx * CI;                    // input-channel base of this block's X patch
    const int ci_out0 = p.x.diag ? 0 : ci_blk;                        // ... and its column base inside the partial slab
    const int tx_n = (xs.W + 15) >> 4, ty_n = (xs.H + 7) >> 3;
    const int pb = bz * patches_per_split;
    int pe = pb + patches_per_split;
    if (pe > n_patches) pe = n_patches;
    const int S = pe > pb ? pe - pb : 0;
    const int Ctot = xs.C0 + xs.C1;

    constexpr int NTAP = PH ? 4 : 9;
    const int ph_ = PH ? c_blk / p.phase_span : 0, ph_y = ph_ >> 1, ph_x = ph_ & 1;
    f32x4 acc[TC][NTAP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient: the first input-channel tile's wn == 0 waves also multiply their dZ fragments with ones (every output column of
    // that MFMA is the fragment's pixel sum)
    const bool do_bias = p.bias_part && bx == 0 && wn == 0;
    f32x4 accb[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (bf16)1.0f;
    const int g = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
    typedef __bf16 trv4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
    typedef __attribute__((address_space(3))) trv4* lds_b4;

    // Patch cursor (advanced by compare-and-wrap) and, per patch, a 56-entry LDS table of its source rows / columns in both operands
    // (reflect / clamp / up-sampling resolved once by 56 threads, one patch ahead): done per piece and patch, the coordinate divisions and
    // border arithmetic were ~400 instructions per patch and wave against 64-144 MFMAs -- the loader, not the MFMA pipe, set the pace.
    int* tab = reinterpret_cast<int*>(smem + 2 * STAGE);              // [2][56]: r0[10] | c0[18] | r1[10] | c1[18]
    int cptx, cpty, cn;
    {
        int t = pb;
        cptx = t % tx_n;
        t /= tx_n;
        cpty = t % ty_n;
        cn = t / ty_n;
    }
    // per-piece constants of the X patch loader: patch pixel (py, px) and channel of each of the thread's pieces
    int xpy[XL], xpx[XL], xc[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int e = tid + 512 * i;
        const int px_ = e / XNP;
        const int c = ci_blk + tn_swz<CI>(px_, e - px_ * XNP) * 8;
        xpy[i] = px_ / 18;
        xpx[i] = px_ - xpy[i] * 18;
        xc[i] = (px_ < XPIX && c < Ctot) ? c : -1;
    }
    if (S > 0) fill_patch_table(tab, tid, xs, cn, cpty, cptx);
    for (int it = 0; it <= S; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it < S) {
            const int n = cn;
            const int oy0 = cpty * 8, ox0 = cptx * 16;
            char* sZ = smem + (it & 1) * STAGE;
            char* sX = sZ + ZB;
#pragma unroll
            for (int i = 0; i < ZL; ++i) {
                if (512 * i + 64 * wave < 128 * (BC / 8)) {           // wave-uniform
                    const int e = tid + 512 * i;
                    const int row = e / (BC / 8), cp = tn_swz<BC>(row, e - row * (BC / 8));
                    const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
                    const int co = c_blk + cp * 8;
                    const bf16* src = (oy < xs.H && ox < xs.W && co < p.Nout) ? p.dz + ((long)(n * xs.H + oy) * xs.W + ox) * p.ldz + co : g_zero_piece;
                    glds16(src, sZ + (512 * i + 64 * wave) * 16);
                }
            }
            const int* tb = tab + (it & 1) * 56;
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const bf16* src = g_zero_piece;
                const int c = xc[i];
                if (c >= 0) {
                    int r = tb[xpy[i]], cc = tb[10 + xpx[i]];
                    const bf16* base = xs.x0 + c;
                    long ldx = xs.ld0;
                    if (xs.C1 > 0 && c >= xs.C0) {                     // the full-resolution (skip) operand of a concat conv
                        r = tb[28 + xpy[i]];
                        cc = tb[38 + xpx[i]];
                        base = xs.x1 + (c - xs.C0);
                        ldx = xs.ld1;
                    }
                    if ((r | cc) >= 0) src = base + (long)(r + cc) * ldx;
                }
                glds16(src, sX + (512 * i + 64 * wave) * 16);
            }
            if (it + 1 < S) {                                          // next patch: cursor, then its table (read after the next barrier)
                if (++cptx == tx_n) { cptx = 0; if (++cpty == ty_n) { cpty = 0; ++cn; } }
                fill_patch_table(tab + ((it + 1) & 1) * 56, tid, xs, cn, cpty, cptx);
            }
        }
        if (it > 0) {
            const char* sZ = smem + ((it - 1) & 1) * STAGE;
            const char* sX = sZ + ZB;
#pragma unroll 1
            for (int ks = wk; ks < 4; ks += KSPLIT) {                  // 32 pixels = patch rows 2ks, 2ks+1 (rolled: 144 accumulator VGPRs)
                bf16x8 a[TC];
                const int rlo = ks * 32 + g * 4 + q, rhi = rlo + 16;
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int piece = (wc * WCO + i * 16) / 8 + (pp >> 1);
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + rlo * (BC * 2) + tn_swz<BC>(rlo, piece) * 16 + (pp & 1) * 8));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sZ + rhi * (BC * 2) + tn_swz<BC>(rhi, piece) * 16 + (pp & 1) * 8));
                    a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < TC; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
                }
                const int bpiece = (wn * 16) / 8 + (pp >> 1);
                bf16x8 b[NTAP];
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const int ky = PH ? ph_y + (t >> 1) : t / 3, kx = PH ? ph_x + (t & 1) : t - 3 * (t / 3);
                    const int plo = (2 * ks + ky) * 18 + kx + g * 4 + q, phi = plo + 18;
                    const trv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sX + plo * XROW + tn_swz<CI>(plo, bpiece) * 16 + (pp & 1) * 8));
                    const trv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(sX + phi * XROW + tn_swz<CI>(phi, bpiece) * 16 + (pp & 1) * 8));
                    b[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int t = 0; t < NTAP; ++t)
#pragma unroll
                    for (int i = 0; i < TC; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[t], acc[i][t], 0, 0, 0);
            }
        }
    }
    const int Ktot = 9 * p.KP;
    float* part = p.part + ((long)bz * KSPLIT + wk) * p.Nout * Ktot;       // each k-split wave group owns its own partial slab
    if (do_bias && (lane & 15) == 0) {
        float* bp = p.bias_part + ((long)bz * KSPLIT + wk) * p.Nout;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c_blk + wc * WCO + i * 16 + (lane >> 4) * 4 + r;
                if (co < p.Nout) bp[co] = accb[i][r];
            }
    }
    const int ci = ci_out0 + wn * 16 + (lane & 15);
    if (ci < p.KP) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int tap = PH ? (ph_y + (t >> 1)) * 3 + ph_x + (t & 1) : t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = c_blk + wc * WCO + i * 16 + (lane >> 4) * 4 + r;
                    if (co < p.Nout) part[(long)co * Ktot + tap * p.KP + ci] = acc[i][t][r];
                }
            }
        if (PH) {                                                      // the five taps outside the phase: zeros in the slab (the reduce sums all nine)
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                if ((unsigned)(ky - ph_y) < 2u && (unsigned)(kx - ph_x) < 2u) continue;   // workgroup-uniform
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = c_blk + wc * WCO + i * 16 + (lane >> 4) * 4 + r;
                        if (co < p.Nout) part[(long)co * Ktot + tap * p.KP + ci] = 0.f;
                    }
            }
        }
    }
}

template <int BC, int CI, int PH = 0>
__global__ __launch_bounds__(512) void wgrad3x3_patch_kernel(const GemmTN p, int patches_per_split, int n_patches) {
    wgrad3x3_patch_body<BC, CI, PH>(p, patches_per_split, n_patches, xcd_remap(blockIdx.x, gridDim.x));
}

// The grouped 3x3 (group width 8) weight gradients of a whole backbone stage in one launch (hn_gconv_wgrad_group): per XBlock this
// was a 15-28 us launch of 120-190 workgroups plus a slab reduce; deferred to the stage boundary (ops.GradQueue) the jobs fill the chip
// together and need fewer patch splits (= fewer 2 MB fp32 slabs each).
struct PJob { const bf16* x; const bf16* dz; float* part; int n_img, H, W, C, ldx, ldz, gy, pps, n_patches, first_block; };
struct PJobs { PJob j[HN_TN_GROUP_MAX]; int n; };
__global__ __launch_bounds__(512) void gconv_wgrad_group_kernel(const PJobs jobs) {
    const int glid = xcd_remap(blockIdx.x, gridDim.x);
    int ji = 0;
    for (int k = 1; k < jobs.n; ++k)
        if (glid >= jobs.j[k].first_block) ji = k;
    const PJob& jb = jobs.j[ji];
    GemmTN p;
    p.x.x0 = jb.x; p.x.x1 = nullptr; p.x.mode = 2; p.x.H = jb.H; p.x.W = jb.W; p.x.Hi = jb.H; p.x.Wi = jb.W; p.x.C0 = jb.C; p.x.C1 = 0;
    p.x.ld0 = jb.ldx; p.x.ld1 = 0; p.x.up = 0; p.x.M = (long)jb.n_img * jb.H * jb.W; p.x.clamp = 2; p.x.diag = 1;
    p.dz = jb.dz; p.ldz = jb.ldz; p.Nout = jb.C; p.KP = 64; p.taps = 9; p.part = jb.part; p.rows_per_split = jb.pps; p.gy = jb.gy;
    p.phase_span = 0; p.bias_part = nullptr; p.out_ld = 0; p.cin_lim = 0; p.dbg = 0;
    wgrad3x3_patch_body<64, 64, 0>(p, jb.pps, jb.n_patches, glid - jb.first_block);
}

// dbias[co] = sum over the slabs of bias_part[slab][co]: done by the first ceil(Nout / 64) workgroups of whichever reduce kernel follows
// the patch wgrad (64 channels x nthreads / 64 slab lanes, fixed-order LDS fold), before their own columns
__device__ __forceinline__ void reduce_bias_cols(const float* bp, float* db, int slabs, int Nout, int blk, int nthreads) {
    __shared__ float rb[8][64];
    if (!bp || blk * 64 >= Nout) return;                              // workgroup-uniform
    const int c = blk * 64 + (threadIdx.x & 63), l = threadIdx.x >> 6, nl = nthreads >> 6;
    float s = 0.f;
    if (c < Nout)
        for (int k = l; k < slabs; k += nl) s += bp[(long)k * Nout + c];
    rb[l][threadIdx.x & 63] = s;
    __syncthreads();
    if (l == 0 && c < Nout) {
        float t = 0.f;
        for (int k = 0; k < nl; ++k) t += rb[k][threadIdx.x & 63];
        db[c] = t;
    }
    __syncthreads();
}

// dW[co][ci][tap] (PyTorch [Cout][Cin][kh][kw] order) = sum_split part[split][co][tap*KP + ci].
// block = 32 consecutive partial columns x 16 split lanes: coalesced rows, LDS tree over the lanes.
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(const float* part, float* dw, int splits, int Nout, int Cin, int KP, int taps,
                                                            const float* bias_part = nullptr, float* dbias = nullptr) {
    reduce_bias_cols(bias_part, dbias, splits, Nout, blockIdx.x, 512);
    __shared__ float red[16][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long Ktot = (long)taps * KP;
    const long cols = (long)Nout * Ktot;
    const long col = (long)blockIdx.x * 32 + tx;
    float s = 0.f;
    if (col < cols) {
        // a dependent global-load round costs ~1 us at this occupancy: four independent loads per round
        int k = ty;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; k + 48 < splits; k += 64) {
            s += part[(long)k * cols + col];
            s1 += part[(long)(k + 16) * cols + col];
            s2 += part[(long)(k + 32) * cols + col];
            s3 += part[(long)(k + 48) * cols + col];
        }
        for (; k < splits; k += 16) s += part[(long)k * cols + col];
        s = (s + s1) + (s2 + s3);
    }
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

// few splits, many columns (the wide deep layers): thread = 4 consecutive columns, float4 loads, splits walked serially
__device__ __forceinline__ void reduce4_body(const float* part, float* dw, int splits, int Nout, int Cin, int KP, int taps, long blk) {
    const long Ktot = (long)taps * KP;
    const long cols = (long)Nout * Ktot;
    const long col = (blk * 256 + threadIdx.x) * 4;
    if (col >= cols) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(part + col);
#pragma unroll 8
    for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(part + (long)k * cols + col);
    const int co = (int)(col / Ktot);
    const int r = (int)(col - (long)co * Ktot);
    const int tap = r / KP, ci = r - tap * KP;                       // the 4 columns share co and tap (KP is a multiple of 32)
    float* d = dw + ((long)co * Cin + ci) * taps + tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < Cin) d[(long)j * taps] = s[j];
}
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* part, float* dw, int splits, int Nout, int Cin, int KP, int taps,
                                                             const float* bias_part = nullptr, float* dbias = nullptr) {
    reduce_bias_cols(bias_part, dbias, splits, Nout, blockIdx.x, 256);
    reduce4_body(part, dw, splits, Nout, Cin, KP, taps, blockIdx.x);
}

// many splits, few columns, 256 threads: 32 consecutive columns x 8 split lanes (four independent loads per lane and round)
__device__ __forceinline__ void reduce_lanes_body(const float* part, float* dw, int splits, int Nout, int Cin, int KP, int taps, long blk,
                                                  float (*red)[33]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long Ktot = (long)taps * KP;
    const long cols = (long)Nout * Ktot;
    const long col = blk * 32 + tx;
    float s = 0.f;
    if (col < cols) {
        int k = ty;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; k + 24 < splits; k += 32) {
            s += part[(long)k * cols + col];
            s1 += part[(long)(k + 8) * cols + col];
            s2 += part[(long)(k + 16) * cols + col];
            s3 += part[(long)(k + 24) * cols + col];
        }
        for (; k < splits; k += 8) s += part[(long)k * cols + col];
        s = (s + s1) + (s2 + s3);
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && col < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        const int co = (int)(col / Ktot);
        const int r = (int)(col - (long)co * Ktot);
        const int tap = r / KP, ci = r - tap * KP;
        if (ci < Cin) dw[((long)co * Cin + ci) * taps + tap] = t;
    }
}

// 3x3 weights: the partial slabs are [co][tap][KP] (ci contiguous), PyTorch wants [co][ci][3][3] (tap contiguous).  Writing that
// transposition straight from registers is a 4-byte store every 36 bytes: the 2048 x 512 x 9 phase-form gradient of decoder.1 (37.7 MB)
// cost 375 MB of WRITE_SIZE and 100 us.  One workgroup = one cout x 64 input channels x 9 taps: coalesced partial reads per tap, the
// [64][9] result tile is turned in LDS and leaves as contiguous float4 runs.
__global__ __launch_bounds__(256) void wgrad_reduce9_kernel(const float* part, float* dw, int splits, int Nout, int Cin, int KP,
                                                             const float* bias_part = nullptr, float* dbias = nullptr) {
    reduce_bias_cols(bias_part, dbias, splits, Nout, blockIdx.y * gridDim.x + blockIdx.x, 256);
    // 64 input channels x 4 split lanes per workgroup (the splits are walked four at a time per lane: 36 loads in flight)
    __shared__ float tile[4][64 * 9 + 4];
    const long Ktot = 9L * KP, cols = (long)Nout * Ktot;
    const int co = blockIdx.y, ci0 = blockIdx.x * 64, cl = threadIdx.x & 63, kl = threadIdx.x >> 6, ci = ci0 + cl;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.f;
    if (ci < Cin) {
        const float* src = part + (long)co * Ktot + ci;
#pragma unroll 4
        for (int k = kl; k < splits; k += 4) {
#pragma unroll
            for (int t = 0; t < 9; ++t) s[t] += src[(long)k * cols + (long)t * KP];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) tile[kl][cl * 9 + t] = s[t];
    __syncthreads();
    const int nci = Cin - ci0 < 64 ? Cin - ci0 : 64;                   // input channels of this block
    float* dst = dw + ((long)co * Cin + ci0) * 9;                      // nci * 9 contiguous floats
    const int total = nci * 9;
    const bool al = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int e = threadIdx.x * 4; e < total; e += 1024) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = e + j < total ? (tile[0][e + j] + tile[1][e + j]) + (tile[2][e + j] + tile[3][e + j]) : 0.f;
        if (al && e + 4 <= total) *reinterpret_cast<f32x4*>(dst + e) = (f32x4){v[0], v[1], v[2], v[3]};
        else for (int j = 0; j < 4 && e + j < total; ++j) dst[e + j] = v[j];
    }
}

// grouped conv (group width 8) weight gradient from the block-diagonal slabs: dw[co][i][tap] = sum_split part[split][co][tap*64 + ((co&63)>>3)*8 + i]
__device__ __forceinline__ void diag_extract_body(const float* part, float* dw, int splits, int C, long blk) {
    const long idx = blk * 256 + threadIdx.x;
    if (idx >= (long)C * 72) return;
    const int i = (int)(idx & 7), tap = (int)((idx >> 3) % 9), co = (int)(idx / 72);   // i fastest: 8 consecutive floats of a slab row per 8 lanes
    const long col = (long)co * 576 + tap * 64 + ((co & 63) >> 3) * 8 + i;
    const long slab = (long)C * 576;
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < splits; ++k) s += part[(long)k * slab + col];
    dw[((long)co * 8 + i) * 9 + tap] = s;
}
__global__ __launch_bounds__(256) void gconv_diag_extract_kernel(const float* part, float* dw, int splits, int C) {
    diag_extract_body(part, dw, splits, C, blockIdx.x);
}

// The slab reduces of several weight gradients in ONE launch (an XBlock's conv_block_1 / 2 / 3 / shortcut gradients: each reduce is a
// 5-7 us launch of mostly latency).  Jobs travel by value in the kernel arguments (no device table to fill inside a captured graph).
//   kind 0: float4 columns (reduce4_body), 1: split lanes (reduce_lanes_body), 2: grouped-conv diagonal extract
struct RJob { const float* part; float* dw; int splits, Nout, Cin, KP, taps, kind; long first_block; };
struct RJobs { RJob j[4]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const RJobs jobs) {
    __shared__ float red[8][33];
    int ji = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < jobs.n && (long)blockIdx.x >= jobs.j[k].first_block) ji = k;
    const RJob& jb = jobs.j[ji];
    const long blk = (long)blockIdx.x - jb.first_block;
    if (jb.kind == 0) reduce4_body(jb.part, jb.dw, jb.splits, jb.Nout, jb.Cin, jb.KP, jb.taps, blk);
    else if (jb.kind == 1) reduce_lanes_body(jb.part, jb.dw, jb.splits, jb.Nout, jb.Cin, jb.KP, jb.taps, blk, red);
    else diag_extract_body(jb.part, jb.dw, jb.splits, jb.Nout, blk);
}
// fp32 grouped weights [C][8][3][3] -> block-diagonal bf16 operands [C][9][64]: wk for the forward conv, wd for the stride-1 data
// gradient (group-transposed, taps flipped)
__global__ void gconv_pack_diag_kernel(const float* w, bf16* wk, bf16* wd, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)C * 576) return;
    const int j = (int)(idx & 63), tap = (int)((idx >> 6) % 9), co = (int)(idx / 576);
    const int gl = (co & 63) >> 3;                                    // group slot of this row inside its 64-channel tile
    float a = 0.f, b = 0.f;
    if ((j >> 3) == gl) {
        a = w[((long)co * 8 + (j & 7)) * 9 + tap];                    // forward: row = cout co, column = its group's input channel j&7
        const int cosrc = (co & ~63) + j;                             // dgrad: row = input channel co, column = output channel of its group
        if (cosrc < C) b = w[((long)cosrc * 8 + (co & 7)) * 9 + (8 - tap)];
    }
    wk[idx] = f2bf(a);
    wd[idx] = f2bf(b);
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
// Phase-form effective weights of a 3x3 conv over a nearest-x2 up-sampled map (ops/seg.py SegConvUp / SegOutUp): on the low-resolution grid the
// conv has 4*Cout outputs, one Cout-vector per output phase (py,px):
//   W_eff[(py*2+px)*Cout + o][c][dy][dx] = sum over the taps (ky,kx) whose up-sampled source falls on low-res offset (dy,dx)
//   phase 0: k=0 -> d=0, k=1,2 -> d=1;   phase 1: k=0,1 -> d=1, k=2 -> d=2      (per axis; d = offset + 1)
// phase_taps(p, d) = bit mask of the k that land on d.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned phase_taps(int p, int d) { return ((p ? 0x430u : 0x061u) >> (4 * d)) & 7u; }

// value of the (virtual) weight tensor that is being packed: plain channel slice [ci0, ci0+Cin) of w[Cout][Cin_total][taps], or (phase)
// the effective weights above (taps == 9, co in [0, 4*Cout))
__device__ __forceinline__ float packed_w_value(const float* w, int Cout, int Cin_total, int ci0, int taps, int phase, int co, int ci, int tap) {
    if (!phase) return w[((long)co * Cin_total + ci0 + ci) * taps + tap];
    const int ph = co / Cout, o = co - ph * Cout;
    const int dy = tap / 3, dx = tap - 3 * dy;
    const unsigned my = phase_taps(ph >> 1, dy), mx = phase_taps(ph & 1, dx);
    const float* wr = w + ((long)o * Cin_total + ci0 + ci) * 9;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            if (((my >> ky) & 1u) && ((mx >> kx) & 1u)) s += wr[ky * 3 + kx];
    return s;
}

// Wp[CoutE][taps][KPi] | Wt[Cin][taps][KPo] | b_eff[CoutE] (phase: the bias repeated per phase), CoutE = phase ? 4*Cout : Cout
__global__ void pack_w_ex_kernel(const float* w, bf16* wp, bf16* wt, int Cout, int Cin_total, int ci0, int Cin, int taps, int phase, int KPi,
                                 int KPo, const float* bias, float* b_eff) {
    const int CoutE = phase ? 4 * Cout : Cout;
    const long nf = (long)CoutE * taps * KPi;
    const long nt = wt ? (long)Cin * taps * KPo : 0;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nf) {
        const int k = (int)(idx % KPi);
        const long t = idx / KPi;
        const int tap = (int)(t % taps);
        const int co = (int)(t / taps);
        wp[idx] = f2bf(k < Cin ? packed_w_value(w, Cout, Cin_total, ci0, taps, phase, co, k, tap) : 0.f);
    } else if (idx < nf + nt) {
        const long j = idx - nf;
        const int k = (int)(j % KPo);
        const long t = j / KPo;
        const int tap = (int)(t % taps);
        const int ci = (int)(t / taps);
        wt[j] = f2bf(k < CoutE ? packed_w_value(w, Cout, Cin_total, ci0, taps, phase, k, ci, tap) : 0.f);
    } else if (b_eff && idx < nf + nt + CoutE) {
        const int co = (int)(idx - nf - nt);
        b_eff[co] = bias[co % Cout];
    }
}

// gradient of the 3x3 weights from the effective-weight gradient (the transpose of the map above), joined with the skip operand's part:
//   dw[o][c][ky][kx] = c < C0 ? sum_{py,px} dw_eff[(py*2+px)*K + o][c][d(py,ky)][d(px,kx)] : dw1[o][c - C0][ky][kx]
//   db[o] = sum_ph db_eff[ph*K + o]                                                   d(0,k) = k ? 1 : 0,  d(1,k) = k == 2 ? 2 : 1
__global__ void phase_fold_kernel(const float* dw_eff, const float* dw1, const float* db_eff, float* dw, float* db, int K, int C0, int C1) {
    const long total = (long)K * (C0 + C1) * 9;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) {
        const int tap = (int)(idx % 9);
        const long t = idx / 9;
        const int c = (int)(t % (C0 + C1));
        const int o = (int)(t / (C0 + C1));
        if (c >= C0) {
            dw[idx] = dw1[((long)o * C1 + (c - C0)) * 9 + tap];
            return;
        }
        const int ky = tap / 3, kx = tap - 3 * ky;
        float s = 0.f;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            const int dy = py ? (ky == 2 ? 2 : 1) : (ky ? 1 : 0), dx = px ? (kx == 2 ? 2 : 1) : (kx ? 1 : 0);
            s += dw_eff[(((long)ph * K + o) * C0 + c) * 9 + dy * 3 + dx];
        }
        dw[idx] = s;
    } else if (db && idx < total + K) {
        const int o = (int)(idx - total);
        db[o] = (db_eff[o] + db_eff[K + o]) + (db_eff[2 * K + o] + db_eff[3 * K + o]);
    }
}

// all conv weights of a model in ONE launch: jobs[j] = {w, wp, wt, Cout, Cin, taps, first block, ci tiles} (device int64 table, built once).
// One workgroup = one 32-cout x 32-cin tile with all its taps, staged through LDS so that BOTH layouts are written in contiguous runs:
// the transposed operand Wt[ci][tap][co] read column-wise from global memory costs a 64-byte sector per 4-byte element (310 us per step
// for the 43 M parameters of the big cfg against ~60 us of HBM time).
template <int TAPS>
__device__ __forceinline__ void pack_tile(const float* w, bf16* wp, bf16* wt, int Cout, int Cin, int co0, int ci0, bf16* tile) {
    constexpr int row = 32 * TAPS, ldt = row + 2;                     // tile[r][c * TAPS + tap] (bf16: 18.5 KB for 3x3 -> 8 workgroups per CU)
    const int KPi = (Cin + 31) / 32 * 32, KPo = (Cout + 31) / 32 * 32;
    const int tid = threadIdx.x;
    // The texture-address unit takes a wave's 64 lanes at 4 per clock whatever each lane moves: with one element per lane (4-byte loads,
    // 2-byte stores) the 43 M parameters cost 0.7 M load and 2.7 M store instructions = 70 us of address time per CU against 60 us of
    // HBM time.  Four floats per load where the rows allow it, eight bf16 (16 bytes) per store always.
    if (((Cin * TAPS) & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
#pragma unroll 2
        for (int e = tid; e < 8 * row; e += 256) {                    // row / 4 float4 per cout row
            const int r = e / (row / 4), x = (e - r * (row / 4)) * 4;
            const int co = co0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (co < Cout && ci0 * TAPS + x < Cin * TAPS) v = *reinterpret_cast<const f32x4*>(w + ((long)co * Cin + ci0) * TAPS + x);
            bf16x2 lo = {f2bf(v[0]), f2bf(v[1])}, hi = {f2bf(v[2]), f2bf(v[3])};
            *reinterpret_cast<bf16x2*>(tile + r * ldt + x) = lo;
            *reinterpret_cast<bf16x2*>(tile + r * ldt + x + 2) = hi;
        }
    } else {
#pragma unroll 4
        for (int e = tid; e < 32 * row; e += 256) {
            const int r = e / row, x = e - r * row;
            const int co = co0 + r, ci = ci0 + x / TAPS;
            tile[r * ldt + x] = f2bf((co < Cout && ci < Cin) ? w[((long)co * Cin + ci0) * TAPS + x] : 0.f);
        }
    }
    __syncthreads();
    // forward operand Wp[co][tap][KPi]: 32 consecutive ci per (co, tap) = four 16-byte pieces
#pragma unroll 2
    for (int e = tid; e < 128 * TAPS; e += 256) {
        const int c8 = e & 3, rt = e >> 2, tap = rt % TAPS, r = rt / TAPS;
        if (co0 + r < Cout) {
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tile[r * ldt + (c8 * 8 + k) * TAPS + tap];
            st8(wp + ((long)(co0 + r) * TAPS + tap) * KPi + ci0 + c8 * 8, v);
        }
    }
    // data-gradient operand Wt[ci][tap][KPo]: 32 consecutive co per (ci, tap)
    if (wt) {
#pragma unroll 2
        for (int e = tid; e < 128 * TAPS; e += 256) {
            const int r8 = e & 3, ct = e >> 2, tap = ct % TAPS, c = ct / TAPS;
            if (ci0 + c < Cin) {
                bf16x8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[(r8 * 8 + k) * ldt + c * TAPS + tap];
                st8(wt + ((long)(ci0 + c) * TAPS + tap) * KPo + co0 + r8 * 8, v);
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_w_batched_kernel(const long* jobs, int njobs, const int* block_job) {
    __shared__ bf16 tile[32 * (32 * 9 + 2)];
    int lo = 0;
    if (block_job) {
        lo = block_job[blockIdx.x];                                   // one load instead of a 7-step dependent search per workgroup
    } else {
        int hi = njobs - 1;                                           // last job whose first block <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid * 8 + 6] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
    }
    const long* jb = jobs + lo * 8;
    const float* w = reinterpret_cast<const float*>(jb[0]);
    bf16* wp = reinterpret_cast<bf16*>(jb[1]);
    bf16* wt = reinterpret_cast<bf16*>(jb[2]);
    const int Cout = (int)jb[3], Cin = (int)jb[4], taps = (int)jb[5], tiles_ci = (int)jb[7];
    const int t = (int)((long)blockIdx.x - jb[6]);
    const int co0 = (t / tiles_ci) * 32, ci0 = (t % tiles_ci) * 32;
    if (taps == 1) pack_tile<1>(w, wp, wt, Cout, Cin, co0, ci0, tile);
    else if (taps == 9) pack_tile<9>(w, wp, wt, Cout, Cin, co0, ci0, tile);
}

// The remaining per-step weight packs of a model in ONE launch (they were ~70 launches of 4-16 us: depthwise taps, grouped-conv stencil
// and block-diagonal operands, phase-form / channel-slice packs).  jobs[j] = 16 int64:
//   {w, out0, out1, out2, bias, kind, first_block, p0, p1, p2, p3, p4, p5, 0, 0, 0}; one thread per output element, 256 per block.
//   kind 1: depthwise [C][1][3][3] -> out0 = wk[tap][C], out1 = wkf[8-tap][C]                              p0 = C
//   kind 2: grouped [C][8][3][3] -> stencil operands out0 = wk[tap][i][G][o], out1 = wd[tap'][o][G][i]      p0 = G, p1 = flip
//   kind 3: grouped -> the DIAGONAL blocks of the block-diagonal MFMA operands out0 = wk, out1 = wd [C][9][64] (buffers zero-filled by the owner)   p0 = C
//   kind 4: hn_pack_weight_ex (channel slice / phase form): out0 = wp, out1 = wt, out2 = b_eff             p0..p5 = Cout, Cin_total, ci0, Cin, taps, phase
//           (work items: numel(wp) / 8 + numel(wt) / 8 + CoutE -- eight K entries per thread)
__global__ __launch_bounds__(256) void pack_small_batched_kernel(const long* jobs, const int* block_job) {
    const long* jb = jobs + (long)block_job[blockIdx.x] * 16;
    const float* w = reinterpret_cast<const float*>(jb[0]);
    bf16* o0 = reinterpret_cast<bf16*>(jb[1]);
    bf16* o1 = reinterpret_cast<bf16*>(jb[2]);
    const int kind = (int)jb[5];
    const long idx = ((long)blockIdx.x - jb[6]) * 256 + threadIdx.x;
    if (kind == 1) {
        const int C = (int)jb[7];
        if (idx >= 9L * C) return;
        const int c = (int)(idx % C), tap = (int)(idx / C);
        const bf16 v = f2bf(w[c * 9 + tap]);
        o0[idx] = v;
        if (o1) o1[(8 - tap) * C + c] = v;
    } else if (kind == 2) {
        const int G = (int)jb[7], flip = (int)jb[8];
        if (idx >= (long)G * 576) return;
        const int b = (int)(idx & 7);
        long t = idx >> 3;
        const int g = (int)(t % G);
        t /= G;
        const int a = (int)(t & 7), tap = (int)(t >> 3);
        o0[idx] = f2bf(w[((long)(g * 8 + b) * 8 + a) * 9 + tap]);
        if (o1) o1[idx] = f2bf(w[((long)(g * 8 + a) * 8 + b) * 9 + (flip ? 8 - tap : tap)]);
    } else if (kind == 3) {
        // only the 8 x 8 diagonal blocks: the caller's persistent buffers were zero-filled once and the off-diagonal 7/8 never change
        const int C = (int)jb[7];
        if (idx >= (long)C * 72) return;
        const int i = (int)(idx & 7), tap = (int)((idx >> 3) % 9), co = (int)(idx / 72);
        const int gl = (co & 63) >> 3, j = gl * 8 + i;
        const float a = w[((long)co * 8 + i) * 9 + tap];
        const int cosrc = (co & ~63) + j;
        const float b = cosrc < C ? w[((long)cosrc * 8 + (co & 7)) * 9 + (8 - tap)] : 0.f;
        const long dst = ((long)co * 9 + tap) * 64 + j;
        o0[dst] = f2bf(a);
        o1[dst] = f2bf(b);
    } else if (kind == 4) {
        // eight consecutive K entries per thread (one 16-byte store; K is padded to a multiple of 32): with one 2-byte store per lane the
        // 19 M elements of the decoder's phase-form operands were 75 us of address time for 38 MB
        const int Cout = (int)jb[7], Cin_total = (int)jb[8], ci0 = (int)jb[9], Cin = (int)jb[10], taps = (int)jb[11], phase = (int)jb[12];
        float* b_eff = reinterpret_cast<float*>(jb[3]);
        const float* bias = reinterpret_cast<const float*>(jb[4]);
        const int CoutE = phase ? 4 * Cout : Cout;
        const int KPi = (Cin + 31) / 32 * 32, KPo = (CoutE + 31) / 32 * 32;
        const long nf = (long)CoutE * taps * (KPi >> 3);
        const long nt = o1 ? (long)Cin * taps * (KPo >> 3) : 0;
        if (idx < nf) {
            const int k0 = (int)(idx % (KPi >> 3)) * 8;
            const long t = idx / (KPi >> 3);
            const int tap = (int)(t % taps), co = (int)(t / taps);
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f2bf(k0 + j < Cin ? packed_w_value(w, Cout, Cin_total, ci0, taps, phase, co, k0 + j, tap) : 0.f);
            st8(o0 + idx * 8, v);
        } else if (idx < nf + nt) {
            const long jx = idx - nf;
            const int k0 = (int)(jx % (KPo >> 3)) * 8;
            const long t = jx / (KPo >> 3);
            const int tap = (int)(t % taps), ci = (int)(t / taps);
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f2bf(k0 + j < CoutE ? packed_w_value(w, Cout, Cin_total, ci0, taps, phase, k0 + j, ci, tap) : 0.f);
            st8(o1 + jx * 8, v);
        } else if (b_eff && idx < nf + nt + CoutE) {
            const int co = (int)(idx - nf - nt);
            b_eff[co] = bias[co % Cout];
        }
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
    s.clamp = mode == 4 ? 1 : (mode == 5 ? 2 : 0);
    s.diag = mode == 5;
    if (mode == 4 || mode == 5) mode = 2;
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

template <int BC, int BP, int WGC, int WGP, int R, int KG = 1>
static int launch_nt_r(const GemmNT& p, int out_f32, hipStream_t st) {
    dim3 grid(cdiv(p.x.M, BP) * cdiv(p.Nout, BC));
    const size_t tables = p.x.mode >= 2 ? (size_t)(3 * p.x.H + 3 * p.x.W) * 4 * (p.x.C1 ? 2 : 1) : 0;
    const size_t lds = (size_t)(BC + BP) * 128 * R * KG + tables;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> optin{0};             // one per instantiation, one bit per device
        if (!lds_optin(optin, {(const void*)gemm_nt_kernel<BC, BP, WGC, WGP, true, R, false, KG>,
                               (const void*)gemm_nt_kernel<BC, BP, WGC, WGP, false, R, false, KG>,
                               (const void*)gemm_nt_kernel<BC, BP, WGC, WGP, true, R, false, KG, (R == 2)>,
                               (const void*)gemm_nt_kernel<BC, BP, WGC, WGP, false, R, false, KG, (R == 2)>,
                               (const void*)gemm_nt_kernel<BC, BP, WGC, WGP, false, R, false, KG, (R == 2), (R == 2 && BC * BP <= 4096)>}))
            return HN_ERR_LAUNCH;
    }
#ifndef HN_NO_PLAIN
    if constexpr (R == 2) {                                           // plain pixel rows, one tap: the lean instantiation
        if (p.x.mode == 0 && p.taps == 1 && p.x.C1 == 0) {
            if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, true, R, false, KG, true>), grid, dim3(256 * KG), lds, st, p);
            else {
                if constexpr (BC * BP <= 4096) {                      // small tile with late epilogue operands: the prefetching instance
                    if ((p.psum && p.emode >= 1) || (p.addend && !p.add_pre)) {
                        hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, false, R, false, KG, true, true>), grid, dim3(256 * KG), lds, st, p);
                        HN_LAUNCH_CHECK();
                    }
                }
                hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, false, R, false, KG, true>), grid, dim3(256 * KG), lds, st, p);
            }
            HN_LAUNCH_CHECK();
        }
    }
#endif
    if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, true, R, false, KG>), grid, dim3(256 * KG), lds, st, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, false, R, false, KG>), grid, dim3(256 * KG), lds, st, p);
    HN_LAUNCH_CHECK();
}

// direct 3x3 kernel: software-pipelined variant on/off (tools/ A/B hook).  Default OFF: measured on MI355X (tools/bench_seg.py) the
// one-workgroup-per-CU ring-of-3 pipeline is 10-60 % SLOWER than two co-resident double-buffer workgroups on every seg-decoder shape
// (decoder.3 forward 306 vs 231 us): the second resident workgroup hides more latency than the deeper prefetch does.
#ifdef HN_TUNING
static int g_direct_pipe = 0;
extern "C" int hn_debug_direct_pipe(int on) { g_direct_pipe = on; return 0; }
#endif

// operand-transform variant (bf16 output, double buffer): measured slower than LDS-DMA + one extra BatchNorm pass, never launched by the
// product path (ops.XBLOCK_XF_GEMM is a tools/ experiment): compiled with -DHN_TUNING only
#ifdef HN_TUNING
template <int BC, int BP, int WGC, int WGP>
static int launch_nt_xf(const GemmNT& p, hipStream_t st) {
    dim3 grid(cdiv(p.x.M, BP) * cdiv(p.Nout, BC));
    const size_t lds = (size_t)(BC + BP) * 128 * 2 + (size_t)3 * p.KP * sizeof(float);
    if (p.xgate && p.xhw % BP != 0) return HN_ERR_UNSUPPORTED;     // a pixel tile must lie inside one image (per-image gate row in LDS)
    if (lds > 64 * 1024) return HN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gemm_nt_kernel<BC, BP, WGC, WGP, false, 2, true>), grid, dim3(256), lds, st, p);
    HN_LAUNCH_CHECK();
}
#endif

// Tuning hook (tools/ only, -DHN_TUNING): force the cout tile and/or ring depth of the next hn_conv_gemm_nt launches; 0 = automatic.
#ifdef HN_TUNING
static int g_nt_force_bc = 0, g_nt_force_r = 0;
extern "C" int hn_debug_nt_config(int bc, int r) { g_nt_force_bc = bc; g_nt_force_r = r; return 0; }
#else
static constexpr int g_nt_force_bc = 0, g_nt_force_r = 0;
#endif

// Ring depth: R = 2 (double buffer) in production; R = 3/4 stay instantiated behind the tuning hook.
template <int BC, int BP, int WGC, int WGP, int RDEEP>
static int launch_nt(const GemmNT& p, int out_f32, hipStream_t st) {
    const long blocks = (long)cdiv(p.x.M, BP) * cdiv(p.Nout, BC);
    const int stages = (p.taps * (p.KP >> 5) + 1) >> 1;
    int r = 2;     // measured: the deeper rings never beat the double buffer (their LDS footprint costs the second resident workgroup)
    (void)blocks; (void)stages;
#ifdef HN_TUNING
    if (g_nt_force_r && BC >= 32) r = g_nt_force_r;
    if (BC >= 32) {
        if (r == 3) return launch_nt_r<BC, BP, WGC, WGP, (BC >= 32 ? 3 : 2)>(p, out_f32, st);
        if (r == 4) return launch_nt_r<BC, BP, WGC, WGP, (BC >= 32 ? 4 : 2)>(p, out_f32, st);
    }
#endif
    (void)r;
    return launch_nt_r<BC, BP, WGC, WGP, 2>(p, out_f32, st);
}

static int pick_bc(int Nout) {
    if (g_nt_force_bc) return g_nt_force_bc;
    if (Nout <= 16) return 16;
    if (Nout <= 32) return 32;
    if (Nout <= 64) return 64;
    // prefer the tile with the least padding; ties go to the larger tile
    int best = 128, pad = cdiv(Nout, 128) * 128;
    const int p64 = cdiv(Nout, 64) * 64;
    if (p64 < pad) { best = 64; pad = p64; }
    return best;
}

// Few pixel rows x wide cout (the deep backbone stages: 2048...8192 rows, 376/936 channels): 64x64 tiles give >= 2 workgroups per CU,
// which overlap each other's load / wait / MFMA phases (a 64x128 tiling leaves one workgroup per CU waiting on its own loads).  The
// 112-channel convs of the neck / det towers (64 < cout <= 128, a two-stage K loop) take the 64x64 tile up to 256 K rows (the training
// step's 131 072 / 174 592-row maps): three or four small workgroups per CU hide each other's short load -> MFMA -> store chains better
// than two 128x128 ones (712 -> 715 img/s); at the 1.1-1.5 M rows of the 32 x 1152 x 1920 inference maps the large tile wins (1053 vs 1045).
// (knob 0, the TN split target: 1024 -> 2048 workgroups measured +0.75 % on the step -- 785 -> 791 img/s, tools/knob_sweep.sh: the
// L2 -> LDS bound weight-gradient GEMMs want two full rounds of short K loops rather than one round of long ones)
#ifdef HN_TUNING
long g_hn_knob[20] = {2048, 256, 1024, 512, 8192, 262144, 0, 0, 1, 0, 0, 0, 256, 384, 0, 0, 0, 0, 0, 0};
extern "C" int hn_debug_knob(int id, long value) { if (id < 0 || id >= 20) return HN_ERR_ARG; g_hn_knob[id] = value; return HN_OK; }
#else
extern const long g_hn_knob[20] = {2048, 256, 1024, 512, 8192, 262144, 0, 0, 1, 0, 0, 0, 256, 384, 0, 0, 0, 0, 0, 0};   // the shipped heuristics: constants
#endif
static bool small_tile(long M, int Nout) { return !g_nt_force_bc && Nout > 64 && (M <= g_hn_knob[4] || (M <= g_hn_knob[5] && Nout <= 128)); }

// partial statistic rows of a mode-5 (grouped 3x3 on the direct kernel) launch: one per 16x16 output patch
extern "C" int hn_direct_stat_rows(int n_img, int H, int W) { return n_img * cdiv(H, 16) * cdiv(W, 16); }

extern "C" int hn_nt_stat_rows(long M, int Nout) {               // one partial row per pixel tile of the tiling hn_conv_gemm_nt picks
    return small_tile(M, Nout) ? cdiv(M, 64) : cdiv(M, 128);
}
extern "C" int hn_nt_stat_tile(long M, int Nout) { return small_tile(M, Nout) ? 64 : 128; }   // pixel rows per partial row (the last tile may be ragged)

struct NextStat { int mode; const bf16* z; int ldz; const float* coef; const bf16* y; int ldy; };
static thread_local NextStat g_next_stat = {0, nullptr, 0, nullptr, nullptr, 0};   // set by hn_conv_gemm_nt_stat for the launch it makes
struct NextFold { bf16* ring; const bf16* y; int ldy; int form; bf16* out2; int ld2; };   // form: GemmNT::fold (1 plain, 2 space-to-depth, 3 both)
static thread_local NextFold g_next_fold = {nullptr, nullptr, 0, 0, nullptr, 0};   // set by hn_conv3x3_dgrad_fold for the launch it makes
static thread_local long* g_next_amax = nullptr;    // set by hn_conv3x3_out_argmax for the launch it makes (same thread, same call)
struct NextImgW { long stride; long rpi; };
static thread_local NextImgW g_next_imgw = {0, 0};  // set by hn_conv_gemm_nt_imgw for the launch it makes
struct NextLvl { const float* coef; int n; long row[HN_MAX_LEVELS + 1]; int nimg; unsigned hw[HN_MAX_LEVELS]; unsigned pix[HN_MAX_LEVELS]; };
static thread_local NextLvl g_next_lvl = {nullptr, 0, {0}, 0, {0}, {0}};   // set by hn_conv_gemm_nt_lvl for the launch it makes
static int conv_gemm_nt_impl(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                             int up, long M, const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out,
                             int out_f32, int ldc, long rpi, long img_stride, float* psum, float* psq, const float* xscale,
                             const float* xshift, const float* xgate, long xhw, int xact, const void* addend, int ld_add, int add_mode,
                             int phase_mode, int phase_span, hipStream_t st);

extern "C" int hn_conv_gemm_nt(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                               int up, long M, const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out,
                               int out_f32, int ldc, long rpi, long img_stride, float* psum, float* psq, hipStream_t st) {
    return conv_gemm_nt_impl(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M, w, Nout, KP, taps, bias, act, out, out_f32, ldc, rpi,
                             img_stride, psum, psq, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, 0, 0, st);
}

/* hn_conv_gemm_nt for plain rows (mode 0, one tap) with ONE PACKED WEIGHT MATRIX PER IMAGE: rows [n * rows_per_image, (n + 1) *
 * rows_per_image) of x0 use w + n * w_img_stride ([Nout][KP] bf16 each).  rows_per_image % 128 == 0.  addend (optional, row stride
 * ld_add): added BEFORE the activation (the identity branch of an XBlock).  Inference: conv_block_3 with the SE gate folded into its
 * weights per image (hn_scale_weight_gate) -- net/anynet.py:68-75 without the b * gate pass over the activation. */
extern "C" int hn_conv_gemm_nt_imgw(const void* x0, int ld0, long M, int C0, const void* w, long w_img_stride, long rows_per_image, int Nout,
                                    int KP, const float* bias, int act, void* out, int ldc, const void* addend, int ld_add, hipStream_t st) {
    HN_CHECK_ARG(rows_per_image > 0 && rows_per_image % 128 == 0 && M % rows_per_image == 0 && w_img_stride >= (long)Nout * KP);
    g_next_imgw = {w_img_stride, rows_per_image};
    const int rc = conv_gemm_nt_impl(x0, nullptr, 0, 1, 1, (int)(M < (1L << 30) ? M : 1), C0, 0, ld0, 0, 0, M, w, Nout, KP, 1, bias, act, out, 0, ldc, 0, 0,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, addend, addend ? -ld_add : 0, 0, 0, 0, st);
    g_next_imgw = {0, 0};
    return rc;
}

/* hn_conv_gemm_nt for level-packed plain rows (mode 0, one tap) with the per-level eval-mode BatchNorm + activation of the det towers
 * (head_detect/detection.py:60-75) in the epilogue: out = act(coef[l][0][c] * (x W^T + bias) + coef[l][1][c]), l = level of the row.  rows
 * [nlev]: rows of every level in the packed tensor (multiples of 128, as hn_bn_act_levels takes them); coef [nlev][4][Nout]. */
extern "C" int hn_conv_gemm_nt_lvl(const void* x0, int ld0, long M, int C0, const void* w, int Nout, int KP, const float* bias, int act,
                                   void* out, int ldc, const float* coef, int nlev, const long* rows, hipStream_t st) {
    HN_CHECK_ARG(coef && rows && nlev >= 1 && nlev <= HN_MAX_LEVELS);
    NextLvl nl = {coef, nlev, {0}, 0, {0}, {0}};
    for (int l = 0; l < nlev; ++l) {
        HN_CHECK_ARG(rows[l] > 0 && rows[l] % 128 == 0);
        nl.row[l + 1] = nl.row[l] + rows[l];
    }
    HN_CHECK_ARG(nl.row[nlev] == M);
    g_next_lvl = nl;
    const int rc = conv_gemm_nt_impl(x0, nullptr, 0, 1, 1, (int)(M < (1L << 30) ? M : 1), C0, 0, ld0, 0, 0, M, w, Nout, KP, 1, bias, act, out, 0, ldc, 0, 0,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, 0, 0, st);
    g_next_lvl.coef = nullptr;
    return rc;
}

/* hn_conv_gemm_nt for level-packed plain rows (mode 0, one tap) whose fp32 output is the per-image concatenation of the levels
 * (head_detect/detection.py:36-60: Regressor / Classifier run every pyramid level through the same convs and torch.cat the results along the
 * anchor axis): row r = image * H_l W_l + pixel of level l is stored at out + image * img_stride + (sum_{k<l} H_k W_k + pixel) * ldc.
 * The levels start on row_align-aligned rows (row_align a multiple of 128); alignment rows are not stored.  One launch instead of one per
 * level. */
extern "C" int hn_conv_gemm_nt_lvlout(const void* x0, int ld0, long M, int C0, const void* w, int Nout, int KP, const float* bias, int act,
                                      float* out, int ldc, long img_stride, int n_img, int nlev, const int* H, const int* W, int row_align,
                                      hipStream_t st) {
    HN_CHECK_ARG(out && H && W && n_img > 0 && nlev >= 1 && nlev <= HN_MAX_LEVELS && row_align > 0 && row_align % 128 == 0);
    NextLvl nl = {nullptr, nlev, {0}, n_img, {0}, {0}};
    unsigned pix = 0;
    for (int l = 0; l < nlev; ++l) {
        HN_CHECK_ARG(H[l] > 0 && W[l] > 0);
        const long real = (long)n_img * H[l] * W[l];
        nl.row[l + 1] = nl.row[l] + (real + row_align - 1) / row_align * row_align;
        nl.hw[l] = (unsigned)(H[l] * W[l]);
        nl.pix[l] = pix;
        pix += nl.hw[l];
    }
    HN_CHECK_ARG(nl.row[nlev] == M);
    g_next_lvl = nl;
    const int rc = conv_gemm_nt_impl(x0, nullptr, 0, 1, 1, (int)(M < (1L << 30) ? M : 1), C0, 0, ld0, 0, 0, M, w, Nout, KP, 1, bias, act, out, 1, ldc, 0,
                                     img_stride, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, 0, 0, st);
    g_next_lvl.nimg = 0; g_next_lvl.n = 0;
    return rc;
}

// out[n][co][k] = bf16(wp[co][k] * gate[n][k]) (k < C; the K padding stays zero): the per-image operands of hn_conv_gemm_nt_imgw
__global__ __launch_bounds__(256) void scale_weight_gate_kernel(const bf16* wp, const float* gate, bf16* out, int N, int Cout, int C, int KP) {
    const unsigned k8n = (unsigned)KP >> 3;
    const unsigned total = (unsigned)N * (unsigned)Cout * k8n;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned t = idx / k8n, k8 = idx - t * k8n;
        const unsigned n = t / (unsigned)Cout, co = t - n * (unsigned)Cout;
        const bf16x8 v = ld8(wp + (long)co * KP + k8 * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = (int)k8 * 8 + j;
            o[j] = f2bf(k < C ? bf2f(v[j]) * gate[(long)n * C + k] : 0.f);
        }
        st8(out + ((long)n * Cout + co) * KP + k8 * 8, o);
    }
}
extern "C" int hn_scale_weight_gate(const void* wp, const float* gate, void* out, int N, int Cout, int C, int KP, hipStream_t st) {
    HN_CHECK_ARG(wp && gate && out && N > 0 && Cout > 0 && C > 0 && C <= KP && (KP & 31) == 0 && (long)N * Cout * (KP >> 3) < (1L << 31));
    long b = ((long)N * Cout * (KP >> 3) + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(scale_weight_gate_kernel, dim3((unsigned)b), dim3(256), 0, st, (const bf16*)wp, gate, (bf16*)out, N, Cout, C, KP);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_conv_gemm_nt_ex(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                                  int up, long M, const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out,
                                  int out_f32, int ldc, long rpi, long img_stride, float* psum, float* psq, const float* xscale,
                                  const float* xshift, const float* xgate, long xhw, int xact, const void* addend, int ld_add,
                                  int add_mode, hipStream_t st) {
    return conv_gemm_nt_impl(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M, w, Nout, KP, taps, bias, act, out, out_f32, ldc, rpi,
                             img_stride, psum, psq, xscale, xshift, xgate, xhw, xact, addend, ld_add, add_mode, 0, 0, st);
}

/* hn_conv_gemm_nt_ex with a statistics-epilogue operand (GemmNT::emode): the partial rows psum / psq carry, instead of the BatchNorm
 * forward statistics of the output, emode 1: the SE gate-gradient partials sum q * relu(bn(ez)) (psq may be null), or emode 2: the
 * BatchNorm-backward partial sums (sum g, sum g * xhat) with g = q * [bn(ez) > 0] -- the reduce pass over (output, ez) that would
 * otherwise follow the launch.  ez: bf16 [M][ld_ez] on the output's pixel rows, ecoef = [4][Nout] (scale, shift, mean, rstd). */
extern "C" int hn_conv_gemm_nt_stat(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                                    int up, long M, const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out,
                                    int out_f32, int ldc, long rpi, long img_stride, float* psum, float* psq, const void* addend, int ld_add,
                                    int add_mode, int emode, const void* ez, int ld_ez, const float* ecoef, hipStream_t st) {
    HN_CHECK_ARG((emode == 1 || emode == 2) && psum && (psq || emode == 1) && ez && ecoef && (Nout & 7) == 0 && (ld_ez & 3) == 0 && !out_f32 &&
                 act == HN_ACT_NONE && (mode <= 1 || mode == 5) && (reinterpret_cast<uintptr_t>(ez) & 7) == 0);
    g_next_stat = {emode, (const bf16*)ez, ld_ez, ecoef, nullptr, 0};
    const int rc = conv_gemm_nt_impl(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M, w, Nout, KP, taps, bias, act, out, out_f32, ldc, rpi,
                                     img_stride, psum, psq, nullptr, nullptr, nullptr, 0, 0, addend, ld_add, add_mode, 0, 0, st);
    g_next_stat = {0, nullptr, 0, nullptr, nullptr, 0};
    return rc;
}

/* 1x1 data-gradient GEMM of an identity XBlock, dx = dz1 W1^T + g (addend, added after rounding: the staged epilogue), whose statistics rows
 * carry the reduce pass of the PREVIOUS block's masked BatchNorm-3 backward over the dx it produces (GemmNT::emode 3): psum / psq
 * [hn_nt_stat_rows(M, Nout)][Nout] = sum g', sum g' (ez - mu) rs with g' = dx [ey > 0]; ez = that block's pre-BatchNorm conv_block_3 output,
 * ey = its output (= this block's input), ecoef [4][Nout] its BatchNorm-3 coefficients (net/anynet.py:65-76 backward, two blocks at once).
 * Only for the 64 x 64 tiling (hn_nt_stat_rows(M, Nout) == ceil(M / 64)); HN_ERR_ARG otherwise. */
extern "C" int hn_conv_gemm_nt_stat3(const void* x0, int ld0, long M, int C0, const void* w, int Nout, int KP, void* out, int ldc, float* psum,
                                     float* psq, const void* addend, int ld_add, const void* ez, int ld_ez, const void* ey, int ld_ey,
                                     const float* ecoef, hipStream_t st) {
    HN_CHECK_ARG(psum && psq && ez && ey && ecoef && addend && (ld_ez & 7) == 0 && (ld_ey & 7) == 0);
    g_next_stat = {3, (const bf16*)ez, ld_ez, ecoef, (const bf16*)ey, ld_ey};
    const int rc = conv_gemm_nt_impl(x0, nullptr, 0, 1, 1, (int)(M < (1L << 30) ? M : 1), C0, 0, ld0, 0, 0, M, w, Nout, KP, 1, nullptr, HN_ACT_NONE, out, 0,
                                     ldc, 0, 0, psum, psq, nullptr, nullptr, nullptr, 0, 0, addend, ld_add, 0, 0, 0, st);
    g_next_stat = {0, nullptr, 0, nullptr, nullptr, 0};
    return rc;
}

/* Phase form of Conv3x3(ReflectionPad2d(1)(nearest_up2(x0))) on the low-resolution grid (head_seg/segmentation.py:92-104 decoder blocks
 * 1/3/5/7): mode 4 (forward; w = effective weights [4*k][9][KP(C0)], out = bf16 [N][2H][2W][k], bias [4*k], addend (optional, ld_add) =
 * pre-activation partial result of the full-resolution skip operand at the output's layout) or mode 3 (data gradient on the padded
 * low-resolution grid; x0 = space-to-depth gradient [N][H][W][4*k], w = transposed effective weights).  Only the 4 non-zero taps of each
 * phase are visited. */
/* The 4-phase k-class output conv of the seg head (head_seg/segmentation.py:101-104) fused with the deploy arg-max (model/model.py:197):
 * x0 [N][H][W][C0] bf16, w = phase-form effective weights [4k][9][KP], bias [4k]; mask int64 [N][2H][2W] = arg-max over the k logits of
 * every output pixel (first maximum wins).  The fp32 logits are never written (1.4 GB at 32 x 1152 x 1920 x 5). */
extern "C" int hn_conv3x3_out_argmax(const void* x0, int n_img, int H, int W, int C0, int ld0, const void* w, int k, int KP, const float* bias,
                                     long* mask, hipStream_t st) {
    HN_CHECK_ARG(mask && k >= 1 && 4 * k <= 32 && 32 * 32 * k * 4 <= 32768 && KP == 64 && C0 <= 64 && ((2 * W * k) & 3) == 0 && ((32 * k) & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(mask) & 15) == 0);
    g_next_amax = mask;
    // (out is only used for its alignment test; nothing is written to it)
    const int rc = conv_gemm_nt_impl(x0, nullptr, 4, n_img, H, W, C0, 0, ld0, 0, 0, (long)n_img * H * W, w, 4 * k, KP, 9, bias, HN_ACT_NONE,
                                     (void*)mask, 1, 4 * k, 0, -(long)k, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, 0, 0, st);
    g_next_amax = nullptr;
    return rc;
}

extern "C" int hn_conv3x3_phase(const void* x0, int mode, int n_img, int H, int W, int C0, int ld0, const void* w, int Nout, int KP,
                                const float* bias, int act, void* out, int ldc, int k, const void* addend, int ld_add, hipStream_t st) {
    HN_CHECK_ARG((mode == 4 || mode == 3) && k > 0 && (k & 3) == 0);
    if (mode == 4) {
        HN_CHECK_ARG(Nout == 4 * k && (k % 64 == 0) && (!addend || (ld_add & 3) == 0));
        return conv_gemm_nt_impl(x0, nullptr, 4, n_img, H, W, C0, 0, ld0, 0, 0, (long)n_img * H * W, w, Nout, KP, 9, bias, act, out, 0, ldc, 0,
                                 -(long)k, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, addend, -ld_add, 0, 1, k, st);
    }
    HN_CHECK_ARG(C0 == 4 * k && (k % 64 == 0) && !addend);
    return conv_gemm_nt_impl(x0, nullptr, 3, n_img, H, W, C0, 0, ld0, 0, 0, (long)n_img * H * W, w, Nout, KP, 9, bias, act, out, 0, ldc, 0, 0,
                             nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, 2, k, st);
}

static int conv_gemm_nt_impl(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                             int up, long M, const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out,
                             int out_f32, int ldc, long rpi, long img_stride, float* psum, float* psq, const float* xscale,
                             const float* xshift, const float* xgate, long xhw, int xact, const void* addend, int ld_add, int add_mode,
                             int phase_mode, int phase_span, hipStream_t st) {
    HN_CHECK_ARG(x0 && w && out && M > 0 && Nout > 0 && KP > 0 && (KP & 31) == 0 && taps >= 1 && taps <= 9);
    // operand transform: plain / stride-2 row gathers, bf16 output; addend: staged bf16 epilogue only (aligned rows, no per-image mapping)
    HN_CHECK_ARG(!xscale || (xshift && mode <= 1 && !out_f32 && C1 == 0 && (!xgate || xhw > 0)));
    // ld_add < 0: the addend (row stride -ld_add) is added BEFORE the activation (any epilogue form)
    HN_CHECK_ARG(phase_mode == 0 || ((mode == 3 || mode == 4) && phase_span >= 64 && phase_span % 64 == 0 && !psum && !rpi));
    HN_CHECK_ARG(!addend || ld_add < 0 || (!out_f32 && (Nout & 7) == 0 && (ldc & 7) == 0 && (ld_add & 7) == 0 && rpi == 0 && mode <= 1 &&
                                           (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(addend) & 15) == 0));
    HN_CHECK_ARG(!addend || ld_add >= 0 || ((mode <= 1 || phase_mode == 1) && rpi == 0 && ((-ld_add) & 3) == 0));
    HN_CHECK_ARG((C0 & 7) == 0 && (C1 & 7) == 0 && (ld0 & 7) == 0 && (C1 == 0 || (x1 && (ld1 & 7) == 0)));
    HN_CHECK_ARG(mode >= 0 && mode <= 5 && (mode < 4 || (up == 0 && C1 == 0)));
    HN_CHECK_ARG(mode == 5 ? (KP == 64 && Nout == C0 && !rpi) : C0 + C1 <= KP);
    HN_CHECK_ARG(mode == 0 || (long)n_img * H * W == M);
    HN_CHECK_ARG(mode < 2 ? taps == 1 : taps == 9);
    HN_CHECK_ARG(mode != 2 || (H >= 2 && W >= 2));
    GemmNT p;
    p.x = make_xsrc(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M);
    mode = p.x.mode;
    p.w = (const bf16*)w; p.Nout = Nout; p.KP = KP; p.taps = taps;
    p.bias = bias; p.act = act; p.out = out; p.ldc = ldc; p.psum = psum; p.psq = psq;
    p.rpi = rpi; p.img_stride = img_stride;
    p.d2s = 0;
    p.xscale = xscale; p.xshift = xshift; p.xgate = xgate; p.xhw = xhw; p.xact = xact;
    p.addend = (const bf16*)addend; p.ld_add = ld_add < 0 ? -ld_add : ld_add; p.add_pre = ld_add < 0 ? 1 : 0;
    HN_CHECK_ARG(add_mode == 0 || (add_mode == 1 && addend && ld_add > 0 && mode == 0 && !(H & 1) && !(W & 1) && (long)n_img * H * W == M &&
                                   M < (1L << 32)));
    p.add_s2 = add_mode;
    p.phase_mode = phase_mode; p.phase_span = phase_span;
    p.amax = g_next_amax;
    g_next_amax = nullptr;
    p.w_img_stride = g_next_imgw.stride; p.w_rpi = (unsigned)g_next_imgw.rpi;
    g_next_imgw = {0, 0};
    p.lcoef = g_next_lvl.coef; p.ln = g_next_lvl.n;
    for (int l = 0; l <= HN_MAX_LEVELS; ++l) p.lrow[l] = g_next_lvl.row[l];
    p.lo_n = g_next_lvl.nimg;
    for (int l = 0; l < HN_MAX_LEVELS; ++l) { p.lo_hw[l] = g_next_lvl.hw[l]; p.lo_pix[l] = g_next_lvl.pix[l]; }
    g_next_lvl.coef = nullptr; g_next_lvl.nimg = 0;
    HN_CHECK_ARG(!p.lcoef || (mode == 0 && taps == 1 && !psum && !xscale));
    // level-mapped output: the generic (fp32) epilogue only, plain rows
    HN_CHECK_ARG(!p.lo_n || (mode == 0 && taps == 1 && out_f32 && !psum && !xscale && rpi == 0 && !addend && M < (1L << 32)));
    HN_CHECK_ARG(p.w_rpi == 0 || (mode == 0 && taps == 1 && !xscale && p.w_rpi % 128 == 0 && M < (1L << 32)));
    p.emode = g_next_stat.mode; p.ez = g_next_stat.z; p.ld_ez = g_next_stat.ldz; p.ecoef = g_next_stat.coef; p.ey = g_next_stat.y; p.ld_ey = g_next_stat.ldy;
    HN_CHECK_ARG(p.emode != 3 || (mode == 0 && taps == 1 && psum && psq && !out_f32 && addend && ld_add > 0 && !add_mode && (Nout & 7) == 0 &&
                                   (ldc & 7) == 0 && rpi == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && small_tile(M, Nout)));
    p.tile_major = g_hn_knob[11] == 1 ? 1 : 0;
    p.wpre = 0;
    p.dbg = (int)g_hn_knob[14];
    p.dbg_buf = reinterpret_cast<unsigned long long*>(g_hn_knob[15]);          // (read by the kernels with -DHN_TUNING only)
    p.fold = g_next_fold.ring ? g_next_fold.form : 0; p.fold_out2 = g_next_fold.out2; p.ld_fo2 = g_next_fold.ld2; p.ring = g_next_fold.ring; p.fold_y = g_next_fold.y; p.ld_fy = g_next_fold.ldy;
    if (img_stride < 0) {                                            // mode 4: -img_stride = channels per depth-to-space output pixel
        HN_CHECK_ARG(p.x.clamp == 1 && (out_f32 || phase_mode == 1) && rpi == 0 && !psum && Nout == 4 * (int)(-img_stride));
        p.d2s = (int)(-img_stride);
        p.img_stride = 0;
    }
    if (mode >= 2 && (!psum || p.x.diag) && !rpi) {      // statistics epilogue: grouped convs only (one partial row per 16x16 patch)
        // the last decoder block's phase-form conv (64 -> 4 x 64): the persistent form, one phase per workgroup
        if (p.x.clamp == 1 && !p.x.diag && !out_f32 && p.d2s == 64 && phase_mode == 1 && phase_span == 64 && taps == 9 && KP == 64 && C0 == 64 && C1 == 0 &&
            Nout == 256 && !psum && !addend && (act == HN_ACT_ELU || act == HN_ACT_NONE) && (ld0 & 7) == 0 && (ldc & 7) == 0 && ldc >= 64 &&
            (reinterpret_cast<uintptr_t>(out) & 15) == 0 && g_hn_knob[11] != 3) {
            const long npatch = (long)cdiv(W, 16) * cdiv(H, 16) * n_img;
            if (npatch >= 1024) {
                SegPhase q;
                q.x = (const bf16*)x0; q.ldx = ld0; q.N = n_img; q.H = H; q.W = W; q.w = (const bf16*)w; q.bias = bias; q.out = (bf16*)out; q.ldc = ldc;
                q.act = act; q.npatch = (int)npatch;
                q.ppw = cdiv(npatch, 64);                              // 64 patch ranges x 4 phases = 256 workgroups
                const int grid = 4 * cdiv(npatch, q.ppw);
                const size_t lds = 2 * (size_t)((18 * 18 * 8 + 63) / 64) * 1024 + 256 * 128;
                static std::atomic<unsigned long long> optin_sp{0};
                if (!lds_optin(optin_sp, {(const void*)seg_phase64_conv_kernel})) return HN_ERR_LAUNCH;
                hipLaunchKernelGGL(seg_phase64_conv_kernel, dim3(grid), dim3(512), lds, st, q);
                HN_LAUNCH_CHECK();
            }
        }
        // the seg output conv on a map big enough to give every CU several patches: the persistent form with the weights in registers
        if (p.x.clamp == 1 && !p.x.diag && out_f32 && p.d2s && taps == 9 && KP == 64 && C0 == 64 && C1 == 0 && Nout == 4 * p.d2s && Nout <= 32 &&
            !psum && !addend && phase_mode == 0 && act == HN_ACT_NONE && (ld0 & 7) == 0 && ((32 * p.d2s) & 3) == 0 && ((2 * W * p.d2s) & 3) == 0 &&
            (reinterpret_cast<uintptr_t>(out) & 15) == 0 && g_hn_knob[11] != 3) {
            const long npatch = (long)cdiv(W, 16) * cdiv(H, 16) * n_img;
            if (npatch >= 1024) {
                SegOut q;
                q.x = (const bf16*)x0; q.ldx = ld0; q.N = n_img; q.H = H; q.W = W; q.w = (const bf16*)w; q.Nout = Nout; q.k = p.d2s; q.bias = bias;
                q.out = p.amax ? nullptr : (float*)out; q.amax = p.amax; q.npatch = (int)npatch;
                q.ppw = cdiv(npatch, 256);
                const int grid = cdiv(npatch, q.ppw);
                const size_t lds = 2 * (size_t)((18 * 18 * 8 + 63) / 64) * 1024 + (size_t)32 * 32 * p.d2s * 4;
                static std::atomic<unsigned long long> optin_so{0};
                if (!lds_optin(optin_so, {(const void*)seg_out_conv_kernel<true>, (const void*)seg_out_conv_kernel<false>})) return HN_ERR_LAUNCH;
                if (p.amax) hipLaunchKernelGGL(seg_out_conv_kernel<true>, dim3(grid), dim3(512), lds, st, q);
                else hipLaunchKernelGGL(seg_out_conv_kernel<false>, dim3(grid), dim3(512), lds, st, q);
                HN_LAUNCH_CHECK();
            }
        }
        HN_CHECK_ARG(p.x.Wi < 8192 && p.x.Hi < 32768);  // packed patch coordinates of the direct kernel
        int bc = p.x.diag ? 64 : (Nout <= 16 ? 16 : (Nout <= 32 ? 32 : (Nout <= 64 ? 64 : 128)));
        if (phase_mode == 1 && phase_span < bc) bc = 64;            // a cout tile must lie inside one phase
        // bf16 tiles of >= 64 couts leave through the LDS-staged epilogue only (whole 16-byte pieces of aligned rows)
        const bool staged_ok = (ldc & 7) == 0 && (Nout & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                               (!p.d2s || ((p.d2s & 7) == 0 && Nout % bc == 0));
        if (bc >= 64 && !out_f32 && !staged_ok) {       // (odd channel counts / unaligned slices: the 32-cout tile keeps the generic epilogue)
            if (p.x.diag) return HN_ERR_UNSUPPORTED;
            bc = 32;
        }
        dim3 grid((unsigned)(cdiv(Nout, bc) * cdiv(W, 16) * cdiv(H, 16) * n_img));
        // preloaded form: one chunk, all its tap tiles (<= 16) in at most 32 KB next to the patch (two or three workgroups per CU as before)
        const int nsteps = phase_mode ? 4 : 9;
        p.wpre = (!p.x.diag && KP <= 64 && (size_t)nsteps * bc * 128 <= 32768 && g_hn_knob[11] != 2) ? 1 : 0;
        // software-pipelined variant (32-channel chunks, two patch buffers + ring of four weight tiles, counted waits; still two workgroups
        // per CU): the multi-chunk bf16 launches with >= 64 couts per tile
        // (the multi-chunk form measured slower on every seg-decoder shape: tools/bench_seg.py; tuning builds can force it)
        // K <= 32 channels per tap in ONE chunk (the output layer's data gradient: 24 real channels of a 64-channel chunk): 32-channel
        // chunks halve the MFMAs, LDS fragment reads and weight DMA of a tap step, and with a single patch buffer (24 KB) + the ring of
        // four weight tiles (32 KB) two workgroups share a CU
        const bool narrow = bc == 64 && !out_f32 && !p.x.diag && !p.wpre && KP <= 32 && nsteps == 9 && g_hn_knob[18] == 0;
        if (narrow) p.wpre = 1;                                     // all nine 4 KB tap tiles next to the patch: no barrier in the tap loop
#ifdef HN_TUNING
        const bool pipe = narrow || (g_direct_pipe && bc >= 64 && !out_f32 && !p.x.diag && !p.wpre && KP > 64);
#else
        const bool pipe = narrow;
#endif
        const size_t lds = narrow ? (size_t)(((18 * 18 * 4 + 511) / 512) * 512 * 16) + 9 * (size_t)(64 * 64)
                         : pipe ? (size_t)2 * (((18 * 18 * 4 + 511) / 512) * 512 * 16) + 4 * (size_t)(512 * 16)
                                : (size_t)((18 * 18 * 128 + 1023) / 1024 * 1024) + (p.wpre ? nsteps : 2) * (size_t)bc * 128;
        const size_t lds_bias = (size_t)bc * 4 + 36 * 4;            // the tile's bias values + the patch source table behind the operand buffers
        // > 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel (done on the first, un-captured call)
        static std::atomic<unsigned long long> optin{0};
        if (!lds_optin(optin, {(const void*)conv3x3_direct_kernel<16, true, false>, (const void*)conv3x3_direct_kernel<16, false, false>,
                               (const void*)conv3x3_direct_kernel<32, true, false>, (const void*)conv3x3_direct_kernel<32, false, false>,
                               (const void*)conv3x3_direct_kernel<64, true, false>, (const void*)conv3x3_direct_kernel<64, false, false>,
                               (const void*)conv3x3_direct_kernel<128, true, false>, (const void*)conv3x3_direct_kernel<128, false, false>}))
            return HN_ERR_LAUNCH;
        if (narrow) {
            hipLaunchKernelGGL((conv3x3_direct_kernel<64, false, true, 1>), grid, dim3(512), lds + lds_bias, st, p);
            HN_LAUNCH_CHECK();
        }
#ifdef HN_TUNING
        static std::atomic<unsigned long long> optin_pipe{0};
        if (pipe) {
            if (!lds_optin(optin_pipe, {(const void*)conv3x3_direct_kernel<64, false, true>, (const void*)conv3x3_direct_kernel<128, false, true>}))
                return HN_ERR_LAUNCH;
            if (bc == 64) hipLaunchKernelGGL((conv3x3_direct_kernel<64, false, true>), grid, dim3(512), lds + lds_bias, st, p);
            else hipLaunchKernelGGL((conv3x3_direct_kernel<128, false, true>), grid, dim3(512), lds + lds_bias, st, p);
            HN_LAUNCH_CHECK();
        }
#endif
#define DIRECT_CASE(BC_) \
        if (bc == BC_) { \
            if (out_f32) hipLaunchKernelGGL((conv3x3_direct_kernel<BC_, true, false>), grid, dim3(512), lds + lds_bias, st, p); \
            else hipLaunchKernelGGL((conv3x3_direct_kernel<BC_, false, false>), grid, dim3(512), lds + lds_bias, st, p); \
        }
        DIRECT_CASE(16) DIRECT_CASE(32) DIRECT_CASE(64) DIRECT_CASE(128)
#undef DIRECT_CASE
        HN_LAUNCH_CHECK();
    }
    if (xscale) {
#ifdef HN_TUNING
        if (small_tile(M, Nout)) return launch_nt_xf<64, 64, 2, 2>(p, st);
        switch (pick_bc(Nout)) {
            case 16: return launch_nt_xf<16, 128, 1, 4>(p, st);
            case 32: return launch_nt_xf<32, 128, 1, 4>(p, st);
            case 64: return launch_nt_xf<64, 128, 2, 2>(p, st);
            default: return launch_nt_xf<128, 128, 2, 2>(p, st);
        }
#else
        return HN_ERR_UNSUPPORTED;                                   // operand-transform loader: tuning builds only (see launch_nt_xf)
#endif
    }
    if (small_tile(M, Nout)) {
        // 1x1 convs of the deep stages: two K groups per workgroup when the K loop is long enough to split
        // (K >= 512 only: at stage 3 -- K = 376, 8192 rows -- the 768 two-group workgroups of 64 KB LDS do not fit the chip's 512 slots in
        // one round, the 256-thread form's 768 do: 775 -> 781 img/s; knob 6 = 1 turns the form off, > 1 sets the threshold)
        if (mode <= 1 && taps == 1 && KP >= (g_hn_knob[6] > 1 ? g_hn_knob[6] : 512) && g_hn_knob[6] != 1 && !g_nt_force_r)
            return launch_nt_r<64, 64, 2, 2, 2, 2>(p, out_f32, st);
        return launch_nt<64, 64, 2, 2, 4>(p, out_f32, st);
    }
    switch (pick_bc(Nout)) {
        case 16: return launch_nt<16, 128, 1, 4, 2>(p, out_f32, st);
        case 32: return launch_nt<32, 128, 1, 4, 4>(p, out_f32, st);
        case 64: return launch_nt<64, 128, 2, 2, 4>(p, out_f32, st);
        default: return launch_nt<128, 128, 2, 2, 4>(p, out_f32, st);
    }
}

// Border fix-up of a folded data gradient (GemmNT::fold): the gradient of a reflection- (clamp = 0) or replicate-padded (clamp = 1) input
// collects, besides its own position of the padded grid, the padded ring positions that mirror / clamp onto it:
//   reflect: rows 1 and H-2 take padded rows 0 and H+1 (columns alike);  clamp: rows 0 and H-1.
// Thread = 8 channels of one border target pixel (2 W + 2 (H - 2) per image); out += ring sum * ELU'(y).
__global__ __launch_bounds__(256) void seg_ring_fix_kernel(bf16* out, int ldo, const bf16* ring, const bf16* y, int ldy, int N, int H, int W,
                                                           int C, int clamp, int s2d, bf16* out2, int ld2) {
    const int C8 = C >> 3, T = 2 * W + 2 * (H - 2), R = 2 * (W + 2) + 2 * H;
    const long total = (long)N * T * C8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % C8);
    long t = idx / C8;
    const int tt = (int)(t % T);
    const int n = (int)(t / T);
    const int ry0 = clamp ? 0 : 1, ry1 = clamp ? H - 1 : H - 2, rx0 = clamp ? 0 : 1, rx1 = clamp ? W - 1 : W - 2;
    int yy, xx;
    if (tt < W) { yy = ry0; xx = tt; }
    else if (tt < 2 * W) { yy = ry1; xx = tt - W; }
    else {
        int j = tt - 2 * W;
        xx = rx0;
        if (j >= H - 2) { j -= H - 2; xx = rx1; }
        yy = j + (j >= ry0 ? 1 : 0);
        if (yy >= ry1) ++yy;
    }
    int ya[3], xa[3], ny = 0, nx = 0;
    ya[ny++] = yy + 1; xa[nx++] = xx + 1;
    if (yy == ry0) ya[ny++] = 0;
    if (yy == ry1) ya[ny++] = H + 1;
    if (xx == rx0) xa[nx++] = 0;
    if (xx == rx1) xa[nx++] = W + 1;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    const bf16* rg = ring + (long)n * R * C + cg * 8;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            if (a == 0 && b == 0) continue;                           // the pixel's own position: written by the conv epilogue
            const int pa = ya[a], pb = xa[b];
            const int r = pa == 0 ? pb : (pa == H + 1 ? (W + 2) + pb : (pb == 0 ? 2 * (W + 2) + pa - 1 : 2 * (W + 2) + H + pa - 1));
            const bf16x8 v = ld8(rg + (long)r * C);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
        }
    const long pix = (long)(n * H + yy) * W + xx;
    const long s2d_cell = (long)(n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1);
    const int s2d_ch = ((yy & 1) * 2 + (xx & 1)) * C + cg * 8;
    bf16* o = s2d ? out + s2d_cell * ldo + s2d_ch : out + pix * ldo + cg * 8;                  // (GemmNT::fold = 2)
    bf16x8 cur = ld8(o);
    if (y) {
        const bf16x8 yv = ld8(y + pix * ldy + cg * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float e = bf2f(yv[k]); acc[k] = e > 0.f ? acc[k] : acc[k] * (e + 1.0f); }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = f2bf(bf2f(cur[k]) + acc[k]);
    st8(o, cur);
    if (out2) st8(out2 + s2d_cell * ld2 + s2d_ch, cur);                                        // (GemmNT::fold = 3: the same value in both orders)
}

extern "C" long hn_fold_ring_rows(int H, int W) { return 2L * (W + 2) + 2L * H; }

/* Data gradient of a 3x3 conv over a reflection-padded (clamp = 0) or, in phase form, replicate-padded (clamp = 1) input, written
 * straight to the unpadded gradient: dx [N][H][W][Nout] (row stride ldo) = fold(full correlation of dz with the transposed weights)
 * [* ELU'(yprev)] -- the padded-grid tensor and the fold pass of hn_conv_gemm_nt(mode 3) + hn_seg_fold in two launches (conv with a
 * folding epilogue + a border fix-up over 2 (H + W) pixels per image).  phase_k = 0: dz [N][H][W][Cz], wt [Nout][9][KP];  phase_k > 0:
 * dz = space-to-depth gradient [N][H][W][4 k] of a phase-form conv, wt = its transposed effective weights (4 taps per phase).
 * ring: scratch [N][hn_fold_ring_rows(H, W)][Nout] bf16.  Needs Nout % 8 == 0, Nout > 32, H, W >= 4 (else HN_ERR_UNSUPPORTED). */
static int dgrad_fold_impl(const void* dz, int ldz, int Cz, int n_img, int H, int W, const void* wt, int Nout, int KP, int phase_k,
                           int clamp, void* out, int ldo, void* out_s2d, int ld_s2d, const void* yprev, int ldy, void* ring, hipStream_t st) {
    HN_CHECK_ARG(dz && wt && (out || out_s2d) && ring && n_img > 0 && (ldo & 7) == 0 && (!yprev || (ldy & 7) == 0) && (clamp == 0 || clamp == 1));
    HN_CHECK_ARG(phase_k == 0 || (Cz == 4 * phase_k && phase_k % 64 == 0));
    HN_CHECK_ARG(!out_s2d || (!(H & 1) && !(W & 1) && ld_s2d >= 4 * Nout && (ld_s2d & 7) == 0 && (reinterpret_cast<uintptr_t>(out_s2d) & 15) == 0));
    if ((Nout & 7) || Nout <= 32 || H < 4 || W < 4 || (reinterpret_cast<uintptr_t>(out) & 15)) return HN_ERR_UNSUPPORTED;
    const int form = !out_s2d ? 1 : (out ? 3 : 2);
    void* o1 = out ? out : out_s2d;                                  // the tensor the conv epilogue / fix-up address first
    const int l1 = out ? ldo : ld_s2d;
    g_next_fold = {(bf16*)ring, (const bf16*)yprev, ldy, form, form == 3 ? (bf16*)out_s2d : nullptr, ld_s2d};
    const int rc = conv_gemm_nt_impl(dz, nullptr, 3, n_img, H + 2, W + 2, Cz, 0, ldz, 0, 0, (long)n_img * (H + 2) * (W + 2), wt, Nout, KP, 9,
                                     nullptr, HN_ACT_NONE, o1, 0, l1, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0,
                                     phase_k ? 2 : 0, phase_k, st);
    g_next_fold = {nullptr, nullptr, 0, 0, nullptr, 0};
    if (rc != HN_OK) return rc;
    const long total = (long)n_img * (2 * W + 2 * (H - 2)) * (Nout >> 3);
    hipLaunchKernelGGL(seg_ring_fix_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (bf16*)o1, l1, (const bf16*)ring, (const bf16*)yprev, ldy,
                       n_img, H, W, Nout, clamp, form == 2 ? 1 : 0, form == 3 ? (bf16*)out_s2d : nullptr, ld_s2d);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_conv3x3_dgrad_fold(const void* dz, int ldz, int Cz, int n_img, int H, int W, const void* wt, int Nout, int KP, int phase_k,
                                     int clamp, void* out, int ldo, const void* yprev, int ldy, void* ring, hipStream_t st) {
    HN_CHECK_ARG(out);
    return dgrad_fold_impl(dz, ldz, Cz, n_img, H, W, wt, Nout, KP, phase_k, clamp, out, ldo, nullptr, 0, yprev, ldy, ring, st);
}
/* The same gradient (also) written in SPACE-TO-DEPTH order: out_s2d [N][H/2][W/2][4 Nout] (row stride ld_s2d >= 4 Nout), value of pixel
 * (y, x), channel c at row (y/2, x/2), channel ((y&1) 2 + (x&1)) Nout + c -- the operand form in which the phase-form block that produced
 * this conv's input consumes its gradient (hn_conv_gemm_tn_phase / hn_conv3x3_phase mode 3): its hn_space_to_depth_bf16 pass (one read and
 * one write of the whole gradient) is not needed.  out = NULL: only that form.  H and W even; yprev stays [N][H][W] (row stride ldy). */
extern "C" int hn_conv3x3_dgrad_fold_s2d(const void* dz, int ldz, int Cz, int n_img, int H, int W, const void* wt, int Nout, int KP, int phase_k,
                                         int clamp, void* out, int ldo, void* out_s2d, int ld_s2d, const void* yprev, int ldy, void* ring,
                                         hipStream_t st) {
    HN_CHECK_ARG(out_s2d);
    return dgrad_fold_impl(dz, ldz, Cz, n_img, H, W, wt, Nout, KP, phase_k, clamp, out, ldo, out_s2d, ld_s2d, yprev, ldy, ring, st);
}

template <int BC, int BN, int WGC, int WGN>
static int launch_tn(const GemmTN& p, int splits, hipStream_t st) {
    GemmTN q = p;
    q.gy = cdiv(p.Nout, BC);
    dim3 grid(cdiv(p.KP, BN) * p.taps * q.gy * splits);
    const size_t tables = p.x.mode >= 2 ? (size_t)(2 * p.x.H + 2 * p.x.W) * 4 : 0;
    hipLaunchKernelGGL((gemm_tn_kernel<BC, BN, WGC, WGN>), grid, dim3(256), (size_t)64 * (BC + BN) * 2 * 2 + tables, st, q);
    HN_LAUNCH_CHECK();
}

// Tuning hook (tools/ only): force the wgrad tile and split count of later hn_conv_gemm_tn launches; 0 = automatic.
#ifdef HN_TUNING
static int g_tn_force_bc = 0, g_tn_force_bn = 0, g_tn_force_splits = 0;
extern "C" int hn_debug_tn_config(int bc, int bn, int splits) { g_tn_force_bc = bc; g_tn_force_bn = bn; g_tn_force_splits = splits; return 0; }
#else
static constexpr int g_tn_force_bc = 0, g_tn_force_bn = 0, g_tn_force_splits = 0;
#endif

static void tn_tiles(int Nout, int KP, int& bc, int& bn) {
    if (g_tn_force_bc && g_tn_force_bn) { bc = g_tn_force_bc; bn = g_tn_force_bn; return; }
    bc = Nout <= 16 ? 16 : (Nout <= 32 ? 32 : (Nout <= 64 ? 64 : 128));
    bn = KP <= 32 ? 32 : (KP <= 64 ? 64 : 128);
    if (bc == 16 && bn < 64) bn = 64;                       // 4 waves need >= 16 columns each
}

// (KP = 32, the 24-channel skip operand of decoder.5: nine tap-parallel row-gather workgroups re-read dZ nine times -- 173 us; the
// patch kernel reads it once: 95 us with half of a 64-channel patch zero-filled, 69 us with the 32-channel patch)
static bool use_patch_wgrad(int mode, int Nout, int KP) { return mode == 2 && KP >= 32; }
static void patch_tiles(int Nout, int KP, int& bc, int& ci, int& ksplit) {
    if (KP <= 32 && Nout > 64) { bc = 128; ci = 32; ksplit = 2; return; }
    if (Nout <= 16) { bc = 16; ci = 64; ksplit = 2; }
    else if (Nout <= 32) { bc = 32; ci = 64; ksplit = 2; }           // the 4-phase 5-class output conv: 20 couts (62 KB LDS: 2 workgroups per CU)
    else if (Nout <= 64) { bc = 64; ci = 64; ksplit = 2; }           // 78 KB LDS: two workgroups per CU (ci = 128: 124 KB, one)
    else { bc = 128; ci = 64; ksplit = 1; }
}

// plan the pixel split for wgrad: returns splits, rows per split (multiple of 64; patches per split for the 3x3 patch kernel) and the
// fp32 workspace size in bytes
static int wgrad_plan_impl(int mode, int n_img, int H, int W, long M, int Nout, int KP, int taps, int phase_span, int* splits,
                           long* rows_per_split, long* ws_bytes);
extern "C" int hn_wgrad_plan(int mode, int n_img, int H, int W, long M, int Nout, int KP, int taps, int* splits, long* rows_per_split,
                             long* ws_bytes) {
    return wgrad_plan_impl(mode, n_img, H, W, M, Nout, KP, taps, 0, splits, rows_per_split, ws_bytes);
}
/* plan of hn_conv_gemm_tn_phase (phase_span = couts per phase) */
extern "C" int hn_wgrad_plan_phase(int n_img, int H, int W, int Nout, int KP, int phase_span, int* splits, long* rows_per_split, long* ws_bytes) {
    return wgrad_plan_impl(4, n_img, H, W, (long)n_img * H * W, Nout, KP, 9, phase_span, splits, rows_per_split, ws_bytes);
}
static int wgrad_plan_impl(int mode, int n_img, int H, int W, long M, int Nout, int KP, int taps, int phase_span, int* splits,
                           long* rows_per_split, long* ws_bytes) {
    HN_CHECK_ARG(M > 0 && Nout > 0 && KP > 0 && taps > 0 && splits && rows_per_split && ws_bytes);
    const int grouped = mode == 5;
    if (mode == 4 || mode == 5) mode = 2;
    if (use_patch_wgrad(mode, Nout, KP)) {
        int bc, ci, ksplit;
        patch_tiles(Nout, KP, bc, ci, ksplit);
        if (grouped) { bc = 64; ci = 64; ksplit = 2; }
        if (phase_span && phase_span < bc) { bc = 64; ci = 64; ksplit = 2; }      // a cout tile must lie inside one phase
        const long tiles = (long)cdiv(Nout, bc) * cdiv(KP, ci);
        const long patches = (long)n_img * cdiv(H, 8) * cdiv(W, 16);
        long want = (g_hn_knob[12] + tiles - 1) / tiles;          // (knob 12 = 256) one workgroup per CU in total: every split costs a full fp32 slab of dW (write + reduce)
        // ... unless the slab is small: then two workgroups per CU (where their LDS fits) hide each other's DMA waits
        if (bc <= 64 && 2 * want * ksplit * (long)Nout * taps * KP * 4 <= (64L << 20)) want *= 2;
        if (want > patches / 2) want = patches / 2;
        if (want < 1) want = 1;
        const long pps = (patches + want - 1) / want;
        *splits = (int)((patches + pps - 1) / pps) * ksplit;       // number of partial slabs
        *rows_per_split = pps;
        *ws_bytes = (long)(*splits) * Nout * taps * KP * 4 + (long)(*splits) * Nout * 4;     // + the bias-gradient partial rows
        return HN_OK;
    }
    int bc, bn;
    tn_tiles(Nout, KP, bc, bn);
    const long tiles = (long)cdiv(Nout, bc) * cdiv(KP, bn) * taps;
    long want = (g_hn_knob[0] + tiles - 1) / tiles;         // ~4 workgroups per CU in total
    if (tiles <= 2) want = 512 / tiles;                     // one or two output tiles (the 112-channel convs of the neck / det towers): every split costs a
                                                            // whole fp32 slab -- two workgroups per CU instead of four (+0.3 % on the step; 256 and 768 are worse)
    const long max_splits = (M + g_hn_knob[1] - 1) / g_hn_knob[1];   // at least 256 rows per split
    if (want > max_splits) want = max_splits;
    if (g_tn_force_splits) want = g_tn_force_splits;
    if (want < 1) want = 1;
    long rps = ((M + want - 1) / want + 63) / 64 * 64;
    *splits = (int)((M + rps - 1) / rps);
    *rows_per_split = rps;
    *ws_bytes = (long)(*splits) * Nout * taps * KP * 4 + (long)(*splits) * Nout * 4;         // + the bias-gradient partial rows
    return HN_OK;
}

static int conv_gemm_tn_impl(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                             int up, long M, const void* dz, int ldz, int Nout, int KP, int taps, int phase_span, float* workspace, float* dw,
                             float* dbias, hipStream_t st);
extern "C" int hn_conv_gemm_tn(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                               int up, long M, const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw,
                               hipStream_t st) {
    return conv_gemm_tn_impl(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M, dz, ldz, Nout, KP, taps, 0, workspace, dw, nullptr, st);
}
/* hn_conv_gemm_tn without its slab reduce: job [8] (host) receives {part, dw, splits, Nout, Cin, KP, taps, kind} for hn_wgrad_reduce_jobs,
 * which reduces up to four such jobs in one launch (kind -1: the reduce was launched here after all -- the transposing 3x3 form). */
static thread_local long* g_defer_job = nullptr;
extern "C" int hn_conv_gemm_tn_deferred(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                                        int up, long M, const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw,
                                        long* job, hipStream_t st) {
    HN_CHECK_ARG(job);
    g_defer_job = job;
    const int rc = conv_gemm_tn_impl(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M, dz, ldz, Nout, KP, taps, 0, workspace, dw, nullptr, st);
    g_defer_job = nullptr;
    return rc;
}
extern "C" int hn_wgrad_reduce_jobs(const long* jobs, int njobs, hipStream_t st) {
    HN_CHECK_ARG(jobs && njobs >= 0 && njobs <= 4);
    RJobs r;
    r.n = 0;
    long blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + 8 * i;
        const int kind = (int)jb[7];
        if (kind < 0) continue;
        HN_CHECK_ARG(kind <= 2 && jb[0] && jb[1]);
        RJob& d = r.j[r.n++];
        d.part = reinterpret_cast<const float*>(jb[0]); d.dw = reinterpret_cast<float*>(jb[1]);
        d.splits = (int)jb[2]; d.Nout = (int)jb[3]; d.Cin = (int)jb[4]; d.KP = (int)jb[5]; d.taps = (int)jb[6]; d.kind = kind;
        d.first_block = blocks;
        const long cols = (long)d.Nout * d.taps * d.KP;
        blocks += kind == 0 ? cdiv(cols / 4, 256) : (kind == 1 ? cdiv(cols, 32) : cdiv((long)d.Nout * 72, 256));
    }
    if (r.n == 0) return HN_OK;
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, st, r);
    HN_LAUNCH_CHECK();
}

// ---- deferred, grouped 1x1 weight gradients ------------------------------------------------------------------------------------------
struct RJobsN { RJob j[HN_TN_GROUP_MAX]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const RJobsN jobs) {
    __shared__ float red[8][33];
    int ji = 0;
    for (int k = 1; k < jobs.n; ++k)
        if ((long)blockIdx.x >= jobs.j[k].first_block) ji = k;
    const RJob& jb = jobs.j[ji];
    const long blk = (long)blockIdx.x - jb.first_block;
    if (jb.kind == 0) reduce4_body(jb.part, jb.dw, jb.splits, jb.Nout, jb.Cin, jb.KP, jb.taps, blk);
    else if (jb.kind == 1) reduce_lanes_body(jb.part, jb.dw, jb.splits, jb.Nout, jb.Cin, jb.KP, jb.taps, blk, red);
    else diag_extract_body(jb.part, jb.dw, jb.splits, jb.Nout, blk);
}

#define HN_WG_FIELDS 12
struct GroupPlan {
    int bc, bn;
    int splits[HN_TN_GROUP_MAX];
    long rps[HN_TN_GROUP_MAX];
    long ws_off[HN_TN_GROUP_MAX];      // float offset of the job's slabs in the workspace (splits > 1 only)
    long ws_floats;
};
// jobs: host table, HN_WG_FIELDS int64 per job: {x0, dz, dw, mode (0 | 1), n_img, H, W, Cin, ld0, ldz, Nout, M}
static int wgrad_group_plan(const long* jobs, int njobs, GroupPlan& g) {
    HN_CHECK_ARG(jobs && njobs > 0 && njobs <= HN_TN_GROUP_MAX);
    int max_nout = 0, max_kp = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + HN_WG_FIELDS * i;
        const int mode = (int)jb[3], cin = (int)jb[7], nout = (int)jb[10];
        const long M = jb[11];
        HN_CHECK_ARG(jb[0] && jb[1] && jb[2] && (mode == 0 || mode == 1) && cin > 0 && (cin & 7) == 0 && nout > 0 && M > 0 &&
                     (jb[8] & 7) == 0 && (jb[9] & 7) == 0 && jb[9] >= ((nout + 7) & ~7));
        HN_CHECK_ARG(mode == 0 || jb[4] * jb[5] * jb[6] == M);
        if (nout > max_nout) max_nout = nout;
        const int kp = (cin + 31) & ~31;
        if (kp > max_kp) max_kp = kp;
    }
    // ONE tile shape per launch, picked for the largest job (the smaller ones of a stage -- its first block's narrower inputs -- pad)
    tn_tiles(max_nout, max_kp, g.bc, g.bn);
    // Measured on the stage-4 group (29 jobs, 104 GFLOP, tools/bench_wgrad_group.py; hn_debug_knob(10, v) selects the variant): the shipped
    // two-stage 128 x 128 tile 304 us = loads alone 186 us + MFMAs alone 90 us + 64-byte-segment stores 48 us, barely overlapped (1.9 GB
    // through L2 -> LDS at ~40 GB/s per CU, 87 % L2 hits at best).  A 256 x 256 tile (8 waves, one workgroup per CU) halves the bytes but
    // runs 332 us (two-stage) / 378 us (ring of 4 x 32 rows, counted vmcnt): its loads alone still take 194-218 us (one workgroup per CU
    // pulls only 17-19 GB/s, L2 hit rate 56 %: each slab has 4 readers instead of 8).  Rings on the 128 tile: 392 (4 x 32 rows) / 501 us
    // (3 x 64 rows, one workgroup per CU).  So: 128 x 128, two stages, two workgroups per CU.
    if (max_nout >= 640 && max_kp >= 640 && (g_hn_knob[10] == 2 || g_hn_knob[10] == 3 || g_hn_knob[10] == 6)) { g.bc = 256; g.bn = 256; }
    long tiles = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + HN_WG_FIELDS * i;
        tiles += (long)cdiv(jb[10], g.bc) * cdiv(((int)jb[7] + 31) & ~31, g.bn);
    }
    // pixel splits only while the launch would not fill the chip (~4 workgroups per CU, as hn_wgrad_plan): every split costs an fp32 slab
    // (no split at all once the tiles alone come within a quarter of the target: the stage-4 group has 1 856 tiles; splitting it in two
    // bought a second half-empty round but cost 203 MB of slabs and a 78 us reduce launch)
    long want = 4 * tiles >= 3 * g_hn_knob[0] ? 1 : (g_hn_knob[0] + tiles - 1) / tiles;
    g.ws_floats = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + HN_WG_FIELDS * i;
        const long M = jb[11];
        long w = want;
        const long max_splits = (M + g_hn_knob[1] - 1) / g_hn_knob[1];
        if (w > max_splits) w = max_splits;
        if (w < 1) w = 1;
        const long rps = ((M + w - 1) / w + 63) / 64 * 64;
        g.splits[i] = (int)((M + rps - 1) / rps);
        g.rps[i] = rps;
        g.ws_off[i] = g.ws_floats;
        if (g.splits[i] > 1) g.ws_floats += (long)g.splits[i] * jb[10] * (((int)jb[7] + 31) & ~31);
    }
    return HN_OK;
}
extern "C" long hn_wgrad_group_ws_bytes(const long* jobs, int njobs) {
    GroupPlan g;
    if (wgrad_group_plan(jobs, njobs, g) != HN_OK) return -1;
    return g.ws_floats * 4 + 64;
}
/* Up to 32 independent 1x1-conv weight gradients dw_j[Nout][Cin] (fp32) = dz_j^T [Nout x M] . x_j [M x Cin] in one GEMM launch (+ one slab
 * reduce launch when the jobs are too few to fill the chip without a pixel split).  jobs: HOST table, 12 int64 per job (see above):
 * mode 0: x rows = dz rows; mode 1: x is the [n_img][2H][2W] input of a stride-2 1x1 conv whose output grid is [n_img][H][W].
 * workspace: hn_wgrad_group_ws_bytes(jobs, njobs) bytes.  Reference ops: the weight gradients of conv_block_1 / conv_block_3 / shortcut of
 * the XBlocks of one backbone stage (net/anynet.py:29-33,52-60), which autograd produces one by one. */
extern "C" int hn_wgrad_group(const long* jobs, int njobs, float* workspace, hipStream_t st) {
    GroupPlan g;
    const int rc0 = wgrad_group_plan(jobs, njobs, g);
    if (rc0 != HN_OK) return rc0;
    HN_CHECK_ARG(workspace || g.ws_floats == 0);
    TNJobs t;
    RJobsN r;
    t.n = njobs;
    t.dbg = (int)g_hn_knob[9];
    r.n = 0;
    long rblocks = 0, per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int units = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + HN_WG_FIELDS * i;
        TNJob& d = t.j[i];
        const int mode = (int)jb[3], H = (int)jb[5], W = (int)jb[6], cin = (int)jb[7], nout = (int)jb[10], kp = (cin + 31) & ~31;
        d.x0 = reinterpret_cast<const bf16*>(jb[0]); d.dz = reinterpret_cast<const bf16*>(jb[1]);
        d.M = jb[11]; d.rows_per_split = g.rps[i];
        d.mode = mode; d.H = H; d.W = W; d.Hi = mode == 1 ? 2 * H : H; d.Wi = mode == 1 ? 2 * W : W;
        d.C0 = cin; d.ld0 = (int)jb[8]; d.ldz = (int)jb[9]; d.Nout = nout; d.KP = kp;
        d.gy = cdiv(nout, g.bc);
        d.unit0 = units; d.splits = g.splits[i]; d.tiles = cdiv(kp, g.bn) * d.gy;
        for (int s_ = 0; s_ < d.splits; ++s_) per_xcd[(units + s_) & 7] += d.tiles;
        units += d.splits;
        if (g.splits[i] == 1) {
            d.part = reinterpret_cast<float*>(jb[2]); d.out_ld = cin;
        } else {
            d.part = workspace + g.ws_off[i]; d.out_ld = 0;
            RJob& q = r.j[r.n++];
            const long cols = (long)nout * kp;
            q.part = d.part; q.dw = reinterpret_cast<float*>(jb[2]); q.splits = g.splits[i]; q.Nout = nout; q.Cin = cin; q.KP = kp; q.taps = 1;
            q.kind = (g.splits[i] <= 128 && cols >= 65536) ? 0 : 1;
            q.first_block = rblocks;
            rblocks += q.kind == 0 ? cdiv(cols / 4, 256) : cdiv(cols, 32);
        }
    }
    long blocks = 0;
    for (int x = 0; x < 8; ++x) blocks = per_xcd[x] > blocks ? per_xcd[x] : blocks;
    blocks *= 8;                                                     // every XCD gets as many workgroups as the busiest one (the rest exit)
    HN_CHECK_ARG(blocks > 0 && blocks < (1L << 31));
    int rc = HN_OK;
    const int variant = (int)g_hn_knob[10];           // tools/: 0 = shipped choice; 1 = two-stage 128 x 128; 2..5 = ring variants below
    (void)variant;
    const size_t lds = (size_t)64 * (g.bc + g.bn) * 2 * 2;
#define TNG_CASE(BC_, BN_, A_, B_) if (g.bc == BC_ && g.bn == BN_) \
        hipLaunchKernelGGL((gemm_tn_group_kernel<BC_, BN_, A_, B_>), dim3((unsigned)blocks), dim3(64 * A_ * B_), lds, st, t); else
#ifdef HN_TUNING
    // ring / 256 x 256 variants (knob 10): all measured slower than the two-stage 128 x 128 tile with two workgroups per CU
    static std::atomic<unsigned long long> optin{0};
    if (!lds_optin(optin, {(const void*)gemm_tn_group_kernel<256, 256, 2, 4, 32, 4>, (const void*)gemm_tn_group_kernel<256, 256, 2, 4, 32, 3>,
                           (const void*)gemm_tn_group_kernel<256, 256, 2, 4>,
                           (const void*)gemm_tn_group_kernel<128, 128, 2, 2, 64, 3>, (const void*)gemm_tn_group_kernel<128, 128, 2, 2, 32, 4>}))
        return HN_ERR_LAUNCH;
#define TNG_RING(BC_, BN_, A_, B_, BK_, R_) \
        hipLaunchKernelGGL((gemm_tn_group_kernel<BC_, BN_, A_, B_, BK_, R_>), dim3((unsigned)blocks), dim3(64 * A_ * B_), \
                           (size_t)BK_ * (BC_ + BN_) * 2 * R_, st, t)
    if (g.bc == 256 && g.bn == 256 && variant == 2) TNG_RING(256, 256, 2, 4, 32, 4);
    else if (g.bc == 256 && g.bn == 256 && variant == 3) TNG_RING(256, 256, 2, 4, 32, 3);
    else if (g.bc == 128 && g.bn == 128 && variant == 4) TNG_RING(128, 128, 2, 2, 64, 3);
    else if (g.bc == 128 && g.bn == 128 && variant == 5) TNG_RING(128, 128, 2, 2, 32, 4);
    else
    TNG_CASE(256, 256, 2, 4)
#undef TNG_RING
#endif
    // 128 x 128: operand stages prefetched into registers, two ahead (gemm_tn_regs_body); knob 10 = 1 (tuning build): the LDS-DMA double buffer
    if (g.bc == 128 && g.bn == 128 && variant != 1)
        hipLaunchKernelGGL((gemm_tn_group_kernel<128, 128, 2, 2, 64, -2>), dim3((unsigned)blocks), dim3(256), lds, st, t);
    else
    TNG_CASE(128, 128, 2, 2) TNG_CASE(128, 64, 2, 2) TNG_CASE(128, 32, 4, 1)
    TNG_CASE(64, 128, 2, 2) TNG_CASE(64, 64, 2, 2) TNG_CASE(64, 32, 4, 1)
    TNG_CASE(32, 128, 1, 4) TNG_CASE(32, 64, 1, 4) TNG_CASE(32, 32, 2, 2)
    TNG_CASE(16, 128, 1, 4) TNG_CASE(16, 64, 1, 4)
    rc = HN_ERR_UNSUPPORTED;
#undef TNG_CASE
    if (rc != HN_OK) return rc;
    if (r.n) hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)rblocks), dim3(256), 0, st, r);
    HN_LAUNCH_CHECK();
}

// jobs: host table, 9 int64 per job {x, dz, dw, n_img, H, W, C, ldx, ldz}: stride-1 grouped 3x3 conv (group width 8, zero "same" padding)
static int gconv_group_plan(const long* jobs, int njobs, int* psplits, long* pps, long* ws_off, long* ws_floats) {
    HN_CHECK_ARG(jobs && njobs > 0 && njobs <= HN_TN_GROUP_MAX);
    long tiles = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + 9 * i;
        HN_CHECK_ARG(jb[0] && jb[1] && jb[2] && jb[3] > 0 && jb[4] > 0 && jb[5] > 0 && jb[6] >= 8 && (jb[6] & 7) == 0 && (jb[7] & 7) == 0 &&
                     (jb[8] & 7) == 0 && jb[8] >= jb[6]);
        tiles += cdiv(jb[6], 64);
    }
    long want = (g_hn_knob[13] + tiles - 1) / tiles;                  // (knob 13 = 384) ~1.5 workgroups per CU in total (78 KB of LDS each: two fit)
    *ws_floats = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + 9 * i;
        const long patches = jb[3] * cdiv(jb[4], 8) * cdiv(jb[5], 16);
        long w = want;
        if (w > patches / 2) w = patches / 2;
        if (w < 1) w = 1;
        pps[i] = (patches + w - 1) / w;
        psplits[i] = (int)((patches + pps[i] - 1) / pps[i]);
        ws_off[i] = *ws_floats;
        *ws_floats += (long)psplits[i] * 2 * jb[6] * 576;    // two k-split wave groups per workgroup: one block-diagonal slab each
    }
    return HN_OK;
}
extern "C" long hn_gconv_wgrad_group_ws_bytes(const long* jobs, int njobs) {
    int ps[HN_TN_GROUP_MAX]; long pp[HN_TN_GROUP_MAX], off[HN_TN_GROUP_MAX], total;
    if (gconv_group_plan(jobs, njobs, ps, pp, off, &total) != HN_OK) return -1;
    return total * 4 + 64;
}
/* Up to 32 grouped-3x3-conv weight gradients dw_j [C][8][3][3] (fp32) in one patch-kernel launch + one extract launch; jobs: HOST table,
 * 9 int64 per job {x, dz, dw, n_img, H, W, C, ldx, ldz} (x = the conv's bf16 input [n_img][H][W][C], dz = its output gradient, stride 1).
 * Reference op: the weight gradient of XBlock.conv_block_2 (net/anynet.py:34-38) of every identity block of a stage. */
extern "C" int hn_gconv_wgrad_group(const long* jobs, int njobs, float* workspace, hipStream_t st) {
    int ps[HN_TN_GROUP_MAX]; long pp[HN_TN_GROUP_MAX], off[HN_TN_GROUP_MAX], total;
    const int rc0 = gconv_group_plan(jobs, njobs, ps, pp, off, &total);
    if (rc0 != HN_OK) return rc0;
    HN_CHECK_ARG(workspace);
    static std::atomic<unsigned long long> optin{0};
    if (!lds_optin(optin, {(const void*)gconv_wgrad_group_kernel})) return HN_ERR_LAUNCH;
    PJobs t;
    RJobsN r;
    t.n = r.n = njobs;
    long blocks = 0, rblocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + 9 * i;
        PJob& d = t.j[i];
        d.x = reinterpret_cast<const bf16*>(jb[0]); d.dz = reinterpret_cast<const bf16*>(jb[1]); d.part = workspace + off[i];
        d.n_img = (int)jb[3]; d.H = (int)jb[4]; d.W = (int)jb[5]; d.C = (int)jb[6]; d.ldx = (int)jb[7]; d.ldz = (int)jb[8];
        d.gy = cdiv(d.C, 64); d.pps = (int)pp[i]; d.n_patches = d.n_img * cdiv(d.H, 8) * cdiv(d.W, 16);
        d.first_block = (int)blocks;
        blocks += (long)d.gy * ps[i];
        RJob& q = r.j[i];
        q.part = d.part; q.dw = reinterpret_cast<float*>(jb[2]); q.splits = ps[i] * 2; q.Nout = d.C; q.Cin = 8; q.KP = 64; q.taps = 9; q.kind = 2;
        q.first_block = rblocks;
        rblocks += cdiv((long)d.C * 72, 256);
    }
    const size_t xb = (size_t)((180 * 8 + 511) / 512) * 512 * 16;
    const size_t lds = 2 * ((size_t)((128 * 64 * 2 + 1023) / 1024 * 1024) + xb) + 2 * 56 * sizeof(int);    // + the two patch source tables
    hipLaunchKernelGGL(gconv_wgrad_group_kernel, dim3((unsigned)blocks), dim3(512), lds, st, t);
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)rblocks), dim3(256), 0, st, r);
    HN_LAUNCH_CHECK();
}

/* hn_conv_gemm_tn that also returns the conv's bias gradient dbias [Nout] = column sums of dz -- accumulated by one extra MFMA per k-step
 * while the dz fragments are in registers, reduced by the launch that reduces the weight-gradient slabs: no column-statistics pass over
 * dz, no extra reduce launches (not for the grouped mode 5). */
extern "C" int hn_conv_gemm_tn_bias(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                                    int up, long M, const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw,
                                    float* dbias, hipStream_t st) {
    HN_CHECK_ARG(dbias);
    return conv_gemm_tn_impl(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M, dz, ldz, Nout, KP, taps, 0, workspace, dw, dbias, st);
}
/* weight gradient of the phase-form conv (hn_conv3x3_phase mode 4): x0 = low-resolution input [N][H][W][C0], dz = space-to-depth output
 * gradient [N][H][W][Nout = 4*k] (hn_space_to_depth_bf16), dw = gradient of the EFFECTIVE weights fp32 [4*k][C0][3][3] (zeros at the
 * five taps a phase does not use).  workspace from hn_wgrad_plan_phase. */
extern "C" int hn_conv_gemm_tn_phase(const void* x0, int n_img, int H, int W, int C0, int ld0, const void* dz, int ldz, int Nout, int KP,
                                     int phase_span, float* workspace, float* dw, float* dbias_eff, hipStream_t st) {
    HN_CHECK_ARG(phase_span >= 64 && phase_span % 64 == 0 && Nout == 4 * phase_span && KP >= 64);
    return conv_gemm_tn_impl(x0, nullptr, 4, n_img, H, W, C0, 0, ld0, 0, 0, (long)n_img * H * W, dz, ldz, Nout, KP, 9, phase_span, workspace, dw,
                             dbias_eff, st);
}
static int conv_gemm_tn_impl(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1,
                             int up, long M, const void* dz, int ldz, int Nout, int KP, int taps, int phase_span, float* workspace, float* dw,
                             float* dbias, hipStream_t st) {
    HN_CHECK_ARG(x0 && dz && workspace && dw && M > 0 && (KP & 31) == 0 && (ldz & 7) == 0 && ldz >= ((Nout + 7) & ~7));
    HN_CHECK_ARG((C0 & 7) == 0 && (C1 & 7) == 0 && (ld0 & 7) == 0 && ((mode >= 0 && mode <= 2) || ((mode == 4 || mode == 5) && up == 0 && C1 == 0)));
    HN_CHECK_ARG(mode != 5 || (KP == 64 && Nout == C0 && taps == 9));
    const int grouped = mode == 5;
    HN_CHECK_ARG(mode == 0 || (long)n_img * H * W == M);
    int splits; long rps, wsb;
    wgrad_plan_impl(mode, n_img, H, W, M, Nout, KP, taps, phase_span, &splits, &rps, &wsb);
    GemmTN p;
    p.x = make_xsrc(x0, x1, mode, n_img, H, W, C0, C1, ld0, ld1, up, M);
    mode = p.x.mode;
    p.dz = (const bf16*)dz; p.ldz = ldz; p.Nout = Nout; p.KP = KP; p.taps = taps;
    p.part = workspace; p.rows_per_split = rps; p.phase_span = phase_span;
    p.bias_part = nullptr; p.out_ld = 0; p.cin_lim = 0; p.dbg = 0;
    int bc, bn, rc;
    long* defer = g_defer_job;
    g_defer_job = nullptr;
    if (defer) {
        defer[0] = (long)workspace; defer[1] = (long)dw; defer[2] = splits; defer[3] = Nout; defer[4] = C0 + C1; defer[5] = KP; defer[6] = taps;
        defer[7] = -1;
    }
    if (dbias && (grouped || defer)) return HN_ERR_UNSUPPORTED;
    if (dbias) p.bias_part = workspace + (long)splits * Nout * taps * KP;
    if (use_patch_wgrad(mode, Nout, KP)) {
        static std::atomic<unsigned long long> optin{0};
        if (!lds_optin(optin, {(const void*)wgrad3x3_patch_kernel<128, 64>, (const void*)wgrad3x3_patch_kernel<16, 64>,
                               (const void*)wgrad3x3_patch_kernel<64, 64>, (const void*)wgrad3x3_patch_kernel<32, 64>,
                               (const void*)wgrad3x3_patch_kernel<128, 32>, (const void*)wgrad3x3_patch_kernel<128, 64, 1>,
                               (const void*)wgrad3x3_patch_kernel<64, 64, 1>, (const void*)wgrad3x3_patch_kernel<128, 32, 1>}))
            return HN_ERR_LAUNCH;
        int pbc, pci, ksplit;
        patch_tiles(Nout, KP, pbc, pci, ksplit);
        if (grouped) { pbc = 64; pci = 64; ksplit = 2; }
        if (phase_span && phase_span < pbc) { pbc = 64; pci = 64; ksplit = 2; }
        p.gy = cdiv(Nout, pbc);
        const int patches = n_img * cdiv(H, 8) * cdiv(W, 16);
        dim3 grid((unsigned)(cdiv(KP, pci) * p.gy * (splits / ksplit)));
        const size_t xb = (size_t)((180 * (pci / 8) + 511) / 512) * 512 * 16;
        const size_t lds = 2 * ((size_t)((128 * pbc * 2 + 1023) / 1024 * 1024) + xb) + 2 * 56 * sizeof(int);   // + the two patch source tables
        if (grouped) hipLaunchKernelGGL((wgrad3x3_patch_kernel<64, 64>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else if (phase_span && pbc == 128 && pci == 32) hipLaunchKernelGGL((wgrad3x3_patch_kernel<128, 32, 1>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else if (phase_span && pbc == 128) hipLaunchKernelGGL((wgrad3x3_patch_kernel<128, 64, 1>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else if (phase_span) hipLaunchKernelGGL((wgrad3x3_patch_kernel<64, 64, 1>), grid, dim3(512), lds, st, p, (int)rps, patches);   // (phase_span >= 64)
        else if (pbc == 128 && pci == 32) hipLaunchKernelGGL((wgrad3x3_patch_kernel<128, 32>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else if (pbc == 128) hipLaunchKernelGGL((wgrad3x3_patch_kernel<128, 64>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else if (pbc == 64) hipLaunchKernelGGL((wgrad3x3_patch_kernel<64, 64>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else if (pbc == 32) hipLaunchKernelGGL((wgrad3x3_patch_kernel<32, 64>), grid, dim3(512), lds, st, p, (int)rps, patches);
        else hipLaunchKernelGGL((wgrad3x3_patch_kernel<16, 64>), grid, dim3(512), lds, st, p, (int)rps, patches);
        if (hipGetLastError() != hipSuccess) return HN_ERR_LAUNCH;
        const long cols = (long)Nout * taps * KP;
        if (defer && (grouped || !(taps == 9 && splits <= 64))) {
            defer[7] = grouped ? 2 : ((splits <= 128 && cols >= 65536) ? 0 : 1);
            return HN_OK;
        }
        if (grouped)
            hipLaunchKernelGGL(gconv_diag_extract_kernel, dim3(cdiv(Nout * 72, 256)), dim3(256), 0, st, workspace, dw, splits, Nout);
        else if (taps == 9 && splits <= 64)
            hipLaunchKernelGGL(wgrad_reduce9_kernel, dim3(cdiv(C0 + C1, 64), Nout), dim3(256), 0, st, workspace, dw, splits, Nout, C0 + C1, KP,
                               p.bias_part, dbias);
        else if (splits <= 128 && cols >= 65536)
            hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(cdiv(cols / 4, 256)), dim3(256), 0, st, workspace, dw, splits, Nout, C0 + C1, KP, taps,
                               p.bias_part, dbias);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(cols, 32)), dim3(512), 0, st, workspace, dw, splits, Nout, C0 + C1, KP, taps,
                               p.bias_part, dbias);
        HN_LAUNCH_CHECK();
    }
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
    if (defer) {
        defer[7] = (splits <= 128 && cols >= 65536) ? 0 : 1;
        return HN_OK;
    }
    if (splits <= 128 && cols >= 65536)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(cdiv(cols / 4, 256)), dim3(256), 0, st, workspace, dw, splits, Nout, C0 + C1, KP, taps,
                           p.bias_part, dbias);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(cols, 32)), dim3(512), 0, st, workspace, dw, splits, Nout, C0 + C1, KP, taps,
                           p.bias_part, dbias);
    HN_LAUNCH_CHECK();
}

/* jobs: DEVICE table of njobs x 8 int64 {w, wp, wt, Cout, Cin, taps, first_block, ci_tiles}; job j owns blocks [first_block_j,
 * first_block_{j+1}) of 256 threads, one per 32 x 32 (cout, cin) tile: (KP(Cout)/32) * ci_tiles with ci_tiles = KP(Cin)/32; total_blocks =
 * their sum.  taps <= 9. */
extern "C" int hn_pack_weights_batched(const long* jobs, int njobs, long total_blocks, const int* block_job, hipStream_t st) {
    HN_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0);
    hipLaunchKernelGGL(pack_w_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jobs, njobs, block_job);
    HN_LAUNCH_CHECK();
}

/* every other per-step weight pack of a model in one launch: jobs = DEVICE table njobs x 16 int64 (see pack_small_batched_kernel),
 * block_job = DEVICE int32 [total_blocks] job index of every 256-thread block */
extern "C" int hn_pack_small_batched(const long* jobs, const int* block_job, long total_blocks, hipStream_t st) {
    HN_CHECK_ARG(jobs && block_job && total_blocks > 0);
    hipLaunchKernelGGL(pack_small_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jobs, block_job);
    HN_LAUNCH_CHECK();
}

/* grouped 3x3 conv (group width 8) as block-diagonal 64-channel MFMA tiles: w fp32 [C][8][3][3] -> wk, wd bf16 [C][9][64] for
 * hn_conv_gemm_nt mode 5 (forward / stride-1 data gradient) */
extern "C" int hn_gconv_pack_diag(const float* w, void* wk, void* wd, int C, hipStream_t st) {
    HN_CHECK_ARG(w && wk && wd && C > 0 && (C & 7) == 0);
    hipLaunchKernelGGL(gconv_pack_diag_kernel, dim3(cdiv((long)C * 576, 256)), dim3(256), 0, st, w, (bf16*)wk, (bf16*)wd, C);
    HN_LAUNCH_CHECK();
}

/* packing of a channel slice [ci0, ci0+Cin) of w[Cout][Cin_total][taps] (phase = 0), or of the phase-form effective weights of that slice
 * (phase = 1, taps = 9: wp [4*Cout][9][KP(Cin)], wt [Cin][9][KP(4*Cout)], b_eff [4*Cout] = bias repeated per phase, optional) */
extern "C" int hn_pack_weight_ex(const float* w, void* wp, void* wt, int Cout, int Cin_total, int ci0, int Cin, int taps, int phase,
                                 const float* bias, float* b_eff, hipStream_t st) {
    HN_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && ci0 >= 0 && ci0 + Cin <= Cin_total && taps > 0 && (!phase || taps == 9) && (!b_eff || bias));
    const int CoutE = phase ? 4 * Cout : Cout;
    const int KPi = (Cin + 31) / 32 * 32, KPo = (CoutE + 31) / 32 * 32;
    const long total = (long)CoutE * taps * KPi + (wt ? (long)Cin * taps * KPo : 0) + (b_eff ? CoutE : 0);
    hipLaunchKernelGGL(pack_w_ex_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, (bf16*)wp, (bf16*)wt, Cout, Cin_total, ci0, Cin, taps,
                       phase, KPi, KPo, bias, b_eff);
    HN_LAUNCH_CHECK();
}

/* dw [K][C0+C1][3][3] (and db [K], optional) from the effective-weight gradient dw_eff [4K][C0][3][3] (db_eff [4K]) and the skip operand's
 * dw1 [K][C1][3][3] (C1 = 0: none) */
extern "C" int hn_phase_fold(const float* dw_eff, const float* dw1, const float* db_eff, float* dw, float* db, int K, int C0, int C1,
                             hipStream_t st) {
    HN_CHECK_ARG(dw_eff && dw && K > 0 && C0 > 0 && C1 >= 0 && (C1 == 0 || dw1) && (!db || db_eff));
    const long total = (long)K * (C0 + C1) * 9 + (db ? K : 0);
    hipLaunchKernelGGL(phase_fold_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, dw_eff, dw1, db_eff, dw, db, K, C0, C1);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_pack_weight(const float* w, void* wp, void* wt, int Cout, int Cin, int taps, hipStream_t st) {
    HN_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && taps > 0);
    const int KPi = (Cin + 31) / 32 * 32, KPo = (Cout + 31) / 32 * 32;
    const long total = (long)Cout * taps * KPi + (wt ? (long)Cin * taps * KPo : 0);
    hipLaunchKernelGGL(pack_w_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, (bf16*)wp, (bf16*)wt, Cout, Cin, taps, KPi, KPo);
    HN_LAUNCH_CHECK();
}
