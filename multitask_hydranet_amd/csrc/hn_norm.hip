// HBM-bound NHWC bf16 kernels: per-channel reductions (BatchNorm statistics, SE pooling, BN-backward sums),
// BatchNorm finalize / apply / backward, SE gating, and small elementwise helpers.
// All tensors are [rows][C] bf16 with an explicit row stride (so producers can write channel slices of a concat
// buffer); every thread moves 16 bytes (8 channels).  Reductions are deterministic: each workgroup writes one
// partial row, a second tiny kernel reduces the partial rows (no float atomics).
// Reference ops covered: nn.BatchNorm2d training-mode forward/backward incl. running statistics
// (net/anynet.py:13,31,36,54,59; net/common.py:98; net/bifpn.py:60-101; head_detect/detection.py:23,60;
// head_lane/lanedetect.py:47,54,61), ReLU / Swish (net/common.py:11-22) / residual add (net/anynet.py:75),
// SE pooling + gating (net/anynet.py:40-48,68-69), ELU backward (head_seg/segmentation.py:24).
#include "hn_common.h"

// ---------------------------------------------------------------------------------------------------------
// column reductions: block b reduces rows [b*R, (b+1)*R) and writes partial row b of two [prows][C] fp32 arrays
//   MODE 0 (stats):   o1 = sum x            o2 = sum x^2
//   MODE 1 (dot):     o1 = sum a*b          o2 = sum a
//   MODE 2 (bn bwd):  g = dout * act'(pre); o1 = sum g, o2 = sum g * xhat
// ---------------------------------------------------------------------------------------------------------
struct ColRed {
    const bf16* a; int lda;
    const bf16* b; int ldb;      // MODE 1: second operand; MODE 2: z (pre-BN conv output)
    const bf16* y; int ldy;      // MODE 2: optional saved block output (ReLU mask = y > 0)
    const float* scale; const float* shift; const float* mean; const float* rstd;
    int act;
    long M; int C; long R;
    float* o1; float* o2;
    Levels sg;                   // MODE 2 on level-packed rows (sg.n > 1): block's level -> coefficient set at + level*coef_stride
    int coef_stride;
};

template <int MODE>
__global__ __launch_bounds__(256) void colred_kernel(const ColRed p) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    __shared__ float red[4][256 * 8];                 // [0..1]: the block's partial sums; all four: MODE 2 coefficient staging (C <= 2048)
    const int C8 = p.C >> 3;
    const int tid = threadIdx.x;
    const int rpi = 256 / C8 > 0 ? 256 / C8 : 1;      // rows per iteration (C8 <= 256 enforced on the host)
    const int cg = tid % C8, rr = tid / C8;
    const bool active = rr < rpi;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    const long m0 = (long)bidx * p.R;
    long m1 = m0 + p.R;
    if (m1 > p.M) m1 = p.M;
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 2) {                                  // stage the 4 x C coefficients through LDS once per block (red is free until the end)
        float* cf = &red[0][0];
        const long cofs = p.sg.n > 1 ? (long)level_of_row(p.sg, (long)bidx * p.R) * p.coef_stride : 0;
        for (int i = tid; i < p.C; i += 256) {
            cf[i] = p.scale[cofs + i]; cf[p.C + i] = p.shift[cofs + i]; cf[2 * p.C + i] = p.mean[cofs + i]; cf[3 * p.C + i] = p.rstd[cofs + i];
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = cg * 8 + k;
                sc[k] = cf[c]; sh[k] = cf[p.C + c]; mu[k] = cf[2 * p.C + c]; rs[k] = cf[3 * p.C + c];
            }
        }
        __syncthreads();
    }
    if (active) {
        // two rows in flight per thread: the loads of both iterations are issued before either is consumed
        auto accum = [&](const bf16x8& va, const bf16x8& vb, const bf16x8& vy) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float v = bf2f(va[k]); s1[k] += v; s2[k] += v * v; }
            } else if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float v = bf2f(va[k]); s1[k] += v * bf2f(vb[k]); s2[k] += v; }
            } else {
                float z[8], g[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { z[k] = bf2f(vb[k]); g[k] = bf2f(va[k]); }
                if (p.y) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[k] = bf2f(vy[k]) > 0.f ? g[k] : 0.f;
                } else {
                    float pre[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) pre[k] = sc[k] * z[k] + sh[k];
                    act_bwd_n(pre, g, p.act);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    s1[k] += g[k];
                    s2[k] += g[k] * (z[k] - mu[k]) * rs[k];
                }
            }
        };
        long m = m0 + rr;
        for (; m + rpi < m1; m += 2 * rpi) {
            const long mb = m + rpi;
            const bf16x8 va0 = ld8(p.a + m * p.lda + cg * 8), va1 = ld8(p.a + mb * p.lda + cg * 8);
            bf16x8 vb0 = va0, vb1 = va1, vy0 = va0, vy1 = va1;
            if (MODE >= 1) { vb0 = ld8(p.b + m * p.ldb + cg * 8); vb1 = ld8(p.b + mb * p.ldb + cg * 8); }
            if (MODE == 2 && p.y) { vy0 = ld8(p.y + m * p.ldy + cg * 8); vy1 = ld8(p.y + mb * p.ldy + cg * 8); }
            accum(va0, vb0, vy0);
            accum(va1, vb1, vy1);
        }
        if (m < m1) {
            const bf16x8 va0 = ld8(p.a + m * p.lda + cg * 8);
            bf16x8 vb0 = va0, vy0 = va0;
            if (MODE >= 1) vb0 = ld8(p.b + m * p.ldb + cg * 8);
            if (MODE == 2 && p.y) vy0 = ld8(p.y + m * p.ldy + cg * 8);
            accum(va0, vb0, vy0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[0][tid * 8 + k] = s1[k]; red[1][tid * 8 + k] = s2[k]; }
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g8 = c >> 3, k = c & 7;
        float t1 = 0.f, t2 = 0.f;
        for (int r = 0; r < rpi; ++r) {
            t1 += red[0][(r * C8 + g8) * 8 + k];
            t2 += red[1][(r * C8 + g8) * 8 + k];
        }
        p.o1[(long)bidx * p.C + c] = t1;
        p.o2[(long)bidx * p.C + c] = t2;
    }
}

// out[g][c] = alpha * sum_{j < S} in[(g*S + j)][c]      block = 32 columns x 16 row lanes (coalesced rows, LDS tree over the lanes)
__global__ __launch_bounds__(512) void rows_reduce_kernel(const float* in, float* out, int G, int S, int C, float alpha) {
    __shared__ float red[16][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx, g = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const float* src = in + (long)g * S * C + c;
        for (int j = ty; j < S; j += 16) s += src[(long)j * C];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        out[(long)g * C + c] = t * alpha;
    }
}

// two arrays at once, ragged groups: outK[g][c] = sum of inK rows [g*S, min((g+1)*S, rows)), S = ceil(rows / G).  Used to fold the
// (up to M/64) per-wave partial statistic rows of a GEMM epilogue down to G rows before the finalize kernel.
__global__ __launch_bounds__(512) void rows_reduce2_kernel(const float* in1, const float* in2, float* out1, float* out2, int rows, int G,
                                                           int C) {
    __shared__ float red[2][16][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx, g = blockIdx.y;
    const int S = (rows + G - 1) / G;
    const int r0 = g * S;
    int r1 = r0 + S;
    if (r1 > rows) r1 = rows;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        if (in2) {
            for (int j = r0 + ty; j < r1; j += 16) { s1 += in1[(long)j * C + c]; s2 += in2[(long)j * C + c]; }
        } else {
            for (int j = r0 + ty; j < r1; j += 16) s1 += in1[(long)j * C + c];
        }
    }
    red[0][ty][tx] = s1; red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t1 += red[0][k][tx]; t2 += red[1][k][tx]; }
        out1[(long)g * C + c] = t1;
        if (in2) out2[(long)g * C + c] = t2;
    }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm finalize (training): one wave per channel reduces the partial rows, then derives
//   mean, biased var -> rstd, scale = gamma*rstd, shift = beta - mean*scale, and updates the running statistics
//   (unbiased variance, PyTorch momentum convention) -- F.batch_norm semantics.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* psum, const float* psq, int prows, int C, double count,
                                                          const float* gamma, const float* beta, float eps, float momentum,
                                                          float* running_mean, float* running_var, float* scale, float* shift,
                                                          float* mean, float* rstd) {
    __shared__ double r1[8][33], r2[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = ty; r < prows; r += 8) { s1 += psum[(long)r * C + c]; s2 += psq[(long)r * C + c]; }
    r1[ty][tx] = s1; r2[ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { s1 += r1[k][tx]; s2 += r2[k][tx]; }
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float rs = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rs;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mu * sc;
        mean[c] = (float)mu;
        rstd[c] = rs;
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

// eval-mode BN: scale/shift from the running statistics
__global__ void bn_eval_coeff_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                     float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// dgamma = sum g*xhat, dbeta = sum g, and the two per-channel means the apply pass needs
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* pg, const float* pgx, int prows, int C, double count,
                                                              float* dgamma, float* dbeta, float* mg, float* mgx) {
    __shared__ double r1[8][33], r2[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = ty; r < prows; r += 8) { s1 += pg[(long)r * C + c]; s2 += pgx[(long)r * C + c]; }
    r1[ty][tx] = s1; r2[ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { s1 += r1[k][tx]; s2 += r2[k][tx]; }
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        mg[c] = (float)(s1 / count);
        mgx[c] = (float)(s2 / count);
    }
}

// ---------------------------------------------------------------------------------------------------------
// elementwise kernels (grid-stride over 16-byte pieces)
// ---------------------------------------------------------------------------------------------------------
struct BnAct {
    const bf16* z; int ldz; const float* scale; const float* shift;
    const bf16* res; int ldr; const float* rscale; const float* rshift;   // optional residual (+ its own BN)
    int act; bf16* out; int ldo; long M; int C;
};
// thread = one 8-channel group; the block stages the per-channel coefficients in LDS once (a per-thread global fetch of 4 x 32 B of
// coefficients for every 16 B of data dominated the small layers), each thread keeps its slice in registers while it walks rows
// (256 / C8 rows per block pass), two rows in flight.
__global__ __launch_bounds__(256) void bn_act_kernel(const BnAct p) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    extern __shared__ float coefs[];                      // [4][C]: scale, shift, rscale, rshift
    for (int i = threadIdx.x; i < p.C; i += 256) {
        coefs[i] = p.scale ? p.scale[i] : 1.f;
        coefs[p.C + i] = p.scale ? p.shift[i] : 0.f;
        if (p.rscale) { coefs[2 * p.C + i] = p.rscale[i]; coefs[3 * p.C + i] = p.rshift[i]; }
    }
    __syncthreads();
    const int C8 = p.C >> 3;
    const int rpb = 256 / C8 > 0 ? 256 / C8 : 1;
    const int cg = threadIdx.x % C8, rr = threadIdx.x / C8;
    if (rr >= rpb) return;
    const int c = cg * 8;
    float sc[8], sh[8], rsc[8], rsh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = coefs[c + k];
        sh[k] = coefs[p.C + c + k];
        rsc[k] = p.rscale ? coefs[2 * p.C + c + k] : 1.f;
        rsh[k] = p.rscale ? coefs[3 * p.C + c + k] : 0.f;
    }
    auto apply = [&](const bf16x8& vz, const bf16x8& vr, long m) {
        bf16x8 o;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf2f(vz[k]) * sc[k] + sh[k];
        if (p.res) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bf2f(vr[k]) * rsc[k] + rsh[k];
        }
        act_fwd_n(v, p.act);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k]);
        st8(p.out + m * p.ldo + c, o);
    };
    const long step = (long)gridDim.x * rpb;
    long m = (long)bidx * rpb + rr;
    for (; m + step < p.M; m += 2 * step) {
        const long mb = m + step;
        const bf16x8 vz0 = ld8(p.z + m * p.ldz + c), vz1 = ld8(p.z + mb * p.ldz + c);
        bf16x8 vr0 = vz0, vr1 = vz1;
        if (p.res) { vr0 = ld8(p.res + m * p.ldr + c); vr1 = ld8(p.res + mb * p.ldr + c); }
        apply(vz0, vr0, m);
        apply(vz1, vr1, mb);
    }
    if (m < p.M) {
        const bf16x8 vz0 = ld8(p.z + m * p.ldz + c);
        bf16x8 vr0 = vz0;
        if (p.res) vr0 = ld8(p.res + m * p.ldr + c);
        apply(vz0, vr0, m);
    }
}

struct BnBwdApply {
    const bf16* dout; int ldd; const bf16* z; int ldz; const bf16* y; int ldy;
    const float* scale; const float* shift; const float* mean; const float* rstd; const float* mg; const float* mgx;
    int act; bf16* dz; int lddz; bf16* gout; int ldg; long M; int C;
};
// dz = scale * (g - mean(g) - xhat * mean(g*xhat)),  g = dout * act'(pre)   (optionally also emits g)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApply p) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    extern __shared__ float coefs[];                      // [6][C]: scale, shift, mean, rstd, mean(g), mean(g*xhat)
    for (int i = threadIdx.x; i < p.C; i += 256) {
        coefs[i] = p.scale[i]; coefs[p.C + i] = p.shift[i]; coefs[2 * p.C + i] = p.mean[i]; coefs[3 * p.C + i] = p.rstd[i];
        coefs[4 * p.C + i] = p.mg[i]; coefs[5 * p.C + i] = p.mgx[i];
    }
    __syncthreads();
    const int C8 = p.C >> 3;
    const int rpb = 256 / C8 > 0 ? 256 / C8 : 1;
    const int cg = threadIdx.x % C8, rr = threadIdx.x / C8;
    if (rr >= rpb) return;
    const int c = cg * 8;
    float sc[8], sh[8], mu[8], rs[8], mg[8], mgx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = coefs[c + k]; sh[k] = coefs[p.C + c + k]; mu[k] = coefs[2 * p.C + c + k]; rs[k] = coefs[3 * p.C + c + k];
        mg[k] = coefs[4 * p.C + c + k]; mgx[k] = coefs[5 * p.C + c + k];
    }
    auto apply = [&](const bf16x8& vd, const bf16x8& vz, const bf16x8& vy, long m) {
        bf16x8 o, og;
        float z[8], g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { z[k] = bf2f(vz[k]); g[k] = bf2f(vd[k]); }
        if (p.y) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = bf2f(vy[k]) > 0.f ? g[k] : 0.f;
        } else {
            float pre[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pre[k] = sc[k] * z[k] + sh[k];
            act_bwd_n(pre, g, p.act);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (z[k] - mu[k]) * rs[k];
            o[k] = f2bf(sc[k] * (g[k] - mg[k] - xh * mgx[k]));
            og[k] = f2bf(g[k]);
        }
        st8(p.dz + m * p.lddz + c, o);
        if (p.gout) st8(p.gout + m * p.ldg + c, og);
    };
    const long step = (long)gridDim.x * rpb;
    long m = (long)bidx * rpb + rr;
    for (; m + step < p.M; m += 2 * step) {
        const long mb = m + step;
        const bf16x8 vd0 = ld8(p.dout + m * p.ldd + c), vd1 = ld8(p.dout + mb * p.ldd + c);
        const bf16x8 vz0 = ld8(p.z + m * p.ldz + c), vz1 = ld8(p.z + mb * p.ldz + c);
        bf16x8 vy0 = vz0, vy1 = vz1;
        if (p.y) { vy0 = ld8(p.y + m * p.ldy + c); vy1 = ld8(p.y + mb * p.ldy + c); }
        apply(vd0, vz0, vy0, m);
        apply(vd1, vz1, vy1, mb);
    }
    if (m < p.M) {
        const bf16x8 vd0 = ld8(p.dout + m * p.ldd + c);
        const bf16x8 vz0 = ld8(p.z + m * p.ldz + c);
        bf16x8 vy0 = vz0;
        if (p.y) vy0 = ld8(p.y + m * p.ldy + c);
        apply(vd0, vz0, vy0, m);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Level-packed variants (det-head towers): rows of all pyramid levels in one tensor, BatchNorm parameters per level.  A block owns RB
// consecutive rows (RB divides every level's row count), looks its level up once and stages that level's coefficients in LDS.
// coef layout: [level][4][C] = scale, shift, mean, rstd;  red layout: [level][2][C] = mean(g), mean(g*xhat).
// ---------------------------------------------------------------------------------------------------------
struct BnLevels {
    const bf16* z; int ldz; const bf16* dout; int ldd; const bf16* y; int ldy;
    const float* coef; const float* red;
    int act; bf16* out; int ldo; int C; int RB;
    Levels sg;
};
template <bool BWD>
__global__ __launch_bounds__(256) void bn_levels_kernel(const BnLevels p) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    extern __shared__ float coefs[];                      // [4 or 6][C]
    const long m0 = (long)bidx * p.RB;
    const int lv = level_of_row(p.sg, m0);
    const float* cf = p.coef + (long)lv * 4 * p.C;
    for (int i = threadIdx.x; i < 4 * p.C; i += 256) coefs[i] = cf[i];
    if (BWD) {
        const float* rd = p.red + (long)lv * 2 * p.C;
        for (int i = threadIdx.x; i < 2 * p.C; i += 256) coefs[4 * p.C + i] = rd[i];
    }
    __syncthreads();
    const int C8 = p.C >> 3;
    const int rpb = 256 / C8 > 0 ? 256 / C8 : 1;
    const int cg = threadIdx.x % C8, rr = threadIdx.x / C8;
    if (rr >= rpb) return;
    const int c = cg * 8;
    float sc[8], sh[8], mu[8], rs[8], mg[8], mgx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = coefs[c + k]; sh[k] = coefs[p.C + c + k];
        if (BWD) { mu[k] = coefs[2 * p.C + c + k]; rs[k] = coefs[3 * p.C + c + k]; mg[k] = coefs[4 * p.C + c + k]; mgx[k] = coefs[5 * p.C + c + k]; }
    }
    long m1 = m0 + p.RB;
    const long total = p.sg.row_off[p.sg.n];
    if (m1 > total) m1 = total;
    auto one = [&](const bf16x8& vz, const bf16x8& vd, const bf16x8& vy, long m) {
        bf16x8 o;
        if (!BWD) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = bf2f(vz[k]) * sc[k] + sh[k];
            act_fwd_n(v, p.act);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k]);
        } else {
            float z[8], g[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { z[k] = bf2f(vz[k]); g[k] = bf2f(vd[k]); }
            if (p.y) {
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = bf2f(vy[k]) > 0.f ? g[k] : 0.f;
            } else {
                float pre[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pre[k] = sc[k] * z[k] + sh[k];
                act_bwd_n(pre, g, p.act);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = (z[k] - mu[k]) * rs[k];
                o[k] = f2bf(sc[k] * (g[k] - mg[k] - xh * mgx[k]));
            }
        }
        st8(p.out + m * p.ldo + c, o);
    };
    long m = m0 + rr;
    for (; m + rpb < m1; m += 2 * rpb) {
        const long mb = m + rpb;
        const bf16x8 vz0 = ld8(p.z + m * p.ldz + c), vz1 = ld8(p.z + mb * p.ldz + c);
        bf16x8 vd0 = vz0, vd1 = vz1, vy0 = vz0, vy1 = vz1;
        if (BWD) { vd0 = ld8(p.dout + m * p.ldd + c); vd1 = ld8(p.dout + mb * p.ldd + c); }
        if (BWD && p.y) { vy0 = ld8(p.y + m * p.ldy + c); vy1 = ld8(p.y + mb * p.ldy + c); }
        one(vz0, vd0, vy0, m);
        one(vz1, vd1, vy1, mb);
    }
    if (m < m1) {
        const bf16x8 vz0 = ld8(p.z + m * p.ldz + c);
        bf16x8 vd0 = vz0, vy0 = vz0;
        if (BWD) vd0 = ld8(p.dout + m * p.ldd + c);
        if (BWD && p.y) vy0 = ld8(p.y + m * p.ldy + c);
        one(vz0, vd0, vy0, m);
    }
}

// per-level finalize: block (x = 32-channel chunk, y = level) reduces that level's partial rows [row_off[l]/div, row_off[l+1]/div)
struct FinLevels {
    const float* p1; const float* p2; int C; int div; float eps, momentum;
    Levels sg;
    long count[HN_MAX_LEVELS];
    const float* gamma[HN_MAX_LEVELS]; const float* beta[HN_MAX_LEVELS]; float* rm[HN_MAX_LEVELS]; float* rv[HN_MAX_LEVELS];
    float* dgamma[HN_MAX_LEVELS]; float* dbeta[HN_MAX_LEVELS];
    float* out;                  // fwd: coef [level][4][C]; bwd: red [level][2][C]
    float* zero;                 // bwd (optional): C zeros written by level 0 (gradient of the conv bias that feeds these BatchNorms)
    const float* bias;           // fwd, ragged packing: the conv bias -- the (rows - count) alignment rows of a level hold exactly
                                 // bf16(bias[c]) (their conv input is zero), which is subtracted from the sums
};
template <bool BWD>
__global__ __launch_bounds__(1024) void bn_finalize_levels_kernel(const FinLevels p) {
    // block = 8 channels x 128 row lanes: level 0 of the det towers has 2048 partial rows (16 per lane)
    __shared__ double r1[128][9], r2[128][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx, lv = blockIdx.y;
    const long rb = p.sg.row_off[lv] / p.div, re = p.sg.row_off[lv + 1] / p.div;
    double s1 = 0.0, s2 = 0.0;
    if (c < p.C)
        for (long r = rb + ty; r < re; r += 128) { s1 += p.p1[r * p.C + c]; s2 += p.p2[r * p.C + c]; }
    r1[ty][tx] = s1; r2[ty][tx] = s2;
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {
        if (ty < st) { r1[ty][tx] += r1[ty + st][tx]; r2[ty][tx] += r2[ty + st][tx]; }
        __syncthreads();
    }
    if (ty == 0 && c < p.C) {
        s1 = r1[0][tx]; s2 = r2[0][tx];
        const double count = (double)p.count[lv];
        if (!BWD) {
            const long npad = (p.sg.row_off[lv + 1] - p.sg.row_off[lv]) - p.count[lv];
            if (npad > 0 && p.bias) {
                const double q = (double)bfround(p.bias[c]);
                s1 -= (double)npad * q;
                s2 -= (double)npad * q * q;
            }
            float* coef = p.out + (long)lv * 4 * p.C;
            const double mu = s1 / count;
            double var = s2 / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float rs = (float)(1.0 / sqrt(var + (double)p.eps));
            const float sc = p.gamma[lv][c] * rs;
            coef[c] = sc;
            coef[p.C + c] = p.beta[lv][c] - (float)mu * sc;
            coef[2 * p.C + c] = (float)mu;
            coef[3 * p.C + c] = rs;
            if (p.rm[lv]) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                p.rm[lv][c] = (1.f - p.momentum) * p.rm[lv][c] + p.momentum * (float)mu;
                p.rv[lv][c] = (1.f - p.momentum) * p.rv[lv][c] + p.momentum * (float)unb;
            }
        } else {
            float* red = p.out + (long)lv * 2 * p.C;
            p.dbeta[lv][c] = (float)s1;
            p.dgamma[lv][c] = (float)s2;
            if (p.zero && lv == 0) p.zero[c] = 0.f;
            red[c] = (float)(s1 / count);
            red[p.C + c] = (float)(s2 / count);
        }
    }
}

// out[m][c] = x[m][c] * gate[m / HW][c]
__global__ __launch_bounds__(256) void scale_rows_kernel(const bf16* x, int ldx, const float* gate, long HW, bf16* out, int ldo,
                                                         long M, int C) {
    const int C8 = C >> 3;
    const long total = M * C8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / C8;
        const int c = (int)(idx - m * C8) * 8;
        const float* g = gate + (m / HW) * C + c;
        const bf16x8 v = ld8(x + m * ldx + c);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(v[k]) * g[k]);
        st8(out + m * ldo + c, o);
    }
}

// SE backward, data path: db = dout * gate[n][c] + dpool[n][c] / HW
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const bf16* dout, int ldd, const float* gate, const float* dpool, long HW,
                                                           bf16* db, int ldb, long M, int C) {
    const int C8 = C >> 3;
    const long total = M * C8;
    const float inv = 1.0f / (float)HW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / C8;
        const int c = (int)(idx - m * C8) * 8;
        const long n = m / HW;
        const bf16x8 v = ld8(dout + m * ldd + c);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(v[k]) * gate[n * C + c + k] + dpool[n * C + c + k] * inv);
        st8(db + m * ldb + c, o);
    }
}

// generic binary / unary elementwise ops on [M][C] bf16:
//   op 0: out = a + b          op 1: out = a * act'(y) with y = post-activation (ELU: y>0 ? 1 : y+1 ; RELU: y>0)
//   op 2: out = alpha * a      op 3: out = act(a)      op 4: out = a * act'(b) with b = PRE-activation
struct Ew { const bf16* a; int lda; const bf16* b; int ldb; bf16* out; int ldo; long M; int C; int op; int act; float alpha; };
__global__ __launch_bounds__(256) void ew_kernel(const Ew p) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = p.C >> 3;
    const long total = p.M * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / C8;
        const int c = (int)(idx - m * C8) * 8;
        const bf16x8 va = ld8(p.a + m * p.lda + c);
        bf16x8 vb;
        if (p.b) vb = ld8(p.b + m * p.ldb + c);
        bf16x8 o;
        float a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = bf2f(va[k]); b[k] = p.b ? bf2f(vb[k]) : 0.f; }
        if (p.op == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += b[k];
        } else if (p.op == 1) {
            if (p.act == HN_ACT_ELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = b[k] > 0.f ? a[k] : a[k] * (b[k] + 1.0f);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = b[k] > 0.f ? a[k] : 0.f;
            }
        } else if (p.op == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] *= p.alpha;
        } else if (p.op == 3) act_fwd_n(a, p.act);
        else act_bwd_n(b, a, p.act);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(a[k]);
        st8(p.out + m * p.ldo + c, o);
    }
}

// dx[n, 2y, 2x, :] += dxs[n, y, x, :]   (backward of a stride-2 1x1 conv's row gather)
__global__ __launch_bounds__(256) void add_strided2_kernel(bf16* dx, int ldx, const bf16* dxs, int lds_, int N, int Ho, int Wo, int C) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3;
    const long total = (long)N * Ho * Wo * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long ms = idx / C8;
        const int c = (int)(idx - ms * C8) * 8;
        const int ox = (int)(ms % Wo);
        const long t = ms / Wo;
        const int oy = (int)(t % Ho);
        const long n = t / Ho;
        const long m = (n * 2 * Ho + 2 * oy) * (2L * Wo) + 2 * ox;
        const bf16x8 a = ld8(dx + m * ldx + c), b = ld8(dxs + ms * lds_ + c);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(a[k]) + bf2f(b[k]));
        st8(dx + m * ldx + c, o);
    }
}

// fp32 [M][C] (row stride lds) -> bf16 [M][ldo] zero padded; and back
__global__ void cast_pad_kernel(const float* src, int lds_, bf16* dst, int ldo, long M, int C) {
    const long total = M * ldo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / ldo;
        const int c = (int)(idx - m * ldo);
        dst[idx] = f2bf(c < C ? src[m * lds_ + c] : 0.f);
    }
}
// ---------------------------------------------------------------------------------------------------------
// SE excitation MLP on the pooled vectors (net/anynet.py:42-47): hid = relu(W1 p + b1), gate = sigmoid(W2 hid + b2).
// One block per image; forward keeps hid and gate for the backward pass.
// ---------------------------------------------------------------------------------------------------------
// out[n][o] = act(bias[o] + sum_i W[o][i] * in[n][i]) : one wave per (n, o), lanes stride the contraction (coalesced weight rows)
// S > 0: `in` holds S partial rows per image (in[n][i] = alpha * sum_j part[(n*S + j)][i], e.g. the SE squeeze from the per-row-block
// channel sums of hn_bn_apply_fused); the o == 0 wave also stores the assembled vector to `store` [N][I] (kept for the backward pass).
__global__ __launch_bounds__(256) void se_fc_rows_kernel(const float* W, const float* bias, const float* in, float* out, int N, int O, int I, int act) {
    // image-major wave order with the row-order placement convention (hn_common.h): image n's waves run on the XCD that holds its rows
    const long wid = (long)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long)N * O) return;
    const int n = (int)(wid / O), o = (int)(wid - (long)n * O);
    const float* wr = W + (long)o * I;
    const float* xr = in + (long)n * I;
    const float bo = bias[o];                                          // with the operands: behind the sum it is a trip of its own
    float s = 0.f;
    for (int i = lane; i < I; i += 64) s += wr[i] * xr[i];
    s = wave_sum(s);
    if (lane == 0) {
        s += bo;
        out[wid] = act == HN_ACT_RELU ? (s > 0.f ? s : 0.f) : 1.f / (1.f + __expf(-s));
    }
}

// The same layer fed by S partial rows per image: in[n][i] = alpha * sum_j part[n * S + j][i] (the SE squeeze from the per-row-block channel
// sums of hn_bn_apply_fused); the o == 0 wave also stores the assembled vector to `store` [N][I] (kept for the backward pass).  One wave per
// (n, two outputs).  A lane owns elements lane, lane + 64, ... (<= 16 per round of 1024).  The partial rows were just written by another
// kernel (another XCD's L2): every DEPENDENT round of loads is a trip to memory (~2 us), so the weights and the first 1 + TR partial rows
// (S <= 8 is what hn_fused_row_block leaves per image) are all issued before the first add -- one trip.  (The nested "for i { for j < S }"
// form made every load wait for the previous one: 20 us at stage 4; rounds of four rows: 7.3 us.)  Two outputs per wave: half the waves
// re-reading the image's partial rows, and all of them resident at once (stage 4: 1872 waves at three per SIMD).
template <int TR>
__global__ __launch_bounds__(256) void se_fc_parts_kernel(const float* W, const float* bias, const float* in, float* out, int N, int O, int I,
                                                          int act, int S, float alpha, float* store) {
    const int OP = (O + 1) >> 1;                                       // output pairs per image
    const long wid = (long)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long)N * OP) return;
    const int n = (int)(wid / OP), o = 2 * (int)(wid - (long)n * OP);
    const bool two = o + 1 < O;
    const float* wr0 = W + (long)o * I;
    const float* wr1 = W + (long)(two ? o + 1 : o) * I;
    const float bo0 = bias[o], bo1 = bias[two ? o + 1 : o];           // with the operands: behind the sums they are a trip of their own
    float s0 = 0.f, s1 = 0.f;
    for (int base = 0; base < I; base += 1024) {
        float v[16], w0[16], w1[16], t[TR][16];
        const float* xr = in + (long)n * S * I + base + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool ok = base + lane + 64 * k < I;
            v[k] = ok ? xr[64 * k] : 0.f;
            w0[k] = ok ? wr0[base + lane + 64 * k] : 0.f;
            w1[k] = ok ? wr1[base + lane + 64 * k] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < TR; ++jj)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                t[jj][k] = (1 + jj < S && base + lane + 64 * k < I) ? xr[(long)(1 + jj) * I + 64 * k] : 0.f;
#pragma unroll
        for (int jj = 0; jj < TR; ++jj)
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += t[jj][k];
        for (int j0 = 1 + TR; j0 < S; j0 += 4) {                         // more rows: rounds of four
            float u[4][16];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    u[jj][k] = (j0 + jj < S && base + lane + 64 * k < I) ? xr[(long)(j0 + jj) * I + 64 * k] : 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] += u[jj][k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] *= alpha;
            if (store && o == 0 && base + lane + 64 * k < I) store[(long)n * I + base + lane + 64 * k] = v[k];
            s0 += w0[k] * v[k];
            s1 += w1[k] * v[k];
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
        s0 += bo0;
        out[(long)n * O + o] = act == HN_ACT_RELU ? (s0 > 0.f ? s0 : 0.f) : 1.f / (1.f + __expf(-s0));
        if (two) {
            s1 += bo1;
            out[(long)n * O + o + 1] = act == HN_ACT_RELU ? (s1 > 0.f ? s1 : 0.f) : 1.f / (1.f + __expf(-s1));
        }
    }
}

// out[n][o] = mask(o) * sum_i W[i][o] * f(in)[n][i]   (contraction over the ROW index of W: coalesced over o).
// block = 16 outputs x 64 partitions of i (a workgroup streams its weight slab at only ~30 GB/s, so the slabs are kept small and many:
// 64-output tiles took 12.7 us per launch at stage 4).  pre: 0 = in as is, 1 = in * g * (1 - g) with g = aux[n][i] (sigmoid', result also stored to
// `store`).  post: 0 none, 1 = zero where aux2[n][o] <= 0 (ReLU').
// S > 0: `in` holds S partial rows per image (summed on the fly: the SE gate gradient from the per-row-block sums of hn_se_bwd_reduce_fused)
template <int PARTS>
__global__ __launch_bounds__(16 * PARTS) void se_fc_cols_kernel(const float* W, const float* in, const float* aux, float* store, const float* aux2,
                                                                float* out, int N, int O, int I, int pre, int post, int S) {
    __shared__ float red[PARTS][17];
    const int ox = threadIdx.x & 15, part = threadIdx.x >> 4;         // 16 outputs x PARTS partitions of the contraction
    const int lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);      // image-major, contiguous per XCD (hn_common.h)
    const int bx = lid % gridDim.x, n = lid / gridDim.x;
    const int o = bx * 16 + ox;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int i0 = (int)((long)I * part / PARTS), i1 = (int)((long)I * (part + 1) / PARTS);
    const float* inr = in + (long)n * (S > 0 ? S : 1) * I;
    const float* auxr = aux ? aux + (long)n * I : nullptr;
    const bool ov = o < O;
    // the partition's elements (<= 16 per round): the weights, the gate values and up to four partial rows are issued together -- every
    // dependent round of loads of this fresh data is a trip to memory (~2 us; the form with one round per group of rows and the gate
    // loads behind them took three trips: 9.9 us per launch at stage 4).  PARTS == 16 (the second launch: one dense input row, no gate)
    // is the lean form: the extra rows' registers would cost its 3776 waves at stage 4 a second round of workgroups.
    constexpr int TR = PARTS == 64 ? 3 : 0;
    for (int ib = i0; ib < i1; ib += 16) {
        float v[16], wv[16], gv[16], t[TR + 1][16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool ok = ib + k < i1;
            v[k] = ok ? inr[ib + k] : 0.f;
            wv[k] = (ok && ov) ? W[(long)(ib + k) * O + o] : 0.f;
            gv[k] = (TR && ok && pre) ? auxr[ib + k] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < TR; ++jj)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                t[jj][k] = (1 + jj < S && ib + k < i1) ? inr[(long)(1 + jj) * I + ib + k] : 0.f;
#pragma unroll
        for (int jj = 0; jj < TR; ++jj)
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += t[jj][k];
        constexpr int UR = PARTS == 64 ? 2 : 4;                        // more rows: UR per round (1024-thread workgroups: 128 VGPRs, no spills)
        for (int j0 = 1 + TR; j0 < S; j0 += UR) {
            float u[UR][16];
#pragma unroll
            for (int jj = 0; jj < UR; ++jj)
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    u[jj][k] = (j0 + jj < S && ib + k < i1) ? inr[(long)(j0 + jj) * I + ib + k] : 0.f;
#pragma unroll
            for (int jj = 0; jj < UR; ++jj)
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] += u[jj][k];
        }
        if (pre) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (ib + k < i1) {
                    const float g = TR ? gv[k] : auxr[ib + k];
                    v[k] *= g * (1.f - g);
                    if (store && bx == 0 && ox == 0) store[(long)n * I + ib + k] = v[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            s0 += wv[k] * v[k];
            s1 += wv[k + 1] * v[k + 1];
            s2 += wv[k + 2] * v[k + 2];
            s3 += wv[k + 3] * v[k + 3];
        }
    }
    // (requested in front of the barrier, not behind the reduction where it was a trip of its own; not at the top: one register too many)
    const float mask = (post && ov && part == 0) ? aux2[(long)n * O + o] : 1.f;
    red[part][ox] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && ov) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < PARTS; ++k) t += red[k][ox];
        if (post && mask <= 0.f) t = 0.f;
        out[(long)n * O + o] = t;
    }
}

// parameter gradients (sums over the N images) as outer-product tiles on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): D[i][j] = sum_n P[n][i] * Q[n][j]
//   job 0: dW2[c][j] = sum dpre2[n][c] hid[n][j]  (+ db2[c] = sum dpre2[n][c] from the j-tile 0 waves)
//   job 1: dW1[j][c] = sum dpre1[n][j] pooled[n][c]  (+ db1[j])
// one wave per 16x16 tile, 4 waves per workgroup
struct SeOuter { const float* P; int PI; const float* Q; int QJ; float* D; float* dbias; int tiles_j; int tiles; };
__global__ __launch_bounds__(256) void se_mlp_wgrad_kernel(const SeOuter j0, const SeOuter j1, int N) {
    const int lane = threadIdx.x & 63;
    int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const SeOuter& jb = tile < j0.tiles ? j0 : j1;
    if (tile >= j0.tiles) tile -= j0.tiles;
    if (tile >= jb.tiles) return;
    const int ti = tile / jb.tiles_j, tj = tile - ti * jb.tiles_j;
    const int r = lane & 15, kk = lane >> 4;
    const int i = ti * 16 + r, j = tj * 16 + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float colsum = 0.f;
    for (int n0 = 0; n0 < N; n0 += 4) {
        const int n = n0 + kk;
        const float a = (n < N && i < jb.PI) ? jb.P[(long)n * jb.PI + i] : 0.f;
        const float b = (n < N && j < jb.QJ) ? jb.Q[(long)n * jb.QJ + j] : 0.f;
        colsum += a;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oi = ti * 16 + 4 * kk + q;
        if (oi < jb.PI && j < jb.QJ) jb.D[(long)oi * jb.QJ + j] = acc[q];
    }
    if (tj == 0) {                                                    // the bias gradient: sum over the images of P[n][i]
        colsum += __shfl_xor(colsum, 16);
        colsum += __shfl_xor(colsum, 32);
        if (kk == 0 && i < jb.PI) jb.dbias[i] = colsum;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static inline int ew_grid(long pieces) {
    long b = (pieces + 255) / 256;
    if (b > 2048 * 4) b = 2048 * 4;
    if (b < 1) b = 1;
    return (int)b;
}

// grid for the row-walking elementwise kernels: 256/C8 rows per block pass, capped at ~16 blocks per CU
static inline int row_grid(long M, int C) {
    const int C8 = C >> 3;
    const int rpb = 256 / C8 > 0 ? 256 / C8 : 1;
    long b = (M + 2 * rpb - 1) / (2 * rpb);       // >= 2 rows per thread where the tensor allows
    if (b > 2048) b = 2048;
    return (int)(b < 1 ? 1 : b);
}

// rows per block for a column reduction over M rows (<= 1024 partial rows, >= 8 rows each; a divisor of `align` if given)
extern "C" long hn_colred_rows(long M, long align) {
    long R = (M + 1023) / 1024;
    if (R < 8) R = 8;
    if (align > 0) {
        if (R > align) R = align;                 // keep partial rows inside one image (align = H*W)
        while (align % R) --R;
    }
    return R;
}

static int launch_colred(int mode, const ColRed& p, hipStream_t st) {
    HN_CHECK_ARG(p.a && p.o1 && p.o2 && (p.C & 7) == 0 && p.C <= 2048 && p.M > 0 && p.R > 0 && (p.lda & 7) == 0);
    const int grid = cdiv(p.M, p.R);
    if (mode == 0) hipLaunchKernelGGL(colred_kernel<0>, dim3(grid), dim3(256), 0, st, p);
    else if (mode == 1) hipLaunchKernelGGL(colred_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(colred_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_col_stats(const void* x, int ldx, long M, int C, long R, float* psum, float* psq, hipStream_t st) {
    ColRed p = {};
    p.a = (const bf16*)x; p.lda = ldx; p.M = M; p.C = C; p.R = R; p.o1 = psum; p.o2 = psq;
    return launch_colred(0, p, st);
}

extern "C" int hn_col_dot(const void* a, int lda, const void* b, int ldb, long M, int C, long R, float* pdot, float* psum,
                          hipStream_t st) {
    HN_CHECK_ARG(b && (ldb & 7) == 0);
    ColRed p = {};
    p.a = (const bf16*)a; p.lda = lda; p.b = (const bf16*)b; p.ldb = ldb; p.M = M; p.C = C; p.R = R; p.o1 = pdot; p.o2 = psum;
    return launch_colred(1, p, st);
}

extern "C" int hn_rows_reduce(const float* in, float* out, int G, int S, int C, float alpha, hipStream_t st) {
    HN_CHECK_ARG(in && out && G > 0 && S > 0 && C > 0);
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv(C, 32), G), dim3(512), 0, st, in, out, G, S, C, alpha);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_rows_reduce2(const float* in1, const float* in2, float* out1, float* out2, int rows, int G, int C, hipStream_t st) {
    HN_CHECK_ARG(in1 && out1 && (!in2 || out2) && rows > 0 && G > 0 && C > 0);       // in2/out2 optional
    hipLaunchKernelGGL(rows_reduce2_kernel, dim3(cdiv(C, 32), G), dim3(512), 0, st, in1, in2, out1, out2, rows, G, C);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_bn_finalize(const float* psum, const float* psq, int prows, int C, long count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                              float* shift, float* mean, float* rstd, hipStream_t st) {
    HN_CHECK_ARG(psum && psq && prows > 0 && C > 0 && count > 0 && gamma && beta && scale && shift && mean && rstd);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, psum, psq, prows, C, (double)count, gamma, beta, eps, momentum,
                       running_mean, running_var, scale, shift, mean, rstd);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_bn_eval_coeff(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                float* scale, float* shift, hipStream_t st) {
    HN_CHECK_ARG(gamma && beta && rm && rv && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, rm, rv, eps, C, scale, shift);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_bn_act(const void* z, int ldz, const float* scale, const float* shift, const void* res, int ldr,
                         const float* rscale, const float* rshift, int act, void* out, int ldo, long M, int C, hipStream_t st) {
    HN_CHECK_ARG(z && out && M > 0 && (C & 7) == 0 && (ldz & 7) == 0 && (ldo & 7) == 0 && (!res || (ldr & 7) == 0));
    HN_CHECK_ARG(C <= 2048);
    BnAct p = {(const bf16*)z, ldz, scale, shift, (const bf16*)res, ldr, rscale, rshift, act, (bf16*)out, ldo, M, C};
    hipLaunchKernelGGL(bn_act_kernel, dim3(row_grid(M, C)), dim3(256), (size_t)4 * C * sizeof(float), st, p);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_bn_bwd_reduce(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* scale,
                                const float* shift, const float* mean, const float* rstd, int act, long M, int C, long R, float* pg,
                                float* pgx, hipStream_t st) {
    HN_CHECK_ARG(z && scale && shift && mean && rstd && (ldz & 7) == 0 && (!y || (ldy & 7) == 0));
    ColRed p = {};
    p.a = (const bf16*)dout; p.lda = ldd; p.b = (const bf16*)z; p.ldb = ldz; p.y = (const bf16*)y; p.ldy = ldy;
    p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd; p.act = act; p.M = M; p.C = C; p.R = R; p.o1 = pg; p.o2 = pgx;
    return launch_colred(2, p, st);
}

extern "C" int hn_bn_bwd_finalize(const float* pg, const float* pgx, int prows, int C, long count, float* dgamma, float* dbeta,
                                  float* mg, float* mgx, hipStream_t st) {
    HN_CHECK_ARG(pg && pgx && prows > 0 && C > 0 && count > 0 && dgamma && dbeta && mg && mgx);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, pg, pgx, prows, C, (double)count, dgamma, dbeta, mg, mgx);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_bn_bwd_apply(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* scale,
                               const float* shift, const float* mean, const float* rstd, const float* mg, const float* mgx, int act,
                               void* dz, int lddz, void* gout, int ldg, long M, int C, hipStream_t st) {
    HN_CHECK_ARG(dout && z && dz && M > 0 && (C & 7) == 0 && ((ldd | ldz | lddz) & 7) == 0 && (!y || (ldy & 7) == 0) &&
                 (!gout || (ldg & 7) == 0));
    BnBwdApply p = {(const bf16*)dout, ldd, (const bf16*)z, ldz, (const bf16*)y, ldy, scale, shift, mean, rstd, mg, mgx, act,
                    (bf16*)dz, lddz, (bf16*)gout, ldg, M, C};
    HN_CHECK_ARG(C <= 2048 && scale && shift && mean && rstd && mg && mgx);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(row_grid(M, C)), dim3(256), (size_t)6 * C * sizeof(float), st, p);
    HN_LAUNCH_CHECK();
}

static int fill_row_levels(Levels& L, int nlev, const long* rows, long block) {
    HN_CHECK_ARG(nlev >= 1 && nlev <= HN_MAX_LEVELS && rows && block > 0);
    L.n = nlev;
    L.row_off[0] = 0;
    for (int l = 0; l < nlev; ++l) {
        HN_CHECK_ARG(rows[l] > 0 && rows[l] % block == 0);       // blocks never straddle a level
        L.H[l] = 0; L.W[l] = 0;
        L.row_off[l + 1] = L.row_off[l] + rows[l];
    }
    return HN_OK;
}

/* Level-packed BatchNorm (per-level parameters) for the det-head towers; see bn_levels_kernel. */
extern "C" int hn_bn_act_levels(const void* z, int ldz, const float* coef, int act, void* out, int ldo, int C, int nlev, const long* rows,
                                hipStream_t st) {
    HN_CHECK_ARG(z && coef && out && (C & 7) == 0 && C <= 2048 && ((ldz | ldo) & 7) == 0);
    BnLevels p = {};
    p.z = (const bf16*)z; p.ldz = ldz; p.coef = coef; p.act = act; p.out = (bf16*)out; p.ldo = ldo; p.C = C; p.RB = 128;
    const int rc = fill_row_levels(p.sg, nlev, rows, p.RB);
    if (rc != HN_OK) return rc;
    hipLaunchKernelGGL(bn_levels_kernel<false>, dim3(cdiv(p.sg.row_off[nlev], p.RB)), dim3(256), (size_t)4 * C * sizeof(float), st, p);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_bn_bwd_apply_levels(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef,
                                      const float* red, int act, void* dz, int lddz, int C, int nlev, const long* rows, hipStream_t st) {
    HN_CHECK_ARG(dout && z && coef && red && dz && (C & 7) == 0 && C <= 2048 && ((ldd | ldz | lddz) & 7) == 0 && (!y || (ldy & 7) == 0));
    BnLevels p = {};
    p.z = (const bf16*)z; p.ldz = ldz; p.dout = (const bf16*)dout; p.ldd = ldd; p.y = (const bf16*)y; p.ldy = ldy;
    p.coef = coef; p.red = red; p.act = act; p.out = (bf16*)dz; p.ldo = lddz; p.C = C; p.RB = 128;
    const int rc = fill_row_levels(p.sg, nlev, rows, p.RB);
    if (rc != HN_OK) return rc;
    hipLaunchKernelGGL(bn_levels_kernel<true>, dim3(cdiv(p.sg.row_off[nlev], p.RB)), dim3(256), (size_t)6 * C * sizeof(float), st, p);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_bn_bwd_reduce_levels(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act,
                                       int C, long R, int nlev, const long* rows, float* pg, float* pgx, hipStream_t st) {
    HN_CHECK_ARG(z && coef && (ldz & 7) == 0 && (!y || (ldy & 7) == 0));
    ColRed p = {};
    p.a = (const bf16*)dout; p.lda = ldd; p.b = (const bf16*)z; p.ldb = ldz; p.y = (const bf16*)y; p.ldy = ldy;
    p.scale = coef; p.shift = coef + C; p.mean = coef + 2 * C; p.rstd = coef + 3 * C; p.coef_stride = 4 * C;
    p.act = act; p.C = C; p.R = R; p.o1 = pg; p.o2 = pgx;
    const int rc = fill_row_levels(p.sg, nlev, rows, R);
    if (rc != HN_OK) return rc;
    p.M = p.sg.row_off[nlev];
    return launch_colred(2, p, st);
}
static int fill_fin(FinLevels& p, const float* p1, const float* p2, int div, int C, int nlev, const long* rows, const long* count) {
    HN_CHECK_ARG(p1 && p2 && div > 0 && C > 0 && count);
    p.p1 = p1; p.p2 = p2; p.C = C; p.div = div;
    const int rc = fill_row_levels(p.sg, nlev, rows, div);
    if (rc != HN_OK) return rc;
    for (int l = 0; l < nlev; ++l) { HN_CHECK_ARG(count[l] > 0); p.count[l] = count[l]; }
    return HN_OK;
}
/* partial rows psum/psq: one per `div` tensor rows (64 for the hn_conv_gemm_nt epilogue statistics); coef out: [nlev][4][C] */
extern "C" int hn_bn_finalize_levels(const float* psum, const float* psq, int div, int C, int nlev, const long* rows, const long* count,
                                     const void* const* gamma, const void* const* beta, void* const* running_mean,
                                     void* const* running_var, float eps, float momentum, const float* conv_bias, float* coef,
                                     hipStream_t st) {
    HN_CHECK_ARG(gamma && beta && coef);
    FinLevels p = {};
    const int rc = fill_fin(p, psum, psq, div, C, nlev, rows, count);
    if (rc != HN_OK) return rc;
    p.eps = eps; p.momentum = momentum; p.out = coef; p.bias = conv_bias;
    for (int l = 0; l < nlev; ++l) {
        HN_CHECK_ARG(gamma[l] && beta[l]);
        p.gamma[l] = (const float*)gamma[l]; p.beta[l] = (const float*)beta[l];
        p.rm[l] = running_mean ? (float*)running_mean[l] : nullptr;
        p.rv[l] = running_var ? (float*)running_var[l] : nullptr;
    }
    hipLaunchKernelGGL(bn_finalize_levels_kernel<false>, dim3(cdiv(C, 8), nlev), dim3(1024), 0, st, p);
    HN_LAUNCH_CHECK();
}
/* red out: [nlev][2][C] = mean(g), mean(g*xhat); dgamma/dbeta: per-level fp32 [C] */
extern "C" int hn_bn_bwd_finalize_levels(const float* pg, const float* pgx, int div, int C, int nlev, const long* rows, const long* count,
                                         void* const* dgamma, void* const* dbeta, float* red, float* zero_c, hipStream_t st) {
    HN_CHECK_ARG(dgamma && dbeta && red);
    FinLevels p = {};
    const int rc = fill_fin(p, pg, pgx, div, C, nlev, rows, count);
    if (rc != HN_OK) return rc;
    p.out = red;
    p.zero = zero_c;
    for (int l = 0; l < nlev; ++l) {
        HN_CHECK_ARG(dgamma[l] && dbeta[l]);
        p.dgamma[l] = (float*)dgamma[l]; p.dbeta[l] = (float*)dbeta[l];
    }
    hipLaunchKernelGGL(bn_finalize_levels_kernel<true>, dim3(cdiv(C, 8), nlev), dim3(1024), 0, st, p);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_scale_rows(const void* x, int ldx, const float* gate, long HW, void* out, int ldo, long M, int C, hipStream_t st) {
    HN_CHECK_ARG(x && gate && out && HW > 0 && M > 0 && (C & 7) == 0 && ((ldx | ldo) & 7) == 0);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, st, (const bf16*)x, ldx, gate, HW, (bf16*)out, ldo,
                       M, C);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_se_bwd_apply(const void* dout, int ldd, const float* gate, const float* dpool, long HW, void* db, int ldb, long M,
                               int C, hipStream_t st) {
    HN_CHECK_ARG(dout && gate && dpool && db && HW > 0 && M > 0 && (C & 7) == 0 && ((ldd | ldb) & 7) == 0);
    hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, st, (const bf16*)dout, ldd, gate, dpool, HW,
                       (bf16*)db, ldb, M, C);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_eltwise(int op, const void* a, int lda, const void* b, int ldb, void* out, int ldo, long M, int C, int act,
                          float alpha, hipStream_t st) {
    HN_CHECK_ARG(a && out && M > 0 && (C & 7) == 0 && ((lda | ldo) & 7) == 0 && op >= 0 && op <= 4);
    HN_CHECK_ARG((op == 2 || op == 3) || (b && (ldb & 7) == 0));
    Ew p = {(const bf16*)a, lda, (const bf16*)b, ldb, (bf16*)out, ldo, M, C, op, act, alpha};
    hipLaunchKernelGGL(ew_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

// out += b0 [+ b1] [+ b2] in one pass (fp32 sum, one bf16 rounding): the gradient of a tensor with several consumers that do not
// accumulate in place themselves (ops.Share: the last BiFPN cell's maps feed the seg, det and lane heads) -- pairwise adds were one
// launch and one rounding per extra consumer
__global__ __launch_bounds__(256) void add_n_kernel(bf16* out, int ldo, const bf16* b0, int ld0, const bf16* b1, int ld1, const bf16* b2, int ld2,
                                                    long M, int C) {
    const int bidx = xcd_remap(blockIdx.x, gridDim.x);     // row-order placement convention (hn_common.h)
    const int C8 = C >> 3;
    const long total = M * C8;
    for (long idx = (long)bidx * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / C8;
        const int c = (int)(idx - m * C8) * 8;
        const bf16x8 vo = ld8(out + m * ldo + c), v0 = ld8(b0 + m * ld0 + c);
        bf16x8 v1 = zero8(), v2 = zero8();
        if (b1) v1 = ld8(b1 + m * ld1 + c);
        if (b2) v2 = ld8(b2 + m * ld2 + c);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(((bf2f(vo[k]) + bf2f(v0[k])) + bf2f(v1[k])) + bf2f(v2[k]));
        st8(out + m * ldo + c, o);
    }
}
extern "C" int hn_add_n(void* out, int ldo, const void* b0, int ld0, const void* b1, int ld1, const void* b2, int ld2, long M, int C,
                        hipStream_t st) {
    HN_CHECK_ARG(out && b0 && M > 0 && (C & 7) == 0 && ((ldo | ld0) & 7) == 0 && (!b1 || (ld1 & 7) == 0) && (!b2 || (b1 && (ld2 & 7) == 0)));
    hipLaunchKernelGGL(add_n_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, st, (bf16*)out, ldo, (const bf16*)b0, ld0, (const bf16*)b1, ld1,
                       (const bf16*)b2, ld2, M, C);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_add_strided2(void* dx, int ldx, const void* dxs, int lds_, int N, int Ho, int Wo, int C, hipStream_t st) {
    HN_CHECK_ARG(dx && dxs && N > 0 && Ho > 0 && Wo > 0 && (C & 7) == 0 && ((ldx | lds_) & 7) == 0);
    hipLaunchKernelGGL(add_strided2_kernel, dim3(ew_grid((long)N * Ho * Wo * (C >> 3))), dim3(256), 0, st, (bf16*)dx, ldx,
                       (const bf16*)dxs, lds_, N, Ho, Wo, C);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_cast_f32_to_bf16_pad(const float* src, int lds_, void* dst, int ldo, long M, int C, hipStream_t st) {
    HN_CHECK_ARG(src && dst && M > 0 && C > 0 && ldo >= C);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(ew_grid(M * ldo)), dim3(256), 0, st, src, lds_, (bf16*)dst, ldo, M, C);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* hid, float* gate,
                             int N, int C, int Cs, hipStream_t st) {
    HN_CHECK_ARG(pooled && w1 && b1 && w2 && b2 && hid && gate && N > 0 && C > 0 && Cs > 0);
    hipLaunchKernelGGL(se_fc_rows_kernel, dim3(cdiv((long)N * Cs, 4)), dim3(256), 0, st, w1, b1, pooled, hid, N, Cs, C, HN_ACT_RELU);
    hipLaunchKernelGGL(se_fc_rows_kernel, dim3(cdiv((long)N * C, 4)), dim3(256), 0, st, w2, b2, (const float*)hid, gate, N, C, Cs, HN_ACT_SIGMOID);
    HN_LAUNCH_CHECK();
}

// the same MLP fed by S partial squeeze rows per image (pool_part [N*S][C], pooled = alpha * their sum, stored for the backward pass)
extern "C" int hn_se_mlp_fwd_parts(const float* pool_part, int S, float alpha, const float* w1, const float* b1, const float* w2, const float* b2,
                                   float* pooled, float* hid, float* gate, int N, int C, int Cs, hipStream_t st) {
    HN_CHECK_ARG(pool_part && S > 0 && w1 && b1 && pooled && hid && (!gate || (w2 && b2)) && N > 0 && C > 0 && Cs > 0);
    const dim3 grid(cdiv((long)N * ((Cs + 1) / 2), 4));
    if (S <= 4) hipLaunchKernelGGL(se_fc_parts_kernel<3>, grid, dim3(256), 0, st, w1, b1, pool_part, hid, N, Cs, C, HN_ACT_RELU, S, alpha, pooled);
    else hipLaunchKernelGGL(se_fc_parts_kernel<7>, grid, dim3(256), 0, st, w1, b1, pool_part, hid, N, Cs, C, HN_ACT_RELU, S, alpha, pooled);
    if (gate)                                                           // null: the second layer runs in hn_se_gate_apply's prologue
        hipLaunchKernelGGL(se_fc_rows_kernel, dim3(cdiv((long)N * C, 4)), dim3(256), 0, st, w2, b2, (const float*)hid, gate, N, C, Cs,
                           HN_ACT_SIGMOID);
    HN_LAUNCH_CHECK();
}

static int se_mlp_bwd_impl(const float* dgate, int S, const float* gate, const float* hid, const float* pooled, const float* w1, const float* w2,
                           float* dpre2, float* dpre1, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int N, int C, int Cs,
                           hipStream_t st);
extern "C" int hn_se_mlp_bwd(const float* dgate, const float* gate, const float* hid, const float* pooled, const float* w1, const float* w2,
                             float* dpre2, float* dpre1, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int N, int C, int Cs,
                             hipStream_t st) {
    return se_mlp_bwd_impl(dgate, 0, gate, hid, pooled, w1, w2, dpre2, dpre1, dpool, dw1, db1, dw2, db2, N, C, Cs, st);
}
/* dgate given as S partial rows per image ([N*S][C], summed on the fly) */
extern "C" int hn_se_mlp_bwd_parts(const float* dgate_part, int S, const float* gate, const float* hid, const float* pooled, const float* w1,
                                   const float* w2, float* dpre2, float* dpre1, float* dpool, float* dw1, float* db1, float* dw2, float* db2,
                                   int N, int C, int Cs, hipStream_t st) {
    HN_CHECK_ARG(S > 0);
    return se_mlp_bwd_impl(dgate_part, S, gate, hid, pooled, w1, w2, dpre2, dpre1, dpool, dw1, db1, dw2, db2, N, C, Cs, st);
}
static int se_mlp_bwd_impl(const float* dgate, int S, const float* gate, const float* hid, const float* pooled, const float* w1, const float* w2,
                           float* dpre2, float* dpre1, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int N, int C, int Cs,
                           hipStream_t st) {
    HN_CHECK_ARG(dgate && gate && hid && pooled && w1 && w2 && dpre2 && dpre1 && dpool && N > 0 && C > 0 && Cs > 0);
    HN_CHECK_ARG((dw1 && db1 && dw2 && db2) || (!dw1 && !db1 && !dw2 && !db2));       // all four, or none (the caller defers them: hn_grad_tail)
    // dpre1[n][j] = [hid > 0] * sum_c W2[c][j] * (dgate * g (1-g))[n][c]        (also stores dpre2)
    // (contraction over C: 64 partitions; over Cs = C/4 below: 16)
    hipLaunchKernelGGL(se_fc_cols_kernel<64>, dim3(cdiv(Cs, 16), N), dim3(1024), 0, st, w2, dgate, gate, dpre2, hid, dpre1, N, Cs, C, 1, 1, S);
    // dpool[n][c] = sum_j W1[j][c] * dpre1[n][j]
    hipLaunchKernelGGL(se_fc_cols_kernel<16>, dim3(cdiv(C, 16), N), dim3(256), 0, st, w1, (const float*)dpre1, (const float*)nullptr,
                       (float*)nullptr, (const float*)nullptr, dpool, N, C, Cs, 0, 0, 0);
    if (!dw1) { HN_LAUNCH_CHECK(); }
    const int tc = cdiv(C, 16), ts = cdiv(Cs, 16);
    const SeOuter j0 = {dpre2, C, hid, Cs, dw2, db2, ts, tc * ts};
    const SeOuter j1 = {dpre1, Cs, pooled, C, dw1, db1, tc, tc * ts};
    hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3(cdiv(2L * tc * ts, 4)), dim3(256), 0, st, j0, j1, N);
    HN_LAUNCH_CHECK();
}


// =====================================================================================================================================
// Deferred parameter-gradient tails in ONE launch (hn_grad_tail).  The last step of many parameter gradients is a tiny reduction that is
// not on the backward pass's critical path -- the partial-row fold of a depthwise / stride-2 grouped conv weight gradient, the weight
// normalisation Jacobian of a BiFPN fusion node, the two outer products of an SE block's MLP -- and each one used to be its own ~5 us
// launch (~120 of them per step).  They are queued (ops.GradQueue) and run together at a segment boundary; jobs travel by value.
//   kind 0: out[c] = sum_r a[r][c]                      a: [n0 rows][n1 cols]
//   kind 1: BiFPN fusion weights (fuse_dweights_kernel): a = pw [n0 blocks][3], b = raw parameter [n1], out = d(raw) [n1], f0 = eps
//   kind 2: SE outer product (se_mlp_wgrad_kernel): out[i][j] = sum_n a[n][i] * b[n][j], out2[i] = sum_n a[n][i];  a: [n2][n0], b: [n2][n1]
// =====================================================================================================================================
#define HN_TAIL_MAX 64
struct TailJob { const float* a; const float* b; float* out; float* out2; int kind, n0, n1, n2; float f0; int first_block; };
struct TailJobs { TailJob j[HN_TAIL_MAX]; int n; };

__global__ __launch_bounds__(256) void grad_tail_kernel(const TailJobs jobs) {
    __shared__ float red[8][33];                                      // (kind 0 uses it flat: 256 floats)
    int ji = 0;
    for (int k = 1; k < jobs.n; ++k)
        if ((int)blockIdx.x >= jobs.j[k].first_block) ji = k;
    const TailJob& jb = jobs.j[ji];
    const int blk = (int)blockIdx.x - jb.first_block;
    if (jb.kind == 0) {                                              // cw columns x (256 / cw) row lanes, four independent loads per lane and round
        // (cw = jb.n2 = 32 ... 4: tall folds -- the stride-2 grouped conv / stem partial rows, up to 4096 of them -- get more row lanes;
        // with 32 columns x 8 lanes a 4096-row fold was 128 dependent load rounds, 65 us for 27 MB)
        const int cw = jb.n2, nl = 256 / cw;
        const int tx = threadIdx.x % cw, ty = threadIdx.x / cw;
        const int c = blk * cw + tx, R = jb.n0, C = jb.n1;
        float* redf = &red[0][0];
        float s = 0.f;
        if (c < C) {
            const float* src = jb.a + c;
            int r = ty;
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (; r + 3 * nl < R; r += 4 * nl) {
                s += src[(long)r * C];
                s1 += src[(long)(r + nl) * C];
                s2 += src[(long)(r + 2 * nl) * C];
                s3 += src[(long)(r + 3 * nl) * C];
            }
            for (; r < R; r += nl) s += src[(long)r * C];
            s = (s + s1) + (s2 + s3);
        }
        redf[ty * cw + tx] = s;
        __syncthreads();
        if (ty == 0 && c < C) {
            float t = 0.f;
            for (int k = 0; k < nl; ++k) t += redf[k * cw + tx];
            jb.out[c] = t;
        }
    } else if (jb.kind == 1) {
        float a[3] = {0.f, 0.f, 0.f};
        for (int b = threadIdx.x; b < jb.n0; b += 256)
            for (int i = 0; i < 3; ++i) a[i] += jb.a[b * 3 + i];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int i = 0; i < 3; ++i) {
            const float t = wave_sum(a[i]);
            if (lane == 0) red[wave][i] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nw = jb.n1;
            const float eps = jb.f0;
            float dw[3], r[3], sum = 0.f, dot = 0.f;
            for (int i = 0; i < 3; ++i) {
                dw[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
                r[i] = i < nw ? fmaxf(jb.b[i], 0.f) : 0.f;
                sum += r[i];
            }
            for (int i = 0; i < 3; ++i) dot += dw[i] * r[i] / (sum + eps);
            for (int i = 0; i < nw; ++i) jb.out[i] = jb.b[i] > 0.f ? (dw[i] - dot) / (sum + eps) : 0.f;
        }
    } else {
        const int PI = jb.n0, QJ = jb.n1, N = jb.n2;
        const int tiles_j = (QJ + 15) >> 4, tiles = ((PI + 15) >> 4) * tiles_j;
        const int lane = threadIdx.x & 63;
        const int tile = blk * 4 + (threadIdx.x >> 6);
        if (tile >= tiles) return;
        const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
        const int r = lane & 15, kk = lane >> 4;
        const int i = ti * 16 + r, j = tj * 16 + r;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float colsum = 0.f;
        for (int n0 = 0; n0 < N; n0 += 4) {
            const int n = n0 + kk;
            const float a = (n < N && i < PI) ? jb.a[(long)n * PI + i] : 0.f;
            const float b = (n < N && j < QJ) ? jb.b[(long)n * QJ + j] : 0.f;
            colsum += a;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oi = ti * 16 + 4 * kk + q;
            if (oi < PI && j < QJ) jb.out[(long)oi * QJ + j] = acc[q];
        }
        if (tj == 0) {
            colsum += __shfl_xor(colsum, 16);
            colsum += __shfl_xor(colsum, 32);
            if (kk == 0 && i < PI) jb.out2[i] = colsum;
        }
    }
}

/* jobs: HOST table, 8 int64 per job {kind, a, b, out, out2, n0, n1, n2 | float bits of f0 (kind 1)}; njobs <= 64 */
extern "C" int hn_grad_tail(const long* jobs, int njobs, hipStream_t st) {
    HN_CHECK_ARG(jobs && njobs > 0 && njobs <= HN_TAIL_MAX);
    TailJobs t;
    t.n = njobs;
    long blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const long* jb = jobs + 8 * i;
        TailJob& d = t.j[i];
        d.kind = (int)jb[0];
        d.a = reinterpret_cast<const float*>(jb[1]); d.b = reinterpret_cast<const float*>(jb[2]);
        d.out = reinterpret_cast<float*>(jb[3]); d.out2 = reinterpret_cast<float*>(jb[4]);
        d.n0 = (int)jb[5]; d.n1 = (int)jb[6]; d.n2 = 0; d.f0 = 0.f;
        HN_CHECK_ARG(d.kind >= 0 && d.kind <= 2 && d.a && d.out && d.n0 > 0 && d.n1 > 0);
        d.first_block = (int)blocks;
        if (d.kind == 0) {
            int cw = 32;                                             // fewer columns per block = more row lanes for tall folds
            while (cw > 4 && d.n0 > 16 * (256 / cw)) cw >>= 1;
            d.n2 = cw;
            blocks += cdiv(d.n1, cw);
        }
        else if (d.kind == 1) {
            HN_CHECK_ARG(d.b && d.n1 <= 3);
            const unsigned bits = (unsigned)jb[7];
            __builtin_memcpy(&d.f0, &bits, 4);
            blocks += 1;
        } else {
            d.n2 = (int)jb[7];
            HN_CHECK_ARG(d.b && d.out2 && d.n2 > 0);
            blocks += cdiv((long)cdiv(d.n0, 16) * cdiv(d.n1, 16), 4);
        }
    }
    hipLaunchKernelGGL(grad_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t);
    HN_LAUNCH_CHECK();
}
