// Fused BatchNorm passes: the per-channel "finalize" step (partial statistics -> mean / rstd / scale / shift, running-statistics update;
// partial gradient sums -> dgamma / dbeta / the two means) runs in the PROLOGUE of the elementwise kernel that consumes it, so a
// training-mode BatchNorm is two launches forward-free (statistics come out of the producing conv's epilogue) + one apply launch, and
// reduce + apply backward -- no finalize launches, no partial-row fold launches.  SE pooling rides on the BN2 pass of an XBlock, the SE
// gate gradient and the gated operand of conv_block_3's weight gradient come out of one pass over (dbg, z2).
//
// Layout: tensors are [rows][C] bf16 (row stride ld*).  grid = (channel chunks of <= 128 channels, row blocks of RB rows); a workgroup
// owns one chunk (<= 16 lanes of 8 channels: up to two 128-byte lines per row; C <= 128 is ONE chunk so whole rows stay contiguous) and
// RB rows, so its prologue reduces only P x 128 partial values.  Measured on MI355X (tools/bench_fused.py): the passes want ~1000
// workgroups (RB 32..512), the reduce pass of the backward pair a separate, larger RB (its row count is the apply pass's P).
// Everything is deterministic: fixed-order LDS reductions, no float atomics.
// Reference ops: nn.BatchNorm2d training forward/backward (net/anynet.py:31,36,54,59; net/common.py:98; head_lane/lanedetect.py:47-61),
// ReLU / Swish, residual add (net/anynet.py:75), SE squeeze / excite gating (net/anynet.py:40-48,68-69).
#include "hn_common.h"

#define FCH 128

// Logical (channel chunk, row block) of this workgroup: row-block-major logical order, contiguous per XCD like the GEMMs' xcd_remap -- XCD k
// owns rows [k M / 8, (k + 1) M / 8) in the passes AND in the GEMMs that produce / consume them (the row-order placement convention of
// hn_common.h; knob 8 = 0 turns it off for A/B runs: 701 -> 708 img/s from these passes alone).
__device__ __forceinline__ void fused_block(int xcd, int& bx, int& by) {
    if (!xcd) { bx = blockIdx.x; by = blockIdx.y; return; }
    const int nwg = gridDim.x * gridDim.y, hw = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = hw & 7;
    const int lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (hw >> 3);
    bx = lid % gridDim.x; by = lid / gridDim.x;
}

struct BnSrc {
    const float* psum; const float* psq;   // [P][C] partial sums / sums of squares
    int P;                                 // > 0: finalize from the partial rows; 0: read coef; < 0: eval mode (running statistics)
    double count;
    const float* gamma; const float* beta;
    float eps, momentum;
    float* rm; float* rv;                  // running statistics: updated by row block 0 when P > 0; read when P < 0
    float* coef;                           // [4][C] scale, shift, mean, rstd: written by row block 0 (P != 0), read (P == 0); null = identity
};

struct FusedLds {
    double r1[8][FCH + 1], r2[8][FCH + 1];
    float coef[6][FCH];                    // scale, shift, mean, rstd, mean(g), mean(g*xhat)
    float fr[2][2][FCH];
};

// sum of P partial rows for the chunk's channels: out in lds.r1[0][c], lds.r2[0][c] (doubles), c < nch.  32 channel quads x 8 row lanes.
__device__ __forceinline__ void chunk_partial_sums(const float* p1, const float* p2, int P, int C, int c0, int nch, FusedLds& L) {
    const int tid = threadIdx.x, q = tid & 31, rl = tid >> 5;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (q * 4 < nch) {
        const float* a = p1 + c0 + q * 4;
        const float* b = p2 + c0 + q * 4;
        // PSR rows (2 PSR loads) per round: every round is a trip to the producer's partial rows (P = 128, 16 rows per lane: three
        // rounds; four with rounds of 4; eight per round costs the apply kernel its fourth workgroup per CU)
        constexpr int PSR = 6;
        for (int r0 = rl; r0 < P; r0 += 8 * PSR) {
            f32x4 va[PSR], vb[PSR];
#pragma unroll
            for (int i = 0; i < PSR; ++i) {
                const int r = r0 + 8 * i < P ? r0 + 8 * i : rl;          // (a row past the end: re-read the lane's first row, added as zero)
                va[i] = *reinterpret_cast<const f32x4*>(a + (long)r * C);
                vb[i] = *reinterpret_cast<const f32x4*>(b + (long)r * C);
            }
#pragma unroll
            for (int i = 0; i < PSR; ++i) {
                const bool ok = r0 + 8 * i < P;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += ok ? (double)va[i][k] : 0.0; s2[k] += ok ? (double)vb[i][k] : 0.0; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { L.r1[rl][q * 4 + k] = s1[k]; L.r2[rl][q * 4 + k] = s2[k]; }
    __syncthreads();
    if (tid < FCH) {
        double t1 = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { t1 += L.r1[k][tid]; t2 += L.r2[k][tid]; }
        L.r1[0][tid] = t1; L.r2[0][tid] = t2;       // only thread `tid` reads/writes column tid here
    }
    __syncthreads();
}

// coefficient set of the chunk into L.coef[0..3]; `writer` (row block 0) also stores it / updates the running statistics
template <bool STATS = true>   // STATS false: an instance without the partial-statistics path (its loads' registers: 114 vs 85 VGPRs in the apply pass)
__device__ __forceinline__ void chunk_coefs(const BnSrc& b, int C, int c0, int nch, bool writer, FusedLds& L) {
    const int tid = threadIdx.x;
    if (STATS && b.P > 0) {
        // gamma / beta / the running statistics are read BEFORE the partial sums (behind them every one of these loads was a dependent
        // trip of its own -- rm, then rv, in the writing workgroup: ~3 us on the kernel's critical path)
        const int cc = c0 + (tid < nch ? tid : 0);
        const float ga = b.gamma[cc], be = b.beta[cc];
        const bool upd = writer && b.rm && tid < nch;
        const float rm0 = upd ? b.rm[cc] : 0.f, rv0 = upd ? b.rv[cc] : 0.f;
        chunk_partial_sums(b.psum, b.psq, b.P, C, c0, nch, L);
        if (tid < nch) {
            const int c = c0 + tid;
            const double mu = L.r1[0][tid] / b.count;
            double var = L.r2[0][tid] / b.count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float rs = (float)(1.0 / sqrt(var + (double)b.eps));
            const float sc = ga * rs;
            const float sh = be - (float)mu * sc;
            L.coef[0][tid] = sc; L.coef[1][tid] = sh; L.coef[2][tid] = (float)mu; L.coef[3][tid] = rs;
            if (writer) {
                if (b.coef) { b.coef[c] = sc; b.coef[C + c] = sh; b.coef[2 * C + c] = (float)mu; b.coef[3 * C + c] = rs; }
                if (b.rm) {
                    const double unb = b.count > 1.0 ? var * b.count / (b.count - 1.0) : var;
                    b.rm[c] = (1.f - b.momentum) * rm0 + b.momentum * (float)mu;
                    b.rv[c] = (1.f - b.momentum) * rv0 + b.momentum * (float)unb;
                }
            }
        }
    } else if (tid < nch) {
        const int c = c0 + tid;
        if (b.P < 0) {                                                     // eval mode: nn.BatchNorm2d with running statistics
            const float rs = 1.0f / sqrtf(b.rv[c] + b.eps);
            const float sc = b.gamma[c] * rs;
            L.coef[0][tid] = sc; L.coef[1][tid] = b.beta[c] - b.rm[c] * sc; L.coef[2][tid] = b.rm[c]; L.coef[3][tid] = rs;
            if (writer && b.coef) { b.coef[c] = sc; b.coef[C + c] = L.coef[1][tid]; b.coef[2 * C + c] = b.rm[c]; b.coef[3 * C + c] = rs; }
        } else if (b.coef) {
            L.coef[0][tid] = b.coef[c]; L.coef[1][tid] = b.coef[C + c]; L.coef[2][tid] = b.coef[2 * C + c]; L.coef[3][tid] = b.coef[3 * C + c];
        } else {
            L.coef[0][tid] = 1.f; L.coef[1][tid] = 0.f; L.coef[2][tid] = 0.f; L.coef[3][tid] = 1.f;
        }
    }
    __syncthreads();
}

// per-channel sums over the workgroup's row lanes of up to two per-thread accumulators -> o1/o2[blockIdx.y][c] (fixed order)
template <int NARR>
__device__ __forceinline__ void chunk_row_reduce(float (&a1)[8], float (&a2)[8], bool active, int cln, int rln, int cl, int rl, int c0, int nch,
                                                 int C, float* o1, float* o2, float* scratch /* [NARR][256*8] */, FusedLds& L, int by) {
    const int tid = threadIdx.x;
    if (active) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            scratch[(rl * cln + cl) * 8 + k] = a1[k];
            if (NARR == 2) scratch[2048 + (rl * cln + cl) * 8 + k] = a2[k];
        }
    }
    __syncthreads();
    const int c = tid & 127, j = tid >> 7;
    float t1 = 0.f, t2 = 0.f;
    if (c < nch) {
        const int g8 = c >> 3, k = c & 7;
        for (int r = j; r < rln; r += 2) {
            t1 += scratch[(r * cln + g8) * 8 + k];
            if (NARR == 2) t2 += scratch[2048 + (r * cln + g8) * 8 + k];
        }
    }
    L.fr[0][j][c] = t1; L.fr[1][j][c] = t2;
    __syncthreads();
    if (tid < nch) {
        const long o = (long)by * C + c0 + tid;
        o1[o] = L.fr[0][0][tid] + L.fr[0][1][tid];
        if (NARR == 2) o2[o] = L.fr[1][0][tid] + L.fr[1][1][tid];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: out = act(bn(z) [+ res]); optional per-row-block channel sums of the bf16-rounded output (SE squeeze), optional no store
// ---------------------------------------------------------------------------------------------------------------------------------
struct FApply {
    const bf16* z; int ldz;
    BnSrc bn;
    const bf16* res; int ldr;
    int act;
    bf16* out; int ldo;
    float* pool;                    // [gridDim.y][C] or null
    const float* gate; long HW;     // optional: out = bf16(act(bn(z))) * gate[row / HW][c]  (SE excite; RB divides HW)
    long M; int C; long RB; int cw; int xcd;
};

template <bool STATS>              // false: P <= 0 launches (eval-mode / given coefficients: the inference passes, the gated second pass over z2)
__global__ __launch_bounds__(256) void fused_apply_kernel(const FApply p) {
    __shared__ FusedLds L;
    __shared__ float scratch[2048];
    int bx, by;
    fused_block(p.xcd, bx, by);
    const int c0 = bx * p.cw;
    const int nch = p.C - c0 < p.cw ? p.C - c0 : p.cw;
    const int cln = nch >> 3, rln = 256 / cln;
    const int tid = threadIdx.x, cl = tid % cln, rl = tid / cln;
    const bool active = rl < rln;
    const long m0 = (long)by * p.RB;
    long m1 = m0 + p.RB;
    if (m1 > p.M) m1 = p.M;
    const int c = c0 + cl * 8;
    // Everything the workgroup reads first is issued before the prologue's first barrier -- the row pieces of the first iteration and
    // the gate here, the BatchNorm parameters in chunk_coefs -- so that the prologue's trip to the partial statistics and the first trip
    // to the rows are ONE trip (they were two, plus one for the gate: ~1 us each on an 5-9 us launch, 159 launches per training step).
    long m = m0 + rl;
    const bool h0 = active && m < m1, h1 = active && m + rln < m1;
    bf16x8 pz0 = {}, pz1 = {}, pr0 = {}, pr1 = {};
    if (h0) pz0 = ld8(p.z + m * p.ldz + c);
    if (h1) pz1 = ld8(p.z + (m + rln) * p.ldz + c);
    if (p.res) {
        if (h0) pr0 = ld8(p.res + m * p.ldr + c);
        if (h1) pr1 = ld8(p.res + (m + rln) * p.ldr + c);
    }
    float gt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) gt[k] = 1.f;
    if (p.gate && active) {
        const float* gp = p.gate + (m0 / p.HW) * p.C + c;                  // (one 64-bit division, not one per channel)
#pragma unroll
        for (int k = 0; k < 8; ++k) gt[k] = gp[k];
    }
    chunk_coefs<STATS>(p.bn, p.C, c0, nch, by == 0, L);
    float sc[8], sh[8], acc[8], dummy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = L.coef[0][cl * 8 + k]; sh[k] = L.coef[1][cl * 8 + k]; acc[k] = 0.f; dummy[k] = 0.f; }
    auto apply = [&](const bf16x8& vz, const bf16x8& vr, long row) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf2f(vz[k]) * sc[k] + sh[k];
        if (p.res) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bf2f(vr[k]);
        }
        act_fwd_n(v, p.act);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(v[k]);
        if (p.gate) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(o[k]) * gt[k]);
        }
        if (p.pool) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += bf2f(o[k]);
        }
        if (p.out) st8(p.out + row * p.ldo + c, o);
    };
    if (h0) apply(pz0, pr0, m);
    if (h1) apply(pz1, pr1, m + rln);
    if (active) {
        for (m += 2 * rln; m + rln < m1; m += 2 * rln) {
            const long mb = m + rln;
            const bf16x8 vz0 = ld8(p.z + m * p.ldz + c), vz1 = ld8(p.z + mb * p.ldz + c);
            bf16x8 vr0 = {}, vr1 = {};       // (not "= vz0": the copy is a wait for the z loads in front of the residual loads -- two trips per iteration)
            if (p.res) { vr0 = ld8(p.res + m * p.ldr + c); vr1 = ld8(p.res + mb * p.ldr + c); }
            apply(vz0, vr0, m);
            apply(vz1, vr1, mb);
        }
        if (m < m1) {
            const bf16x8 vz0 = ld8(p.z + m * p.ldz + c);
            bf16x8 vr0 = {};
            if (p.res) vr0 = ld8(p.res + m * p.ldr + c);
            apply(vz0, vr0, m);
        }
    }
    if (p.pool) chunk_row_reduce<1>(acc, dummy, active, cln, rln, cl, rl, c0, nch, p.C, p.pool, nullptr, scratch, L, by);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward.  g = upstream gradient at the BatchNorm output:
//   plain:    g = dout * act'(scale*z + shift)            (act NONE: g = dout)
//   masked:   g = dout * [y > 0]                          (y = saved block output: ReLU after the residual add)
//   SE:       g = (dout * gate[n][c] + dpool[n][c] / HW) * [scale*z + shift > 0]     (dout = gradient of the gated tensor)
// reduce: per-row-block partial sums of g and g*xhat;  apply: dz = scale * (g - mean(g) - xhat * mean(g*xhat)), optional copy of g.
// ---------------------------------------------------------------------------------------------------------------------------------
struct FBwd {
    const bf16* dout; int ldd; const bf16* z; int ldz; const bf16* y; int ldy;
    const float* coef;              // [4][C] of the forward pass
    int act;
    const float* gate; const float* dpool; long HW;   // SE variant when gate != null (RB divides HW)
    float* pg; float* pgx;          // reduce: written [gridDim.y][C]; apply: read [P][C]
    int P; double count;
    float* dgamma; float* dbeta; float* zvec;   // zvec (optional): C zeros (the gradient of a conv bias that feeds this BatchNorm)
    bf16* dz; int lddz; bf16* gout; int ldg;
    long M; int C; long RB; int cw; int xcd;
};

// VAR: 0 = plain (g = dout * act'), 1 = masked by the saved block output y, 2 = SE (gate / dpool): one instantiation per form -- the
// all-in-one kernel held every form's operands in registers (142 VGPRs: three workgroups per CU)
template <bool APPLY, int VAR>
__global__ __launch_bounds__(256) void fused_bwd_kernel(const FBwd p) {
    __shared__ FusedLds L;
    __shared__ float scratch[APPLY ? 1 : 4096];
    int bx, by;
    fused_block(p.xcd, bx, by);
    const int c0 = bx * p.cw;
    const int nch = p.C - c0 < p.cw ? p.C - c0 : p.cw;
    const int tid = threadIdx.x;
    const int cln = nch >> 3, rln = 256 / cln;
    const int cl = tid % cln, rl = tid / cln;
    const bool active = rl < rln;
    const int c = c0 + cl * 8;
    const long m0 = (long)by * p.RB;
    long m1 = m0 + p.RB;
    if (m1 > p.M) m1 = p.M;
    // Everything the workgroup reads first is issued before the first barrier (see fused_apply_kernel): the row pieces of the first
    // iteration, the forward coefficients, the SE gate / pooled gradient -- ONE trip together with the prologue's partial sums (the apply
    // pass made three: partial sums, then the coefficients, then the rows).
    long mp = m0 + rl;
    const bool h0 = active && mp < m1, h1 = active && mp + rln < m1;
    bf16x8 qd0 = {}, qd1 = {}, qz0 = {}, qz1 = {}, qy0 = {}, qy1 = {};
    if (h0) { qd0 = ld8(p.dout + mp * p.ldd + c); qz0 = ld8(p.z + mp * p.ldz + c); }
    if (h1) { qd1 = ld8(p.dout + (mp + rln) * p.ldd + c); qz1 = ld8(p.z + (mp + rln) * p.ldz + c); }
    if (VAR == 1) {
        if (h0) qy0 = ld8(p.y + mp * p.ldy + c);
        if (h1) qy1 = ld8(p.y + (mp + rln) * p.ldy + c);
    }
    const int cc = c0 + (tid < nch ? tid : 0);
    const float f0 = p.coef[cc], f1 = p.coef[p.C + cc], f2 = p.coef[2 * p.C + cc], f3 = p.coef[3 * p.C + cc];
    float gt[8], dp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gt[k] = 1.f; dp[k] = 0.f; }
    if (VAR == 2 && active) {
        const long n = m0 / p.HW;
        const float inv = 1.0f / (float)p.HW;
#pragma unroll
        for (int k = 0; k < 8; ++k) { gt[k] = p.gate[n * p.C + c + k]; dp[k] = p.dpool[n * p.C + c + k] * inv; }
    }
    if (APPLY) {
        chunk_partial_sums(p.pg, p.pgx, p.P, p.C, c0, nch, L);
        if (tid < nch) {
            const double s1 = L.r1[0][tid], s2 = L.r2[0][tid];
            L.coef[4][tid] = (float)(s1 / p.count);
            L.coef[5][tid] = (float)(s2 / p.count);
            if (by == 0) {
                p.dbeta[c0 + tid] = (float)s1; p.dgamma[c0 + tid] = (float)s2;
                if (p.zvec) p.zvec[c0 + tid] = 0.f;
            }
        }
    }
    if (tid < nch) { L.coef[0][tid] = f0; L.coef[1][tid] = f1; L.coef[2][tid] = f2; L.coef[3][tid] = f3; }
    __syncthreads();
    float sc[8], sh[8], mu[8], rs[8], mg[8], mgx[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = L.coef[0][cl * 8 + k]; sh[k] = L.coef[1][cl * 8 + k]; mu[k] = L.coef[2][cl * 8 + k]; rs[k] = L.coef[3][cl * 8 + k];
        mg[k] = APPLY ? L.coef[4][cl * 8 + k] : 0.f; mgx[k] = APPLY ? L.coef[5][cl * 8 + k] : 0.f;
        s1[k] = 0.f; s2[k] = 0.f;
    }
    auto one = [&](const bf16x8& vd, const bf16x8& vz, const bf16x8& vy, long m) {
        float z[8], g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { z[k] = bf2f(vz[k]); g[k] = bf2f(vd[k]); }
        if (VAR == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = bfround(g[k] * gt[k] + dp[k]);       // db, rounded where the unfused path stored it
        }
        if (VAR == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = bf2f(vy[k]) > 0.f ? g[k] : 0.f;
        } else {
            float pre[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pre[k] = sc[k] * z[k] + sh[k];
            if constexpr (VAR == 2) {                                  // (the SE form is ReLU only: checked by the launchers)
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = pre[k] > 0.f ? g[k] : 0.f;
            } else {
                act_bwd_n(pre, g, p.act);
            }
        }
        if (!APPLY) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1[k] += g[k]; s2[k] += g[k] * (z[k] - mu[k]) * rs[k]; }
        } else {
            bf16x8 o, og;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = (z[k] - mu[k]) * rs[k];
                o[k] = f2bf(sc[k] * (g[k] - mg[k] - xh * mgx[k]));
                og[k] = f2bf(g[k]);
            }
            st8(p.dz + m * p.lddz + c, o);
            if (p.gout) st8(p.gout + m * p.ldg + c, og);
        }
    };
    if (h0) one(qd0, qz0, qy0, mp);
    if (h1) one(qd1, qz1, qy1, mp + rln);
    if (active) {
        long m = mp + 2 * rln;
        for (; m + rln < m1; m += 2 * rln) {
            const long mb = m + rln;
            const bf16x8 vd0 = ld8(p.dout + m * p.ldd + c), vd1 = ld8(p.dout + mb * p.ldd + c);
            const bf16x8 vz0 = ld8(p.z + m * p.ldz + c), vz1 = ld8(p.z + mb * p.ldz + c);
            bf16x8 vy0 = vz0, vy1 = vz1;
            if (VAR == 1) { vy0 = ld8(p.y + m * p.ldy + c); vy1 = ld8(p.y + mb * p.ldy + c); }
            one(vd0, vz0, vy0, m);
            one(vd1, vz1, vy1, mb);
        }
        if (m < m1) {
            const bf16x8 vd0 = ld8(p.dout + m * p.ldd + c);
            const bf16x8 vz0 = ld8(p.z + m * p.ldz + c);
            bf16x8 vy0 = vz0;
            if (VAR == 1) vy0 = ld8(p.y + m * p.ldy + c);
            one(vd0, vz0, vy0, m);
        }
    }
    if (!APPLY) chunk_row_reduce<2>(s1, s2, active, cln, rln, cl, rl, c0, nch, p.C, p.pg, p.pgx, scratch, L, by);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// per-channel statistics of a bf16 tensor (producers without a statistics epilogue: stem, grouped / depthwise convs):
// psum / psq [gridDim.y][C] of the stored (bf16) values
// ---------------------------------------------------------------------------------------------------------------------------------
struct FStats { const bf16* x; int ldx; float* psum; float* psq; long M; int C; long RB; int cw; int xcd; };

__global__ __launch_bounds__(256) void fused_stats_kernel(const FStats p) {
    __shared__ FusedLds L;
    __shared__ float scratch[4096];
    int bx, by;
    fused_block(p.xcd, bx, by);
    const int c0 = bx * p.cw;
    const int nch = p.C - c0 < p.cw ? p.C - c0 : p.cw;
    const int cln = nch >> 3, rln = 256 / cln;
    const int tid = threadIdx.x, cl = tid % cln, rl = tid / cln;
    const bool active = rl < rln;
    const int c = c0 + cl * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    const long m0 = (long)by * p.RB;
    long m1 = m0 + p.RB;
    if (m1 > p.M) m1 = p.M;
    if (active) {
        long m = m0 + rl;
        for (; m + 3 * rln < m1; m += 4 * rln) {
            bf16x8 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ld8(p.x + (m + i * rln) * p.ldx + c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float f = bf2f(v[i][k]); s1[k] += f; s2[k] += f * f; }
        }
        for (; m < m1; m += rln) {
            const bf16x8 v = ld8(p.x + m * p.ldx + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float f = bf2f(v[k]); s1[k] += f; s2[k] += f * f; }
        }
    }
    chunk_row_reduce<2>(s1, s2, active, cln, rln, cl, rl, c0, nch, p.C, p.psum, p.psq, scratch, L, by);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SE backward, first pass over (dbg, z2): b = relu(bn2(z2)) (bf16-rounded as the forward pass saw it),
//   dgate partial[rb][c] = sum_rows dbg * b          (gradient of the gate, per row block inside one image)
//   bg = b * gate[n][c]   (bf16)                      (the operand of conv_block_3's weight gradient, materialised only here)
// ---------------------------------------------------------------------------------------------------------------------------------
struct FSeBwd {
    const bf16* dbg; int ldd; const bf16* z; int ldz; const float* coef; const float* gate; long HW;
    bf16* bg; int ldb; float* pdot; long M; int C; long RB; int cw; int xcd;
};

__global__ __launch_bounds__(256) void fused_se_bwd_kernel(const FSeBwd p) {
    __shared__ FusedLds L;
    __shared__ float scratch[2048];
    int bx, by;
    fused_block(p.xcd, bx, by);
    const int c0 = bx * p.cw;
    const int nch = p.C - c0 < p.cw ? p.C - c0 : p.cw;
    const int tid = threadIdx.x;
    if (tid < nch) { L.coef[0][tid] = p.coef[c0 + tid]; L.coef[1][tid] = p.coef[p.C + c0 + tid]; }
    __syncthreads();
    const int cln = nch >> 3, rln = 256 / cln;
    const int cl = tid % cln, rl = tid / cln;
    const bool active = rl < rln;
    const int c = c0 + cl * 8;
    const long m0 = (long)by * p.RB;
    long m1 = m0 + p.RB;
    if (m1 > p.M) m1 = p.M;
    float sc[8], sh[8], gt[8], acc[8], dummy[8];
    const long n = m0 / p.HW;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = L.coef[0][cl * 8 + k]; sh[k] = L.coef[1][cl * 8 + k]; acc[k] = 0.f; dummy[k] = 0.f;
        gt[k] = active ? p.gate[n * p.C + c + k] : 0.f;
    }
    auto one = [&](const bf16x8& vd, const bf16x8& vz, long m) {
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float b = bf2f(vz[k]) * sc[k] + sh[k];
            b = bfround(b > 0.f ? b : 0.f);
            acc[k] += bf2f(vd[k]) * b;
            o[k] = f2bf(b * gt[k]);
        }
        if (p.bg) st8(p.bg + m * p.ldb + c, o);
    };
    if (active) {
        long m = m0 + rl;
        for (; m + rln < m1; m += 2 * rln) {
            const long mb = m + rln;
            const bf16x8 vd0 = ld8(p.dbg + m * p.ldd + c), vd1 = ld8(p.dbg + mb * p.ldd + c);
            const bf16x8 vz0 = ld8(p.z + m * p.ldz + c), vz1 = ld8(p.z + mb * p.ldz + c);
            one(vd0, vz0, m);
            one(vd1, vz1, mb);
        }
        if (m < m1) one(ld8(p.dbg + m * p.ldd + c), ld8(p.z + m * p.ldz + c), m);
    }
    chunk_row_reduce<1>(acc, dummy, active, cln, rln, cl, rl, c0, nch, p.C, p.pdot, nullptr, scratch, L, by);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SE excite + gated apply in ONE launch (net/anynet.py:44-48,68-69): gate[n][c] = sigmoid(b2[c] + W2[c][:] . hid[n][:]) for the workgroup's
// channels, then out = bf16(act(sc z + sh)) * gate over its rows.  A workgroup owns cw <= 64 channels (so its weight slice is <= 30 KB: one
// round of loads) and RB rows of one image; the second excitation layer as a launch of its own cost 5.7 us + the gap to the apply pass
// (r05 kernel stats), here it is ~1.5 us of prologue that the first row loads already overlap.  The summation order per channel is
// se_fc_rows_kernel's (lanes stride the contraction, wave_sum), so the gate is the value that kernel gives.
// ---------------------------------------------------------------------------------------------------------------------------------
struct FGateApply {
    const bf16* z; int ldz;
    const float* coef;              // [4][C] scale, shift (null = identity)
    int act;
    const float* hid; const float* w2; const float* b2; int Cs;
    float* gate;                    // [N][C], written by the first row block of every image (kept for the backward pass); may be null
    bf16* out; int ldo;
    long HW; long M; int C; long RB; int cw; int xcd;
};

template <int GA_JMAX, int GA_KS>   // channels per wave (cw / 4); 64-lane pieces of the contraction whose loads are issued together
__global__ __launch_bounds__(256) void se_gate_apply_kernel(const FGateApply p) {
    __shared__ float sgate[4 * GA_JMAX];
    int bx, by;
    fused_block(p.xcd, bx, by);
    const int c0 = bx * p.cw;
    const int nch = p.C - c0 < p.cw ? p.C - c0 : p.cw;
    const long m0 = (long)by * p.RB;
    long m1 = m0 + p.RB;
    if (m1 > p.M) m1 = p.M;
    const long n = m0 / p.HW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cln = nch >> 3, rln = 256 / cln;
    const int cl = tid % cln, rl = tid / cln;
    const bool active = rl < rln;
    const int c = c0 + cl * 8;
    // the first two row pieces and the BatchNorm coefficients are in flight while the gate is computed
    long m = m0 + rl;
    const bool h0 = active && m < m1, h1 = active && m + rln < m1;
    bf16x8 vz0 = {}, vz1 = {};
    if (h0) vz0 = ld8(p.z + m * p.ldz + c);
    if (h1) vz1 = ld8(p.z + (m + rln) * p.ldz + c);
    // (no select on the loaded values here: a v_cndmask behind a load is a wait for ALL loads in flight, in front of the weight loads)
    const bool hc = p.coef && active;
    const float* cp = hc ? p.coef + c : p.b2;                          // any valid address when there are no coefficients
    const int cstep = hc ? p.C : 0;
    float rsc[8], rsh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { rsc[k] = cp[hc ? k : 0]; rsh[k] = cp[cstep + (hc ? k : 0)]; }
    // wave wv: channels wv, wv + 4, ... of the chunk; all weight loads of a 256-wide slab of the contraction issued together
    float acc[GA_JMAX], bias[GA_JMAX];
#pragma unroll
    for (int jj = 0; jj < GA_JMAX; ++jj) {
        acc[jj] = 0.f;
        bias[jj] = p.b2[c0 + (wv + 4 * jj < nch ? wv + 4 * jj : 0)];   // with the weights: behind the sums each would be a trip of its own
    }
    const float* hr = p.hid + n * p.Cs;
    for (int base = 0; base < p.Cs; base += 64 * GA_KS) {
        float hv[GA_KS];
#pragma unroll
        for (int k = 0; k < GA_KS; ++k) hv[k] = base + lane + 64 * k < p.Cs ? hr[base + lane + 64 * k] : 0.f;
        float wvv[GA_JMAX][GA_KS];
#pragma unroll
        for (int jj = 0; jj < GA_JMAX; ++jj) {
            const int j = wv + 4 * jj;
            const float* wr = p.w2 + (long)(c0 + (j < nch ? j : 0)) * p.Cs + base + lane;
#pragma unroll
            for (int k = 0; k < GA_KS; ++k) wvv[jj][k] = (j < nch && base + lane + 64 * k < p.Cs) ? wr[64 * k] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < GA_JMAX; ++jj)
#pragma unroll
            for (int k = 0; k < GA_KS; ++k) acc[jj] += wvv[jj][k] * hv[k];
    }
#pragma unroll
    for (int jj = 0; jj < GA_JMAX; ++jj) {
        const int j = wv + 4 * jj;
        if (j < nch) {                                                  // wave-uniform
            const float s = wave_sum(acc[jj]) + bias[jj];
            const float g = 1.f / (1.f + __expf(-s));
            if (lane == 0) {
                sgate[j] = g;
                if (p.gate && m0 == n * p.HW) p.gate[n * p.C + c0 + j] = g;
            }
        }
    }
    __syncthreads();
    float gt[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        gt[k] = active ? sgate[cl * 8 + k] : 1.f;
        sc[k] = hc ? rsc[k] : 1.f;
        sh[k] = hc ? rsh[k] : 0.f;
    }
    auto apply = [&](const bf16x8& vz, long row) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf2f(vz[k]) * sc[k] + sh[k];
        act_fwd_n(v, p.act);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(f2bf(v[k])) * gt[k]);
        st8(p.out + row * p.ldo + c, o);
    };
    if (h0) apply(vz0, m);
    if (h1) apply(vz1, m + rln);
    if (active) {
        for (m += 2 * rln; m + rln < m1; m += 2 * rln) {
            const bf16x8 a = ld8(p.z + m * p.ldz + c), b = ld8(p.z + (m + rln) * p.ldz + c);
            apply(a, m);
            apply(b, m + rln);
        }
        if (m < m1) apply(ld8(p.z + m * p.ldz + c), m);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
// chunk width: C <= 128 is one chunk (whole rows contiguous); wider tensors are cut into equal chunks of <= 128 channels
static int chunk_width(int C) {
    const int n = (C + FCH - 1) / FCH;
    return (((C >> 3) + n - 1) / n) << 3;
}
static int chunk_count(int C) { const int cw = chunk_width(C); return (C + cw - 1) / cw; }

// Rows per row block (policy measured with tools/bench_fused.py on MI355X).
//   kind 0 (apply passes, forward and backward): ~1000 workgroups over (chunks x row blocks), RB a power of two in [32, 1024]; larger when
//           the prologue has many partial rows to reduce (P > 128);
//   kind 1 (reduce passes: their row-block count is the P of the apply that follows): RB = clamp(M / 128, 32, 128), then max(.., M / 512).
// With align (rows per image) RB divides align and is >= align / 8 (<= 8 SE partial rows per image).
extern "C" long hn_fused_row_block(long M, int C, long align, int P, int kind) {
    long RB;
    if (kind == 1) {
        // 32 ... 128 rows for the few-row tensors of the deep stages (2048 rows: 64 blocks per chunk instead of 16 -- one load round per
        // thread, still <= 128 partial rows for the apply prologue; 709 -> 714 img/s), 128 and up for the large ones
        RB = 32;
        while (RB < 128 && RB * 128 < M) RB <<= 1;
        while (RB * g_hn_knob[3] < M) RB <<= 1;
    } else {
        const long want = (M * chunk_count(C) + g_hn_knob[2] - 1) / g_hn_knob[2];
        RB = 32;
        while (RB < want && RB < 1024) RB <<= 1;
        if (P > 128 && RB < 128) RB = 128;
    }
    if (RB > M) RB = M;
    if (align > 0) {
        if (RB > align) RB = align;
        if (RB < (align + 7) / 8) RB = (align + 7) / 8;
        while (align % RB) ++RB;                        // smallest divisor of align that is >= RB (terminates at RB == align)
    }
    return RB;
}

static dim3 fused_grid(long M, int C, long RB) { return dim3((unsigned)chunk_count(C), (unsigned)((M + RB - 1) / RB)); }

extern "C" int hn_bn_apply_fused(const void* z, int ldz, long M, int C, const float* psum, const float* psq, int P, long count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* rm, float* rv, float* coef, const void* res, int ldr,
                                 int act, void* out, int ldo, float* pool, const float* gate, long HW, long RB, hipStream_t st) {
    HN_CHECK_ARG(z && M > 0 && C > 0 && (C & 7) == 0 && (ldz & 7) == 0 && RB > 0 && (out || pool));
    HN_CHECK_ARG(!gate || (HW > 0 && HW % RB == 0));
    HN_CHECK_ARG(P <= 0 || (psum && psq && gamma && beta && count > 0));
    HN_CHECK_ARG(P >= 0 || (gamma && beta && rm && rv));
    HN_CHECK_ARG((!res || (ldr & 7) == 0) && (!out || (ldo & 7) == 0));
    FApply p;
    p.z = (const bf16*)z; p.ldz = ldz;
    p.bn.psum = psum; p.bn.psq = psq; p.bn.P = P; p.bn.count = (double)count; p.bn.gamma = gamma; p.bn.beta = beta; p.bn.eps = eps;
    p.bn.momentum = momentum; p.bn.rm = rm; p.bn.rv = rv; p.bn.coef = coef;
    p.res = (const bf16*)res; p.ldr = ldr; p.act = act; p.out = (bf16*)out; p.ldo = ldo; p.pool = pool; p.M = M; p.C = C; p.RB = RB;
    p.gate = gate; p.HW = HW; p.cw = chunk_width(C); p.xcd = (int)g_hn_knob[8];
    if (P > 0) hipLaunchKernelGGL(fused_apply_kernel<true>, fused_grid(M, C, RB), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(fused_apply_kernel<false>, fused_grid(M, C, RB), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

// rows per workgroup of hn_se_gate_apply: every workgroup pays the gate prologue (~3 us of dependent loads), so the pass wants ONE round of
// workgroups -- ~512 over (chunks x row blocks), measured with tools/bench_gate_apply.py: 128 rows at stage 4 ... 1024 at stage 0 -- and a
// divisor of HW (a workgroup's rows lie inside one image)
static long gate_apply_rows(long M, long HW, int chunks) {
    long RB = 64;
    while (RB * 512 < M * chunks) RB <<= 1;
    if (g_hn_knob[17]) RB = g_hn_knob[17];
    if (RB > HW) RB = HW;
    while (HW % RB) ++RB;
    return RB;
}

extern "C" int hn_se_gate_apply(const void* z, int ldz, const float* coef, int act, const float* hid, const float* w2, const float* b2, float* gate,
                                void* out, int ldo, int N, long HW, int C, int Cs, hipStream_t st) {
    HN_CHECK_ARG(z && hid && w2 && b2 && out && N > 0 && HW > 0 && C > 0 && Cs > 0 && (C & 7) == 0 && (ldz & 7) == 0 && (ldo & 7) == 0);
    FGateApply p;
    p.z = (const bf16*)z; p.ldz = ldz; p.coef = coef; p.act = act; p.hid = hid; p.w2 = w2; p.b2 = b2; p.Cs = Cs; p.gate = gate;
    p.out = (bf16*)out; p.ldo = ldo; p.HW = HW; p.M = (long)N * HW; p.C = C;
    // channels per workgroup: the weight slice cw * Cs * 4 B stays one round of <= 40 loads per lane (knob 16 overrides: tools/bench_gate_apply.py)
    p.cw = g_hn_knob[16] ? (int)g_hn_knob[16] : 32;
    const int chunks = (C + p.cw - 1) / p.cw;
    p.RB = gate_apply_rows(p.M, HW, chunks);
    p.xcd = (int)g_hn_knob[8];
    const dim3 grid((unsigned)chunks, (unsigned)(p.M / p.RB));
    if (p.cw == 32) hipLaunchKernelGGL((se_gate_apply_kernel<8, 4>), grid, dim3(256), 0, st, p);
    else if (p.cw == 64) hipLaunchKernelGGL((se_gate_apply_kernel<16, 2>), grid, dim3(256), 0, st, p);
    else if (p.cw == 160) hipLaunchKernelGGL((se_gate_apply_kernel<40, 1>), grid, dim3(256), 0, st, p);
    else return HN_ERR_ARG;
    HN_LAUNCH_CHECK();
}

static int fill_bwd(FBwd& p, const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act,
                    const float* gate, const float* dpool, long HW, float* pg, float* pgx, long M, int C, long RB) {
    HN_CHECK_ARG(dout && z && coef && pg && pgx && M > 0 && C > 0 && (C & 7) == 0 && (ldd & 7) == 0 && (ldz & 7) == 0 && RB > 0);
    HN_CHECK_ARG((!y || (ldy & 7) == 0) && (!gate || (dpool && HW > 0 && HW % RB == 0 && act == HN_ACT_RELU && !y)));
    p = FBwd{};
    p.dout = (const bf16*)dout; p.ldd = ldd; p.z = (const bf16*)z; p.ldz = ldz; p.y = (const bf16*)y; p.ldy = ldy; p.coef = coef; p.act = act;
    p.gate = gate; p.dpool = dpool; p.HW = HW; p.pg = pg; p.pgx = pgx; p.M = M; p.C = C; p.RB = RB; p.cw = chunk_width(C); p.xcd = (int)g_hn_knob[8];
    return HN_OK;
}

extern "C" int hn_bn_bwd_reduce_fused(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act,
                                      const float* gate, const float* dpool, long HW, long M, int C, long RB, float* pg, float* pgx,
                                      hipStream_t st) {
    FBwd p;
    const int rc = fill_bwd(p, dout, ldd, z, ldz, y, ldy, coef, act, gate, dpool, HW, pg, pgx, M, C, RB);
    if (rc) return rc;
    if (gate) hipLaunchKernelGGL((fused_bwd_kernel<false, 2>), fused_grid(M, C, RB), dim3(256), 0, st, p);
    else if (y) hipLaunchKernelGGL((fused_bwd_kernel<false, 1>), fused_grid(M, C, RB), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fused_bwd_kernel<false, 0>), fused_grid(M, C, RB), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_bn_bwd_apply_fused(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act,
                                     const float* gate, const float* dpool, long HW, const float* pg, const float* pgx, int P, long count,
                                     float* dgamma, float* dbeta, void* dz, int lddz, void* gout, int ldg, long M, int C, long RB,
                                     float* zero_c, hipStream_t st) {
    FBwd p;
    const int rc = fill_bwd(p, dout, ldd, z, ldz, y, ldy, coef, act, gate, dpool, HW, (float*)pg, (float*)pgx, M, C, RB);
    if (rc) return rc;
    HN_CHECK_ARG(P > 0 && count > 0 && dgamma && dbeta && dz && (lddz & 7) == 0 && (!gout || (ldg & 7) == 0));
    p.P = P; p.count = (double)count; p.dgamma = dgamma; p.dbeta = dbeta; p.zvec = zero_c; p.dz = (bf16*)dz; p.lddz = lddz; p.gout = (bf16*)gout; p.ldg = ldg;
    if (gate) hipLaunchKernelGGL((fused_bwd_kernel<true, 2>), fused_grid(M, C, RB), dim3(256), 0, st, p);
    else if (y) hipLaunchKernelGGL((fused_bwd_kernel<true, 1>), fused_grid(M, C, RB), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fused_bwd_kernel<true, 0>), fused_grid(M, C, RB), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_col_stats_fused(const void* x, int ldx, long M, int C, long RB, float* psum, float* psq, hipStream_t st) {
    HN_CHECK_ARG(x && psum && psq && M > 0 && C > 0 && (C & 7) == 0 && (ldx & 7) == 0 && RB > 0);
    FStats p = {(const bf16*)x, ldx, psum, psq, M, C, RB, chunk_width(C), (int)g_hn_knob[8]};
    hipLaunchKernelGGL(fused_stats_kernel, fused_grid(M, C, RB), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_se_bwd_reduce_fused(const void* dbg, int ldd, const void* z, int ldz, const float* coef, const float* gate, long HW, void* bg,
                                      int ldb, float* pdot, long M, int C, long RB, hipStream_t st) {
    HN_CHECK_ARG(dbg && z && coef && gate && pdot && M > 0 && C > 0 && (C & 7) == 0 && (ldd & 7) == 0 && (ldz & 7) == 0);
    HN_CHECK_ARG(RB > 0 && HW > 0 && HW % RB == 0 && (!bg || (ldb & 7) == 0));
    FSeBwd p = {(const bf16*)dbg, ldd, (const bf16*)z, ldz, coef, gate, HW, (bf16*)bg, ldb, pdot, M, C, RB, chunk_width(C), (int)g_hn_knob[8]};
    hipLaunchKernelGGL(fused_se_bwd_kernel, fused_grid(M, C, RB), dim3(256), 0, st, p);
    HN_LAUNCH_CHECK();
}
