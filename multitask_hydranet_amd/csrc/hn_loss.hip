// Loss kernels of HydraNet.cal_loss (model/model.py:201-264) on device, no host synchronisation.
//   * segmentation: weighted cross entropy with ignore_index and "top-k hardest pixels" (head_seg/segmentation_loss.py:48-65).  The
//     reference sorts the 524,288 per-pixel losses of every image and averages the first k; the mean only needs the k-th largest value,
//     so a 3-level (11+11+10 bit) radix select on the fp32 bit pattern replaces the sort: exact, deterministic (integer atomics only).
//   * deploy-mode argmax over the class logits (model/model.py:197) -> int64 mask.
#include "hn_common.h"

#define HN_SEG_MAXC 16

// per pixel: loss = w[y] * (logsumexp(l) - l[y]) (0 for ignored pixels); also the level-0 histogram (top 11 bits) per image
__global__ __launch_bounds__(256) void seg_ce_fwd_kernel(const float* logits, int ldl, int C, const void* target, int target_is_float,
                                                         const float* cw, int ignore_index, long HW, long M, float* loss,
                                                         unsigned int* hist /* [N][2048] or null */) {
    __shared__ unsigned int sh[2048];
    const bool do_hist = hist != nullptr;
    if (do_hist) {
        for (int i = threadIdx.x; i < 2048; i += 256) sh[i] = 0;
        __syncthreads();
    }
    // one block handles a contiguous pixel range inside ONE image (grid = N * blocks_per_image)
    const int bpi = gridDim.x / (int)(M / HW);
    const int n = blockIdx.x / bpi, b = blockIdx.x - n * bpi;
    const long per = (HW + bpi - 1) / bpi;
    const long p0 = (long)n * HW + b * per;
    long p1 = p0 + per;
    if (p1 > (long)(n + 1) * HW) p1 = (long)(n + 1) * HW;
    for (long m = p0 + threadIdx.x; m < p1; m += 256) {
        const int y = target_is_float ? (int)reinterpret_cast<const float*>(target)[m] : (int)reinterpret_cast<const long*>(target)[m];
        float l = 0.f;
        if (y != ignore_index && y >= 0 && y < C) {
            const float* row = logits + m * ldl;
            float mx = row[0];
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += __expf(row[c] - mx);
            l = cw[y] * (mx + __logf(se) - row[y]);
            if (l < 0.f) l = 0.f;                     // guards the unsigned-bit-pattern ordering (rounding can give -0)
        }
        loss[m] = l;
        if (do_hist) atomicAdd(&sh[__float_as_uint(l) >> 21], 1u);
    }
    if (do_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2048; i += 256)
            if (sh[i]) atomicAdd(&hist[(long)n * 2048 + i], sh[i]);
    }
}

// compile-time class count, four consecutive pixels per thread: 4*CT contiguous floats = CT float4 loads, float4 targets, float4 loss store
template <int CT>
__global__ __launch_bounds__(256) void seg_ce_fwd_vec_kernel(const float* logits, const float* target, const float* cw, int ignore_index,
                                                             long HW, long M, float* loss, unsigned int* hist) {
    __shared__ unsigned int sh[2048];
    const bool do_hist = hist != nullptr;
    if (do_hist) {
        for (int i = threadIdx.x; i < 2048; i += 256) sh[i] = 0;
        __syncthreads();
    }
    const int bpi = gridDim.x / (int)(M / HW);
    const int n = blockIdx.x / bpi, b = blockIdx.x - n * bpi;
    const long per = ((HW / 4 + bpi - 1) / bpi) * 4;                  // pixels per block, a multiple of 4 (HW % 4 == 0)
    const long p0 = (long)n * HW + b * per;
    long p1 = p0 + per;
    if (p1 > (long)(n + 1) * HW) p1 = (long)(n + 1) * HW;
    for (long m = p0 + 4 * threadIdx.x; m < p1; m += 1024) {
        float lg[4 * CT];
        const f32x4* row = reinterpret_cast<const f32x4*>(logits + m * CT);
#pragma unroll
        for (int k = 0; k < CT; ++k) { const f32x4 v = row[k]; lg[4 * k] = v[0]; lg[4 * k + 1] = v[1]; lg[4 * k + 2] = v[2]; lg[4 * k + 3] = v[3]; }
        const f32x4 tg = *reinterpret_cast<const f32x4*>(target + m);
        f32x4 lo;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = (int)tg[u];
            float l = 0.f;
            if (y != ignore_index && y >= 0 && y < CT) {
                const float* r = &lg[u * CT];
                float mx = r[0];
#pragma unroll
                for (int c = 1; c < CT; ++c) mx = fmaxf(mx, r[c]);
                float se = 0.f, ry = 0.f;
#pragma unroll
                for (int c = 0; c < CT; ++c) { se += __expf(r[c] - mx); ry = c == y ? r[c] : ry; }
                l = cw[y] * (mx + __logf(se) - ry);
                if (l < 0.f) l = 0.f;
            }
            lo[u] = l;
            if (do_hist) atomicAdd(&sh[__float_as_uint(l) >> 21], 1u);
        }
        *reinterpret_cast<f32x4*>(loss + m) = lo;
    }
    if (do_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2048; i += 256)
            if (sh[i]) atomicAdd(&hist[(long)n * 2048 + i], sh[i]);
    }
}

// radix-select state per image: prefix (bits fixed so far), remaining rank r (how many of the still-ambiguous bin are needed)
struct SelState { unsigned int prefix; unsigned int remaining; };

// level L histogram of the elements whose higher bits equal the prefix.  level 1: bits 20..10 (11 bits), level 2: bits 9..0 (10 bits)
__global__ __launch_bounds__(256) void seg_hist_kernel(const float* loss, long HW, long M, const SelState* st, int level, unsigned int* hist) {
    __shared__ unsigned int sh[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sh[i] = 0;
    __syncthreads();
    const int bpi = gridDim.x / (int)(M / HW);
    const int n = blockIdx.x / bpi, b = blockIdx.x - n * bpi;
    const long per = (HW + bpi - 1) / bpi;
    const long p0 = (long)n * HW + b * per;
    long p1 = p0 + per;
    if (p1 > (long)(n + 1) * HW) p1 = (long)(n + 1) * HW;
    const unsigned int prefix = st[n].prefix;
    for (long m = p0 + threadIdx.x; m < p1; m += 256) {
        const unsigned int u = __float_as_uint(loss[m]);
        if (level == 1) { if ((u >> 21) == (prefix >> 21)) atomicAdd(&sh[(u >> 10) & 2047], 1u); }
        else if ((u >> 10) == (prefix >> 10)) atomicAdd(&sh[u & 1023], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (sh[i]) atomicAdd(&hist[(long)n * 2048 + i], sh[i]);
}

// one wave per image: find the bin (from the top) in which the remaining rank falls and fix that bin's bits.  The 2048 bins are staged in
// LDS, every lane sums its contiguous chunk, a wave suffix-scan locates the chunk and its lane walks at most 32 bins (a single thread
// walking 2048 dependent global loads took 60 us).  Semantics of the serial walk: b runs from the top bin down to 1 and stops at the first
// bin with h[b] >= need, otherwise need -= h[b]; b = 0 if no bin stops it.
__global__ __launch_bounds__(64) void seg_select_kernel(unsigned int* hist, SelState* st, int level, unsigned int k) {
    __shared__ unsigned int sh[2048];
    __shared__ int s_b;
    __shared__ unsigned int s_need;
    const int n = blockIdx.x, lane = threadIdx.x;
    unsigned int* h = hist + (long)n * 2048;
    const int bins = level == 2 ? 1024 : 2048;
    const int per = bins / 64;
    for (int i = lane; i < bins; i += 64) sh[i] = h[i];
    const unsigned int need0 = level == 0 ? k : st[n].remaining;
    if (lane == 0) { s_b = 0; s_need = 0; }
    __syncthreads();
    unsigned int local = 0;
    for (int j = 0; j < per; ++j) {
        const int b = lane * per + j;
        if (b >= 1) local += sh[b];                                   // bin 0 never stops the walk, it only receives what is left
    }
    // above = sum of the chunks of all higher lanes (inclusive suffix scan minus own)
    unsigned int suf = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_down(suf, o);
        if (lane + o < 64) suf += v;
    }
    const unsigned int above = suf - local;
    // the walk stops inside this lane's chunk iff above < need0 <= above + local  (cumulative count from the top reaches need0 here)
    const bool mine = above < need0 && need0 <= above + local;
    if (mine) {
        unsigned int need = need0 - above;
        int b = lane * per + per - 1;
        for (; b > lane * per && b > 0; --b) {
            if (sh[b] >= need) break;
            need -= sh[b];
        }
        // b is now either the stopping bin or the lowest bin of the chunk (which must stop the walk because the chunk total reaches need)
        s_b = b;
        s_need = need;
    }
    const unsigned long long any = __ballot(mine);
    if (!any && lane == 0) {                                          // fewer than need0 elements above bin 0: the walk ends at b = 0
        s_b = 0;
        s_need = need0 - suf;                                         // lane 0's inclusive suffix = everything in bins >= 1
    }
    __syncthreads();
    if (lane == 0) {
        const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
        const unsigned int prefix = (level == 0 ? 0u : st[n].prefix) | ((unsigned int)s_b << shift);
        st[n].prefix = prefix;
        st[n].remaining = s_need;                      // elements of bin b (at this resolution) that still belong to the top-k
    }
    for (int i = lane; i < 2048; i += 64) h[i] = 0;                   // ready for the next level
}

// per image: sum of the losses strictly above the threshold (partial sums per block); the `remaining` ties at the threshold are added
// in the finalize kernel
__global__ __launch_bounds__(256) void seg_topk_sum_kernel(const float* loss, long HW, long M, const SelState* st, int use_topk, float* psum) {
    __shared__ float red[4];
    const int bpi = gridDim.x / (int)(M / HW);
    const int n = blockIdx.x / bpi, b = blockIdx.x - n * bpi;
    const long per = (HW + bpi - 1) / bpi;
    const long p0 = (long)n * HW + b * per;
    long p1 = p0 + per;
    if (p1 > (long)(n + 1) * HW) p1 = (long)(n + 1) * HW;
    const unsigned int thr = use_topk ? st[n].prefix : 0u;
    float s = 0.f;
    for (long m = p0 + threadIdx.x; m < p1; m += 256) {
        const float v = loss[m];
        if (!use_topk || __float_as_uint(v) > thr) s += v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) psum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out = ( sum_blocks psum + sum_n remaining_n * thr_n ) / denom
__global__ __launch_bounds__(256) void seg_loss_finalize_kernel(const float* psum, int nblocks, const SelState* st, int N, int use_topk,
                                                                double denom, float* out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += psum[i];
    if (use_topk)
        for (int n = threadIdx.x; n < N; n += 256) s += (double)st[n].remaining * (double)__uint_as_float(st[n].prefix);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / denom);
}

// backward: dlogits[m][c] = gout/denom * sel(m) * w[y] * (softmax_c - [c == y]);  sel = 1 above the threshold, remaining/ties at it, 0 below
__global__ __launch_bounds__(256) void seg_ce_bwd_kernel(const float* logits, int ldl, int C, const void* target, int target_is_float,
                                                         const float* cw, int ignore_index, long HW, long M, const float* loss,
                                                         const SelState* st, const unsigned int* ties, int use_topk, const float* gout,
                                                         float inv_denom, float* dlogits, int ldd) {
    const float gs = gout[0] * inv_denom;
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const int n = (int)(m / HW);
        const int y = target_is_float ? (int)reinterpret_cast<const float*>(target)[m] : (int)reinterpret_cast<const long*>(target)[m];
        float sel = 0.f;
        if (y != ignore_index && y >= 0 && y < C) {
            sel = 1.f;
            if (use_topk) {
                const unsigned int u = __float_as_uint(loss[m]), thr = st[n].prefix;
                if (u < thr) sel = 0.f;
                else if (u == thr) sel = (float)st[n].remaining / (float)(ties[n] > 0 ? ties[n] : 1u);
            }
        }
        float* d = dlogits + m * ldd;
        if (sel == 0.f) {
            for (int c = 0; c < C; ++c) d[c] = 0.f;
            continue;
        }
        const float* row = logits + m * ldl;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float e[HN_SEG_MAXC], se = 0.f;
        for (int c = 0; c < C; ++c) { e[c] = __expf(row[c] - mx); se += e[c]; }
        const float k = gs * sel * cw[y] / se;
        for (int c = 0; c < C; ++c) d[c] = k * e[c] - (c == y ? gs * sel * cw[y] : 0.f);
    }
}

// the same gradient written straight into the operand the phase-form output conv's backward consumes (ops.SegOutUp): bf16
// [N][H/2][W/2][ldz], channel (py*2+px)*C + c of low-res pixel (y, x) = dlogits(2y+py, 2x+px, c), zeros in [4C, ldz).  One thread per
// low-res pixel writes one contiguous row; the fp32 dlogits tensor (168 MB at 16 x 512 x 1024 x 5) and the separate space-to-depth pass
// over it are never materialised.
__global__ __launch_bounds__(256) void seg_ce_bwd_s2d_kernel(const float* logits, int ldl, int C, const void* target, int target_is_float,
                                                             const float* cw, int ignore_index, int H, int W, long M4, const float* loss,
                                                             const SelState* st, const unsigned int* ties, int use_topk, const float* gout,
                                                             float inv_denom, bf16* dz, int ldz) {
    const float gs = gout[0] * inv_denom;
    const int h = H >> 1, w = W >> 1;
    const long HW = (long)H * W;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < M4; q += (long)gridDim.x * 256) {
        const int x = (int)(q % w);
        const long t = q / w;
        const int y = (int)(t % h);
        const int n = (int)(t / h);
        bf16* d = dz + q * ldz;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const long m = (long)n * HW + (long)(2 * y + (ph >> 1)) * W + 2 * x + (ph & 1);
            const int yy = target_is_float ? (int)reinterpret_cast<const float*>(target)[m] : (int)reinterpret_cast<const long*>(target)[m];
            float sel = 0.f;
            if (yy != ignore_index && yy >= 0 && yy < C) {
                sel = 1.f;
                if (use_topk) {
                    const unsigned int u = __float_as_uint(loss[m]), thr = st[n].prefix;
                    if (u < thr) sel = 0.f;
                    else if (u == thr) sel = (float)st[n].remaining / (float)(ties[n] > 0 ? ties[n] : 1u);
                }
            }
            if (sel == 0.f) {
                for (int c = 0; c < C; ++c) d[ph * C + c] = f2bf(0.f);
                continue;
            }
            const float* row = logits + m * ldl;
            float mx = row[0];
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
            float e[HN_SEG_MAXC], se = 0.f;
            for (int c = 0; c < C; ++c) { e[c] = __expf(row[c] - mx); se += e[c]; }
            const float k = gs * sel * cw[yy] / se;
            for (int c = 0; c < C; ++c) d[ph * C + c] = f2bf(k * e[c] - (c == yy ? gs * sel * cw[yy] : 0.f));
        }
        for (int c = 4 * C; c < ldz; ++c) d[c] = f2bf(0.f);
    }
}

// The same for a compile-time class count with vector memory operations: per image row of the quad the two pixels' logits are 2*CT
// contiguous floats (8-byte aligned: float2 loads), targets / losses float2, and the 4*CT (+ padding) bf16 results leave as 16-byte stores
// (the generic kernel issues 4*CT two-byte stores and 4*CT + 8 scalar loads per thread: 265 us for 2 M pixels).
template <int CT, int LDZ>
__global__ __launch_bounds__(256) void seg_ce_bwd_s2d_vec_kernel(const float* logits, const float* target, const float* cw, int ignore_index,
                                                                 int H, int W, long M4, const float* loss, const SelState* st,
                                                                 const unsigned int* ties, int use_topk, const float* gout, float inv_denom,
                                                                 bf16* dz) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float gs = gout[0] * inv_denom;
    const int h = H >> 1, w = W >> 1;
    const long HW = (long)H * W;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < M4; q += (long)gridDim.x * 256) {
        const int x = (int)(q % w);
        const long t = q / w;
        const int y = (int)(t % h);
        const int n = (int)(t / h);
        const unsigned int thr = use_topk ? st[n].prefix : 0u;
        const float share = use_topk ? (float)st[n].remaining / (float)(ties[n] > 0 ? ties[n] : 1u) : 1.f;
        float lg[2][2 * CT];
        f32x2 tg[2], ls[2];
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const long m = (long)n * HW + (long)(2 * y + py) * W + 2 * x;
            const f32x2* row = reinterpret_cast<const f32x2*>(logits + m * CT);
#pragma unroll
            for (int k = 0; k < CT; ++k) { const f32x2 v = row[k]; lg[py][2 * k] = v[0]; lg[py][2 * k + 1] = v[1]; }
            tg[py] = *reinterpret_cast<const f32x2*>(target + m);
            ls[py] = use_topk ? *reinterpret_cast<const f32x2*>(loss + m) : (f32x2){0.f, 0.f};
        }
        bf16 out[LDZ];
#pragma unroll
        for (int c = 0; c < LDZ; ++c) out[c] = f2bf(0.f);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            const int yy = (int)tg[py][px];
            float sel = 0.f;
            if (yy != ignore_index && yy >= 0 && yy < CT) {
                sel = 1.f;
                if (use_topk) {
                    const unsigned int u = __float_as_uint(ls[py][px]);
                    if (u < thr) sel = 0.f;
                    else if (u == thr) sel = share;
                }
            }
            if (sel != 0.f) {
                const float* r = &lg[py][px * CT];
                float mx = r[0];
#pragma unroll
                for (int c = 1; c < CT; ++c) mx = fmaxf(mx, r[c]);
                float e[CT], se = 0.f;
#pragma unroll
                for (int c = 0; c < CT; ++c) { e[c] = __expf(r[c] - mx); se += e[c]; }
                const float wy = cw[yy];
                const float k = gs * sel * wy / se;
#pragma unroll
                for (int c = 0; c < CT; ++c) out[ph * CT + c] = f2bf(k * e[c] - (c == yy ? gs * sel * wy : 0.f));
            }
        }
        bf16x8* d = reinterpret_cast<bf16x8*>(dz + q * LDZ);
#pragma unroll
        for (int k = 0; k < LDZ / 8; ++k) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = out[k * 8 + j];
            d[k] = v;
        }
    }
}

// number of elements exactly at the threshold, per image (for the tie share in backward): it is hist level-2 bin = prefix&1023, saved
// by the select kernel before clearing; simpler: count again
__global__ __launch_bounds__(256) void seg_count_ties_kernel(const float* loss, long HW, long M, const SelState* st, unsigned int* ties) {
    const int bpi = gridDim.x / (int)(M / HW);
    const int n = blockIdx.x / bpi, b = blockIdx.x - n * bpi;
    const long per = (HW + bpi - 1) / bpi;
    const long p0 = (long)n * HW + b * per;
    long p1 = p0 + per;
    if (p1 > (long)(n + 1) * HW) p1 = (long)(n + 1) * HW;
    const unsigned int thr = st[n].prefix;
    unsigned int c = 0;
    for (long m = p0 + threadIdx.x; m < p1; m += 256) c += __float_as_uint(loss[m]) == thr ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&ties[n], c);
}

// ---------------------------------------------------------------------------------------------------------
// Detection loss (FocalLoss.forward, head_detect/detection_loss.py:132-267): IoU assignment (neg < 0.4, pos >= 0.5, else ignored),
// focal BCE (alpha 0.25, gamma 2) on probabilities clamped to [1e-4, 1-1e-4] divided by max(#pos, 1), smooth-L1 (beta 1/9) on
// (dy, dx, dh, dw) averaged over positives x 4; batch mean.  One thread per (image, anchor); the per-image Python loop and the
// A x M IoU matrix of the reference never exist.  assign[n][a]: >= 0 matched annotation, -1 background, -2 ignored.
// ---------------------------------------------------------------------------------------------------------
#define HN_DET_MAXK 32
__device__ __forceinline__ int det_assign(const float* anc, const float* ann, int Mx, float& best_iou) {
    const float ay1 = anc[0], ax1 = anc[1], ay2 = anc[2], ax2 = anc[3];
    const float aarea = (ay2 - ay1) * (ax2 - ax1);
    float best = -1.f;
    int arg = -1;
    bool any = false;
    for (int j = 0; j < Mx; ++j) {
        const float* b = ann + j * 5;
        if (b[4] == -1.f) continue;
        any = true;
        const float area = (b[2] - b[0]) * (b[3] - b[1]);
        float iw = fminf(ax2, b[2]) - fmaxf(ax1, b[0]);
        float ih = fminf(ay2, b[3]) - fmaxf(ay1, b[1]);
        iw = fmaxf(iw, 0.f);
        ih = fmaxf(ih, 0.f);
        float ua = aarea + area - iw * ih;
        ua = fmaxf(ua, 1e-8f);
        const float iou = iw * ih / ua;
        if (iou > best) { best = iou; arg = j; }                      // first maximum wins, like torch.max
    }
    best_iou = best;
    if (!any) return -1;                                              // image without boxes: every anchor is background
    if (best >= 0.5f) return arg;
    if (best < 0.4f) return -1;
    return -2;
}

__global__ __launch_bounds__(256) void det_loss_fwd_kernel(const float* cls, const float* reg, const float* anchors, const float* ann, int A,
                                                           int K, int Mx, short* assign, float* part /* [N][blocks][3] */) {
    __shared__ float red[4][3];
    const int n = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    float s_cls = 0.f, s_reg = 0.f, s_pos = 0.f;
    if (a < A) {
        const float* anc = anchors + (long)a * 4;
        const float* an = ann + (long)n * Mx * 5;
        float biou;
        const int as = det_assign(anc, an, Mx, biou);
        assign[(long)n * A + a] = (short)as;
        if (as != -2) {
            const float* c = cls + ((long)n * A + a) * K;
            const int cid = as >= 0 ? (int)an[as * 5 + 4] : -1;
            for (int k = 0; k < K; ++k) {
                const float p = fminf(fmaxf(c[k], 1e-4f), 1.f - 1e-4f);
                if (k == cid) s_cls += 0.25f * (1.f - p) * (1.f - p) * -__logf(p);
                else s_cls += 0.75f * p * p * -__logf(1.f - p);
            }
        }
        if (as >= 0) {
            s_pos = 1.f;
            const float* b = an + as * 5;
            const float aw = anc[3] - anc[1], ah = anc[2] - anc[0];
            const float acx = anc[1] + 0.5f * aw, acy = anc[0] + 0.5f * ah;
            float gw = b[2] - b[0], gh = b[3] - b[1];
            const float gcx = b[0] + 0.5f * gw, gcy = b[1] + 0.5f * gh;
            gw = fmaxf(gw, 1.f);
            gh = fmaxf(gh, 1.f);
            const float t[4] = {(gcy - acy) / ah, (gcx - acx) / aw, __logf(gh / ah), __logf(gw / aw)};
            const float* r = reg + ((long)n * A + a) * 4;
            for (int q = 0; q < 4; ++q) {
                const float d = fabsf(t[q] - r[q]);
                s_reg += d <= 1.f / 9.f ? 0.5f * 9.f * d * d : d - 0.5f / 9.f;
            }
        }
    }
    s_cls = wave_sum(s_cls); s_reg = wave_sum(s_reg); s_pos = wave_sum(s_pos);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s_cls; red[threadIdx.x >> 6][1] = s_reg; red[threadIdx.x >> 6][2] = s_pos; }
    __syncthreads();
    if (threadIdx.x < 3) part[((long)n * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// per image: cls = sum_cls / max(npos, 1), reg = sum_reg / max(4 npos, 1) (0 without positives); outputs = batch means; npos kept for bwd
__global__ __launch_bounds__(1024) void det_loss_finalize_kernel(const float* part, int blocks, int N, float* npos, float* out /* [2] */) {
    // one wave per image (16 images walked one after the other by a single wave took 30 us of dependent load rounds), images beyond 16
    // in further passes; the batch means are summed in image order by one thread
    __shared__ float pc[64], pr[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float tc = 0.f, tr = 0.f;
    for (int n0 = 0; n0 < N; n0 += 16) {
        const int n = n0 + wave;
        float c = 0.f, r = 0.f, p = 0.f;
        if (n < N)
            for (int b = lane; b < blocks; b += 64) {
                const float* q = part + ((long)n * blocks + b) * 3;
                c += q[0]; r += q[1]; p += q[2];
            }
        c = wave_sum(c); r = wave_sum(r); p = wave_sum(p);
        __syncthreads();
        if (lane == 0 && n < N) {
            npos[n] = p;
            pc[wave] = c / fmaxf(p, 1.f);
            pr[wave] = p > 0.f ? r / (4.f * p) : 0.f;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < 16 && n0 + k < N; ++k) { tc += pc[k]; tr += pr[k]; }
    }
    if (threadIdx.x == 0) { out[0] = tc / N; out[1] = tr / N; }
}

__global__ __launch_bounds__(256) void det_loss_bwd_kernel(const float* cls, const float* reg, const float* anchors, const float* ann, int N, int A,
                                                           int K, int Mx, const short* assign, const float* npos, const float* gout /* [2] */,
                                                           float* dcls, float* dreg) {
    // the block's 256 x K probabilities / gradients are one contiguous run in memory: they move through LDS with coalesced 4-byte
    // accesses (a thread walking its own K floats 36 bytes from its neighbour's wrote partial lines: 0.9 TB/s, 91 us per step)
    extern __shared__ float tile[];                                    // [256][K]
    const int n = blockIdx.y;
    const int a0 = blockIdx.x * 256;
    const int a = a0 + threadIdx.x;
    const int nv = A - a0 < 256 ? A - a0 : 256;
    const float* cblk = cls + ((long)n * A + a0) * K;
    for (int i = threadIdx.x; i < nv * K; i += 256) tile[i] = cblk[i];
    __syncthreads();
    const float p_n = npos[n];
    const float fc = gout[0] / (N * fmaxf(p_n, 1.f));
    const float fr = p_n > 0.f ? gout[1] / (N * 4.f * p_n) : 0.f;
    const int as = a < A ? assign[(long)n * A + a] : -2;
    const float* an = ann + (long)n * Mx * 5;
    if (a < A) {
        float* c = tile + threadIdx.x * K;                             // (K odd or not: stride-K rows, each thread only touches its own)
        const int cid = as >= 0 ? (int)an[as * 5 + 4] : -1;
        for (int k = 0; k < K; ++k) {
            float g = 0.f;
            const float p = c[k];
            if (as != -2 && p > 1e-4f && p < 1.f - 1e-4f) {            // the clamp has zero slope outside its range
                if (k == cid) g = 0.25f * (2.f * (1.f - p) * __logf(p) - (1.f - p) * (1.f - p) / p);
                else g = 0.75f * (-2.f * p * __logf(1.f - p) + p * p / (1.f - p));
            }
            c[k] = fc * g;
        }
    }
    __syncthreads();
    float* dblk = dcls + ((long)n * A + a0) * K;
    for (int i = threadIdx.x; i < nv * K; i += 256) dblk[i] = tile[i];
    if (a >= A) return;
    float* dr = dreg + ((long)n * A + a) * 4;
    if (as >= 0) {
        const float* anc = anchors + (long)a * 4;
        const float* b = an + as * 5;
        const float aw = anc[3] - anc[1], ah = anc[2] - anc[0];
        const float acx = anc[1] + 0.5f * aw, acy = anc[0] + 0.5f * ah;
        float gw = b[2] - b[0], gh = b[3] - b[1];
        const float gcx = b[0] + 0.5f * gw, gcy = b[1] + 0.5f * gh;
        gw = fmaxf(gw, 1.f);
        gh = fmaxf(gh, 1.f);
        const float t[4] = {(gcy - acy) / ah, (gcx - acx) / aw, __logf(gh / ah), __logf(gw / aw)};
        const float* r = reg + ((long)n * A + a) * 4;
        for (int q = 0; q < 4; ++q) {
            const float d = t[q] - r[q];
            const float ad = fabsf(d);
            dr[q] = fr * (ad <= 1.f / 9.f ? -9.f * d : (d > 0.f ? -1.f : 1.f));
        }
    } else {
        dr[0] = dr[1] = dr[2] = dr[3] = 0.f;
    }
}

// deploy-mode argmax over C channel logits (NHWC fp32, first maximum wins like torch.argmax) -> int64
__global__ void argmax_kernel(const float* logits, int ldl, int C, long M, long* out) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const float* row = logits + m * ldl;
        float best = row[0];
        int arg = 0;
        for (int c = 1; c < C; ++c)
            if (row[c] > best) { best = row[c]; arg = c; }
        out[m] = arg;
    }
}

// ---------------------------------------------------------------------------------------------------------
static inline int bpi_for(long HW) {
    long b = (HW + 4095) / 4096;
    if (b > 64) b = 64;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int hn_seg_loss_blocks(int N, long HW) { return N * bpi_for(HW); }

// workspace layout (caller allocates, all zero-initialised by this call where needed):
//   loss fp32[M] | hist u32[N*2048] | state {u32,u32}[N] | ties u32[N] | psum fp32[blocks]
extern "C" long hn_seg_loss_ws_bytes(int N, long HW) {
    const long M = (long)N * HW;
    return M * 4 + (long)N * 2048 * 4 + (long)N * 8 + (long)N * 4 + (long)hn_seg_loss_blocks(N, HW) * 4 + 64;
}

__global__ void zero_u32_kernel(unsigned int* p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// forward: returns the scalar loss in out[0]; ws is kept for the backward pass.  k = int(top_k_ratio * HW) when use_topk.
extern "C" int hn_seg_loss_fwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw,
                               int ignore_index, int N, long HW, int use_topk, long k, void* ws, float* out, hipStream_t st) {
    HN_CHECK_ARG(logits && target && cw && ws && out && C >= 1 && C <= HN_SEG_MAXC && N > 0 && HW > 0 && (!use_topk || (k >= 1 && k <= HW)));
    const long M = (long)N * HW;
    char* w = (char*)ws;
    float* loss = (float*)w;
    unsigned int* hist = (unsigned int*)(w + M * 4);
    SelState* state = (SelState*)(w + M * 4 + (long)N * 2048 * 4);
    unsigned int* ties = (unsigned int*)(w + M * 4 + (long)N * 2048 * 4 + (long)N * 8);
    float* psum = (float*)(w + M * 4 + (long)N * 2048 * 4 + (long)N * 12);
    const int blocks = hn_seg_loss_blocks(N, HW);
    // zero the histogram + selection state with a kernel, not hipMemsetAsync: inside a captured hipGraph the memset node was observed to
    // lose its ordering against the following kernel (first replay after a host sync while another process shares the GPU -> NaN loss)
    {
        const long words = (long)N * 2048 + (long)N * 3;
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, hist, words);
    }
    if (C == 5 && ldl == 5 && target_is_float && (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(target) & 15) == 0)
        hipLaunchKernelGGL(seg_ce_fwd_vec_kernel<5>, dim3(blocks), dim3(256), 0, st, logits, (const float*)target, cw, ignore_index, HW, M, loss,
                           use_topk ? hist : (unsigned int*)nullptr);
    else
        hipLaunchKernelGGL(seg_ce_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, target, target_is_float, cw, ignore_index, HW, M,
                           loss, use_topk ? hist : (unsigned int*)nullptr);
    if (use_topk) {
        hipLaunchKernelGGL(seg_select_kernel, dim3(N), dim3(64), 0, st, hist, state, 0, (unsigned int)k);
        hipLaunchKernelGGL(seg_hist_kernel, dim3(blocks), dim3(256), 0, st, loss, HW, M, state, 1, hist);
        hipLaunchKernelGGL(seg_select_kernel, dim3(N), dim3(64), 0, st, hist, state, 1, (unsigned int)k);
        hipLaunchKernelGGL(seg_hist_kernel, dim3(blocks), dim3(256), 0, st, loss, HW, M, state, 2, hist);
        hipLaunchKernelGGL(seg_select_kernel, dim3(N), dim3(64), 0, st, hist, state, 2, (unsigned int)k);
        hipLaunchKernelGGL(seg_count_ties_kernel, dim3(blocks), dim3(256), 0, st, loss, HW, M, state, ties);
    }
    hipLaunchKernelGGL(seg_topk_sum_kernel, dim3(blocks), dim3(256), 0, st, loss, HW, M, state, use_topk, psum);
    const double denom = use_topk ? (double)N * (double)k : (double)M;
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, st, psum, blocks, state, N, use_topk, denom, out);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_seg_loss_bwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw,
                               int ignore_index, int N, long HW, int use_topk, long k, const void* ws, const float* gout, float* dlogits,
                               int ldd, hipStream_t st) {
    HN_CHECK_ARG(logits && target && cw && ws && gout && dlogits && C >= 1 && C <= HN_SEG_MAXC && N > 0 && HW > 0);
    const long M = (long)N * HW;
    const char* w = (const char*)ws;
    const float* loss = (const float*)w;
    const SelState* state = (const SelState*)(w + M * 4 + (long)N * 2048 * 4);
    const unsigned int* ties = (const unsigned int*)(w + M * 4 + (long)N * 2048 * 4 + (long)N * 8);
    const double denom = use_topk ? (double)N * (double)k : (double)M;
    long blocks = (M + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(seg_ce_bwd_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, target, target_is_float, cw, ignore_index, HW, M, loss,
                       state, ties, use_topk, gout, (float)(1.0 / denom), dlogits, ldd);
    HN_LAUNCH_CHECK();
}

/* hn_seg_loss_bwd writing the space-to-depth bf16 operand of the phase-form output conv's backward instead of fp32 dlogits:
 * dz [N][H/2][W/2][ldz], ldz >= 4*C (see seg_ce_bwd_s2d_kernel); H, W even */
extern "C" int hn_seg_loss_bwd_s2d(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw,
                                   int ignore_index, int N, int H, int W, int use_topk, long k, const void* ws, const float* gout, void* dz,
                                   int ldz, hipStream_t st) {
    HN_CHECK_ARG(logits && target && cw && ws && gout && dz && C >= 1 && C <= HN_SEG_MAXC && N > 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) &&
                 ldz >= 4 * C);
    const long HW = (long)H * W, M = (long)N * HW;
    const char* w = (const char*)ws;
    const float* loss = (const float*)w;
    const SelState* state = (const SelState*)(w + M * 4 + (long)N * 2048 * 4);
    const unsigned int* ties = (const unsigned int*)(w + M * 4 + (long)N * 2048 * 4 + (long)N * 8);
    const double denom = use_topk ? (double)N * (double)k : (double)M;
    long blocks = (M / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (C == 5 && ldl == 5 && ldz == 24 && target_is_float && (reinterpret_cast<uintptr_t>(logits) & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(target) & 7) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0) {
        // the shipped 5-class cfgs (cityscapes-style class list): vector loads / stores
        hipLaunchKernelGGL((seg_ce_bwd_s2d_vec_kernel<5, 24>), dim3(blocks), dim3(256), 0, st, logits, (const float*)target, cw, ignore_index, H, W,
                           M / 4, loss, state, ties, use_topk, gout, (float)(1.0 / denom), (bf16*)dz);
        HN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(seg_ce_bwd_s2d_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, target, target_is_float, cw, ignore_index, H, W,
                       M / 4, loss, state, ties, use_topk, gout, (float)(1.0 / denom), (bf16*)dz, ldz);
    HN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------
// Focal variant of the seg loss (head_seg/segmentation_loss.py:31-46, `use_focal: True` of cfgs/hydranet_joint_small_backbone.yml):
//   p = softmax(l) + 1e-8;  t = one_hot(y) + 1e-8;  loss_pix = sum_c t_c * (-alpha * (1 - p_c)^gamma * log(p_c) * w_c);  mean over all pixels
// (every class contributes through the +1e-8 of the one-hot; no ignore_index on this path, as in the reference).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float focal_pow(float q, float gamma) { return gamma == 2.0f ? q * q : powf(q, gamma); }

__global__ __launch_bounds__(256) void seg_focal_fwd_kernel(const float* logits, int ldl, int C, const void* target, int target_is_float,
                                                            const float* cw, float gamma, float alpha, long M, float* psum) {
    __shared__ float red[4];
    float s = 0.f;
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const int y = target_is_float ? (int)reinterpret_cast<const float*>(target)[m] : (int)reinterpret_cast<const long*>(target)[m];
        const float* row = logits + m * ldl;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float e[HN_SEG_MAXC], se = 0.f;
        for (int c = 0; c < C; ++c) { e[c] = expf(row[c] - mx); se += e[c]; }
        float l = 0.f;
        for (int c = 0; c < C; ++c) {
            const float p = e[c] / se + 1e-8f;
            const float t = (c == y ? 1.0f : 0.0f) + 1e-8f;
            l += t * (-alpha * focal_pow(1.0f - p, gamma) * logf(p) * cw[c]);
        }
        s += l;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) psum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dlogits_j = gout/M * s_j * (g_j - sum_c g_c s_c),  g_c = dloss/dp_c = t_c w_c alpha (gamma (1-p_c)^(gamma-1) log p_c - (1-p_c)^gamma / p_c)
__global__ __launch_bounds__(256) void seg_focal_bwd_kernel(const float* logits, int ldl, int C, const void* target, int target_is_float,
                                                            const float* cw, float gamma, float alpha, long M, const float* gout,
                                                            float inv_denom, float* dlogits, int ldd) {
    const float gs = gout[0] * inv_denom;
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const int y = target_is_float ? (int)reinterpret_cast<const float*>(target)[m] : (int)reinterpret_cast<const long*>(target)[m];
        const float* row = logits + m * ldl;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float sm[HN_SEG_MAXC], g[HN_SEG_MAXC], se = 0.f, dot = 0.f;
        for (int c = 0; c < C; ++c) { sm[c] = expf(row[c] - mx); se += sm[c]; }
        for (int c = 0; c < C; ++c) {
            sm[c] /= se;
            const float p = sm[c] + 1e-8f, q = 1.0f - p;
            const float t = (c == y ? 1.0f : 0.0f) + 1e-8f;
            const float qg1 = gamma == 2.0f ? q : powf(q, gamma - 1.0f);
            g[c] = t * cw[c] * alpha * (gamma * qg1 * logf(p) - focal_pow(q, gamma) / p);
            dot += g[c] * sm[c];
        }
        float* d = dlogits + m * ldd;
        for (int c = 0; c < C; ++c) d[c] = gs * sm[c] * (g[c] - dot);
    }
}

/* focal seg loss: logits fp32 [N*HW][ldl] (C classes), target float32 or int64 class ids [N*HW] (no ignore_index on this path, as in the
 * reference), ws = fp32 [hn_seg_loss_blocks(N, HW)] partial sums, out[0] = mean over all N*HW pixels */
extern "C" int hn_seg_focal_fwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw, float gamma,
                                float alpha, int N, long HW, void* ws, float* out, hipStream_t st) {
    HN_CHECK_ARG(logits && target && cw && ws && out && C >= 1 && C <= HN_SEG_MAXC && N > 0 && HW > 0);
    const long M = (long)N * HW;
    const int blocks = hn_seg_loss_blocks(N, HW);
    hipLaunchKernelGGL(seg_focal_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, target, target_is_float, cw, gamma, alpha, M,
                       (float*)ws);
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, blocks, (const SelState*)nullptr, N, 0, (double)M, out);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_seg_focal_bwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw, float gamma,
                                float alpha, int N, long HW, const float* gout, float* dlogits, int ldd, hipStream_t st) {
    HN_CHECK_ARG(logits && target && cw && gout && dlogits && C >= 1 && C <= HN_SEG_MAXC && N > 0 && HW > 0);
    const long M = (long)N * HW;
    long blocks = (M + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(seg_focal_bwd_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, target, target_is_float, cw, gamma, alpha, M, gout,
                       (float)(1.0 / (double)M), dlogits, ldd);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_argmax_channels(const float* logits, int ldl, int C, long M, long* out, hipStream_t st) {
    HN_CHECK_ARG(logits && out && C >= 1 && M > 0);
    long blocks = (M + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(argmax_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, M, out);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_det_loss_blocks(int A) { return (A + 255) / 256; }

// fwd: out[0] = classification loss, out[1] = regression loss (batch means); assign int16 [N][A], part fp32 [N][blocks][3], npos fp32 [N]
extern "C" int hn_det_loss_fwd(const float* cls, const float* reg, const float* anchors, const float* ann, int N, int A, int K, int Mx,
                               void* assign, float* part, float* npos, float* out, hipStream_t st) {
    HN_CHECK_ARG(cls && reg && anchors && ann && assign && part && npos && out && N > 0 && A > 0 && K > 0 && K <= HN_DET_MAXK && Mx > 0);
    const int blocks = hn_det_loss_blocks(A);
    hipLaunchKernelGGL(det_loss_fwd_kernel, dim3(blocks, N), dim3(256), 0, st, cls, reg, anchors, ann, A, K, Mx, (short*)assign, part);
    hipLaunchKernelGGL(det_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, part, blocks, N, npos, out);
    HN_LAUNCH_CHECK();
}

extern "C" int hn_det_loss_bwd(const float* cls, const float* reg, const float* anchors, const float* ann, int N, int A, int K, int Mx,
                               const void* assign, const float* npos, const float* gout, float* dcls, float* dreg, hipStream_t st) {
    HN_CHECK_ARG(cls && reg && anchors && ann && assign && npos && gout && dcls && dreg && N > 0 && A > 0 && K > 0 && Mx > 0);
    HN_CHECK_ARG(K <= 48);                                             // 256 x K floats of LDS
    hipLaunchKernelGGL(det_loss_bwd_kernel, dim3(hn_det_loss_blocks(A), N), dim3(256), 256 * (size_t)K * sizeof(float), st, cls, reg, anchors, ann, N, A, K, Mx,
                       (const short*)assign, npos, gout, dcls, dreg);
    HN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------
// Lane losses (head_lane/lanedetect_loss.py:5-78): OHEM classification loss and masked Huber location loss.
//   cls: M = N*hw anchors, logits [M][2], target one-hot [M][2] (positive = t[1] > 0).  NEGATIVE_RATIO 15, ALPHA 10.
//        neg_num = clamp(15 * #pos, 1, #neg); thr = neg_num-th smallest background log-prob among the negatives (detached);
//        pos = -alpha * sum_{pos} log p_fg / max(#pos,1);  neg = -alpha * sum_{neg, log p_bg <= thr} log p_bg / max(#pos,1)
//   One workgroup of 1024 threads does the whole problem (M is 8 192 at batch 16): log-softmax, counts, a 4-pass byte radix select of the
//   k-th smallest value in LDS histograms (instead of torch.sort), and both sums.  aux[0..3] = {thr, max(#pos,1), #pos, #neg}.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int f2ord(float f) {              // order-preserving float -> uint
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o) {
    const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ float block_sum_1024(float v, float* red) {  // red: >= 16 floats of LDS; all threads get the total
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
// k-th element's bin of a 256-bin LDS histogram, found by all threads together: thread b < 256 takes bin b, an inclusive prefix sum runs
// inside each of the four waves (shuffles) and across them (LDS), and the one thread whose bin satisfies excl < k <= incl publishes
// (bin, k - excl).  (One thread walking the 256 bins was ~10 us per radix pass: 256 dependent LDS reads.)  All 1024 threads call it.
__device__ __forceinline__ void hist_select_256(const unsigned int* hist, unsigned int k, unsigned int* wsum /* [4] LDS */, unsigned int* out_bin,
                                                unsigned int* out_k) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned int v = tid < 256 ? hist[tid] : 0u, incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (tid < 256 && lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid < 256) {
        unsigned int base = 0;
        for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
        incl += base;
        const unsigned int excl = incl - v;
        if (excl < k && k <= incl) { *out_bin = (unsigned int)tid; *out_k = k - excl; }
    }
    __syncthreads();
}
__global__ __launch_bounds__(1024) void lane_cls_fwd_kernel(const float* logits, const float* target, long M, float neg_ratio, float alpha,
                                                            float* lsm, unsigned char* pmask, float* out, float* aux) {
    __shared__ unsigned int hist[256];
    __shared__ float red[16];
    __shared__ unsigned int s_bin, s_k, s_wsum[4];
    const int tid = threadIdx.x;
    float npos = 0.f, nneg = 0.f;
    for (long i = tid; i < M; i += 1024) {
        const float z0 = logits[2 * i], z1 = logits[2 * i + 1];
        const float mx = fmaxf(z0, z1);
        const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
        lsm[2 * i] = z0 - lse;
        lsm[2 * i + 1] = z1 - lse;
        const bool p = target[2 * i + 1] > 0.f;
        pmask[i] = p ? 1 : 0;
        npos += p ? 1.f : 0.f;
        nneg += p ? 0.f : 1.f;
    }
    npos = block_sum_1024(npos, red);
    nneg = block_sum_1024(nneg, red);
    const float posn = fmaxf(npos, 1.f);
    long kk = (long)fmaxf(fminf(npos * neg_ratio, nneg), 1.f);       // 1-based rank among the negatives
    float thr = __builtin_inff();
    if (nneg >= 1.f) {
        // byte-wise radix select, most significant byte first, over the order-preserving keys of the negatives' background log-probs
        unsigned int prefix = 0, k = (unsigned int)kk;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (long i = tid; i < M; i += 1024) {
                if (pmask[i]) continue;
                const unsigned int key = f2ord(lsm[2 * i]);
                if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
            }
            __syncthreads();
            hist_select_256(hist, k, s_wsum, &s_bin, &s_k);
            prefix |= s_bin << shift;
            k = s_k;
        }
        thr = ord2f(prefix);
    }
    float sp = 0.f, sn = 0.f;
    for (long i = tid; i < M; i += 1024) {
        const float bg = lsm[2 * i], fg = lsm[2 * i + 1];
        if (pmask[i]) sp += fg;
        else if (bg <= thr) sn += bg;
    }
    sp = block_sum_1024(sp, red);
    sn = block_sum_1024(sn, red);
    if (tid == 0) {
        out[0] = -alpha * sp / posn;
        out[1] = -alpha * sn / posn;
        aux[0] = thr; aux[1] = posn; aux[2] = npos; aux[3] = nneg;
    }
}
// The same selection with every row held in registers (M <= 1024 * RPT): the seven passes of the kernel above each made a global-memory
// round trip over the rows with one workgroup (58 us for 32 768 rows); here the rows are read once, the radix passes run on registers.
template <int RPT>
__global__ __launch_bounds__(1024) void lane_cls_fwd_reg_kernel(const float* logits, const float* target, long M, float neg_ratio, float alpha,
                                                                float* lsm, unsigned char* pmask, float* out, float* aux) {
    __shared__ unsigned int hist[256];
    __shared__ float red[16];
    __shared__ unsigned int s_bin, s_k, s_wsum[4];
    const int tid = threadIdx.x;
    float bg[RPT], fg[RPT];
    unsigned int pos = 0, valid = 0;                                  // bit j: row tid + 1024 j is a positive / exists
    float npos = 0.f, nneg = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < RPT; j0 += 8) {                              // eight rows' loads in flight at a time
        float2 zz[8], tt[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const long i = tid + 1024L * (j0 + jj);
            zz[jj] = make_float2(0.f, 0.f); tt[jj] = make_float2(0.f, 0.f);
            if (i < M) {
                zz[jj] = *reinterpret_cast<const float2*>(logits + 2 * i);
                tt[jj] = *reinterpret_cast<const float2*>(target + 2 * i);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = j0 + jj;
            const long i = tid + 1024L * j;
            const float z0 = zz[jj].x, z1 = zz[jj].y;
            const float mx = fmaxf(z0, z1);
            const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
            bg[j] = z0 - lse;
            fg[j] = z1 - lse;
            if (i < M) {
                const bool p = tt[jj].y > 0.f;
                valid |= 1u << j;
                pos |= p ? (1u << j) : 0u;
                *reinterpret_cast<float2*>(lsm + 2 * i) = make_float2(bg[j], fg[j]);
                pmask[i] = p ? 1 : 0;
                npos += p ? 1.f : 0.f;
                nneg += p ? 0.f : 1.f;
            }
        }
    }
    npos = block_sum_1024(npos, red);
    nneg = block_sum_1024(nneg, red);
    const float posn = fmaxf(npos, 1.f);
    long kk = (long)fmaxf(fminf(npos * neg_ratio, nneg), 1.f);
    float thr = __builtin_inff();
    const unsigned int negs = valid & ~pos;
    if (nneg >= 1.f) {
        unsigned int prefix = 0, k = (unsigned int)kk;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                if (!((negs >> j) & 1u)) continue;
                const unsigned int key = f2ord(bg[j]);
                if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
            }
            __syncthreads();
            hist_select_256(hist, k, s_wsum, &s_bin, &s_k);
            prefix |= s_bin << shift;
            k = s_k;
        }
        thr = ord2f(prefix);
    }
    float sp = 0.f, sn = 0.f;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        if ((pos >> j) & 1u) sp += fg[j];
        else if (((negs >> j) & 1u) && bg[j] <= thr) sn += bg[j];
    }
    sp = block_sum_1024(sp, red);
    sn = block_sum_1024(sn, red);
    if (tid == 0) {
        out[0] = -alpha * sp / posn;
        out[1] = -alpha * sn / posn;
        aux[0] = thr; aux[1] = posn; aux[2] = npos; aux[3] = nneg;
    }
}
// dlogits = gpos * d(pos)/dz + gneg * d(neg)/dz
__global__ void lane_cls_bwd_kernel(const float* lsm, const unsigned char* pmask, const float* aux, const float* gpos, const float* gneg,
                                    float alpha, long M, float* dlogits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float thr = aux[0], posn = aux[1];
    const float bg = lsm[2 * i], fg = lsm[2 * i + 1];
    const float p0 = expf(bg), p1 = expf(fg);
    float d0 = 0.f, d1 = 0.f;
    if (pmask[i]) {                                    // -alpha/P * log p_fg
        const float c = -alpha / posn * gpos[0];
        d0 = -p0 * c;
        d1 = (1.f - p1) * c;
    } else if (bg <= thr) {                            // -alpha/P * log p_bg
        const float c = -alpha / posn * gneg[0];
        d0 = (1.f - p0) * c;
        d1 = -p1 * c;
    }
    dlogits[2 * i] = d0;
    dlogits[2 * i + 1] = d1;
}
// location loss: per row r (positive anchors only): sum_c huber(p - t) * w_c * [t != 0] / max(#[t != 0], 1); w = alpha at columns wcol, wcol+1.
// block = 8 rows x 32 lanes; rowloss[r] (0 for negatives) and the per-row normaliser go to workspace; a 1-block finalize sums the rows.
__global__ __launch_bounds__(256) void lane_loc_fwd_kernel(const float* pred, const float* tgt, const unsigned char* pmask, long M, int L,
                                                           int wcol, float alpha, float* rowloss, float* rownorm) {
    const int lane = threadIdx.x & 31, rr = threadIdx.x >> 5;
    const long r = (long)blockIdx.x * 8 + rr;
    float s = 0.f, cnt = 0.f;
    if (r < M) {
        const bool pos = pmask[r] != 0;
        for (int c = lane; c < L; c += 32) {
            const float t = tgt[r * L + c];
            if (t != 0.f) {
                cnt += 1.f;
                if (pos) {
                    const float d = pred[r * L + c] - t, ad = fabsf(d);
                    const float hv = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
                    s += hv * ((c == wcol || c == wcol + 1) ? alpha : 1.f);
                }
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o); cnt += __shfl_xor(cnt, o); }
    if (lane == 0 && r < M) {
        const float nrm = fmaxf(cnt, 1.f);
        rownorm[r] = nrm;
        rowloss[r] = s / nrm;
    }
}
__global__ __launch_bounds__(1024) void lane_loc_finalize_kernel(const float* rowloss, long M, const float* aux, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < M; i += 1024) s += rowloss[i];
    s = block_sum_1024(s, red);
    if (threadIdx.x == 0) out[0] = s / aux[1];
}
__global__ void lane_loc_bwd_kernel(const float* pred, const float* tgt, const unsigned char* pmask, const float* rownorm, const float* aux,
                                    const float* gout, long M, int L, int wcol, float alpha, float* dpred) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * L) return;
    const long r = idx / L;
    const int c = (int)(idx - r * L);
    float g = 0.f;
    const float t = tgt[idx];
    if (pmask[r] && t != 0.f) {
        const float d = pred[idx] - t;
        const float dh = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        g = dh * ((c == wcol || c == wcol + 1) ? alpha : 1.f) / rownorm[r] / aux[1] * gout[0];
    }
    dpred[idx] = g;
}

extern "C" int hn_lane_cls_loss_fwd(const float* logits, const float* target, long M, float neg_ratio, float alpha, float* lsm,
                                    void* pmask, float* out, float* aux, hipStream_t st) {
    HN_CHECK_ARG(logits && target && lsm && pmask && out && aux && M > 0 && M <= (1L << 22));
    const bool al8 = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(lsm)) & 7) == 0;
    if (al8 && M <= 1024L * 8)
        hipLaunchKernelGGL(lane_cls_fwd_reg_kernel<8>, dim3(1), dim3(1024), 0, st, logits, target, M, neg_ratio, alpha, lsm, (unsigned char*)pmask, out, aux);
    else if (al8 && M <= 1024L * 32)
        hipLaunchKernelGGL(lane_cls_fwd_reg_kernel<32>, dim3(1), dim3(1024), 0, st, logits, target, M, neg_ratio, alpha, lsm, (unsigned char*)pmask, out, aux);
    else
        hipLaunchKernelGGL(lane_cls_fwd_kernel, dim3(1), dim3(1024), 0, st, logits, target, M, neg_ratio, alpha, lsm, (unsigned char*)pmask, out, aux);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_lane_cls_loss_bwd(const float* lsm, const void* pmask, const float* aux, const float* gpos, const float* gneg, float alpha,
                                    long M, float* dlogits, hipStream_t st) {
    HN_CHECK_ARG(lsm && pmask && aux && gpos && gneg && dlogits && M > 0);
    hipLaunchKernelGGL(lane_cls_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, lsm, (const unsigned char*)pmask, aux, gpos, gneg,
                       alpha, M, dlogits);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_lane_loc_loss_fwd(const float* pred, const float* target, const void* pmask, const float* aux, long M, int L, int wcol,
                                    float alpha, float* rowloss, float* rownorm, float* out, hipStream_t st) {
    HN_CHECK_ARG(pred && target && pmask && aux && rowloss && rownorm && out && M > 0 && L > 0 && wcol >= 0 && wcol + 1 < L);
    hipLaunchKernelGGL(lane_loc_fwd_kernel, dim3((unsigned)((M + 7) / 8)), dim3(256), 0, st, pred, target, (const unsigned char*)pmask, M, L, wcol,
                       alpha, rowloss, rownorm);
    hipLaunchKernelGGL(lane_loc_finalize_kernel, dim3(1), dim3(1024), 0, st, rowloss, M, aux, out);
    HN_LAUNCH_CHECK();
}
extern "C" int hn_lane_loc_loss_bwd(const float* pred, const float* target, const void* pmask, const float* rownorm, const float* aux,
                                    const float* gout, long M, int L, int wcol, float alpha, float* dpred, hipStream_t st) {
    HN_CHECK_ARG(pred && target && pmask && rownorm && aux && gout && dpred && M > 0 && L > 0);
    hipLaunchKernelGGL(lane_loc_bwd_kernel, dim3((unsigned)((M * L + 255) / 256)), dim3(256), 0, st, pred, target, (const unsigned char*)pmask,
                       rownorm, aux, gout, M, L, wcol, alpha, dpred);
    HN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------
// Device NMS for the detection post-process (head_detect/detection_loss.py:70-108; torchvision.ops.batched_nms semantics restated in
// postprocess.py): boxes are already sorted by descending score (stable) and carry their class offset.
//   nms_mask_kernel: bit (i, j) of mask[i][j/64] = (j > i) and IoU(i, j) > thr, IoU = inter / (area_i + area_j - inter) in separately
//                    rounded fp32 operations (no FMA contraction), so the comparisons are bit-identical to the numpy/torch host path;
//   nms_scan_kernel: one wave walks the boxes in order, keeps box i unless an earlier kept box suppressed it (64-bit word per lane).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_rn(const float4 a, const float4 b) {
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    float iw = __fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x));
    float ih = __fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y));
    iw = iw > 0.f ? iw : 0.f;
    ih = ih > 0.f ? ih : 0.f;
    const float inter = __fmul_rn(iw, ih);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
}
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* boxes, int K, float thr, unsigned long long* mask, int words) {
    const int i = blockIdx.y;                         // row box
    const int wj = blockIdx.x;                        // 64-column word
    const int j = wj * 64 + threadIdx.x;
    bool sup = false;
    if (j < K && j > i) sup = iou_rn(boxes[i], boxes[j]) > thr;
    const unsigned long long bits = __ballot(sup);
    if (threadIdx.x == 0) mask[(long)i * words + wj] = bits;
}
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* mask, int K, int words, unsigned char* keep) {
    // lane l owns removed-words l, l+64, ...  (K <= 64 * 64 * NW)
    constexpr int NW = 8;
    unsigned long long removed[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) removed[w] = 0ull;
    const int lane = threadIdx.x;
    for (int i = 0; i < K; ++i) {
        const int wi = i >> 6;
        // is box i still alive?  its bit lives in word wi, owned by lane wi % 64, slot wi / 64
        unsigned long long wv = 0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) if ((wi >> 6) == w) wv = removed[w];
        const unsigned long long word = __shfl(wv, wi & 63);
        const bool alive = !((word >> (i & 63)) & 1ull);
        if (lane == 0) keep[i] = alive ? 1 : 0;
        if (alive) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int ww = w * 64 + lane;
                if (ww < words) removed[w] |= mask[(long)i * words + ww];
            }
        }
    }
}
extern "C" long hn_nms_mask_words(int K) { return (long)K * ((K + 63) / 64); }
/* boxes: fp32 [K][4] (x1, y1, x2, y2), sorted by descending score, class offsets already added; mask: hn_nms_mask_words(K) uint64;
 * keep: K bytes (1 = kept).  K <= 32768. */
extern "C" int hn_nms_sorted(const float* boxes, int K, float iou_threshold, void* mask, void* keep, hipStream_t st) {
    HN_CHECK_ARG(boxes && mask && keep && K > 0 && K <= 32768);
    const int words = (K + 63) / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, K), dim3(64), 0, st, (const float4*)boxes, K, iou_threshold, (unsigned long long*)mask, words);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)mask, K, words, (unsigned char*)keep);
    HN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------
// Weighted multitask loss sum (HydraTrainer.cal_total_loss, model/train.py:192-203) in one launch:
//   total = sum_g ( sum_{i in g} x_i * w_i ) * gw_g        evaluated left to right in fp32, no fused multiply-add (the reference's separate
//   mul / add kernels round after every operation); backward: dx_i = (gout * gw_g) * w_i
// ---------------------------------------------------------------------------------------------------------
#define HN_MAX_LOSS_TERMS 8
struct WSum {
    const float* x[HN_MAX_LOSS_TERMS]; float w[HN_MAX_LOSS_TERMS]; float gw[HN_MAX_LOSS_TERMS]; int grp[HN_MAX_LOSS_TERMS];
    int n; const float* gout; float* out; float* grads;
};
__global__ void weighted_sum_kernel(const WSum p) {
    if (threadIdx.x != 0) return;
    if (p.out) {
        float tot = 0.f;
        int i = 0;
        while (i < p.n) {
            const int g = p.grp[i];
            float s = __fmul_rn(p.x[i][0], p.w[i]);
            for (++i; i < p.n && p.grp[i] == g; ++i) s = __fadd_rn(s, __fmul_rn(p.x[i][0], p.w[i]));
            tot = __fadd_rn(tot, __fmul_rn(s, p.gw[g]));
        }
        p.out[0] = tot;
    }
    if (p.grads) {
        const float go = p.gout[0];
        for (int i = 0; i < p.n; ++i) p.grads[i] = __fmul_rn(__fmul_rn(go, p.gw[p.grp[i]]), p.w[i]);
    }
}
/* xs: HOST array of n device pointers (one fp32 scalar each); w, gw, grp: HOST arrays (term weight, group weight indexed by group id, group
 * id per term; terms of a group are consecutive); out (optional): the total; grads (optional, with gout): n gradients */
extern "C" int hn_weighted_sum(const void* const* xs, const float* w, const float* gw, const int* grp, int n, const float* gout, float* out,
                               float* grads, hipStream_t st) {
    HN_CHECK_ARG(xs && w && gw && grp && n > 0 && n <= HN_MAX_LOSS_TERMS && (out || grads) && (!grads || gout));
    WSum p = {};
    for (int i = 0; i < n; ++i) {
        HN_CHECK_ARG(xs[i] && grp[i] >= 0 && grp[i] < HN_MAX_LOSS_TERMS && (i == 0 || grp[i] >= grp[i - 1]));
        p.x[i] = (const float*)xs[i]; p.w[i] = w[i]; p.grp[i] = grp[i];
    }
    for (int g = 0; g <= grp[n - 1]; ++g) p.gw[g] = gw[g];
    p.n = n; p.gout = gout; p.out = out; p.grads = grads;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, st, p);
    HN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Adam step of ALL parameters in one launch (torch.optim.Adam as the reference constructs it, model/train.py:147: L2 weight decay added to
// the gradient, bias-corrected moments, eps outside the square root).  The foreach implementation is ~10 launches over 693 tensor lists
// (3.9 ms per step on the big cfg); this is one pass over p, g, m, v (1.2 GB: ~0.35 ms).  jobs (device): n x {p, g, m, v, numel,
// first_block}; a block = 256 threads x 4 consecutive elements; block_job: job index of every block.
// Same operation order as torch's single-tensor formula (lerp for the first moment, mul + addcmul for the second, sqrt / bias2_sqrt +
// eps, addcdiv), with explicitly rounded steps (no fused multiply-add), so that it tracks torch.optim.Adam to the last bit or two.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_step_kernel(const long* jobs, const int* block_job, float lr_over_bc1, float w1, float b2, float w2,
                                                        float eps, float wd, float bc2_sqrt) {
    const long* jb = jobs + (long)block_job[blockIdx.x] * 6;
    float* p = reinterpret_cast<float*>(jb[0]);
    const float* g = reinterpret_cast<const float*>(jb[1]);
    float* m = reinterpret_cast<float*>(jb[2]);
    float* v = reinterpret_cast<float*>(jb[3]);
    const long n = jb[4];
    const long i0 = (((long)blockIdx.x - jb[5]) * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    auto one = [&](float pv, float gv, float& mv, float& vv) {
        if (wd != 0.f) gv = __fadd_rn(gv, __fmul_rn(wd, pv));
        mv = __fadd_rn(mv, __fmul_rn(w1, __fsub_rn(gv, mv)));                          // exp_avg.lerp_(grad, 1 - beta1)
        vv = __fadd_rn(__fmul_rn(vv, b2), __fmul_rn(w2, __fmul_rn(gv, gv)));           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv), bc2_sqrt), eps);
        return __fsub_rn(pv, __fmul_rn(lr_over_bc1, __fdiv_rn(mv, denom)));            // param.addcdiv_(exp_avg, denom, value = -step_size)
    };
    // ONE arithmetic instruction stream for both operand forms (whole aligned float4s / element by element): the vector and the scalar
    // branch used to carry their own copies of `one`, and the two compiled forms differed by an ulp on a few elements per tensor -- a
    // data-parallel run (gradients = views at arbitrary offsets of a flat bucket) then drifted from the single-GPU run bit by bit
    const bool vec = i0 + 4 <= n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                      reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    float pv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + i0), b = *reinterpret_cast<const f32x4*>(g + i0);
        const f32x4 c = *reinterpret_cast<const f32x4*>(m + i0), d = *reinterpret_cast<const f32x4*>(v + i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) { pv[k] = a[k]; gv[k] = b[k]; mv[k] = c[k]; vv[k] = d[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n) { pv[k] = p[i0 + k]; gv[k] = g[i0 + k]; mv[k] = m[i0 + k]; vv[k] = v[i0 + k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) pv[k] = one(pv[k], gv[k], mv[k], vv[k]);
    if (vec) {
        *reinterpret_cast<f32x4*>(p + i0) = (f32x4){pv[0], pv[1], pv[2], pv[3]};
        *reinterpret_cast<f32x4*>(m + i0) = (f32x4){mv[0], mv[1], mv[2], mv[3]};
        *reinterpret_cast<f32x4*>(v + i0) = (f32x4){vv[0], vv[1], vv[2], vv[3]};
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n) { p[i0 + k] = pv[k]; m[i0 + k] = mv[k]; v[i0 + k] = vv[k]; }
    }
}

extern "C" int hn_adam_step(const long* jobs, const int* block_job, long total_blocks, double lr, double beta1, double beta2, double eps,
                            double weight_decay, long step, hipStream_t st) {
    HN_CHECK_ARG(jobs && block_job && total_blocks > 0 && step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
    // the scalars as torch forms them from Python doubles: 1 - beta, 1 - beta ** step, lr / bias_correction1, sqrt(bias_correction2) in
    // double, rounded to fp32 once (1.0f - 0.999f differs from float(0.001) by 1.3e-5)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jobs, block_job, (float)(lr / bc1), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2));
    HN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Many fp32 tensors copied (or rounded to bf16) in ONE launch: the gather of a gradient bucket before its all-reduce and the scatter of
// a reduced-precision payload back to fp32 (torch._foreach_copy_ issues one launch per ~50 tensors: 14 launches, 0.5 ms per step for the
// 693 gradients).  jobs (device): n x {src, dst, numel, first_block}; block = 256 threads x 4 elements; kind: 0 f32 -> f32, 1 f32 -> bf16,
// 2 bf16 -> f32.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_many_kernel(const long* jobs, const int* block_job, int kind) {
    const long* jb = jobs + (long)block_job[blockIdx.x] * 4;
    const long n = jb[2];
    const long i0 = (((long)blockIdx.x - jb[3]) * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    const int cnt = n - i0 < 4 ? (int)(n - i0) : 4;
    if (kind == 0) {
        const float* s = reinterpret_cast<const float*>(jb[0]) + i0;
        float* d = reinterpret_cast<float*>(jb[1]) + i0;
        if (cnt == 4 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(s);
        else for (int k = 0; k < cnt; ++k) d[k] = s[k];
    } else if (kind == 1) {
        const float* s = reinterpret_cast<const float*>(jb[0]) + i0;
        bf16* d = reinterpret_cast<bf16*>(jb[1]) + i0;
        for (int k = 0; k < cnt; ++k) d[k] = f2bf(s[k]);
    } else {
        const bf16* s = reinterpret_cast<const bf16*>(jb[0]) + i0;
        float* d = reinterpret_cast<float*>(jb[1]) + i0;
        for (int k = 0; k < cnt; ++k) d[k] = bf2f(s[k]);
    }
}

extern "C" int hn_copy_many(const long* jobs, const int* block_job, long total_blocks, int kind, hipStream_t st) {
    HN_CHECK_ARG(jobs && block_job && total_blocks > 0 && kind >= 0 && kind <= 2);
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jobs, block_job, kind);
    HN_LAUNCH_CHECK();
}
