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

// one block per image: walk the histogram from the top bin down until the remaining rank falls inside a bin; fix that bin's bits
__global__ __launch_bounds__(64) void seg_select_kernel(unsigned int* hist, SelState* st, int level, unsigned int k) {
    const int n = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned int* h = hist + (long)n * 2048;
        unsigned int need = level == 0 ? k : st[n].remaining;
        const int bins = level == 2 ? 1024 : 2048;
        int b = bins - 1;
        for (; b > 0; --b) {
            if (h[b] >= need) break;
            need -= h[b];
        }
        const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
        const unsigned int prefix = (level == 0 ? 0u : st[n].prefix) | ((unsigned int)b << shift);
        st[n].prefix = prefix;
        st[n].remaining = need;                        // elements of bin b (at this resolution) that still belong to the top-k
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) hist[(long)n * 2048 + i] = 0;     // ready for the next level
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
    if (hipMemsetAsync(hist, 0, (size_t)N * 2048 * 4 + (size_t)N * 12, st) != hipSuccess) return HN_ERR_LAUNCH;
    hipLaunchKernelGGL(seg_ce_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, target, target_is_float, cw, ignore_index, HW, M, loss,
                       use_topk ? hist : (unsigned int*)nullptr);
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

extern "C" int hn_argmax_channels(const float* logits, int ldl, int C, long M, long* out, hipStream_t st) {
    HN_CHECK_ARG(logits && out && C >= 1 && M > 0);
    long blocks = (M + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(argmax_kernel, dim3(blocks), dim3(256), 0, st, logits, ldl, C, M, out);
    HN_LAUNCH_CHECK();
}
