// Device-side stages either side of the forward/backward hot path:
//   1. detection post-process: box decode + clip + per-anchor max / arg-max class + score threshold compaction + stable score sort +
//      class-offset greedy NMS + gather, for a whole batch, no host round trip between the stages
//      (head_detect/detection_loss.py:7-108, model/model.py:193-198; torchvision.ops.batched_nms semantics, see postprocess.py)
//   2. lane decode + lane NMS (head_lane/lane_codec.py:116-219, head_lane/lane_codec_utils.py:487-543, head_lane/lanedetect.py:103-116)
//   3. input pre-processing: BGR uint8 HWC frame -> bilinear resize -> RGB -> /255, ImageNet mean / std -> fp32 NCHW
//      (demo.py:26-50,191-196; dataset/utility.py:213-227)
//   4. streaming confusion counts for the segmentation mIoU (head_seg/seg_metrics.py:12-101)
// Index / ordering logic is exact integer work; floating-point decisions (IoU > thr, score > thr, distance <= thr) use separately rounded
// fp32 operations (no FMA contraction) in the reference's operation order, so they are bit-identical to the host path.
#include "hn_common.h"

// =====================================================================================================================================
// 1. detection post-process
// =====================================================================================================================================
struct DetCand {            // one above-threshold anchor
    float score; int anchor; int cls; int pad;
    float x1, y1, x2, y2;
};

__device__ __forceinline__ int float_order_key(float f) {        // monotone float -> int map (for atomicMax on floats of any sign)
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float float_from_key(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// per (image, anchor): score = max_k cls, class = first arg-max, keep when score > thr; decode the box (BBoxTransform) and clip it
// (ClipBoxes); compact with a per-image counter (the order is fixed later by the sort key (score desc, anchor asc)); track the maximum
// coordinate of the image's kept boxes for the class offset of batched_nms.
__global__ __launch_bounds__(256) void det_select_kernel(const float* anchors, const float* reg, const float* cls, int A, int K, float thr,
                                                         float wmax, float hmax, DetCand* cand, int cap, int* count, int* maxkey) {
    const int n = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const float* c = cls + ((long)n * A + a) * K;
    float best = c[0];
    int bi = 0;
    for (int k = 1; k < K; ++k) {
        const float v = c[k];
        if (v > best) { best = v; bi = k; }
    }
    if (!(best > thr)) return;
    const float* an = anchors + (long)a * 4;                  // (y1, x1, y2, x2)
    const float* r = reg + ((long)n * A + a) * 4;             // (dy, dx, dh, dw)
    const float yca = __fdiv_rn(__fadd_rn(an[0], an[2]), 2.f), xca = __fdiv_rn(__fadd_rn(an[1], an[3]), 2.f);
    const float ha = __fsub_rn(an[2], an[0]), wa = __fsub_rn(an[3], an[1]);
    const float w = __fmul_rn(expf(r[3]), wa), h = __fmul_rn(expf(r[2]), ha);
    const float yc = __fadd_rn(__fmul_rn(r[0], ha), yca), xc = __fadd_rn(__fmul_rn(r[1], wa), xca);
    float x1 = __fsub_rn(xc, __fdiv_rn(w, 2.f)), y1 = __fsub_rn(yc, __fdiv_rn(h, 2.f));
    float x2 = __fadd_rn(xc, __fdiv_rn(w, 2.f)), y2 = __fadd_rn(yc, __fdiv_rn(h, 2.f));
    x1 = x1 < 0.f ? 0.f : x1;
    y1 = y1 < 0.f ? 0.f : y1;
    x2 = x2 > wmax ? wmax : x2;
    y2 = y2 > hmax ? hmax : y2;
    const int slot = atomicAdd(count + n, 1);
    if (slot < cap) {
        DetCand d;
        d.score = best; d.anchor = a; d.cls = bi; d.pad = 0; d.x1 = x1; d.y1 = y1; d.x2 = x2; d.y2 = y2;
        cand[(long)n * cap + slot] = d;
    }
    const float m = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
    atomicMax(maxkey + n, float_order_key(m));
}

// rank by counting: position of candidate i in the stable descending-score order = #{j : s_j > s_i or (s_j == s_i and anchor_j < anchor_i)};
// scatter the candidate there and write its class-offset box (box + cls * (max_coord + 1)) for the suppression kernel
__global__ __launch_bounds__(256) void det_rank_kernel(const DetCand* cand, int cap, const int* count, const int* maxkey, DetCand* sorted,
                                                       float4* sboxes) {
    __shared__ float ss[256];
    __shared__ int sa[256];
    const int n = blockIdx.y;
    int cnt = count[n];
    if (cnt > cap) cnt = cap;
    if (blockIdx.x * 256 >= cnt) return;
    const DetCand* cn = cand + (long)n * cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    DetCand me;
    if (i < cnt) me = cn[i];
    int rank = 0;
    for (int j0 = 0; j0 < cnt; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        if (j < cnt) { ss[threadIdx.x] = cn[j].score; sa[threadIdx.x] = cn[j].anchor; }
        __syncthreads();
        const int lim = cnt - j0 < 256 ? cnt - j0 : 256;
        if (i < cnt)
            for (int t = 0; t < lim; ++t) rank += (ss[t] > me.score || (ss[t] == me.score && sa[t] < me.anchor)) ? 1 : 0;
    }
    if (i < cnt) {
        sorted[(long)n * cap + rank] = me;
        const float off = __fmul_rn((float)me.cls, __fadd_rn(float_from_key(maxkey[n]), 1.0f));
        sboxes[(long)n * cap + rank] = make_float4(__fadd_rn(me.x1, off), __fadd_rn(me.y1, off), __fadd_rn(me.x2, off), __fadd_rn(me.y2, off));
    }
}

__device__ __forceinline__ float iou_rn4(const float4 a, const float4 b) {
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    float iw = __fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x));
    float ih = __fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y));
    iw = iw > 0.f ? iw : 0.f;
    ih = ih > 0.f ? ih : 0.f;
    const float inter = __fmul_rn(iw, ih);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
}

// suppression bit mask of image n: bit (i, j) = (j > i) and IoU(i, j) > thr.  grid (word columns, row chunks of 64, image)
__global__ __launch_bounds__(64) void det_mask_kernel(const float4* sboxes, int cap, const int* count, float thr, unsigned long long* mask,
                                                      int words_cap) {
    const int n = blockIdx.z;
    int cnt = count[n];
    if (cnt > cap) cnt = cap;
    const int wj = blockIdx.x, i0 = blockIdx.y * 64;
    if (i0 >= cnt || wj * 64 >= cnt) return;
    const float4* b = sboxes + (long)n * cap;
    const int j = wj * 64 + threadIdx.x;
    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < cnt) bj = b[j];
    const int i1 = i0 + 64 < cnt ? i0 + 64 : cnt;
    for (int i = i0; i < i1; ++i) {
        bool sup = false;
        if (j < cnt && j > i) sup = iou_rn4(b[i], bj) > thr;
        const unsigned long long bits = __ballot(sup);
        if (threadIdx.x == 0) mask[((long)n * cap + i) * words_cap + wj] = bits;
    }
}

// one wave per image walks the boxes in order (greedy NMS) and writes the kept candidates compactly, in descending-score order
__global__ __launch_bounds__(64) void det_scan_kernel(const unsigned long long* mask, int cap, int words_cap, const int* count, const DetCand* sorted,
                                                      float* rois, long* class_ids, float* scores, int* kept) {
    constexpr int NW = 8;                                  // up to 64 * 64 * 8 = 32768 candidates per image
    const int n = blockIdx.x;
    int cnt = count[n];
    if (cnt > cap) cnt = cap;
    const int words = (cnt + 63) >> 6;
    unsigned long long removed[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) removed[w] = 0ull;
    const int lane = threadIdx.x;
    int nk = 0;
    for (int i = 0; i < cnt; ++i) {
        const int wi = i >> 6;
        unsigned long long wv = 0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) if ((wi >> 6) == w) wv = removed[w];
        const unsigned long long word = __shfl(wv, wi & 63);
        const bool alive = !((word >> (i & 63)) & 1ull);
        if (alive) {
            if (lane == 0) {
                const DetCand d = sorted[(long)n * cap + i];
                float* r = rois + ((long)n * cap + nk) * 4;
                r[0] = d.x1; r[1] = d.y1; r[2] = d.x2; r[3] = d.y2;
                class_ids[(long)n * cap + nk] = d.cls;
                scores[(long)n * cap + nk] = d.score;
            }
            ++nk;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int ww = w * 64 + lane;
                if (ww < words) removed[w] |= mask[((long)n * cap + i) * words_cap + ww];
            }
        }
    }
    if (lane == 0) kept[n] = nk;
}

__global__ void zero_i32_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" long hn_det_post_ws_bytes(int N, int cap) {
    const long words = (cap + 63) / 64;
    return (long)N * cap * (2 * sizeof(DetCand) + sizeof(float4)) + (long)N * cap * words * 8 + 2L * N * sizeof(int) + 256;
}

/* anchors fp32 [A][4] (y1,x1,y2,x2), regression [N][A][4], classification [N][A][K] (post-sigmoid).  cap <= 32768 = capacity per image.
 * Outputs, per image n: kept[n] boxes in descending-score order: rois [N][cap][4] (x1,y1,x2,y2), class_ids int64 [N][cap], scores [N][cap];
 * total[n] = number of anchors over the threshold (> cap means the image overflowed the capacity: the caller raises).
 * ws: hn_det_post_ws_bytes(N, cap) bytes. */
extern "C" int hn_det_postprocess(const float* anchors, const float* regression, const float* classification, int N, int A, int K, int img_h,
                                  int img_w, float threshold, float iou_threshold, int cap, void* ws, float* rois, long* class_ids,
                                  float* scores, int* kept, int* total, hipStream_t st) {
    HN_CHECK_ARG(anchors && regression && classification && ws && rois && class_ids && scores && kept && total);
    HN_CHECK_ARG(N > 0 && A > 0 && K > 0 && cap > 0 && cap <= 32768);
    char* w = (char*)ws;
    DetCand* cand = (DetCand*)w; w += (long)N * cap * sizeof(DetCand);
    DetCand* sorted = (DetCand*)w; w += (long)N * cap * sizeof(DetCand);
    float4* sboxes = (float4*)w; w += (long)N * cap * sizeof(float4);
    const int words = (cap + 63) / 64;
    unsigned long long* mask = (unsigned long long*)w; w += (long)N * cap * words * 8;
    int* maxkey = (int*)w;
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, total, N, 0);
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, maxkey, N, (int)0x80000000);
    hipLaunchKernelGGL(det_select_kernel, dim3(cdiv(A, 256), N), dim3(256), 0, st, anchors, regression, classification, A, K, threshold,
                       (float)(img_w - 1), (float)(img_h - 1), cand, cap, total, maxkey);
    hipLaunchKernelGGL(det_rank_kernel, dim3(cdiv(cap, 256), N), dim3(256), 0, st, (const DetCand*)cand, cap, (const int*)total,
                       (const int*)maxkey, sorted, sboxes);
    hipLaunchKernelGGL(det_mask_kernel, dim3(words, cdiv(cap, 64), N), dim3(64), 0, st, (const float4*)sboxes, cap, (const int*)total,
                       iou_threshold, mask, words);
    hipLaunchKernelGGL(det_scan_kernel, dim3(N), dim3(64), 0, st, (const unsigned long long*)mask, cap, words, (const int*)total,
                       (const DetCand*)sorted, rois, class_ids, scores, kept);
    HN_LAUNCH_CHECK();
}

// =====================================================================================================================================
// 2. lane decode + lane NMS: one workgroup (1024 threads) per image; the anchors (any count: 512x1024 has 512, the 1152x1920 deploy
//    resolution 2160) are walked with a workgroup stride, their bookkeeping lives in dynamic LDS (21 bytes per anchor)
// =====================================================================================================================================
struct LaneGeo {
    int fw, fh, stride, ppl, W, H, L;
    float interval;          // input_height / points_per_line
    double ppa;              // points_per_line / feature_height
    float margin;
};

__global__ __launch_bounds__(1024) void lane_decode_nms_kernel(const float* cls, const float* loc, LaneGeo g, float exist_thr, float nms_thr,
                                                               int use_mean, float* X, float* prob_out, int* start_out, int* end_out,
                                                               int* order_out, int* keep_out, int* counts) {
    extern __shared__ __attribute__((aligned(16))) char lane_smem[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int hw = g.fw * g.fh;
    float* s_prob = reinterpret_cast<float*>(lane_smem);
    int* s_valid = reinterpret_cast<int*>(s_prob + hw);
    int* s_order = s_valid + hw;
    int* s_start = s_order + hw;
    int* s_end = s_start + hw;
    int* s_cntp = s_end + hw;                                       // (no static LDS beside the dynamic array: the opt-in above 64 KiB
    unsigned char* s_sup = reinterpret_cast<unsigned char*>(s_cntp + 4);   //  asks for the whole 160 KiB as dynamic memory)
    float* Xn = X + (long)n * hw * g.ppl;
    if (tid == 0) *s_cntp = 0;
    for (int a = tid; a < hw; a += 1024) {
        float prob;
        int start = 0, end = 0, valid = 0;
        const float l0 = cls[((long)n * hw + a) * 2], l1 = cls[((long)n * hw + a) * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        prob = __fdiv_rn(e1, __fadd_rn(e0, e1));
        if (!(prob < exist_thr)) {
            const int h = a / g.fw, w = a - h * g.fw;
            const int ypos = (int)((double)(g.fh - 1 - h) * g.ppa);
            const float cx = (float)((1.0 * w + 0.5) * g.stride);
            const float* row = loc + ((long)n * hw + a) * g.L;
            const float rel_down = row[g.ppl], rel_up = row[g.ppl + 1];
            end = start = ypos;
            float* xr = Xn + (long)a * g.ppl;
            for (int i = 0; i < g.ppl; ++i) {                       // up anchor: positions ypos, ypos+1, ...
                if ((float)i >= rel_up || ypos + i >= g.ppl) break;
                const float ax = __fadd_rn(cx, __fmul_rn(row[g.ppl + 2 + i], g.interval));
                if (ax < 0.f || ax >= (float)g.W) break;
                xr[ypos + i] = ax;
                end = ypos + i + 1;
            }
            for (int i = 0; i < ypos; ++i) {                        // down anchor: positions ypos-1, ypos-2, ...
                if ((float)i >= rel_down || ypos - 1 - i < 0) break;
                const float ax = __fadd_rn(cx, __fmul_rn(row[i], g.interval));
                if (ax < 0.f || ax >= (float)g.W + g.margin) break;
                xr[ypos - 1 - i] = ax;
                start = ypos - 1 - i;
            }
            valid = (end - start) >= 2 ? 1 : 0;
        }
        s_prob[a] = prob; s_valid[a] = valid; s_start[a] = start; s_end[a] = end;
        s_sup[a] = 0;
        prob_out[(long)n * hw + a] = prob;
        start_out[(long)n * hw + a] = start;
        end_out[(long)n * hw + a] = end;
    }
    __threadfence_block();                                          // this workgroup's X rows are read back below by other threads
    __syncthreads();
    // stable descending-prob order of the valid anchors (Python's sorted() on Lane.__lt__ = prob > other.prob keeps raster order on ties)
    for (int a = tid; a < hw; a += 1024) {
        if (!s_valid[a]) continue;
        const float prob = s_prob[a];
        int rank = 0;
        for (int b = 0; b < hw; ++b)
            rank += (s_valid[b] && (s_prob[b] > prob || (s_prob[b] == prob && b < a))) ? 1 : 0;
        s_order[rank] = a;
        atomicAdd(s_cntp, 1);
    }
    __syncthreads();
    const int cnt = *s_cntp;
    // greedy suppression in that order: candidate k (if still alive) suppresses every later candidate t whose distance is <= thr
    for (int k = 0; k < cnt; ++k) {
        if (!s_sup[k]) {                                            // uniform: s_sup[k] was settled by earlier iterations
            const int la = s_order[k];
            const float* xa = Xn + (long)la * g.ppl;
            for (int t = k + 1 + tid; t < cnt; t += 1024) {
                const int lb = s_order[t];
                const int lo = s_start[la] > s_start[lb] ? s_start[la] : s_start[lb];
                const int hi = s_end[la] < s_end[lb] ? s_end[la] : s_end[lb];
                if (hi <= lo || lo < 0 || hi < 1) continue;
                const float* xb = Xn + (long)lb * g.ppl;
                float dis = 0.f;
                for (int i = lo; i < hi; ++i) dis = __fadd_rn(dis, fabsf(__fsub_rn(xa[i], xb[i])));
                dis = __fdiv_rn(dis, (float)(hi - lo));
                if (!use_mean) {
                    dis = fmaxf(dis, fabsf(__fsub_rn(xa[lo], xb[lo])));
                    dis = fmaxf(dis, fabsf(__fsub_rn(xa[hi - 1], xb[hi - 1])));
                }
                if (dis <= nms_thr) s_sup[t] = 1;
            }
        }
        __syncthreads();
    }
    for (int a = tid; a < cnt; a += 1024) {
        order_out[(long)n * hw + a] = s_order[a];
        keep_out[(long)n * hw + a] = s_sup[a] ? 0 : 1;
    }
    if (tid == 0) counts[n] = cnt;
}

/* predict_cls fp32 [N][hw][2] (logits), predict_loc fp32 [N][hw][L = 2*ppl+2]; hw = (W/stride)*(H/stride) <= 7168 (21 bytes of LDS per
 * anchor).  Outputs: X [N][hw][ppl] = x coordinate of anchor a at position p (valid for start[a] <= p < end[a]), prob / start / end [N][hw]
 * per anchor, order [N][hw] = candidate anchors in descending-prob order (counts[n] entries), keep [N][hw] = 1 where the candidate
 * survives the NMS. */
extern "C" int hn_lane_decode_nms(const float* predict_cls, const float* predict_loc, int N, int W, int H, int stride, int ppl,
                                  float exist_threshold, float nms_threshold, int use_mean, float margin, float* X, float* prob, int* start,
                                  int* end, int* order, int* keep, int* counts, hipStream_t st) {
    HN_CHECK_ARG(predict_cls && predict_loc && X && prob && start && end && order && keep && counts && N > 0 && stride > 0 && ppl > 0);
    LaneGeo g;
    g.fw = W / stride; g.fh = H / stride; g.stride = stride; g.ppl = ppl; g.W = W; g.H = H; g.L = 2 * ppl + 2;
    g.interval = (float)((double)H / ppl);
    g.ppa = (double)ppl / g.fh;
    g.margin = margin;
    const long hw = (long)g.fw * g.fh;
    HN_CHECK_ARG(hw > 0);
    if (hw > 7168) return HN_ERR_UNSUPPORTED;
    const size_t lds = (size_t)hw * 21 + 32;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> optin{0};
        if (!lds_optin(optin, {(const void*)lane_decode_nms_kernel})) return HN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(lane_decode_nms_kernel, dim3(N), dim3(1024), lds, st, predict_cls, predict_loc, g, exist_threshold, nms_threshold, use_mean,
                       X, prob, start, end, order, keep, counts);
    HN_LAUNCH_CHECK();
}

// =====================================================================================================================================
// 3. pre-processing: uint8 BGR HWC frame(s) -> fp32 RGB NCHW, bilinear resize in cv2's fixed-point form (INTER_LINEAR on 8-bit images:
//    11-bit coefficients, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2), then (v/255 - mean)/std evaluated in double like numpy.
// =====================================================================================================================================
__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int& s0, short& c0, short& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    s0 = s;
    const float a0 = (1.f - f) * 2048.f, a1 = f * 2048.f;
    c0 = (short)__float2int_rn(a0);
    c1 = (short)__float2int_rn(a1);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* src, int Hs, int Ws, float* dst, int Hd, int Wd, int N) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * Hd * Wd;
    if (idx >= total) return;
    const int x = (int)(idx % Wd);
    const long t = idx / Wd;
    const int y = (int)(t % Hd);
    const int n = (int)(t / Hd);
    const unsigned char* im = src + (long)n * Hs * Ws * 3;
    int v[3];
    if (Hs == Hd && Ws == Wd) {
        const unsigned char* p = im + ((long)y * Ws + x) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
        int sx, sy;
        short ax0, ax1, by0, by1;
        lin_coef(x, (double)Ws / Wd, Ws, sx, ax0, ax1);
        lin_coef(y, (double)Hs / Hd, Hs, sy, by0, by1);
        const int sx1 = sx + 1 < Ws ? sx + 1 : Ws - 1, sy1 = sy + 1 < Hs ? sy + 1 : Hs - 1;
        const unsigned char* r0 = im + (long)sy * Ws * 3;
        const unsigned char* r1 = im + (long)sy1 * Ws * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;       // horizontal pass, 11-bit fixed point
            const int h1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
            v[c] = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v[c] = v[c] < 0 ? 0 : (v[c] > 255 ? 255 : v[c]);
        }
    }
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    const long plane = (long)Hd * Wd;
    float* o = dst + (long)n * 3 * plane + (long)y * Wd + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                                // output channel c = R,G,B = source channel 2-c (BGR)
        const double q = ((double)v[2 - c] / 255.0 - mean[c]) / sd[c];
        o[c * plane] = (float)q;
    }
}

/* src: uint8 [N][Hs][Ws][3] BGR (cv2.imread layout); dst: fp32 [N][3][Hd][Wd] RGB, ImageNet-normalised (demo.py:186-196) */
extern "C" int hn_preprocess_bgr(const void* src, int N, int Hs, int Ws, float* dst, int Hd, int Wd, hipStream_t st) {
    HN_CHECK_ARG(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0);
    const long total = (long)N * Hd * Wd;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const unsigned char*)src, Hs, Ws, dst, Hd, Wd, N);
    HN_LAUNCH_CHECK();
}

// =====================================================================================================================================
// 4. segmentation confusion counts (streaming mIoU): conf[(p * (C+1)) + t] += 1 with p, t clamped to C (the ignore bucket)
// =====================================================================================================================================
__global__ __launch_bounds__(256) void seg_confusion_kernel(const long* pred, const void* target, int target_is_float, long M, int C,
                                                            unsigned long long* conf) {
    extern __shared__ unsigned int hist[];                // (C+1)^2
    const int nb = (C + 1) * (C + 1);
    for (int i = threadIdx.x; i < nb; i += 256) hist[i] = 0;
    __syncthreads();
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        long p = pred[m];
        long t = target_is_float ? (long)((const float*)target)[m] : ((const long*)target)[m];
        p = p > C ? C : (p < 0 ? 0 : p);
        t = t > C ? C : (t < 0 ? 0 : t);
        atomicAdd(&hist[p * (C + 1) + t], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
        if (hist[i]) atomicAdd(conf + i, (unsigned long long)hist[i]);
}

/* pred: int64 class ids [M] (deploy-mode arg-max); target: int64 or float32 class ids [M]; conf: uint64 [(C+1)*(C+1)], ACCUMULATED
 * (zero it once, call per batch): conf[p*(C+1)+t].  Integer atomics: exact and order-independent. */
extern "C" int hn_seg_confusion(const long* pred, const void* target, int target_is_float, long M, int C, void* conf, hipStream_t st) {
    HN_CHECK_ARG(pred && target && conf && M > 0 && C > 0 && C <= 63);
    long blocks = (M + 256 * 16 - 1) / (256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(seg_confusion_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(C + 1) * (C + 1) * 4, st, pred, target, target_is_float,
                       M, C, (unsigned long long*)conf);
    HN_LAUNCH_CHECK();
}

// =====================================================================================================================================
// 5. segmentation overlay (SegmentHeader.decode, head_seg/segmentation.py:107-125; C++ twin deploy/src/model/hydranet_model.cpp:758):
//    class ids -> colour LUT -> 8-bit bilinear resize to the frame size -> saturating blend 0.8 * frame + 0.5 * colours.
//    The reference calls cv2.resize(vis_seg, org_size, cv2.INTER_NEAREST): the third POSITIONAL parameter of cv2.resize is `dst`, so the
//    flag never reaches `interpolation` and the resize runs with the default INTER_LINEAR -- restated here in cv2's fixed-point form
//    (lin_coef above, as in the pre-processing kernel).  cv2.addWeighted on 8-bit images: float32 arithmetic, round half to even,
//    saturate.  cv2 is absent from this image: parity of both steps is UNPINNED (checked against the oracle's restatement only).
//    One thread per output pixel; the colour image at network resolution is never materialised (the four neighbours' colours are looked
//    up from their class ids).
// =====================================================================================================================================
__global__ __launch_bounds__(256) void seg_overlay_kernel(const long* mask, int H, int W, const unsigned char* lut, int ncls,
                                                          const unsigned char* frames, unsigned char* out, int Ho, int Wo, int N) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * Ho * Wo;
    if (idx >= total) return;
    const int x = (int)(idx % Wo);
    const long t = idx / Wo;
    const int y = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const long* m = mask + (long)n * H * W;
    auto colour = [&](int yy, int xx, int c) -> int {
        const long k = m[(long)yy * W + xx];
        return (k >= 0 && k < ncls) ? (int)lut[k * 3 + c] : 0;
    };
    int v[3];
    if (H == Ho && W == Wo) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = colour(y, x, c);
    } else {
        int sx, sy;
        short ax0, ax1, by0, by1;
        lin_coef(x, (double)W / Wo, W, sx, ax0, ax1);
        lin_coef(y, (double)H / Ho, H, sy, by0, by1);
        const int sx1 = sx + 1 < W ? sx + 1 : W - 1, sy1 = sy + 1 < H ? sy + 1 : H - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = colour(sy, sx, c) * ax0 + colour(sy, sx1, c) * ax1;
            const int h1 = colour(sy1, sx, c) * ax0 + colour(sy1, sx1, c) * ax1;
            int r = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
    }
    const unsigned char* f = frames + idx * 3;
    unsigned char* o = out + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float s = __fadd_rn(__fmul_rn((float)f[c], 0.8f), __fmul_rn((float)v[c], 0.5f));
        int r = __float2int_rn(s);                                              // cvRound: round half to even
        o[c] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

/* mask: int64 class ids [N][H][W] (arg-max of the seg logits); lut: uint8 [ncls][3] colours in the frame's channel order (ids without an
 * entry, and ids outside [0, ncls), stay black as in the reference's zero-initialised vis_seg); frames / out: uint8 [N][Ho][Wo][3]. */
extern "C" int hn_seg_overlay(const long* mask, int N, int H, int W, const void* lut, int ncls, const void* frames, void* out, int Ho, int Wo,
                              hipStream_t st) {
    HN_CHECK_ARG(mask && lut && frames && out && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && ncls > 0);
    const long total = (long)N * Ho * Wo;
    hipLaunchKernelGGL(seg_overlay_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, mask, H, W, (const unsigned char*)lut, ncls,
                       (const unsigned char*)frames, (unsigned char*)out, Ho, Wo, N);
    HN_LAUNCH_CHECK();
}

// =====================================================================================================================================
// 6. lane F1 (head_lane/lane_metric.py:166-266): bitwise IoU of two lanes drawn as thick polylines.  The reference rasterises with
//    cv2.line(img, p_i, p_{i+1}, 255, lane_width) -- OpenCV's ThickLine: the quadrilateral of half-width lane_width / 2 around the segment
//    plus filled circles of radius round(lane_width / 2) at both end points = the pixels within lane_width / 2 of the segment.  cv2 is
//    absent: restated as that distance test on pixel centres (parity with OpenCV's polygon / circle fill at the boundary pixels: UNPINNED).
//    One workgroup per polyline segment paints its bounding box into the lane's uint8 mask; a second kernel counts |A|, |B|, |A & B| for
//    every (ground truth, prediction) pair with integer atomics (exact, order independent).
// =====================================================================================================================================
__global__ __launch_bounds__(256) void lane_raster_kernel(const int* pts, const int* seg_lane, const int* seg_first, int nseg, int width2,
                                                          int H, int W, unsigned char* masks) {
    const int s = blockIdx.x;
    if (s >= nseg) return;
    const int lane = seg_lane[s], i = seg_first[s];
    const int x0 = pts[2 * i], y0 = pts[2 * i + 1], x1 = pts[2 * i + 2], y1 = pts[2 * i + 3];
    const int r = (width2 + 3) / 4 + 1;                                  // width2 = 2 * lane_width: radius lane_width / 2, rounded up, + 1
    const int bx0 = max(min(x0, x1) - r, 0), bx1 = min(max(x0, x1) + r, W - 1);
    const int by0 = max(min(y0, y1) - r, 0), by1 = min(max(y0, y1) + r, H - 1);
    if (bx1 < bx0 || by1 < by0) return;
    const int bw = bx1 - bx0 + 1, n = bw * (by1 - by0 + 1);
    const long dx = x1 - x0, dy = y1 - y0, len2 = dx * dx + dy * dy;
    unsigned char* m = masks + (long)lane * H * W;
    for (int k = threadIdx.x; k < n; k += 256) {
        const int x = bx0 + k % bw, y = by0 + k / bw;
        const long px = x - x0, py = y - y0;
        // squared distance from (x, y) to the segment, scaled by len2 (integers: exact).  t = clamp(dot / len2, 0, 1)
        long num;                                                       // distance^2 * len2  (or distance^2 when len2 == 0)
        long den;
        const long dot = px * dx + py * dy;
        if (len2 == 0 || dot <= 0) { num = px * px + py * py; den = 1; }
        else if (dot >= len2) { const long qx = x - x1, qy = y - y1; num = qx * qx + qy * qy; den = 1; }
        else { const long cr = px * dy - py * dx; num = cr * cr; den = len2; }
        // inside: distance <= lane_width / 2  <=>  4 * distance^2 <= lane_width^2  (width2 = 2 * lane_width -> (width2 / 2)^2)
        if (16 * num <= (long)width2 * width2 * den) m[(long)y * W + x] = 255;
    }
}

__global__ __launch_bounds__(256) void lane_iou_kernel(const unsigned char* masks, int G, int P, long HW, unsigned long long* inter,
                                                       unsigned long long* area) {
    extern __shared__ unsigned int cnt[];                                // [G * P + G + P]
    const int nc = G * P + G + P;
    for (int i = threadIdx.x; i < nc; i += 256) cnt[i] = 0;
    __syncthreads();
    for (long px = (long)blockIdx.x * 256 + threadIdx.x; px < HW; px += (long)gridDim.x * 256) {
        unsigned int gbits = 0, pbits = 0;                               // (G, P <= 32)
        for (int g = 0; g < G; ++g) gbits |= (masks[(long)g * HW + px] ? 1u : 0u) << g;
        for (int p = 0; p < P; ++p) pbits |= (masks[(long)(G + p) * HW + px] ? 1u : 0u) << p;
        if (!(gbits | pbits)) continue;
        for (int g = 0; g < G; ++g)
            if ((gbits >> g) & 1u) {
                atomicAdd(&cnt[G * P + g], 1u);
                for (int p = 0; p < P; ++p)
                    if ((pbits >> p) & 1u) atomicAdd(&cnt[g * P + p], 1u);
            }
        for (int p = 0; p < P; ++p)
            if ((pbits >> p) & 1u) atomicAdd(&cnt[G * P + G + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += 256)
        if (cnt[i]) atomicAdd(i < G * P ? inter + i : area + (i - G * P), (unsigned long long)cnt[i]);
}

/* pts: int32 [npts][2] = the (x, y) pixel coordinates of every lane's interpolated polyline, lanes back to back (already truncated with
 * int() as lane_metric.py:198-203 does); seg_lane / seg_first: int32 [nseg] = the lane index and the index of the first point of every
 * segment; masks: uint8 [n_lanes][H][W], ZEROED by the caller.  lane_width = the reference's `lane_width` (30). */
extern "C" int hn_lane_raster(const int* pts, const int* seg_lane, const int* seg_first, int nseg, int lane_width, int H, int W, void* masks,
                              hipStream_t st) {
    HN_CHECK_ARG(pts && seg_lane && seg_first && masks && nseg >= 0 && lane_width > 0 && H > 0 && W > 0);
    if (nseg == 0) return HN_OK;
    hipLaunchKernelGGL(lane_raster_kernel, dim3((unsigned)nseg), dim3(256), 0, st, pts, seg_lane, seg_first, nseg, 2 * lane_width, H, W,
                       (unsigned char*)masks);
    HN_LAUNCH_CHECK();
}

/* masks: uint8 [G + P][HW] (ground-truth lanes first); inter: uint64 [G][P], area: uint64 [G + P], both ZEROED by the caller: pixel counts of
 * mask_g & mask_p and of every mask.  IoU(g, p) = inter / (area_g + area_p - inter) (lane_metric.py:204-209).  G, P <= 32. */
extern "C" int hn_lane_iou(const void* masks, int G, int P, long HW, void* inter, void* area, hipStream_t st) {
    HN_CHECK_ARG(masks && inter && area && G > 0 && P > 0 && G <= 32 && P <= 32 && HW > 0);
    long blocks = (HW + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lane_iou_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(G * P + G + P) * 4, st, (const unsigned char*)masks, G, P, HW,
                       (unsigned long long*)inter, (unsigned long long*)area);
    HN_LAUNCH_CHECK();
}
