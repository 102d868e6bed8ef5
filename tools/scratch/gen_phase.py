s=open('/root/repo/multitask_hydranet_amd/csrc/hn_gemm.hip').read()
s=s.replace('#include "hn_common.h"','#include "../../multitask_hydranet_amd/csrc/hn_common.h"\n__device__ unsigned long long g_dbg[32];\n#define STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("" ::: "memory"); g_dbg[i] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); }',1)
k=s.index("__global__ __launch_bounds__(256) void gemm_nt_kernel")
e=s.index("// Direct 3x3 convolution (im2col-free)")
body=s[k:e]
def rep(old,new):
    global body
    assert old in body, old
    body=body.replace(old,new,1)
rep("    constexpr int WC = BC / WGC, WP = BP / WGP, TC = WC / 16, TP = WP / 16;","    STAMP(0)\n    constexpr int WC = BC / WGC, WP = BP / WGP, TC = WC / 16, TP = WP / 16;")
rep("    f32x4 acc[TC][TP];","    STAMP(1)\n    f32x4 acc[TC][TP];")
rep("        if (it < S) {\n            char* sW","        if (it < 6) STAMP(8 + 3 * it)\n        if (it < S) {\n            char* sW")
rep("        if (it >= R - 1) {\n            const char* sW","        if (it < 6) STAMP(9 + 3 * it)\n        if (it >= R - 1) {\n            const char* sW")
rep("    // ---- epilogue: bias, activation, optional BN partial statistics; store.","    STAMP(2)\n    // ---- epilogue: bias, activation, optional BN partial statistics; store.")
rep("    act_fwd_n(vv, p.act);","    STAMP(3)\n    act_fwd_n(vv, p.act);\n    STAMP(4)")
rep("    if (staged) {\n        __syncthreads();\n        bf16* outp","    STAMP(5)\n    if (staged) {\n        __syncthreads();\n        bf16* outp")
i=body.rindex("}\n")
body=body[:i]+"    STAMP(6)\n}\n"+body[i+2:]
s=s[:k]+body+s[e:]
s+='''
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 512, K = argc > 2 ? atoi(argv[2]) : 112, N = argc > 3 ? atoi(argv[3]) : 112;
    int bc = argc > 4 ? atoi(argv[4]) : 0, r = argc > 5 ? atoi(argv[5]) : 0;
    int KP = (K + 31) / 32 * 32;
    void *x, *w, *out; float *ps, *pq;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * KP * 2); hipMalloc(&out, (size_t)M * N * 2);
    hipMalloc(&ps, 4u << 20); hipMalloc(&pq, 4u << 20);
    hipMemset(x, 0, (size_t)M * K * 2); hipMemset(w, 0, (size_t)N * KP * 2);
    hn_debug_nt_config(bc, r);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, st);
        int rc = hn_conv_gemm_nt(x, nullptr, 0, 1, 1, M, K, 0, K, 0, 0, M, w, N, KP, 1, nullptr, 0, out, 0, N, 0, 0, ps, pq, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long d[32]; hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d));
        auto us = [&](int a, int b) { return (double)(long long)(d[b] - d[a]) * 0.01; };
        printf("rc %d event %.1f us | setup %.2f loop %.2f bias+stats %.2f act %.2f stage/store-regs %.2f store %.2f | it0: top->issue %.2f; it1 wait %.2f issue %.2f ; it2 wait %.2f issue %.2f\\n", rc, ms * 1e3,
               us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(8, 9), us(9, 11), us(11, 12), us(12, 14), us(14, 15));
    }
    return 0;
}
'''
open('nt_phase.hip','w').write(s)
