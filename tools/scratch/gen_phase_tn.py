s=open('/root/repo/multitask_hydranet_amd/csrc/hn_gemm.hip').read()
s=s.replace('#include "hn_common.h"','#include "../../multitask_hydranet_amd/csrc/hn_common.h"\n__device__ unsigned long long g_dbg[32];\n#define STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("" ::: "memory"); g_dbg[i] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); }',1)
k=s.index("__global__ __launch_bounds__(256) void gemm_tn_kernel")
e=s.index("// 3x3 weight gradient with patch reuse")
body=s[k:e]
def rep(old,new):
    global body
    assert old in body, old
    body=body.replace(old,new,1)
rep("    constexpr int ZL = ","    STAMP(0)\n    constexpr int ZL = ")
rep("    for (int it = 0; it <= S; ++it) {\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");","    STAMP(1)\n    for (int it = 0; it <= S; ++it) {\n        if (it < 6) STAMP(8 + 3 * it)\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");")
rep("        if (it < S) {\n            char* sZ = smem + (it & 1) * STAGE;","        if (it < 6) STAMP(9 + 3 * it)\n        if (it < S) {\n            char* sZ = smem + (it & 1) * STAGE;")
rep("        if (it > 0) {\n            const char* sZ = smem + ((it - 1) & 1) * STAGE;","        if (it < 6) STAMP(10 + 3 * it)\n        if (it > 0) {\n            const char* sZ = smem + ((it - 1) & 1) * STAGE;")
rep("    const int Ktot = p.taps * p.KP;\n    float* part = p.part","    STAMP(2)\n    const int Ktot = p.taps * p.KP;\n    float* part = p.part")
i=body.rindex("}\n\n")
body=body[:i]+"    STAMP(3)\n}\n\n"+body[i+3:]
s=s[:k]+body+s[e:]
s+='''
#include <cstdio>
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 2048, K = argc > 2 ? atoi(argv[2]) : 936, N = argc > 3 ? atoi(argv[3]) : 936;
    int bc = argc > 4 ? atoi(argv[4]) : 0, bn = argc > 5 ? atoi(argv[5]) : 0, sp = argc > 6 ? atoi(argv[6]) : 0;
    int KP = (K + 31) / 32 * 32;
    void *x, *dz; float *ws, *dw;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&dz, (size_t)M * N * 2); hipMalloc(&ws, 256u << 20); hipMalloc(&dw, (size_t)N * K * 4);
    hipMemset(x, 0, (size_t)M * K * 2); hipMemset(dz, 0, (size_t)M * N * 2);
    hn_debug_tn_config(bc, bn, sp);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, st);
        int rc = hn_conv_gemm_tn(x, nullptr, 0, 1, 1, M, K, 0, K, 0, 0, M, dz, N, N, KP, 1, ws, dw, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long d[32]; hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d));
        auto us = [&](int a, int b) { return (double)(long long)(d[b] - d[a]) * 0.01; };
        printf("rc %d event(TN+reduce) %.1f us | setup %.2f loop %.2f epilogue %.2f | it0: wait %.2f issue %.2f | it1: wait %.2f issue %.2f compute(+) %.2f | it2: wait %.2f issue %.2f compute %.2f\\n", rc, ms * 1e3,
               us(0, 1), us(1, 2), us(2, 3), us(8, 9), us(9, 10), us(11, 12), us(12, 13), us(13, 14), us(14, 15), us(15, 16), us(16, 17));
    }
    return 0;
}
'''
open('tn_phase.hip','w').write(s)
