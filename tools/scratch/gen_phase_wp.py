s=open('/root/repo/multitask_hydranet_amd/csrc/hn_gemm.hip').read()
s=s.replace('#include "hn_common.h"','#include "../../multitask_hydranet_amd/csrc/hn_common.h"\n__device__ unsigned long long g_dbg[32];\n#define STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("" ::: "memory"); g_dbg[i] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); }',1)
k=s.index("__global__ __launch_bounds__(512) void wgrad3x3_patch_kernel")
e=s.index("// dW[co][ci][tap] (PyTorch [Cout][Cin][kh][kw] order) = sum_split")
body=s[k:e]
def rep(old,new):
    global body
    assert old in body, old
    body=body.replace(old,new,1)
rep("    constexpr int WCO = BC >= 64 ? 64 : BC, TC = WCO / 16;","    STAMP(0)\n    constexpr int WCO = BC >= 64 ? 64 : BC, TC = WCO / 16;")
rep("    for (int it = 0; it <= S; ++it) {\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __syncthreads();","    STAMP(1)\n    for (int it = 0; it <= S; ++it) {\n        if (it < 5) STAMP(8 + 3 * it)\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __syncthreads();\n        if (it < 5) STAMP(9 + 3 * it)")
rep("        if (it > 0) {\n            const char* sZ = smem + ((it - 1) & 1) * STAGE;","        if (it < 5) STAMP(10 + 3 * it)\n        if (it > 0) {\n            const char* sZ = smem + ((it - 1) & 1) * STAGE;")
rep("    const int Ktot = 9 * p.KP;\n    float* part = p.part + ((long)bz * KSPLIT + wk)","    STAMP(2)\n    const int Ktot = 9 * p.KP;\n    float* part = p.part + ((long)bz * KSPLIT + wk)")
i=body.rindex("}\n\n")
body=body[:i]+"    STAMP(3)\n}\n\n"+body[i+3:]
s=s[:k]+body+s[e:]
s+='''
#include <cstdio>
int main(int argc, char** argv) {
    int N = 16, H = 64, W = 128, C0 = 256, C1 = 112, Cout = 256;           // seg decoder.3 weight gradient
    int Cin = C0 + C1, KP = (Cin + 31) / 32 * 32;
    long M = (long)N * H * W;
    void *x0, *x1, *dz; float *ws, *dw;
    hipMalloc(&x0, (size_t)N * (H / 2) * (W / 2) * C0 * 2); hipMalloc(&x1, (size_t)M * C1 * 2); hipMalloc(&dz, (size_t)M * Cout * 2);
    hipMemset(x0, 0, (size_t)N * (H / 2) * (W / 2) * C0 * 2); hipMemset(x1, 0, (size_t)M * C1 * 2); hipMemset(dz, 0, (size_t)M * Cout * 2);
    int splits; long rps, wsb;
    hn_wgrad_plan(2, N, H, W, M, Cout, KP, 9, &splits, &rps, &wsb);
    hipMalloc(&ws, wsb); hipMalloc(&dw, (size_t)Cout * Cin * 9 * 4);
    printf("splits %d patches/split %ld ws %.1f MB\\n", splits, rps, wsb / 1e6);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, st);
        int rc = hn_conv_gemm_tn(x0, x1, 2, N, H, W, C0, C1, C0, C1, 1, M, dz, Cout, Cout, KP, 9, ws, dw, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long d[32]; hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d));
        auto us = [&](int a, int b) { return (double)(long long)(d[b] - d[a]) * 0.01; };
        printf("rc %d event(wgrad+reduce) %.1f us | setup %.2f loop %.2f epilogue %.2f | it1: wait %.2f issue %.2f compute %.2f | it2: wait %.2f issue %.2f compute %.2f | it3: wait %.2f issue %.2f compute %.2f\\n", rc, ms * 1e3,
               us(0, 1), us(1, 2), us(2, 3), us(11, 12), us(12, 13), us(13, 14), us(14, 15), us(15, 16), us(16, 17), us(17, 18), us(18, 19), us(19, 20));
    }
    return 0;
}
'''
open('wp_phase.hip','w').write(s)
