// ground-truth probe for ds_read_b64_tr_b16 lane semantics on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) short t[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) t[i] = (short)((i / pitch) * 100 + (i % pitch));   // T[r][c] = r*100 + c
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, t16 = lane & 15, q = t16 >> 2, pp = t16 & 3;
    // lane 4q+pp of group g supplies &T[g*4 + q][pp*4]
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(t + (g * 4 + q) * pitch + pp * 4));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int pitch : {16, 72}) {
        probe<<<1, 64>>>(d, pitch);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pitch %d\n", pitch);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
