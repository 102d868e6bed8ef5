// What does it cost a workgroup to stream its image's rows (written a moment ago by the other workgroups of the same XCD) from the
// XCD's L2 into LDS?  The question behind the GEMM phases of csrc/hn_xstage.hip: 30 workgroups per XCD (2 images x 15 channel slices)
// each read the whole [128][936] bf16 image (240 KB) for every 1x1 conv.
// Per round: every workgroup writes its [128 rows][64 channels] slice (plain stores, whole 16-byte pieces), XCD-local arrive / poll, then
// streams the image through an LDS ring (LDS-DMA, 16-byte pieces) with `DEPTH` stages of `KS` channels in flight; the elapsed real-time
// ticks of the streaming part are averaged.  Variants: row stride 936 (slices straddle 128-byte lines) or 960 (whole lines).
// hipcc --offload-arch=gfx950 -O3 -o feed_probe feed_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }

template <int KS, int DEPTH, int RING, int SWZ = 0>
__global__ __launch_bounds__(512) void feed_kernel(unsigned short* buf, int ld, int rounds, unsigned* ctr, unsigned long long* ticks, int do_mfma, int fresh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int info[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        const unsigned x = xcc_id();
        info[0] = (int)x;
        info[1] = (int)atomicAdd(ctr + 64 + 32 * x, 1u);
    }
    __syncthreads();
    const int xcc = info[0], t = info[1];
    if (t >= 30) return;
    const int img = xcc * 2 + t / 15, slice = t % 15;
    constexpr int STAGE = 128 * KS * 2;                 // bytes per stage: 128 rows x KS channels
    constexpr int NI = STAGE / 1024 / 8;                // LDS-DMA instructions per wave and stage
    constexpr int PPR = KS / 8;                         // 16-byte pieces per row
    const int S = 960 / KS;
    gu32* cnt = (gu32*)(ctr + 1024 + img * 32);
    unsigned long long total = 0;
    float sink = 0.f;
    for (int r = 0; r < rounds; ++r) {
        unsigned short* B = buf + (size_t)(fresh == 1 ? r : (r & 1)) * 16 * 128 * ld;
        if (fresh == 2) {                               // pre-touch: read the lines this workgroup is about to write (fresh buffer per round)
            B = buf + (size_t)r * 16 * 128 * ld;
            const int nch = slice == 14 ? 40 : 64;
            unsigned acc = 0;
            for (int idx = tid; idx < 128 * (nch / 8); idx += 512) {
                const int row = idx / (nch / 8), pc = idx % (nch / 8);
                acc += *reinterpret_cast<const unsigned*>(B + ((size_t)img * 128 + row) * ld + slice * 64 + pc * 8);
            }
            if (acc == 0x12345u) sink += 1.f;
        }         // two buffers: this round's data is never in a stale L1 line
        // write this workgroup's slice: 128 rows x 64 channels (40 in the last slice) = 8 pieces per row
        {
            const int nch = slice == 14 ? 40 : 64;
            for (int idx = tid; idx < 128 * (nch / 8); idx += 512) {
                const int row = idx / (nch / 8), pc = idx % (nch / 8);
                uint4 v = {(unsigned)r, (unsigned)idx, 1u, 2u};
                *reinterpret_cast<uint4*>(B + ((size_t)img * 128 + row) * ld + slice * 64 + pc * 8) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int n = 0; n < (1 << 22); ++n) {
                if ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 15u * (r + 1)) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned short* X = B + (size_t)img * 128 * ld;
        auto issue = [&](int s) {
            char* dst = smem + (s % RING) * STAGE;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int q = u * 8 + wave;                                   // 1 KB chunk of the stage
                const int row = q * (64 / PPR) + lane / PPR, pc = lane % PPR;
                const int k = s * KS + (SWZ ? (pc ^ (row & (PPR - 1))) : pc) * 8;
                const unsigned short* src = k < 936 ? X + (size_t)row * ld + k : X;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
            }
        };
        for (int s = 0; s < DEPTH && s < S; ++s) issue(s);
        for (int it = 0; it < S; ++it) {
            const int newer = (it + DEPTH - 1 < S ? it + DEPTH - 1 : S - 1) - it;
            // wait for stage `it`: `newer` stages may stay in flight
            const int w = newer * NI;
            switch (w < 15 ? w : 15) {
                case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __syncthreads();
            if (it + DEPTH < S) issue(it + DEPTH);
            if (do_mfma) sink += *reinterpret_cast<const float*>(smem + (it % RING) * STAGE + tid * 4);
        }
        __syncthreads();
        total += __builtin_amdgcn_s_memrealtime() - t0;
    }
    if (tid == 0) ticks[blockIdx.x] = total;
    if (sink == 123.456f) ticks[0] = 0;
}

template <int KS, int DEPTH, int RING, int SWZ = 0>
static void run(const char* name, int ld, int fresh = 0) {
    unsigned short* buf; unsigned* ctr; unsigned long long* ticks;
    const int rounds = 40;
    hipMalloc(&buf, (size_t)(fresh ? rounds : 2) * 16 * 128 * ld * 2);
    hipMemset(buf, 0, (size_t)(fresh ? rounds : 2) * 16 * 128 * ld * 2);
    hipMalloc(&ctr, 8192 * 4);
    hipMalloc(&ticks, 256 * 8);
    hipFuncSetAttribute((const void*)feed_kernel<KS, DEPTH, RING, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(ctr, 0, 8192 * 4);
        hipMemset(ticks, 0, 256 * 8);
        hipLaunchKernelGGL((feed_kernel<KS, DEPTH, RING, SWZ>), dim3(256), dim3(512), (size_t)RING * 128 * KS * 2, 0, buf, ld, rounds, ctr, ticks, 1, fresh);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
    std::vector<double> us;
    for (auto v : h) if (v) us.push_back(v / 100.0 / rounds);
    std::sort(us.begin(), us.end());
    if (us.empty()) { printf("%-44s no data (%s)\n", name, hipGetErrorString(hipGetLastError())); return; }
    printf("%-44s ld %4d  workgroups %3zu  stream of 240 KB: median %.2f us  min %.2f  max %.2f   (%.0f GB/s per CU)\n", name, ld, us.size(),
           us[us.size() / 2], us.front(), us.back(), 240e3 / us[us.size() / 2] / 1e3);
    hipFree(buf); hipFree(ctr); hipFree(ticks);
}

int main() {
    for (int ld : {936, 960}) {
        run<64, 1, 2, 1>("KS 64, reused buffers", ld, 0);
        run<64, 1, 2, 1>("KS 64, fresh buffer every round", ld, 1);
        run<64, 1, 2, 1>("KS 64, fresh buffer, lines read before written", ld, 2);
    }
    return 0;
}
