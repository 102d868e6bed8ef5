// What would a persistent "one chain per XCD" kernel pay per synchronisation?  
// 256 workgroups (one per CU, all co-resident) run ITERS rounds of one of three barriers and the host divides the elapsed time:
//   mode 0  global barrier through memory: every workgroup bumps one device-scope counter and spins on it (sc1 atomics / loads)
//   mode 1  XCD-local barrier: the workgroups with the same (blockIdx.x & 7) -- dealt to the same XCD, sharing its L2 -- bump and poll a counter
//           of their own with workgroup-scope (L2-resident) atomics; no cross-XCD traffic
//   mode 2  what a BatchNorm statistic exchange needs: every workgroup publishes a 1 KB partial row (sc1 stores), global barrier, every
//           workgroup reads the 8 XCD-leader rows
// Every spin is bounded (a barrier that does not complete in 2^22 polls makes the kernel give up and report it): a hang is a lost box.
// hipcc --offload-arch=gfx950 -O3 -o xcd_sync_probe xcd_sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SPIN_LIMIT (1 << 22)

__device__ __forceinline__ bool spin_until(unsigned* ctr, unsigned target, bool device_scope) {
    for (int i = 0; i < SPIN_LIMIT; ++i) {
        // XCD-local polls are read-modify-writes of zero: an RMW always executes in the L2 (a workgroup-scope LOAD may be served by the polling
        // CU's own vector cache and never see another CU's store)
        const unsigned v = device_scope ? __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : __hip_atomic_fetch_add(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)(v - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__global__ __launch_bounds__(256) void sync_probe(int mode, int iters, unsigned* gctr, unsigned* xctr, float* rows, float* sink, int* failed) {
    const int xcd = blockIdx.x & 7, nwg = gridDim.x, per_xcd = nwg / 8;
    float acc = 0.f;
    for (int it = 1; it <= iters; ++it) {
        if (mode == 2) {                                              // publish this workgroup's partial row (256 floats), write-through
            __hip_atomic_store(rows + (long)blockIdx.x * 256 + threadIdx.x, (float)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            bool ok = true;
            if (mode == 1) {
                __hip_atomic_fetch_add(xctr + xcd * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ok = spin_until(xctr + xcd * 64, (unsigned)it * per_xcd, false);
            } else {
                __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                ok = spin_until(gctr, (unsigned)it * nwg, true);
            }
            if (!ok) atomicCAS(failed, 0, 1);
        }
        __syncthreads();
        if (*(volatile int*)failed) return;
        if (mode == 2) {                                              // read the rows of the eight XCD leaders (workgroups 0..7)
            for (int k = 0; k < 8; ++k) acc += __hip_atomic_load(rows + (long)k * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (mode == 2) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

int main() {
    const int nwg = 256, iters = 2000;
    unsigned *gctr, *xctr; float *rows, *sink; int* failed;
    hipMalloc(&gctr, 4); hipMalloc(&xctr, 8 * 64 * 4); hipMalloc(&rows, nwg * 256 * 4); hipMalloc(&sink, nwg * 256 * 4); hipMalloc(&failed, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"global barrier (device-scope counter, all 256 workgroups)", "XCD-local barrier (32 workgroups sharing an L2, workgroup-scope atomics)",
                            "statistic exchange (1 KB row per workgroup, global barrier, read 8 leader rows)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(gctr, 0, 4); hipMemset(xctr, 0, 8 * 64 * 4); hipMemset(failed, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(sync_probe, dim3(nwg), dim3(256), 0, 0, mode, iters, gctr, xctr, rows, sink, failed);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            int f = 0; hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
            if (rep == 1) printf("mode %d  %-88s %s %.2f us per round (code %d)\n", mode, names[mode], f ? "GAVE UP (spin limit);" : "", ms * 1e3 / iters, f);
        }
    }
    // the reference: a dependent launch boundary (eager stream launches of a one-thread kernel; a hipGraph replays them at ~1.6 us per node)
    const int n = 2000;
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("eager back-to-back one-thread kernels: %.2f us per launch\n", ms * 1e3 / n);
    return 0;
}
