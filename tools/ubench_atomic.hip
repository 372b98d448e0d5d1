// ubench_atomic.hip -- throughput of scattered global atomic adds (no return value) on gfx950: 32-bit vs 64-bit
// counters, as a function of the active lanes per wave instruction and the size of the counter array.  The
// training histogram (tree_train_hip.hip) is bound by this rate on its deep levels.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_atomic tools/ubench_atomic.hip && tools/bin/ubench_atomic
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <typename T>
__global__ __launch_bounds__(256) void k_atomics(T *bins, uint32_t n_bins, int nact, int iters)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (((lane * 37u) & 63u) >= (uint32_t)nact) return;
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        atomicAdd(&bins[(h >> 8) % n_bins], (T)1);
    }
}

int main()
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    void *buf;
    if (hipMalloc(&buf, 1u << 30) != hipSuccess) return 2;
    (void)hipMemset(buf, 0, 1u << 30);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = cus * 8, iters = 2000;
    printf("type  bins      nact   G atomics/s\n");
    for (int type = 0; type < 2; ++type)
        for (uint32_t n_bins : {1u << 10, 1u << 16, 1u << 20, 1u << 24})
            for (int nact : {8, 32, 64}) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0, 0);
                    if (type == 0) hipLaunchKernelGGL(k_atomics<unsigned int>, dim3(grid), dim3(256), 0, 0, (unsigned int *)buf, n_bins, nact, iters);
                    else hipLaunchKernelGGL(k_atomics<unsigned long long>, dim3(grid), dim3(256), 0, 0, (unsigned long long *)buf, n_bins, nact, iters);
                    (void)hipEventRecord(e1, 0);
                    (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                const double n = (double)grid * 4 * nact * iters;
                printf("%s %9u  %4d   %8.2f\n", type ? "u64" : "u32", n_bins, nact, n / (ms * 1e-3) / 1e9);
            }
    return 0;
}
