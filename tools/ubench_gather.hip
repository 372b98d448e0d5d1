// ubench_gather.hip -- what the memory side of an MI355X delivers to the deep-block walk's access pattern: every lane of a
// wave needs ONE random 128-byte line of a table (a deep block), waits for it, and asks for the next one, which depends on
// the data (a walk).  The forest kernel's roofline for forests whose deep levels are occupied is this rate, not the 8 TB/s of
// the HBM data sheet: the guide measures 5.5-5.8 TB/s for random 1,152-byte rows from a table beyond the Infinity Cache and
// 7.4-7.9 TB/s from a 151-MB one; 128-byte rows are not in its tables.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_gather tools/ubench_gather.hip
//   tools/bin/ubench_gather [steps per wave, default 400] [only: coop16 | coop8 | seven24 | one24] [table in MB] [think]
//   (with `only`, ONE kernel shape runs on ONE table: what a rocprofv3 --pmc pass is pointed at, tools/calibrate_gather.sh;
//    `think` = iterations of a dependent integer hash between two fetches of a wave: the request rate falls BELOW the ceiling, which is
//    where the forest kernel runs -- what TA_TA_BUSY charges per L2 miss there calibrates tools/roofline.py's ta_busy_model)
//
// Table sizes: 146 MB (a T4/D20 forest's deep blocks: Infinity-Cache-sized), 1.2 GB (T8/D22's: HBM), 4 GB.
// Fetch shapes, each with a dependent chain per wave (the next 64 line numbers are a hash of the data just read):
//   coop     the round-5 walk: eight LDS-DMA loads of 1 KB per step, lanes 8 i .. 8 i + 7 fetch the eight 16-byte slots of lane
//            8 k + i's line (offset by ds_bpermute) into an 8-KB slab per wave; 512-thread workgroups, 2 per CU (16 waves)
//            and 1 per CU (8 waves);
//   seven    the round-4 walk: every lane loads seven 16-byte slots of its own line into registers; 24 and 16 waves per CU;
//   one      one 16-byte load per lane and line (the least a lane-per-line fetch can issue); 24 waves per CU.
// Prints, per table and shape: lines x 128 B / time in GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

enum { kCoop = 0, kSeven = 1, kOne = 2 };

template <int MODE>
__global__ __launch_bounds__(512, MODE == kCoop ? 4 : 6) void k_gather(const char *table, uint32_t n_lines, int steps, int think, uint32_t *out)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t state = mix(blockIdx.x * 512u + threadIdx.x + 1u);
    uint32_t acc = 0;
    unsigned char *slab = lds + wave * 8192u;
    const unsigned char *my_block = slab + ((uint32_t)lane << 7);
    const uint32_t my_swz = ((uint32_t)lane >> 1) & 7u;
    const uint32_t src_rec = (((uint32_t)lane & 7u) ^ ((uint32_t)lane >> 4)) << 4;
    for (int s = 0; s < steps; ++s) {
        const uint32_t line = __umulhi(state, n_lines);
        uint4 v;
        if (MODE == kCoop) {
            // (tables of up to 4 GiB: a line's byte offset fits 32 bits... of a 4-GiB table it does not: 64-bit here)
            const uint32_t lo = line << 7, hi = line >> 25;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t olo = (uint32_t)__shfl((int)lo, k * 8 + (lane >> 3));
                const uint32_t ohi = (uint32_t)__shfl((int)hi, k * 8 + (lane >> 3));
                const char *p = table + (((size_t)ohi << 32) | olo) + (src_rec ^ ((k & 1) ? 64u : 0u));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                 (__attribute__((address_space(3))) void *)(slab + k * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            v = *reinterpret_cast<const uint4 *>(my_block + (((state >> 29) ^ my_swz) << 4));
        } else {
            const uint4 *p = reinterpret_cast<const uint4 *>(table + ((size_t)line << 7));
            if (MODE == kSeven) {
                uint4 q[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) q[i] = p[i];
                asm volatile("" : "+v"(q[0].x), "+v"(q[1].x), "+v"(q[2].x), "+v"(q[3].x), "+v"(q[4].x), "+v"(q[5].x), "+v"(q[6].x));
                v = q[0];
#pragma unroll
                for (int i = 1; i < 7; ++i) { v.x ^= q[i].x; v.y ^= q[i].y; v.z ^= q[i].z; v.w ^= q[i].w; }
            } else {
                v = p[state >> 29];
            }
        }
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
        state = mix(state + (v.x & 1u) + 0x9e3779b9u);      // the next line depends on the data
        for (int t = 0; t < think; ++t) state = mix(state);  // (a walk's arithmetic between two fetches)
    }
    if (acc == 0x12345679u) out[0] = acc;
}

template <int MODE>
static double run(const char *table, uint32_t n_lines, int steps, int blocks_per_cu, int cus, uint32_t *out, int think = 0)
{
    const int lds = MODE == kCoop ? 8 * 8192 : (blocks_per_cu == 2 ? 80000 : 53000);   // (the LDS footprint sets the residency)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds > 65536 ? lds : 65536));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gather<MODE>, 512, (size_t)lds));
    if (per_cu < blocks_per_cu) fprintf(stderr, "  (only %d workgroups per CU fit, %d asked)\n", per_cu, blocks_per_cu);
    const int grid = cus * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(grid), dim3(512), lds, 0, table, n_lines, steps / 4, think, out);      // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather<MODE>, dim3(grid), dim3(512), lds, 0, table, n_lines, steps, think, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    const double lines = (double)grid * 512.0 * steps;
    return lines * 128.0 / (best * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 400;
    const char *only = argc > 2 ? argv[2] : nullptr;
    const size_t only_mb = argc > 3 ? (size_t)atoi(argv[3]) : 1200;
    const int think = argc > 4 ? atoi(argv[4]) : 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t sizes[3] = {(size_t)146 << 20, (size_t)1200 << 20, (size_t)4095 << 20};
    char *table = nullptr;
    uint32_t *out = nullptr;
    CK(hipMalloc(&table, sizes[2] + 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(table, 0x5A, sizes[2] + 4096));
    printf("random 128-byte line gather, dependent chain per wave, %d CUs, %d steps per wave; GB/s = lines x 128 B / time\n", cus, steps);
    if (only) {
        const uint32_t n_lines = (uint32_t)((only_mb << 20) >> 7);
        double r = 0;
        if (!strcmp(only, "coop16")) r = run<kCoop>(table, n_lines, steps, 2, cus, out, think);
        else if (!strcmp(only, "coop8")) r = run<kCoop>(table, n_lines, steps, 1, cus, out, think);
        else if (!strcmp(only, "seven24")) r = run<kSeven>(table, n_lines, steps, 3, cus, out, think);
        else if (!strcmp(only, "one24")) r = run<kOne>(table, n_lines, steps, 3, cus, out, think);
        printf("table %5zu MB: %s think %d: %7.0f GB/s\n", only_mb, only, think, r);
        return 0;
    }
    for (int t = 0; t < 3; ++t) {
        const uint32_t n_lines = (uint32_t)(sizes[t] >> 7);
        printf("table %5zu MB:", sizes[t] >> 20);
        printf("  coop 16 waves/CU %7.0f", run<kCoop>(table, n_lines, steps, 2, cus, out));
        printf("  coop 8 waves/CU %7.0f", run<kCoop>(table, n_lines, steps, 1, cus, out));
        printf("  seven loads 24 waves/CU %7.0f", run<kSeven>(table, n_lines, steps, 3, cus, out));
        printf("  seven loads 16 waves/CU %7.0f", run<kSeven>(table, n_lines, steps, 2, cus, out));
        printf("  one load 24 waves/CU %7.0f\n", run<kOne>(table, n_lines, steps, 3, cus, out));
        fflush(stdout);
    }
    CK(hipFree(table));
    CK(hipFree(out));
    return 0;
}
