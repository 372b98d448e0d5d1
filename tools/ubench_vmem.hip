// ubench_vmem.hip -- how many cycles does one wave-level global load cost the CU's address/L1 path (TA/TCP) on
// gfx950, as a function of the load width, the number of active lanes and where the data sits?  The forest
// kernel issues two kinds of divergent loads (16-byte node records with all 64 lanes active, 2-byte "far"
// probes with ~12 scattered lanes active); this measures what each costs so that DESIGN.md can say which one
// fills the 88 % TA busy time.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_vmem tools/ubench_vmem.hip && tools/bin/ubench_vmem
//
// Every lane keeps 8 fixed addresses (one 128-byte line apart, scattered inside a window of `window` bytes)
// and re-reads them `iters` times; loads are independent, results are xor-ed.  Reported: cycles per wave-level
// load instruction per CU at 2.4 GHz with 5 workgroups x 4 waves per CU (the forest kernel's occupancy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int WIDTH, int MODE>   // WIDTH 2 or 16 bytes; MODE 0 plain, 1 sc1, 2 sc0 sc1, 3 nt
__global__ __launch_bounds__(256) void k_loads(const char *buf, uint32_t window, int nact, int iters, uint32_t *out, uint32_t distinct = 64, int runs = 0,
                                                 unsigned long long *spans = nullptr)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const bool active = ((lane * 37u) & 63u) < (uint32_t)nact;   // scattered lanes, like far probes
    uint32_t off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t h = (wave_id * 64u + (runs ? lane / (64u / distinct) : lane % distinct)) * 2654435761u + (uint32_t)k * 40503u;   // `distinct` lines per instruction
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        off[k] = (h % (window / 128u)) * 128u + (lane & 7u) * 16u;
    }
    uint32_t acc = 0;
    if (active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (WIDTH == 2) {
                    uint32_t v;
                    if (MODE == 0) asm volatile("global_load_ushort %0, %1, %2" : "=v"(v) : "v"(off[k]), "s"(buf) : "memory");
                    if (MODE == 1) asm volatile("global_load_ushort %0, %1, %2 sc1" : "=v"(v) : "v"(off[k]), "s"(buf) : "memory");
                    if (MODE == 2) asm volatile("global_load_ushort %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(off[k]), "s"(buf) : "memory");
                    if (MODE == 3) asm volatile("global_load_ushort %0, %1, %2 nt" : "=v"(v) : "v"(off[k]), "s"(buf) : "memory");
                    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    acc ^= v;
                } else {
                    uint4 v;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off[k]), "s"(buf) : "memory");
                    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    acc ^= v.x ^ v.w;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345u) out[0] = acc;   // keep the loads
    if (spans && lane == 0) spans[wave_id] = __builtin_amdgcn_s_memtime() - t0;   // this wave's span on the shader clock
}

int main()
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t bytes = 256u << 20;
    char *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * 5, iters = 2000;
    const uint32_t windows[3] = {16u << 10, 2u << 20, 128u << 20};   // L1-resident, L2-resident, beyond L2
    const char *wname[3] = {"16KB(L1)", "2MB(L2)", "128MB(MALL/HBM)"};
    printf("(width 18/20/22 = 2-byte loads with sc1 / sc0 sc1 / nt)\nwidth window nact  cycles_per_wave_instr_per_CU  (5 WG x 4 waves per CU, 2.4 GHz assumed)\n");
    for (int width = 2; width <= 16 + 3 * 2; width += (width < 16 ? 14 : 2))   // 2, 16, then 18/20/22 = 2-byte loads with sc1 / sc0 sc1 / nt
        for (int w = 0; w < 3; ++w)
            for (int nact : {4, 12, 32, 64}) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0, 0);
                    if (width == 2) hipLaunchKernelGGL((k_loads<2, 0>), dim3(grid), dim3(256), 0, 0, buf, windows[w], nact, iters, out);
                    else if (width == 16) hipLaunchKernelGGL((k_loads<16, 0>), dim3(grid), dim3(256), 0, 0, buf, windows[w], nact, iters, out);
                    else if (width == 18) hipLaunchKernelGGL((k_loads<2, 1>), dim3(grid), dim3(256), 0, 0, buf, windows[w], nact, iters, out);
                    else if (width == 20) hipLaunchKernelGGL((k_loads<2, 2>), dim3(grid), dim3(256), 0, 0, buf, windows[w], nact, iters, out);
                    else hipLaunchKernelGGL((k_loads<2, 3>), dim3(grid), dim3(256), 0, 0, buf, windows[w], nact, iters, out);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    if (rep == 0) continue;
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                    const double instr_per_cu = 5.0 * 4.0 * iters * 8.0;   // every wave has at least one active lane
                    printf("%5d %-16s %4d  %8.1f   (%.3f ms)\n", width, wname[w], nact, ms * 1e-3 * 2.4e9 / instr_per_cu, ms);
                }
            }
    // lanes sharing lines: 16-byte loads, all 64 lanes active, L1-resident window, `distinct` different lines per instruction
    printf("distinct_lines_per_instr  cycles_per_wave_instr_per_CU (16-byte loads, 64 lanes, 16 KB window); first interleaved\n"
           "(lane %% n), then in runs of consecutive lanes (lane / (64/n))\n");
    for (int runs = 0; runs < 2; ++runs)
    for (uint32_t distinct : {1u, 4u, 16u, 64u}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_loads<16, 0>), dim3(grid), dim3(256), 0, 0, buf, windows[0], 64, iters, out, distinct, runs);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s %8u  %8.1f\n", runs ? "runs       " : "interleaved", distinct, ms * 1e-3 * 2.4e9 / (5.0 * 4.0 * iters * 8.0));
    }
    // the two constants of tools/roofline.py's L1/TA level, on the shader clock itself (s_memtime) and as wall time:
    // 64 lanes, 64 distinct lines per instruction, served by L1 (16 KB window) and by L2 (2 MB window)
    unsigned long long *spans; hipMalloc(&spans, sizeof(unsigned long long) * grid * 4);
    std::vector<unsigned long long> h(grid * 4);
    printf("calibration: width window  cycles_per_line_per_CU (median wave span, s_memtime)  ns_per_line_per_CU (wall)  implied GHz\n");
    for (int width : {2, 16})
        for (int w = 0; w < 2; ++w) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (width == 2) hipLaunchKernelGGL((k_loads<2, 0>), dim3(grid), dim3(256), 0, 0, buf, windows[w], 64, iters, out, 64u, 0, spans);
                else hipLaunchKernelGGL((k_loads<16, 0>), dim3(grid), dim3(256), 0, 0, buf, windows[w], 64, iters, out, 64u, 0, spans);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h.data(), spans, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double lines_per_cu = 5.0 * 4.0 * iters * 8.0 * 64.0;
            const double cyc = (double)h[h.size() / 2] / lines_per_cu, ns = ms * 1e6 / lines_per_cu;
            printf("calibration: %5d %-10s  %6.3f  %6.3f  %5.2f\n", width, wname[w], cyc, ns, cyc / ns);
        }
    return 0;
}
