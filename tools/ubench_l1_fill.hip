// ubench_l1_fill.hip -- calibration of tools/roofline.py's L1/TA level: what one 128-byte line costs a CU's vector
// memory pipeline when it is served by the L1 and when it is filled from the L2, in SHADER CYCLES and in wall time,
// with the shader clock measured during the run itself.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_l1_fill tools/ubench_l1_fill.hip && tools/bin/ubench_l1_fill
//
// tools/ubench_vmem.hip (round 1) turned wall time into cycles with an assumed 2.4 GHz.  Here every wave stamps
// s_memtime (shader clock) and s_memrealtime (constant 100 MHz) before and after its loop, and HW_ID/XCC_ID say which
// CU it ran on.  Per CU: span = last end - first start; cycles per line = span / lines the CU's waves loaded; the
// clock = memtime ticks per memrealtime tick x 100 MHz.  Cases: lines served by the L1 (16 KB window), filled from L2
// (2 MB window per XCD... a window every XCD's L2 holds), and mixes of the two at the forest kernel's ratio, with 20 and
// 24 waves per CU (5 x 256 and 3 x 512 threads: the forest kernel's two geometries).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long t0, t1, r0, r1; uint32_t hw, xcc; };

// every lane keeps 8 line addresses; of each 8 loads `fills_of_8` go to the big (L2) window and the rest to a 16 KB
// (L1-resident) window.  WIDTH 2 or 16 bytes.
template <int WIDTH>
__global__ void k_lines(const char *buf, uint32_t l2_window, int fills_of_8, int iters, Stamp *stamps, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t h = (wave_id * 64u + lane) * 2654435761u + (uint32_t)k * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint32_t window = k < fills_of_8 ? l2_window : (16u << 10);
        off[k] = (h % (window / 128u)) * 128u + (lane & 7u) * 16u;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (WIDTH == 2) {
                uint32_t v;
                asm volatile("global_load_ushort %0, %1, %2" : "=v"(v) : "v"(off[k]), "s"(buf) : "memory");
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
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x1234567u) out[0] = acc;
    if (lane == 0) {
        Stamp s;
        s.t0 = t0; s.t1 = t1; s.r0 = r0; s.r1 = r1;
        s.hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID: cu 11:8, sh 12, se 15:13
        s.xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
        stamps[wave_id] = s;
    }
}

int main()
{
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t bytes = 64u << 20;
    char *buf; uint32_t *out; Stamp *stamps;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMalloc(&out, 4));
    const int max_waves = cus * 24;
    CHECK(hipMalloc(&stamps, sizeof(Stamp) * max_waves));
    std::vector<Stamp> h(max_waves);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    printf("width block wg/CU fills_of_8 | CUs waves/CU(min-max) | clock GHz | cycles/line/CU (median CU, min CU) | ns/line/CU (wall) | "
           "solved: hit cycles, fill cycles\n");
    for (int geom = 0; geom < 2; ++geom) {
        const int block = geom ? 512 : 256, per_cu = geom ? 3 : 5;
        for (int width : {2, 16}) {
            double cyc_hit = 0;
            for (int fills : {0, 8, 6, 4, 3, 2, 1}) {
                float ms = 0;
                const int grid = cus * per_cu;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    if (width == 2) hipLaunchKernelGGL((k_lines<2>), dim3(grid), dim3(block), 0, 0, buf, 2u << 20, fills, iters, stamps, out);
                    else hipLaunchKernelGGL((k_lines<16>), dim3(grid), dim3(block), 0, 0, buf, 2u << 20, fills, iters, stamps, out);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                const int waves = grid * block / 64;
                CHECK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
                struct Cu { unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0; int waves = 0; };
                std::map<uint32_t, Cu> per;
                for (int w = 0; w < waves; ++w) {
                    Cu &c = per[((h[w].xcc & 7u) << 8) | ((h[w].hw >> 8) & 0xFFu)];
                    c.t0 = std::min(c.t0, h[w].t0); c.t1 = std::max(c.t1, h[w].t1);
                    c.r0 = std::min(c.r0, h[w].r0); c.r1 = std::max(c.r1, h[w].r1);
                    c.waves++;
                }
                std::vector<double> cyc, ghz;
                int wmin = 1 << 30, wmax = 0;
                for (auto &kv : per) {
                    const Cu &c = kv.second;
                    const double lines = (double)c.waves * iters * 8.0 * 64.0;
                    cyc.push_back((double)(c.t1 - c.t0) / lines);
                    ghz.push_back((double)(c.t1 - c.t0) / ((double)(c.r1 - c.r0) / 100e6) / 1e9);
                    wmin = std::min(wmin, c.waves); wmax = std::max(wmax, c.waves);
                }
                std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
                const double med = cyc[cyc.size() / 2];
                const double ns = ms * 1e6 / ((double)per_cu * (block / 64) * iters * 8.0 * 64.0);
                char solved[96] = "";
                if (fills == 0) cyc_hit = med;
                else snprintf(solved, sizeof solved, "hit %.3f, fill %.3f", cyc_hit, (med - cyc_hit * (8 - fills) / 8.0) * 8.0 / fills);
                printf("%5d %5d %5d %10d | %4zu %3d-%-3d | %5.3f | %6.3f %6.3f | %6.3f | %s\n", width, block, per_cu, fills, per.size(), wmin, wmax,
                       ghz[ghz.size() / 2], med, cyc.front(), ns, solved);
            }
        }
    }
    return 0;
}
