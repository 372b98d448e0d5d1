// ubench_fetch.hip -- calibrates the rocprofv3 counters the roofline is built from (FETCH_SIZE, TCC_EA0_RDREQ*,
// TCC_MISS/REQ, TCP_TCC_READ_REQ, TCP_TOTAL_CACHE_ACCESSES) on the forest kernel's OWN access patterns, as
// MI355X_MICROARCH.md (HBM section) asks before an absolute is trusted: gfx950 tallies a 128-byte fabric request
// at 64 bytes for wide coalesced streams, other shapes are uncalibrated.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_fetch tools/ubench_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/bin/ubench_fetch      (one --pmc pass per counter set)
//
// Every kernel touches each 128-byte line of the first GiB of a 2-GiB buffer exactly once (lines are visited in a
// bijective pseudo-random order, so nothing is served by a cache: the buffer was just overwritten end to end and only
// its LAST 256 MiB can sit in the Infinity Cache), so the number of lines (8 388 608) and the bytes a correct counter
// would report (lines x 128 B if the L2 fills whole lines, lines x 64 B if it fills 64-byte halves) are known:
//   k_stream16      coalesced 16 B per lane, the guide's calibration shape            (1 GiB really moved)
//   k_line<0>      2 B at offset 0 of a random line per lane = a far depth probe
//   k_line<1>      2 B at offsets 0 and 64 of the same random line: if the counters double against k_line<0> the L2
//                   fetches 64-byte halves, if they stay it fetches whole 128-byte lines
//   k_line<2>      16 B at a random 16-byte slot of a random line per lane = a node record
// The wall time of each kernel is printed too (lines x 128 B / time = the rate at which the memory side serves line fills).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kLines = 1u << 23;          // 1 GiB / 128 B
constexpr uint32_t kMul = 2654435761u | 1u;    // odd: i -> i * kMul mod 2^23 is a bijection

__device__ __forceinline__ uint32_t line_of(uint32_t i) { return (i * kMul) & (kLines - 1u); }

__global__ __launch_bounds__(256) void k_stream16(const uint4 *buf, uint32_t *out)
{
    // 8 M lines x 8 lanes of 16 B; each thread takes 8 slots, coalesced across the wave
    uint32_t acc = 0;
    const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint4 v = buf[base + (size_t)k * 256];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>   // 0: 2 B at offset 0; 1: 2 B at offsets 0 and 64; 2: 16 B at a random slot
__global__ __launch_bounds__(256) void k_line(const char *buf, uint32_t *out)
{
    uint32_t acc = 0;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t i = t * 8u + (uint32_t)k;
        const char *p = buf + (size_t)line_of(i) * 128u;
        if (MODE == 0) acc ^= *reinterpret_cast<const uint16_t *>(p);
        if (MODE == 1) acc ^= *reinterpret_cast<const uint16_t *>(p) ^ *reinterpret_cast<const uint16_t *>(p + 64);
        if (MODE == 2) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + ((i >> 7) & 7u) * 16u);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x1234u) out[0] = acc;      // (a 16-bit value: the compiler must keep the 2-byte loads)
}

__global__ __launch_bounds__(256) void k_fill(uint4 *buf, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t v = (uint32_t)i * 2246822519u + seed;
        buf[i] = make_uint4(v, v ^ 0x9e3779b9u, v + 1u, v + 2u);
    }
}

template <typename F>
void timed(const char *name, char *buf, size_t total, F launch)
{
    // overwrite the whole buffer first: afterwards only its tail can be cache-resident
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, reinterpret_cast<uint4 *>(buf), total / 16, 12345u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-14s %8.3f ms   lines %u   lines*128 B / time = %7.1f GB/s   lines*64 B / time = %7.1f GB/s\n", name, ms, kLines,
           (double)kLines * 128.0 / (ms * 1e-3) / 1e9, (double)kLines * 64.0 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t total = (size_t)2 << 30;
    char *buf; uint32_t *out;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&out, 64));
    const unsigned grid = kLines / (256 * 8);   // 8 lines (or 8 x 16 B x ... ) per thread
    for (int rep = 0; rep < 2; ++rep) {
        timed("k_stream16", buf, total, [&] { hipLaunchKernelGGL(k_stream16, dim3(kLines * 8 / (256 * 8)), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), out); });
        timed("k_line<0> 2B", buf, total, [&] { hipLaunchKernelGGL(k_line<0>, dim3(grid), dim3(256), 0, 0, buf, out); });
        timed("k_line<1> 2Bx2", buf, total, [&] { hipLaunchKernelGGL(k_line<1>, dim3(grid), dim3(256), 0, 0, buf, out); });
        timed("k_line<2> 16B", buf, total, [&] { hipLaunchKernelGGL(k_line<2>, dim3(grid), dim3(256), 0, 0, buf, out); });
    }
    return 0;
}
