// verify_magic.hip -- exhaustive check, on the GPU, of the forest kernel's divide-floor in ONE fma
// (3d-beats_amd/csrc/rdf_hip.hip, NodeRec16 / the level loop).  Companion of verify_intoffset.hip and verify_fastdiv.hip,
// which tie integer floor division to the reference's floor(IEEE (s*u)/d) (decision_tree_common.hpp:15-18).
//
// Record word:  w = (x << 9) | g << 8 | p    x = floor(s*u) as a signed integer, |x| < 2^21; p = 8 bits of unrelated
//                                            payload (threshold / flags); g = NOT bit 7 of p, so the nine low bits are
//                                            a value in [128, 383]
// Decode:       n = v_cvt_f32_i32(w) = 512 * (x + e),  1/4 <= e <= 3/4, in round-down mode (the level loop's mode)
// Divide:       t = v_fma_f32(n, r/512, 1.5 * 2^23) in round-down mode, r = the pixel's refined reciprocal
//               (v_rcp_f32 and one Newton step, computed in round-to-nearest)
// Claim A:      bits(t) - bits(1.5 * 2^23) == floor(x / d)   (integer floor division)
//               for EVERY x in [-2^21, 2^21), EVERY payload byte p and EVERY depth d in [1, 65535].
// Claim B (--ieee): a lane on such a node inside a wave that takes the IEEE path (some other lane holds a kFlagExact
//               node) computes floor(n / (512 d)) with the correctly rounded fp32 divide in round-to-nearest:
//               v_cvt_i32_f32(v_floor_f32(q)) == floor(x / d) for the same triples.
// Why A can hold: (x + e)/d is at least 1/(4d) away from the integers on either side of floor(x/d); the exact product
// n * r/512 differs from it by the reciprocal's relative error (<= 2^-23) times |x + e| / d < 2^21 / d * 2^-23 = 1/(4d);
// the fma rounds the exact sum once, downwards, to a multiple of one.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/verify_magic tools/verify_magic.hip && tools/bin/verify_magic [--ieee] [d_lo d_hi]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

struct Counters {
    unsigned long long mismatches, mismatches_ieee, triples;
    int ex_x[8]; unsigned ex_d[8], ex_p[8], ex_kind[8], n_ex;
};

constexpr uint32_t kMagicBits = 0x4B400000u;   // 1.5 * 2^23

__device__ __forceinline__ uint32_t guarded_payload(uint32_t byte) { return (byte & 0xFFu) | ((~byte & 0x80u) << 1); }

template <bool IEEE>
__global__ __launch_bounds__(256) void k_check(unsigned d_lo, unsigned d_n, Counters *c)
{
    // thread -> (x, d); loops over the 256 payload bytes
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const int x = (int)(gid & 0x3FFFFFu) - (1 << 21);
    const unsigned di = (unsigned)(gid >> 22);
    if (di >= d_n) return;
    const unsigned du = d_lo + di;
    const float d = (float)du;
    // the kernel's reciprocal, in round-to-nearest
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    float rs = r * (1.0f / 512.0f);
    // integer floor division
    int ref = x / (int)du;
    if ((x % (int)du) != 0 && x < 0) --ref;
    unsigned long long bad = 0, bad_ieee = 0;

    if (IEEE) {
        // claim B, in round-to-nearest; the numerator as the round-down decode leaves it (largest fp32 <= w)
        const float ds = d * 512.0f;
        for (unsigned p = 0; p < 256; ++p) {
            const int w = (int)(((uint32_t)x << 9) | guarded_payload(p));
            float n = (float)w;
            if ((long long)n > (long long)w) n = __uint_as_float(__float_as_uint(n) + (w < 0 ? 1u : 0xFFFFFFFFu));   // one step towards -inf
            const float q = n / ds;   // IEEE divide (no fast-math)
            const float fl = __builtin_floorf(q);
            int k;
            asm("v_cvt_i32_f32 %0, %1" : "=v"(k) : "v"(fl));
            if (k != ref) {
                ++bad_ieee;
                const unsigned e = atomicAdd(&c->n_ex, 1u);
                if (e < 8) { c->ex_x[e] = x; c->ex_d[e] = du; c->ex_p[e] = p; c->ex_kind[e] = 1; }
            }
        }
    }

    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\ts_nop 0" : "+v"(rs), "+v"(bad_ieee));
    for (unsigned p = 0; p < 256; ++p) {
        const uint32_t w = ((uint32_t)x << 9) | guarded_payload(p);
        float n, t;
        asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(n) : "v"(w));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(n), "v"(rs), "s"(12582912.0f));
        const int k = (int)(__float_as_uint(t) - kMagicBits);
        if (k != ref) {
            ++bad;
            const unsigned e = atomicAdd(&c->n_ex, 1u);
            if (e < 8) { c->ex_x[e] = x; c->ex_d[e] = du; c->ex_p[e] = p; c->ex_kind[e] = 0; }
        }
    }
    unsigned long long keep = bad;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 0" : "+v"(keep));
    bad = keep;

    unsigned long long n = 256;
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_down(bad, o); bad_ieee += __shfl_down(bad_ieee, o); n += __shfl_down(n, o); }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(&c->mismatches, bad);
        if (bad_ieee) atomicAdd(&c->mismatches_ieee, bad_ieee);
        atomicAdd(&c->triples, n);
    }
}

int main(int argc, char **argv)
{
    bool ieee = false;
    int ai = 1;
    if (argc > 1 && strcmp(argv[1], "--ieee") == 0) { ieee = true; ai = 2; }
    unsigned d_lo = argc > ai + 1 ? (unsigned)atoi(argv[ai]) : 1u;
    unsigned d_hi = argc > ai + 1 ? (unsigned)atoi(argv[ai + 1]) : 65535u;
    Counters *c;
    if (hipMalloc(&c, sizeof(Counters)) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    if (hipMemset(c, 0, sizeof(Counters)) != hipSuccess) return 2;
    const unsigned chunk = 64;
    auto t0 = std::chrono::steady_clock::now();
    unsigned launches = 0;
    Counters h;
    for (unsigned d = d_lo; d <= d_hi; d += chunk) {
        const unsigned n = (d + chunk - 1 <= d_hi) ? chunk : d_hi - d + 1;
        const unsigned long long threads = (unsigned long long)n << 22;
        if (ieee) hipLaunchKernelGGL(k_check<true>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, d, n, c);
        else hipLaunchKernelGGL(k_check<false>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, d, n, c);
        if (++launches % 64 == 0 || d + chunk > d_hi) {
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
            if (hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("d<=%u  (x,p,d) triples=%llu  mismatches: fma %llu, ieee %llu   (%.0f s)\n", d + n - 1, h.triples, h.mismatches,
                   h.mismatches_ieee, s);
            fflush(stdout);
        }
    }
    if (hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (unsigned k = 0; k < h.n_ex && k < 8; ++k)
        printf("example mismatch (%s): x=%d p=%u d=%u\n", h.ex_kind[k] ? "ieee" : "fma", h.ex_x[k], h.ex_p[k], h.ex_d[k]);
    printf("RESULT x [-2^21, 2^21) p [0,255] d [%u,%u]%s: triples %llu | mismatches: round-down fma %llu, ieee path %llu\n", d_lo, d_hi,
           ieee ? " (with the IEEE path)" : "", h.triples, h.mismatches, h.mismatches_ieee);
    return (h.mismatches == 0 && h.mismatches_ieee == 0) ? 0 : 1;
}
