// verify_payload.hip -- companion of verify_fastdiv.hip / verify_intoffset.hip.  Exhaustive check, on the GPU, of the
// 16-byte node record's ONE-instruction numerator decode.
//
// Record word:  w = (x << 9) | p     x = floor(s*u) as a 23-bit signed integer (|x| <= 2^22), bit 8 = 0,
//                                    p = 8 bits of unrelated payload (threshold / flags)
// Decode:       n = v_cvt_f32_i32(w) = 512 * (x + e),  0 <= e <= 1/2   (the payload and the convert's rounding
//                                    only ever ADD a fraction of one unit of x: 255/512 rounds to at most 256/512)
// Divide:       q = fastdiv(n, 512*d) with the pixel's refined reciprocal r/512 (exact scalings of the verified
//               sequence): q0 = n*r'; rem = fma(-512 d, q0, n); q = fma(rem, r', q0)
// Claim:        v_cvt_flr_i32_f32(q) == floor(x / d)   (integer floor division)
//               for EVERY x in [-2^22, 2^22), EVERY payload byte p and EVERY depth d in [1, 65535].
// Why it can hold: for integers x, d and 0 <= e < 1, floor((x+e)/d) = floor(x/d); the nearest integer above
// (x+e)/d is at least (1-e)/d >= 1/(2d) away, more than half an ulp of the quotient because |x| + d < 2^23.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/verify_payload tools/verify_payload.hip && tools/bin/verify_payload [d_lo d_hi]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

struct Counters {
    unsigned long long mismatches, triples;
    int ex_x[8]; unsigned ex_d[8], ex_p[8], n_ex;
};

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_check(unsigned d_lo, unsigned d_n, Counters *c)
{
    // thread -> (x, d); loops over the 256 payload bytes
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const int x = (int)(gid & 0x7FFFFFu) - (1 << 22);
    const unsigned di = (unsigned)(gid >> 23);
    if (di >= d_n) return;
    const unsigned du = d_lo + di;
    const float d = (float)du;
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    const float rs = r * (1.0f / 512.0f), nds = -512.0f * d;
    // integer floor division
    int ref = x / (int)du;
    if ((x % (int)du) != 0 && x < 0) --ref;
    unsigned long long bad = 0;
    for (unsigned p = 0; p < 256; p += 2) {   // two payloads per packed instruction, like the kernel
        const uint32_t w0 = ((uint32_t)x << 9) | p, w1 = w0 | 1u;
        f2 n;
        asm("v_cvt_f32_i32 %0, %1" : "=v"(n.x) : "v"(w0));
        asm("v_cvt_f32_i32 %0, %1" : "=v"(n.y) : "v"(w1));
        const f2 rr = {rs, rs}, nd = {nds, nds};
        const f2 q0 = n * rr;
        const f2 q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q0, n), rr, q0);
        int f0, f1;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(f0) : "v"(q.x));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(f1) : "v"(q.y));
        const bool m0 = f0 != ref, m1 = f1 != ref;
        bad += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
        if (m0 || m1) {
            const unsigned k = atomicAdd(&c->n_ex, 1u);
            if (k < 8) { c->ex_x[k] = x; c->ex_d[k] = du; c->ex_p[k] = m0 ? p : p + 1; }
        }
    }
    unsigned long long n = 256;
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_down(bad, o); n += __shfl_down(n, o); }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(&c->mismatches, bad);
        atomicAdd(&c->triples, n);
    }
}

int main(int argc, char **argv)
{
    unsigned d_lo = argc > 2 ? (unsigned)atoi(argv[1]) : 1u;
    unsigned d_hi = argc > 2 ? (unsigned)atoi(argv[2]) : 65535u;
    Counters *c;
    if (hipMalloc(&c, sizeof(Counters)) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    if (hipMemset(c, 0, sizeof(Counters)) != hipSuccess) return 2;
    const unsigned chunk = 32;
    auto t0 = std::chrono::steady_clock::now();
    unsigned launches = 0;
    Counters h;
    for (unsigned d = d_lo; d <= d_hi; d += chunk) {
        const unsigned n = (d + chunk - 1 <= d_hi) ? chunk : d_hi - d + 1;
        const unsigned long long threads = (unsigned long long)n << 23;
        hipLaunchKernelGGL(k_check, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, d, n, c);
        if (++launches % 64 == 0 || d + chunk > d_hi) {
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
            if (hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("d<=%u  (x,p,d) triples=%llu  floor mismatches %llu   (%.0f s)\n", d + n - 1, h.triples, h.mismatches, s);
            fflush(stdout);
        }
    }
    if (hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (unsigned k = 0; k < h.n_ex && k < 8; ++k) printf("example mismatch: x=%d p=%u d=%u\n", h.ex_x[k], h.ex_p[k], h.ex_d[k]);
    printf("RESULT x [-2^22, 2^22) p [0,255] d [%u,%u]: triples %llu | floor mismatches %llu\n", d_lo, d_hi, h.triples, h.mismatches);
    return h.mismatches == 0 ? 0 : 1;
}
