// ubench_valu.hip -- how many cycles does one wave64 VALU instruction hold a gfx950 SIMD's issue port, as a function
// of the number of waves resident on that SIMD?  MI355X_MICROARCH.md says 2 cycles (SIMD-32, 32 lanes per cycle) once
// more than one wave shares the SIMD and 4 for a wave that is alone; DESIGN.md (round 1) priced the forest kernel's
// 2.33e9 VALU wave-instructions at 4 cycles each although it runs 5 waves per SIMD.  This settles it for the
// instructions the kernel's walk is made of.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_valu tools/ubench_valu.hip && tools/bin/ubench_valu
//
// One workgroup per CU (a census of HW_ID checks that), W waves per SIMD (workgroups of 256*W threads, two
// workgroups per CU for W = 8); every wave runs `iters` x 64 independent instructions of one kind (8 destination
// registers in turn, inline asm so nothing is folded) between two s_memtime stamps.  Reported per kind and W:
// SIMD cycles per wave-instruction = (latest stamp - earliest stamp on the CU's clock) / (iters * 64 * W), and the
// same from the hipEvent wall time at the clock GRBM would report (printed as the implied GHz).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { FMA = 0, PK_FMA, CVT_F32_I32, CVT_FLR_I32_F32, ADD_U32, PERM_B32, CNDMASK, MAD_U32_U24, LSHL_ADD,
          CMP_SGPR, CNDMASK_SGPR, MUL_U32_U24, ADD_LSHL, OR3, BFE_I32, LSHLREV, MAX_I32, PK_MUL, MOV, FLOOR_F32, MED3_I32, PK_ADD_I16, CNDMASK_SDWA, ADD_F32, MUL_F32, CMP_CND_VCC, CMP_CND_SGPR, CND_VCC_NEWDST, CND_E64_VCC, SUB_CO_SGPR, ADD_CO_VCC, ADDC_CO_SGPR, CMP_LT_I32_SGPR, CMP_EQ_U32_SGPR, N_OPS };
static const char *kOpName[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_i32", "v_cvt_flr_i32_f32", "v_add_u32",
                                     "v_perm_b32", "v_cndmask_b32(vcc)", "v_mad_u32_u24", "v_lshl_add_u32",
                                     "v_cmp_gt_u32 -> sgpr", "v_cndmask_b32(sgpr)", "v_mul_u32_u24", "v_add_lshl_u32", "v_or3_b32",
                                     "v_bfe_i32", "v_lshlrev_b32", "v_max_i32", "v_pk_mul_f32", "v_mov_b32", "v_floor_f32",
                                     "v_med3_i32", "v_pk_add_i16", "v_cndmask_b32_sdwa(vcc)", "v_add_f32", "v_mul_f32",
                                     "v_cmp->vcc + v_cndmask(vcc) [pair]", "v_cmp->sgpr + v_cndmask(sgpr) [pair]",
                                     "v_cndmask_b32(vcc) dst!=src", "v_cndmask_b32_e64(vcc)",
                                     "v_sub_co_u32_e64 -> sgpr (borrow = a < b)", "v_add_co_u32_e32 -> vcc", "v_addc_co_u32_e64 (sgpr in/out)",
                                     "v_cmp_lt_i32 -> sgpr", "v_cmp_eq_u32 -> sgpr"};

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(1024) void k_valu(int iters, unsigned long long *stamps, uint32_t *where, float seed)
{
    float a[8];
    f2 p[8];
    uint32_t u[8];
    unsigned long long m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        m[k] = 0x5555555555555555ull * (unsigned long long)(k + 1);
        a[k] = seed + (float)k + (float)threadIdx.x;
        p[k] = f2{a[k], a[k] + 0.5f};
        u[k] = (uint32_t)threadIdx.x * 2654435761u + (uint32_t)k;
    }
    const float b = seed * 0.999f, c = seed * 0.001f;
    const f2 b2 = {b, b}, c2 = {c, c};
    const uint32_t ub = (uint32_t)(seed * 1000.f) | 1u;
    // the kernel must own VCC (inline asm below reads it without telling the compiler: in a kernel that never mentions
    // VCC the register is not reserved, and v_cndmask ... vcc then took 19.5 cycles, an artefact)
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(u[0]), "v"(ub) : "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == FMA) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                REP8(S)
#undef S
            } else if (OP == PK_FMA) {
#define S(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(b2), "v"(c2));
                REP8(S)
#undef S
            } else if (OP == CVT_F32_I32) {
#define S(k) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                REP8(S)
#undef S
            } else if (OP == CVT_FLR_I32_F32) {
#define S(k) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u[k]) : "v"(a[k]));
                REP8(S)
#undef S
            } else if (OP == ADD_U32) {
#define S(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == PERM_B32) {
#define S(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(ub), "v"(0x0c040100u));
                REP8(S)
#undef S
            } else if (OP == CNDMASK) {
#define S(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == MAD_U32_U24) {
#define S(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[k]) : "v"(ub), "v"(ub));
                REP8(S)
#undef S
            } else if (OP == LSHL_ADD) {
#define S(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == CMP_SGPR) {
#define S(k) asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m[k]) : "v"(u[k]), "v"(ub));
                REP8(S)
#undef S
            } else if (OP == CNDMASK_SGPR) {
#define S(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[k]) : "v"(ub), "s"(m[k]));
                REP8(S)
#undef S
            } else if (OP == MUL_U32_U24) {
#define S(k) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == ADD_LSHL) {
#define S(k) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == OR3) {
#define S(k) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(ub), "v"(ub));
                REP8(S)
#undef S
            } else if (OP == BFE_I32) {
#define S(k) asm volatile("v_bfe_i32 %0, %0, 2, 24" : "+v"(u[k]));
                REP8(S)
#undef S
            } else if (OP == LSHLREV) {
#define S(k) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(u[k]));
                REP8(S)
#undef S
            } else if (OP == MAX_I32) {
#define S(k) asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == PK_MUL) {
#define S(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(b2));
                REP8(S)
#undef S
            } else if (OP == MOV) {
#define S(k) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == FLOOR_F32) {
#define S(k) asm volatile("v_floor_f32_e32 %0, %0" : "+v"(a[k]));
                REP8(S)
#undef S
            } else if (OP == MED3_I32) {
#define S(k) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(ub), "v"(ub));
                REP8(S)
#undef S
            } else if (OP == PK_ADD_I16) {
#define S(k) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == CNDMASK_SDWA) {
#define S(k) asm volatile("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == ADD_F32) {
#define S(k) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                REP8(S)
#undef S
            } else if (OP == MUL_F32) {
#define S(k) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                REP8(S)
#undef S
            } else if (OP == CMP_CND_VCC) {
#define S(k) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(ub) : "vcc");
                REP8(S)
#undef S
            } else if (OP == CMP_CND_SGPR) {
#define S(k) asm volatile("v_cmp_gt_u32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(u[k]), "=&s"(m[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == CND_VCC_NEWDST) {
#define S(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(ub), "v"(ub));
                REP8(S)
#undef S
            } else if (OP == CND_E64_VCC) {
#define S(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == SUB_CO_SGPR) {
#define S(k) asm volatile("v_sub_co_u32_e64 %0, %1, %0, %2" : "+v"(u[k]), "=s"(m[k]) : "v"(ub));
                REP8(S)
#undef S
            } else if (OP == ADD_CO_VCC) {
#define S(k) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(u[k]) : "v"(ub) : "vcc");
                REP8(S)
#undef S
            } else if (OP == ADDC_CO_SGPR) {
#define S(k) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(u[k]), "+s"(m[k]));
                REP8(S)
#undef S
            } else if (OP == CMP_LT_I32_SGPR) {
#define S(k) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m[k]) : "v"(u[k]), "v"(ub));
                REP8(S)
#undef S
            } else {
#define S(k) asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m[k]) : "v"(u[k]), "v"(ub));
                REP8(S)
#undef S
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    uint32_t usink = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sink += a[k] + p[k].x + p[k].y; usink ^= u[k] ^ (uint32_t)m[k] ^ (uint32_t)(m[k] >> 32); }
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63u) == 0u) {
        stamps[2 * wave] = t0;
        stamps[2 * wave + 1] = t1;
        where[2 * wave] = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
        where[2 * wave + 1] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); // HW_REG_XCC_ID
    }
    if (sink == 123.456f && usink == 77u) stamps[0] = 0;   // keeps the chains alive
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
void run_op(int cus, int iters, unsigned long long *d_st, uint32_t *d_wh)
{
    const int ws[] = {1, 2, 5, 8};
    for (int W : ws) {
        // W waves per SIMD: 4*W waves per CU.  W <= 4: one workgroup of 256*W threads per CU; W = 5: five workgroups of 256
        // (the forest kernel's shape); W = 8: two workgroups of 1024.
        int block = 256 * W, per_cu = 1;
        if (W == 5) { block = 256; per_cu = 5; }
        if (W == 8) { block = 1024; per_cu = 2; }
        const int grid = cus * per_cu;
        const int n_waves = grid * block / 64;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_valu<OP>, dim3(grid), dim3(block), 0, 0, iters / 8, d_st, d_wh, 1.25f);   // warm-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_valu<OP>, dim3(grid), dim3(block), 0, 0, iters, d_st, d_wh, 1.25f);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * (size_t)n_waves);
        std::vector<uint32_t> wh(2 * (size_t)n_waves);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(wh.data(), d_wh, wh.size() * 4, hipMemcpyDeviceToHost));
        // census: waves per (xcc, se, sh, cu, simd)
        std::map<uint32_t, int> per_simd;
        std::map<uint32_t, std::pair<unsigned long long, unsigned long long>> span;   // per CU: earliest t0, latest t1
        for (int w = 0; w < n_waves; ++w) {
            const uint32_t hw = wh[2 * w], xcc = wh[2 * w + 1] & 0xFu;
            const uint32_t cu_key = (xcc << 8) | ((hw >> 8) & 0xFFu);
            per_simd[(cu_key << 2) | ((hw >> 4) & 3u)]++;
            auto it = span.find(cu_key);
            if (it == span.end()) span[cu_key] = {st[2 * w], st[2 * w + 1]};
            else { it->second.first = std::min(it->second.first, st[2 * w]); it->second.second = std::max(it->second.second, st[2 * w + 1]); }
        }
        int simd_min = 1 << 30, simd_max = 0;
        for (auto &kv : per_simd) { simd_min = std::min(simd_min, kv.second); simd_max = std::max(simd_max, kv.second); }
        std::vector<double> cyc;
        for (auto &kv : span) cyc.push_back((double)(kv.second.second - kv.second.first));
        std::sort(cyc.begin(), cyc.end());
        const double med = cyc[cyc.size() / 2];
        const double n_instr = (double)iters * 64.0;
        printf("%-18s W=%d  CUs seen %3zu  waves/SIMD min %d max %d   %.3f cycles per wave-instruction per SIMD (median CU span / (instr * W));"
               "  one wave's own span %.3f cycles/instr;  wall %.3f ms -> %.2f GHz implied\n",
               kOpName[OP], W, span.size(), simd_min, simd_max, med / (n_instr * W),
               (double)(st[1] - st[0]) / n_instr, ms, med / (ms * 1e-3) / 1e9);
        fflush(stdout);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned long long *d_st; uint32_t *d_wh;
    const size_t max_waves = (size_t)cus * 2 * 16;
    CK(hipMalloc(&d_st, max_waves * 16));
    CK(hipMalloc(&d_wh, max_waves * 8));
    printf("gfx950 VALU issue cost, %d CUs, %d x 64 instructions per wave\n", cus, iters);
    run_op<FMA>(cus, iters, d_st, d_wh);
    run_op<PK_FMA>(cus, iters, d_st, d_wh);
    run_op<CVT_F32_I32>(cus, iters, d_st, d_wh);
    run_op<CVT_FLR_I32_F32>(cus, iters, d_st, d_wh);
    run_op<ADD_U32>(cus, iters, d_st, d_wh);
    run_op<PERM_B32>(cus, iters, d_st, d_wh);
    run_op<CNDMASK>(cus, iters, d_st, d_wh);
    run_op<MAD_U32_U24>(cus, iters, d_st, d_wh);
    run_op<LSHL_ADD>(cus, iters, d_st, d_wh);
    run_op<CMP_SGPR>(cus, iters, d_st, d_wh);
    run_op<CNDMASK_SGPR>(cus, iters, d_st, d_wh);
    run_op<MUL_U32_U24>(cus, iters, d_st, d_wh);
    run_op<ADD_LSHL>(cus, iters, d_st, d_wh);
    run_op<OR3>(cus, iters, d_st, d_wh);
    run_op<BFE_I32>(cus, iters, d_st, d_wh);
    run_op<LSHLREV>(cus, iters, d_st, d_wh);
    run_op<MAX_I32>(cus, iters, d_st, d_wh);
    run_op<PK_MUL>(cus, iters, d_st, d_wh);
    run_op<MOV>(cus, iters, d_st, d_wh);
    run_op<FLOOR_F32>(cus, iters, d_st, d_wh);
    run_op<MED3_I32>(cus, iters, d_st, d_wh);
    run_op<PK_ADD_I16>(cus, iters, d_st, d_wh);
    run_op<CNDMASK_SDWA>(cus, iters, d_st, d_wh);
    run_op<ADD_F32>(cus, iters, d_st, d_wh);
    run_op<MUL_F32>(cus, iters, d_st, d_wh);
    run_op<CMP_CND_VCC>(cus, iters, d_st, d_wh);
    run_op<CMP_CND_SGPR>(cus, iters, d_st, d_wh);
    run_op<CND_VCC_NEWDST>(cus, iters, d_st, d_wh);
    run_op<CND_E64_VCC>(cus, iters, d_st, d_wh);
    run_op<SUB_CO_SGPR>(cus, iters, d_st, d_wh);
    run_op<ADD_CO_VCC>(cus, iters, d_st, d_wh);
    run_op<ADDC_CO_SGPR>(cus, iters, d_st, d_wh);
    run_op<CMP_LT_I32_SGPR>(cus, iters, d_st, d_wh);
    run_op<CMP_EQ_U32_SGPR>(cus, iters, d_st, d_wh);
    return 0;
}
