// mean_shift_hip.hip -- gfx950 kernels for the consumer of the composite label map:
// per-class 2-D mean-shift mode finding and fingertip heights.
//
// Reference: src/cuda/mean_shift.cu:3-48 (one pass over the label image per round, fp64 atomicAdd
// into a [classes][3] array) driven by the host loop of src/cuda/mean_shift.py:35-59, which per
// round zero-fills the sums, launches, copies sums and means to the host, divides, adds and copies
// the means back (12 transfers for 6 rounds); heights: src/3d_bz.py:503-522 on the host.
//
// Here everything stays on the device and is bitwise reproducible:
//   * round r is ONE launch; every workgroup first rebuilds the previous round's means from that
//     round's per-workgroup partial sums (one per lane, added by a fixed shuffle tree, so all workgroups
//     get identical values), then accumulates its pixels with fixed-order wave reductions -- no atomics,
//     so no order dependence (the reference's fp64 atomics make its last bits run-dependent);
//   * kernel boundaries are the only inter-workgroup hand-off (no in-kernel fences needed).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/rdf_hip.h"

namespace {

constexpr int kMsBlocks = 64;      // workgroups per round (fixed: it defines the summation order)
constexpr int kMsThreads = 1024;
constexpr int kMsWaves = kMsThreads / 64;
constexpr int kMsMaxClasses = 64;
constexpr uint32_t kNoLabel = 65535u;

// workspace layout (doubles): means[rounds + 1][L][2] (slot 0 = start = zeros) | partials[2][kMsBlocks][L][3]
__host__ __device__ inline size_t ms_means_off(int slot, int L) { return (size_t)slot * L * 2; }
__host__ __device__ inline size_t ms_part_off(int rounds, int L, int parity)
{
    return (size_t)(rounds + 1) * L * 2 + (size_t)parity * kMsBlocks * L * 3;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);   // fixed tree => reproducible
    return v;
}

// round == 0: centroid sums (mean_shift.cu:32-35); round >= 1: kernel-weighted shift sums (:36-47).
// round == num_rounds: no pixel pass, only the final means.
__global__ __launch_bounds__(kMsThreads) void k_mean_shift_round(const uint16_t *labels, int dim_x, int dim_y, int L,
                                                                 const float *variances, int round, int num_rounds,
                                                                 double *ws, double *means_out)
{
    __shared__ double s_means[kMsMaxClasses][2];
    __shared__ double s_acc[kMsWaves][kMsMaxClasses][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- means entering this round = means[round-1 slot] + shift of round-1 (mean_shift.py:54-57) ----
    // The previous round left one partial sum per workgroup; lane b of a wave fetches workgroup b's and a fixed
    // shuffle tree adds them (same inputs, same tree in every workgroup: identical means everywhere).  An earlier
    // version added the 64 partials one after the other in one lane of every workgroup: 64 dependent round trips to L2
    // made a round cost 13 us for 2 us of work (profiles/r02_pipeline_kernel_stats.csv).
    static_assert(kMsBlocks == 64, "one partial per lane");
    if (round > 0) {
        const double *part = ws + ms_part_off(num_rounds, L, (round - 1) & 1);
        for (int c = wave; c < L; c += kMsWaves) {
            const double *p = part + ((size_t)lane * L + c) * 3;
            const double sx = wave_sum(p[0]), sy = wave_sum(p[1]), sw = wave_sum(p[2]);
            if (lane == 0) {
                const double *prev = ws + ms_means_off(round - 1, L) + (size_t)c * 2;
                s_means[c][0] = prev[0] + sx / sw;      // 0/0 = NaN for a class without pixels, as in the reference
                s_means[c][1] = prev[1] + sy / sw;
            }
        }
    } else if (tid < L) {
        s_means[tid][0] = 0.0;
        s_means[tid][1] = 0.0;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < L) {
        double *cur = ws + ms_means_off(round, L) + (size_t)tid * 2;
        cur[0] = s_means[tid][0]; cur[1] = s_means[tid][1];
        if (round == num_rounds) { means_out[tid * 2] = s_means[tid][0]; means_out[tid * 2 + 1] = s_means[tid][1]; }
    }
    if (round == num_rounds) return;
    for (int i = tid; i < kMsWaves * kMsMaxClasses * 3; i += kMsThreads) (&s_acc[0][0][0])[i] = 0.0;
    __syncthreads();

    // ---- this workgroup's pixels: a contiguous slab, 64 consecutive pixels per wave step ----
    const long long n_px = (long long)dim_x * dim_y;
    const long long per_block = (n_px + kMsBlocks - 1) / kMsBlocks;
    const long long begin = per_block * blockIdx.x;
    const long long end = begin + per_block < n_px ? begin + per_block : n_px;
    for (long long base = begin + (long long)wave * 64; base < end; base += kMsThreads) {
        const long long p = base + lane;
        uint32_t l = 0u;
        double cx = 0.0, cy = 0.0, cw = 0.0;
        if (p < end) {
            l = labels[p];
            if (l != 0u && l != kNoLabel && l <= (uint32_t)L) {
                const double x = (double)(int)(p % dim_x), y = (double)(int)(p / dim_x);
                if (round == 0) {
                    cx = x; cy = y; cw = 1.0;
                } else {
                    const double dx = x - s_means[l - 1][0], dy = y - s_means[l - 1][1];
                    const double dist_sq = (dx * dx) + (dy * dy);
                    const float vf = variances[l - 1];
                    const double v_2 = (double)(vf * vf);          // float product, as mean_shift.cu:42
                    const double w = exp(-dist_sq / (2 * v_2));
                    cx = dx * w; cy = dy * w; cw = w;
                }
            } else {
                l = 0u;
            }
        }
        unsigned long long present = 0ull;   // classes seen by this wave step
        for (int c = 0; c < L; ++c) {
            if (__any(l == (uint32_t)(c + 1))) present |= 1ull << c;
        }
        for (int c = 0; c < L; ++c) {
            if (!((present >> c) & 1ull)) continue;
            const bool mine = l == (uint32_t)(c + 1);
            const double sx = wave_sum(mine ? cx : 0.0), sy = wave_sum(mine ? cy : 0.0), sw = wave_sum(mine ? cw : 0.0);
            if (lane == 0) { s_acc[wave][c][0] += sx; s_acc[wave][c][1] += sy; s_acc[wave][c][2] += sw; }
        }
    }
    __syncthreads();
    if (tid < L * 3) {
        const int c = tid / 3, k = tid % 3;
        double s = 0.0;
        for (int w = 0; w < kMsWaves; ++w) s += s_acc[w][c][k];
        ws[ms_part_off(num_rounds, L, round & 1) + ((size_t)blockIdx.x * L + c) * 3 + k] = s;
    }
}

// Heights of the requested classes' modes above the calibrated plane (3d_bz.py:503-522):
//   px,py = int(mean) * labels_reduce; off-frame -> NaN ("reset"); z = depth[py][px];
//   pt = z * ((px-ppx)/fx, (py-ppy)/fy, 1)  (librealsense rs2_deproject_pixel_to_point, no distortion, fp32);
//   height = -(plane @ [pt,1]).z  with the fp32 plane matrix promoted to fp64 like numpy's float32 @ float64.
__global__ void k_fingertip_heights(const double *means, int L, const int *class_ids, int n_ids, const uint16_t *depth,
                                    int dim_x, int dim_y, int labels_reduce, float fx, float fy, float ppx, float ppy,
                                    const float *plane, double *heights)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ids) return;
    const int c = class_ids[i];
    double out = nan("");
    if (c >= 1 && c <= L) {
        const double mx = means[(c - 1) * 2], my = means[(c - 1) * 2 + 1];
        if (mx == mx && my == my && fabs(mx) < 1e9 && fabs(my) < 1e9) {
            const long long px = (long long)mx * labels_reduce, py = (long long)my * labels_reduce;   // trunc toward 0
            if (px >= 0 && py >= 0 && px < dim_x && py < dim_y) {
                const float z = (float)depth[py * dim_x + px];
                const float x = ((float)px - ppx) / fx, y = ((float)py - ppy) / fy;
                const double pt[4] = {(double)(z * x), (double)(z * y), (double)z, 1.0};
                double pz = 0.0;
                for (int k = 0; k < 4; ++k) pz += (double)plane[2 * 4 + k] * pt[k];
                out = -pz;
            }
        }
    }
    heights[i] = out;
}

} // namespace

extern "C" {

size_t rdf_mean_shift_workspace_bytes(int num_classes, int num_rounds)
{
    if (num_classes < 0 || num_rounds < 0) return 0;
    return (ms_part_off(num_rounds, num_classes, 0) + (size_t)2 * kMsBlocks * num_classes * 3) * sizeof(double);
}

int rdf_mean_shift(const uint16_t *labels, int dim_x, int dim_y, int num_classes, const float *variances,
                   int num_rounds, double *means_out, void *workspace, void *stream)
{
    if (dim_x < 0 || dim_y < 0 || num_classes < 0 || num_classes > kMsMaxClasses || num_rounds < 0) return RDF_ERR_BAD_ARG;
    if (num_classes == 0) return RDF_OK;
    if (!means_out || !workspace || (num_rounds > 0 && (!labels || !variances))) return RDF_ERR_NULL_PTR;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double *ws = reinterpret_cast<double *>(workspace);
    for (int r = 0; r <= num_rounds; ++r) {
        const int blocks = r == num_rounds ? 1 : kMsBlocks;
        hipLaunchKernelGGL(k_mean_shift_round, dim3(blocks), dim3(kMsThreads), 0, st, labels, dim_x, dim_y, num_classes,
                           variances, r, num_rounds, ws, means_out);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return RDF_OK;
}

int rdf_fingertip_heights(const double *means, int num_classes, const int *class_ids, int n_ids,
                          const uint16_t *depth, int dim_x, int dim_y, int labels_reduce, float fx, float fy,
                          float ppx, float ppy, const float *plane, double *heights_out, void *stream)
{
    if (num_classes < 0 || n_ids < 0 || dim_x < 0 || dim_y < 0 || labels_reduce < 1) return RDF_ERR_BAD_ARG;
    if (n_ids == 0) return RDF_OK;
    if (!means || !class_ids || !depth || !plane || !heights_out) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_fingertip_heights, dim3((n_ids + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       means, num_classes, class_ids, n_ids, depth, dim_x, dim_y, labels_reduce, fx, fy, ppx, ppy, plane,
                       heights_out);
    return (int)hipGetLastError();
}

} // extern "C"
