// mean_shift_hip.hip -- gfx950 kernels for the consumer of the composite label map:
// per-class 2-D mean-shift mode finding and fingertip heights.
//
// Reference: src/cuda/mean_shift.cu:3-48 (one pass over the label image per round, fp64 atomicAdd
// into a [classes][3] array) driven by the host loop of src/cuda/mean_shift.py:35-59, which per
// round zero-fills the sums, launches, copies sums and means to the host, divides, adds and copies
// the means back (12 transfers for 6 rounds); heights: src/3d_bz.py:503-522 on the host.
//
// Here everything stays on the device, in ONE launch, and is bitwise reproducible: one workgroup per class lists the
// class's pixels in LDS once and runs every round on that list with fixed-order sums -- no atomics, so no order
// dependence (the reference's fp64 atomics make its last bits run-dependent), and no hand-off between workgroups.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <mutex>

#include "../../include/rdf_hip.h"

namespace {

constexpr int kMsThreads = 1024;
constexpr int kMsWaves = kMsThreads / 64;
constexpr int kMsMaxClasses = 64;
constexpr uint32_t kMsListCap = 32768;   // pixels of one class kept in LDS (131 072 B of the CU's 160 KB)
constexpr uint32_t kMsTabCap = 3072;     // dim_x + dim_y the per-round weight tables cover (24 576 B)
constexpr int kMsSteps = 13;             // 16 waves x 13 steps x 512 pixels = 106 496 >= the app's 424 x 240 label map

// One step of a row-wise (16 lanes) sum: v += the value `shr` lanes to the left in the same row (0.0 where there is none).
// DPP moves of the two halves: a VALU-speed lane exchange (ds_bpermute, which __shfl_down compiles to, is an LDS round trip).
template <int SHR>
__device__ __forceinline__ double row_add_shr(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SHR, 0xF, 0xF, true);   // row_shr:SHR, bound_ctrl: 0 outside
    const int phi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SHR, 0xF, 0xF, true);
    return v + __hiloint2double(phi, plo);
}

// Sum over the wave in a fixed order: a shift-add scan inside each row of 16 lanes (lane 15, 31, 47, 63 then hold their
// row's total), the four row totals added in order.  Every lane returns the same value.
__device__ __forceinline__ double wave_sum(double v)
{
    v = row_add_shr<1>(v); v = row_add_shr<2>(v); v = row_add_shr<4>(v); v = row_add_shr<8>(v);
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * q + 15),
                                __builtin_amdgcn_readlane(__double2loint(v), 16 * q + 15));
    return ((r[0] + r[1]) + r[2]) + r[3];
}

// Sum of every thread's (a, b, c) in a fixed order: inside a wave as above, then the waves one after the other.
// Every thread returns with the totals in tot[0..2].  s_red is double-buffered by the caller's round parity, so one
// barrier per call is enough: a buffer is rewritten two calls later, after every thread has passed the barrier between.
__device__ __forceinline__ void block_sum3(double a, double b, double c, double (*s_red)[3], double *tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if (lane == 0) { s_red[wave][0] = a; s_red[wave][1] = b; s_red[wave][2] = c; }
    __syncthreads();
    double ta = 0.0, tb = 0.0, tc = 0.0;
    for (int w = 0; w < kMsWaves; ++w) { ta += s_red[w][0]; tb += s_red[w][1]; tc += s_red[w][2]; }   // broadcast reads
    tot[0] = ta; tot[1] = tb; tot[2] = tc;
}

// All rounds of one class in ONE workgroup, all classes in one launch: classes never interact (mean_shift.cu:3-48 sums
// per class), so no round needs anything from another workgroup.  The workgroup lists its class's pixels once, in pixel
// order, in LDS (x | y << 16) and then iterates over the list: round 0 = centroid (:32-35), rounds >= 1 = kernel-weighted
// shift (:36-47), means += sums[:2] / sums[2] after each (mean_shift.py:54-57).  A class with more than kMsListCap
// pixels is not listed: its rounds rescan the label image.  Every sum is
// taken in a fixed order (thread-strided partial sums, shuffle tree, waves in order): bitwise reproducible.
// An earlier version ran one launch per round over 64 workgroups with a cross-workgroup partial-sum hand-off at every
// kernel boundary: 7 launches, 74 us for the app's 424x240 label map (profiles/r02_pipeline_kernel_stats_before.csv).
struct HeightArgs {
    const int *class_ids;      // 1-based labels (device); n_ids of them
    int n_ids;
    const uint16_t *depth;     // the ORIGINAL depth frame (3d_bz.py:515)
    int dim_x, dim_y, labels_reduce;
    float fx, fy, ppx, ppy;
    const float *plane;        // 4x4 row-major (device)
    double *heights;           // [n_ids]
};

__device__ __forceinline__ double height_of_mode(double mx, double my, const HeightArgs &h)
{
    double out = nan("");
    if (mx == mx && my == my && fabs(mx) < 1e9 && fabs(my) < 1e9) {
        const long long px = (long long)mx * h.labels_reduce, py = (long long)my * h.labels_reduce;   // trunc toward 0
        if (px >= 0 && py >= 0 && px < h.dim_x && py < h.dim_y) {
            const float z = (float)h.depth[py * h.dim_x + px];
            const float x = ((float)px - h.ppx) / h.fx, y = ((float)py - h.ppy) / h.fy;
            const double pt[4] = {(double)(z * x), (double)(z * y), (double)z, 1.0};
            double pz = 0.0;
            for (int k = 0; k < 4; ++k) pz += (double)h.plane[2 * 4 + k] * pt[k];
            out = -pz;
        }
    }
    return out;
}

template <bool HEIGHTS>
__global__ __launch_bounds__(kMsThreads) void k_mean_shift_fused(const uint16_t *labels, int dim_x, int dim_y, int L,
                                                                 const float *variances, int num_rounds, double *means_out,
                                                                 const HeightArgs hp)
{
    extern __shared__ uint32_t s_list[];
    __shared__ double s_tab[kMsTabCap];      // round's weights by column, then by row (see the rounds below)
    __shared__ double s_red[2][kMsWaves][3];
    __shared__ uint32_t s_wave_cnt[kMsWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t want = blockIdx.x + 1u;
    const uint32_t n_px = (uint32_t)dim_x * (uint32_t)dim_y;

    // ---- list this class's pixels.  A wave takes kMsSteps x 512 pixels per batch (512 consecutive ones per step, the
    // sixteen waves interleaved), eight per lane and step
    // (one 16-byte load; a step's eight match bits are one byte of a packed register); a lane's pixels go to consecutive
    // list slots, the lanes of a wave, the waves and the batches follow each other in order -- positions come from a
    // wave scan of the per-lane counts and the waves' totals, so the order does not depend on timing (the rounds' sums
    // depend on the list order, and must be reproducible).  A class that does not fit the list (kMsListCap pixels) is
    // not listed at all: its rounds rescan the label image. ----
    const bool vec_ok = (reinterpret_cast<uintptr_t>(labels) & 15u) == 0u;
    uint32_t n_class = 0;
    for (uint32_t batch0 = 0; batch0 < n_px; batch0 += kMsWaves * kMsSteps * 512u) {
        // step k of wave w covers the 512 pixels from batch0 + (k * 16 + w) * 512: a class is a few blobs, and with the
        // waves interleaved every 512 pixels all sixteen share the work of listing it (a wave alone on its SIMD issues
        // an instruction every ~8 cycles: with 16 consecutive rows per wave, the one or two waves that held a fingertip
        // took 13 000 cycles to list 140 pixels while the others idled)
        const uint32_t wbase = batch0 + (uint32_t)wave * 512u;
        uint32_t mp[(kMsSteps + 3) / 4];      // byte k & 3 of word k >> 2: which of the lane's eight pixels of step k match
#pragma unroll
        for (int q = 0; q < (kMsSteps + 3) / 4; ++q) mp[q] = 0u;
        uint32_t cnt = 0u;
#pragma unroll
        for (int k = 0; k < kMsSteps; ++k) {
            const uint32_t p0 = wbase + ((uint32_t)k * (kMsWaves * 64u) + (uint32_t)lane) * 8u;
            uint32_t bits = 0u;
            if (p0 + 8u <= n_px && vec_ok) {
                const uint4 v = *reinterpret_cast<const uint4 *>(labels + p0);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bits |= ((w4[q] & 0xFFFFu) == want ? 1u << (2 * q) : 0u) | ((w4[q] >> 16) == want ? 2u << (2 * q) : 0u);
            } else {
                for (int j = 0; j < 8; ++j)
                    if (p0 + (uint32_t)j < n_px && labels[p0 + j] == want) bits |= 1u << j;
            }
            mp[k >> 2] |= bits << (8 * (k & 3));
            cnt += (uint32_t)__builtin_popcount(bits);
        }
        // exclusive scan of the lanes' counts (fixed shuffle pattern) and the wave's total
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const uint32_t lane_off = incl - cnt, wave_total = __shfl(incl, 63);
        __syncthreads();             // the previous batch's s_wave_cnt has been read
        if (lane == 0) s_wave_cnt[wave] = wave_total;
        __syncthreads();
        uint32_t at = n_class + lane_off, batch_total = 0u;
        for (int w = 0; w < kMsWaves; ++w) {
            if (w < wave) at += s_wave_cnt[w];
            batch_total += s_wave_cnt[w];
        }
        if (cnt != 0u && at < kMsListCap) {      // (a lane whose first slot is beyond the list writes nothing: the class overflows)
#pragma unroll
            for (int k = 0; k < kMsSteps; ++k) {
                const uint32_t mk = (mp[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                if (mk == 0u) continue;
                const uint32_t p0 = wbase + ((uint32_t)k * (kMsWaves * 64u) + (uint32_t)lane) * 8u;
                const uint32_t y0 = p0 / (uint32_t)dim_x, x0 = p0 - y0 * (uint32_t)dim_x;   // one division per eight pixels
                if (dim_x >= 8) {
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j) {      // the eight pixels sit in at most two rows
                        if ((mk >> j) & 1u) {
                            const uint32_t xj = x0 + j, wrap = xj >= (uint32_t)dim_x ? 1u : 0u;
                            const uint32_t slot = at + (uint32_t)__builtin_popcount(mk & ((1u << j) - 1u));
                            if (slot < kMsListCap) s_list[slot] = (xj - (wrap ? (uint32_t)dim_x : 0u)) | ((y0 + wrap) << 16);
                        }
                    }
                } else {
                    uint32_t rest = mk, slot = at;
                    while (rest) {
                        const uint32_t j = (uint32_t)__builtin_ctz(rest);
                        rest &= rest - 1u;
                        uint32_t x = x0 + j, y = y0;
                        while (x >= (uint32_t)dim_x) { x -= (uint32_t)dim_x; ++y; }      // (rows narrower than 8 wrap more than once)
                        if (slot < kMsListCap) s_list[slot] = x | (y << 16);
                        ++slot;
                    }
                }
                at += (uint32_t)__builtin_popcount(mk);
            }
        }
        n_class += batch_total;
    }
    __syncthreads();
    const bool listed = n_class <= kMsListCap;
    const uint32_t n_list = listed ? n_class : 0u, rescan_from = listed ? n_px : 0u;

    double mx = 0.0, my = 0.0;
    const float vf = variances ? variances[blockIdx.x] : 0.0f;
    const double v_2 = (double)(vf * vf);          // float product, as mean_shift.cu:42
    // The kernel weight factorises: exp(-((x-mx)^2 + (y-my)^2) / 2v^2) = exp(-(x-mx)^2 / 2v^2) * exp(-(y-my)^2 / 2v^2).
    // A round therefore needs dim_x + dim_y exponentials per class (one table entry per thread), not one per pixel, and a
    // pixel's weight is the product of two table entries (a few ulp from the exponential of the sum; the reference's own
    // sums differ more from run to run).  That took the app's 424x240 map from ~6 us to ~1 us per round.
    const bool tables = (uint32_t)dim_x + (uint32_t)dim_y <= kMsTabCap && v_2 > 0.0;
    for (int round = 0; round < num_rounds; ++round) {
        double sx = 0.0, sy = 0.0, sw = 0.0;
        if (round > 0 && tables) {
            // (no barrier before rewriting the tables: every thread has passed the barrier of the previous round's
            // block_sum3, which it reaches only after its last table read)
            for (uint32_t i = tid; i < (uint32_t)dim_x + (uint32_t)dim_y; i += kMsThreads) {
                const double c = i < (uint32_t)dim_x ? (double)(int)i - mx : (double)(int)(i - (uint32_t)dim_x) - my;
                s_tab[i] = exp(-(c * c) / (2 * v_2));
            }
            __syncthreads();
        }
        auto term = [&](uint32_t xi, uint32_t yi) {
            const double x = (double)(int)xi, y = (double)(int)yi;
            if (round == 0) {
                sx += x; sy += y; sw += 1.0;
            } else {
                const double dx = x - mx, dy = y - my;
                double w;
                if (tables) {
                    w = s_tab[xi] * s_tab[(uint32_t)dim_x + yi];
                } else {
                    const double dist_sq = (dx * dx) + (dy * dy);
                    w = exp(-dist_sq / (2 * v_2));
                }
                sx += dx * w; sy += dy * w; sw += w;
            }
        };
        if (round > 0 && tables) {
            // four list entries per trip, their table reads in flight together; added in the same order as one by one
            for (uint32_t i = tid; i < n_list; i += 4u * kMsThreads) {
                uint32_t e[4];
                double wx[4], wy[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) e[u] = i + u * kMsThreads < n_list ? s_list[i + u * kMsThreads] : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) { wx[u] = s_tab[e[u] & 0xFFFFu]; wy[u] = s_tab[(uint32_t)dim_x + (e[u] >> 16)]; }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    if (i + u * kMsThreads < n_list) {
                        const double dx = (double)(int)(e[u] & 0xFFFFu) - mx, dy = (double)(int)(e[u] >> 16) - my;
                        const double w = wx[u] * wy[u];
                        sx += dx * w; sy += dy * w; sw += w;
                    }
                }
            }
        } else {
            for (uint32_t i = tid; i < n_list; i += kMsThreads) {
                const uint32_t e = s_list[i];
                term(e & 0xFFFFu, e >> 16);
            }
        }
        for (uint32_t p = rescan_from + tid; p < n_px; p += kMsThreads)
            if (labels[p] == want) term(p % (uint32_t)dim_x, p / (uint32_t)dim_x);
        double tot[3];
        block_sum3(sx, sy, sw, s_red[round & 1], tot);
        mx = mx + tot[0] / tot[2];                 // 0/0 = NaN for a class without pixels, as in the reference
        my = my + tot[1] / tot[2];
    }
    if (tid == 0) { means_out[blockIdx.x * 2] = mx; means_out[blockIdx.x * 2 + 1] = my; }
    if (HEIGHTS) {
        // the heights of this class's fingertips ride on its workgroup (rdf_mean_shift_heights: one launch less per hand
        // and frame); ids that name no class are the first workgroup's
        if (tid < hp.n_ids) {
            const int c = hp.class_ids[tid];
            if (c == (int)want) hp.heights[tid] = height_of_mode(mx, my, hp);
            else if (blockIdx.x == 0 && (c < 1 || c > L)) hp.heights[tid] = nan("");
        }
    }
}

// Heights of the requested classes' modes above the calibrated plane (3d_bz.py:503-522):
//   px,py = int(mean) * labels_reduce; off-frame -> NaN ("reset"); z = depth[py][px];
//   pt = z * ((px-ppx)/fx, (py-ppy)/fy, 1)  (librealsense rs2_deproject_pixel_to_point, no distortion, fp32);
//   height = -(plane @ [pt,1]).z  with the fp32 plane matrix promoted to fp64 like numpy's float32 @ float64.
__global__ void k_fingertip_heights(const double *means, int L, const HeightArgs h)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h.n_ids) return;
    const int c = h.class_ids[i];
    h.heights[i] = (c >= 1 && c <= L) ? height_of_mode(means[(c - 1) * 2], means[(c - 1) * 2 + 1], h) : nan("");
}

} // namespace

extern "C" {

size_t rdf_mean_shift_workspace_bytes(int num_classes, int num_rounds)
{
    (void)num_classes; (void)num_rounds;
    return 0;      // every class iterates inside one workgroup: nothing is handed between workgroups any more
}

static int mean_shift_launch(const uint16_t *labels, int dim_x, int dim_y, int num_classes, const float *variances,
                             int num_rounds, double *means_out, const HeightArgs *heights, void *stream)
{
    if (dim_x < 0 || dim_y < 0 || dim_x > 65535 || dim_y > 65535 || num_classes < 0 || num_classes > kMsMaxClasses ||
        num_rounds < 0)
        return RDF_ERR_BAD_ARG;
    if (num_classes == 0) return RDF_OK;
    if ((long long)dim_x * dim_y >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    // the kernel lists every class's pixels and reads variances[0 .. num_classes) whatever the number of rounds
    if (!means_out || (((long long)dim_x * dim_y > 0) && !labels) || !variances) return RDF_ERR_NULL_PTR;
    const int lds_bytes = (int)(kMsListCap * sizeof(uint32_t));
    const void *kp = heights ? reinterpret_cast<const void *>(k_mean_shift_fused<true>)
                             : reinterpret_cast<const void *>(k_mean_shift_fused<false>);
    {   // more than 64 KB of dynamic LDS has to be allowed once per device and kernel
        static std::mutex mu;
        static unsigned long long allowed[2] = {0ull, 0ull};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return RDF_ERR_NO_DEVICE;
        std::lock_guard<std::mutex> lock(mu);
        unsigned long long &bits = allowed[heights ? 1 : 0];
        if (dev >= 64 || !((bits >> dev) & 1ull)) {
            const hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return (int)e;
            if (dev < 64) bits |= 1ull << dev;
        }
    }
    if (heights)
        hipLaunchKernelGGL(k_mean_shift_fused<true>, dim3((unsigned)num_classes), dim3(kMsThreads), lds_bytes,
                           reinterpret_cast<hipStream_t>(stream), labels, dim_x, dim_y, num_classes, variances, num_rounds,
                           means_out, *heights);
    else
        hipLaunchKernelGGL(k_mean_shift_fused<false>, dim3((unsigned)num_classes), dim3(kMsThreads), lds_bytes,
                           reinterpret_cast<hipStream_t>(stream), labels, dim_x, dim_y, num_classes, variances, num_rounds,
                           means_out, HeightArgs{});
    return (int)hipGetLastError();
}

int rdf_mean_shift(const uint16_t *labels, int dim_x, int dim_y, int num_classes, const float *variances,
                   int num_rounds, double *means_out, void *workspace, void *stream)
{
    (void)workspace;
    return mean_shift_launch(labels, dim_x, dim_y, num_classes, variances, num_rounds, means_out, nullptr, stream);
}

int rdf_fingertip_heights(const double *means, int num_classes, const int *class_ids, int n_ids,
                          const uint16_t *depth, int dim_x, int dim_y, int labels_reduce, float fx, float fy,
                          float ppx, float ppy, const float *plane, double *heights_out, void *stream)
{
    if (num_classes < 0 || n_ids < 0 || dim_x < 0 || dim_y < 0 || labels_reduce < 1) return RDF_ERR_BAD_ARG;
    if (n_ids == 0) return RDF_OK;
    if (!means || !class_ids || !depth || !plane || !heights_out) return RDF_ERR_NULL_PTR;
    const HeightArgs h = {class_ids, n_ids, depth, dim_x, dim_y, labels_reduce, fx, fy, ppx, ppy, plane, heights_out};
    hipLaunchKernelGGL(k_fingertip_heights, dim3((n_ids + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       means, num_classes, h);
    return (int)hipGetLastError();
}

int rdf_mean_shift_heights(const uint16_t *labels, int dim_x, int dim_y, int num_classes, const float *variances,
                           int num_rounds, double *means_out, const int *class_ids, int n_ids, const uint16_t *depth,
                           int depth_dim_x, int depth_dim_y, int labels_reduce, float fx, float fy, float ppx, float ppy,
                           const float *plane, double *heights_out, void *stream)
{
    if (n_ids < 0 || n_ids > kMsThreads || depth_dim_x < 0 || depth_dim_y < 0 || labels_reduce < 1) return RDF_ERR_BAD_ARG;
    if (n_ids == 0 || num_classes == 0) {
        // nothing to fuse: the two calls, one of which does nothing (no class at all: every id is "reset")
        int rc = mean_shift_launch(labels, dim_x, dim_y, num_classes, variances, num_rounds, means_out, nullptr, stream);
        if (rc != RDF_OK || n_ids == 0) return rc;
        return rdf_fingertip_heights(means_out, num_classes, class_ids, n_ids, depth, depth_dim_x, depth_dim_y,
                                     labels_reduce, fx, fy, ppx, ppy, plane, heights_out, stream);
    }
    if (!class_ids || !depth || !plane || !heights_out) return RDF_ERR_NULL_PTR;
    const HeightArgs h = {class_ids, n_ids, depth, depth_dim_x, depth_dim_y, labels_reduce, fx, fy, ppx, ppy, plane,
                          heights_out};
    return mean_shift_launch(labels, dim_x, dim_y, num_classes, variances, num_rounds, means_out, &h, stream);
}

} // extern "C"
