// points_ops_hip.hip -- the element-wise kernels either side of the forest: the ones that
// define the forest's input convention (0 / filtered point -> 65535, per-hand stencil, x flip) and the
// label colouring.  Semantics follow src/cuda/points_ops.cu:117-127 (convert_0s_to_maxuint), :149-165
// (setup_depth_image_for_forest), :440-463 (stencil_depth_image_by_group), :466-483 (flip_x), :258-281
// (make_rgba_from_labels); all are byte/integer exact.  Each is a pure HBM stream: one lane per pixel
// with 2-byte accesses is latency- not bandwidth-limited at these sizes (a 848x480 frame is 814 KB), so
// lanes take 8 pixels (16 B) where alignment allows.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rdf_hip.h"

namespace {

constexpr uint32_t kNoPixel = 65535u;

__global__ __launch_bounds__(256) void k_convert_0s(uint16_t *depth, size_t n)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(depth);
    size_t head = ((16 - (addr & 15)) & 15) / 2;
    if (head > n) head = n;
    const size_t nvec = (n - head) / 8;
    uint4 *vp = reinterpret_cast<uint4 *>(depth + head);
    for (size_t i = gid; i < nvec; i += stride) {
        uint4 v = vp[i];
        uint32_t *w = reinterpret_cast<uint32_t *>(&v);
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
            any |= lo == 0u || hi == 0u;
            lo = lo == 0u ? kNoPixel : lo;
            hi = hi == 0u ? kNoPixel : hi;
            w[k] = lo | (hi << 16);
        }
        if (any) vp[i] = v;
    }
    for (size_t i = gid; i < head; i += stride) if (depth[i] == 0) depth[i] = (uint16_t)kNoPixel;
    for (size_t i = head + nvec * 8 + gid; i < n; i += stride) if (depth[i] == 0) depth[i] = (uint16_t)kNoPixel;
}

__global__ __launch_bounds__(256) void k_setup_depth(const float4 *pts, uint16_t *depth, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = depth[i];
    const float w = pts[i].w;
    if (d == 0u || w == 0.0f) depth[i] = (uint16_t)kNoPixel;
}

__global__ __launch_bounds__(256) void k_stencil(int dim_x, int dim_y, int level, int group, const uint16_t *g_in,
                                                 const uint16_t *d_in, uint16_t *d_out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dim_x || y >= dim_y) return;
    const int gw = dim_x >> level, gh = dim_y >> level;   // IMG_DIM / f with f = 1 << level
    const int gx = x >> level, gy = y >> level;
    const uint32_t g = (gx < gw && gy < gh) ? g_in[(size_t)gy * gw + gx] : 0u;   // Array2d::get default 0 when OOB
    if ((int)g != group) return;
    d_out[(size_t)y * dim_x + x] = d_in[(size_t)y * dim_x + x];
}

__global__ __launch_bounds__(256) void k_flip_x(int dim_x, int dim_y, const uint16_t *in, uint16_t *out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dim_x || y >= dim_y) return;
    out[(size_t)y * dim_x + (dim_x - (x + 1))] = in[(size_t)y * dim_x + x];
}

__global__ __launch_bounds__(256) void k_rgba(int dim_x, int dim_y, int num_colors, const uint16_t *labels,
                                              const uint32_t *colors, uint32_t *image)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dim_x || y >= dim_y) return;
    const uint32_t l = labels[(size_t)y * dim_x + x];
    if (l == 0u || l == kNoPixel || l > (uint32_t)num_colors) return;   // (> num_colors: the reference memcpy's from nullptr)
    image[(size_t)y * dim_x + x] = colors[l - 1];
}

// The app's per-hand input chain in ONE read and ONE write (3d_bz.py:396-420: fill(0) -> stencil_depth_image_by_group ->
// flip_x or copy -> convert_0s_to_maxuint): out[y][x'] = (group(x, y) == g and d != 0) ? d : 65535, x' = W-1-x when
// flipping.  One lane = eight consecutive pixels of a row (16 bytes in, 16 bytes out where the row pitch allows).
__global__ __launch_bounds__(256) void k_prepare_hand(int dim_x, int dim_y, int level, int group, const uint16_t *g_in,
                                                      const uint16_t *d_in, uint16_t *d_out, int flip, int vec_ok)
{
    const int x8 = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x0 = x8 * 8;
    if (x0 >= dim_x || y >= dim_y) return;
    const int gw = dim_x >> level, gh = dim_y >> level;
    const int gy = y >> level;
    const size_t row = (size_t)y * dim_x;
    uint32_t v[8];
    const int n = min(8, dim_x - x0);
    if (vec_ok && n == 8) {
        const uint4 w = *reinterpret_cast<const uint4 *>(d_in + row + x0);
        v[0] = w.x & 0xFFFFu; v[1] = w.x >> 16; v[2] = w.y & 0xFFFFu; v[3] = w.y >> 16;
        v[4] = w.z & 0xFFFFu; v[5] = w.z >> 16; v[6] = w.w & 0xFFFFu; v[7] = w.w >> 16;
    } else {
        for (int k = 0; k < 8; ++k) v[k] = k < n ? d_in[row + x0 + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int gx = (x0 + k) >> level;
        const uint32_t g = (k < n && gx < gw && gy < gh) ? g_in[(size_t)gy * gw + gx] : 0u;   // Array2d::get: 0 out of bounds
        if ((int)g != group || v[k] == 0u) v[k] = kNoPixel;
    }
    if (vec_ok && n == 8) {
        uint4 o;
        if (flip) {
            o = make_uint4(v[7] | (v[6] << 16), v[5] | (v[4] << 16), v[3] | (v[2] << 16), v[1] | (v[0] << 16));
            *reinterpret_cast<uint4 *>(d_out + row + (dim_x - x0 - 8)) = o;
        } else {
            o = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
            *reinterpret_cast<uint4 *>(d_out + row + x0) = o;
        }
    } else {
        for (int k = 0; k < n; ++k) d_out[row + (flip ? dim_x - 1 - (x0 + k) : x0 + k)] = (uint16_t)v[k];
    }
}

dim3 grid2d(int dim_x, int dim_y) { return dim3((dim_x + 63) / 64, (dim_y + 3) / 4); }

} // namespace

extern "C" {

int rdf_convert_0s_to_maxuint(uint16_t *depth, size_t num_pixels, void *stream)
{
    if (num_pixels == 0) return RDF_OK;
    if (!depth) return RDF_ERR_NULL_PTR;
    size_t blocks = (num_pixels / 8 + 255) / 256 + 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_convert_0s, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth,
                       num_pixels);
    return (int)hipGetLastError();
}

int rdf_setup_depth_image_for_forest(const float *pts_xyzw, uint16_t *depth, size_t num_pixels, void *stream)
{
    if (num_pixels == 0) return RDF_OK;
    if (!pts_xyzw || !depth) return RDF_ERR_NULL_PTR;
    if (num_pixels >= ((size_t)1 << 39)) return RDF_ERR_TOO_LARGE;
    hipLaunchKernelGGL(k_setup_depth, dim3((unsigned)((num_pixels + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4 *>(pts_xyzw), depth, num_pixels);
    return (int)hipGetLastError();
}

int rdf_stencil_depth_image_by_group(int dim_x, int dim_y, int mipmap_level, int group, const uint16_t *groups_in,
                                     const uint16_t *depth_in, uint16_t *depth_out, void *stream)
{
    if (dim_x < 0 || dim_y < 0 || mipmap_level < 0 || mipmap_level > 30) return RDF_ERR_BAD_ARG;
    if (dim_x == 0 || dim_y == 0) return RDF_OK;
    if (!groups_in || !depth_in || !depth_out) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_stencil, grid2d(dim_x, dim_y), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dim_x, dim_y,
                       mipmap_level, group, groups_in, depth_in, depth_out);
    return (int)hipGetLastError();
}

int rdf_prepare_hand_depth(int dim_x, int dim_y, int mipmap_level, int group, const uint16_t *groups_in,
                           const uint16_t *depth_in, uint16_t *depth_out, int flip_x, void *stream)
{
    if (dim_x < 0 || dim_y < 0 || mipmap_level < 0 || mipmap_level > 30) return RDF_ERR_BAD_ARG;
    if (dim_x == 0 || dim_y == 0) return RDF_OK;
    if (!groups_in || !depth_in || !depth_out) return RDF_ERR_NULL_PTR;
    if (depth_in == depth_out && flip_x) return RDF_ERR_BAD_ARG;   // a flip in place would read what it has overwritten
    const int vec_ok = dim_x % 8 == 0 && ((reinterpret_cast<uintptr_t>(depth_in) | reinterpret_cast<uintptr_t>(depth_out)) & 15u) == 0;
    hipLaunchKernelGGL(k_prepare_hand, dim3((dim_x + 511) / 512, (dim_y + 3) / 4), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), dim_x, dim_y, mipmap_level, group, groups_in, depth_in,
                       depth_out, flip_x ? 1 : 0, vec_ok);
    return (int)hipGetLastError();
}

int rdf_flip_x(int dim_x, int dim_y, const uint16_t *in, uint16_t *out, void *stream)
{
    if (dim_x < 0 || dim_y < 0) return RDF_ERR_BAD_ARG;
    if (dim_x == 0 || dim_y == 0) return RDF_OK;
    if (!in || !out) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_flip_x, grid2d(dim_x, dim_y), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dim_x, dim_y, in,
                       out);
    return (int)hipGetLastError();
}

int rdf_make_rgba_from_labels(int dim_x, int dim_y, int num_colors, const uint16_t *labels, const uint8_t *colors_rgba,
                              uint8_t *image_rgba, void *stream)
{
    if (dim_x < 0 || dim_y < 0 || num_colors < 0) return RDF_ERR_BAD_ARG;
    if (dim_x == 0 || dim_y == 0) return RDF_OK;
    if (!labels || !image_rgba || (num_colors > 0 && !colors_rgba)) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_rgba, grid2d(dim_x, dim_y), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dim_x, dim_y,
                       num_colors, labels, reinterpret_cast<const uint32_t *>(colors_rgba),
                       reinterpret_cast<uint32_t *>(image_rgba));
    return (int)hipGetLastError();
}

} // extern "C"
