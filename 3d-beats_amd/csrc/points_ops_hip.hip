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
