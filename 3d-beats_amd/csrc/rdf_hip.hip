// rdf_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for per-pixel randomized-decision-forest
// inference.  Written from the semantics of 3d-beats' tree_eval.cu (see include/rdf_hip.h for the
// reference lines each entry point replaces); not a translation of it:
//
//   * one LANE owns one label pixel and walks kGroup trees interleaved, level-synchronously, so
//     that kGroup node fetches and 2*kGroup depth probes are in flight per lane (the reference
//     uses one thread per (pixel,tree), shared-memory float atomics and two block barriers);
//   * a wave is 64 consecutive label pixels of one row: the centre-depth read, the label store
//     and -- for smooth surfaces -- each probe are one or two 128-byte lines;
//   * per-tree leaf PDFs are added in registers in tree order (canonical order, no atomics);
//   * the top levels of every tree live in LDS as 32-byte records {s*u, s*v, thresh, flags};
//     deeper levels are fetched from a packed 32-byte-record table (rdf_forest_pack) or, for
//     the unpacked entry point, straight from the reference's 7+2C-float records;
//   * tiles are dealt to workgroups through an XCD-aware bijective remap so the tiles of one
//     frame share one XCD's L2.
//
// Bit-exactness: (s*u)/d is one fp32 multiply and one IEEE-correct fp32 divide (hipcc's default
// v_div_scale/v_div_fmas/v_div_fixup sequence; never build this file with -ffast-math), floor +
// saturating convert is v_floor_f32 + v_cvt_i32_f32, coordinate adds wrap, bounds are checked per axis.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rdf_hip.h"

namespace {

constexpr int kGroup = 4;          // trees walked interleaved by one lane
constexpr int kDefaultLdsBudget = 80 * 1024;
constexpr uint32_t kNoPixel = 65535u;

struct alignas(16) NodeRec {       // 32 bytes
    float sux, suy, svx, svy;      // scale_factor * (u.x, u.y, v.x, v.y)
    float thresh;
    uint32_t flags;                // bit0: left child continues, bit1: right child continues
    uint32_t pad0, pad1;
};
static_assert(sizeof(NodeRec) == 32, "NodeRec must be 32 bytes");

struct EvalArgs {
    const uint16_t *depth;
    const float *forest;
    const NodeRec *packed;
    const uint16_t *filter;
    uint16_t *labels;
    unsigned long long *stats;
    uint32_t total;        // label pixels in this launch
    uint32_t n_tiles;
    uint32_t per_img_l;    // Wl*Hl
    uint32_t per_img_d;    // W*H
    int W, H, Wl, r;
    int T, D, C, E;
    int nodes;             // 2^D - 1
    int lds_levels;        // top levels held in LDS
    int filter_class;
    int keep_if_no_leaf;   // single-tree semantics: no leaf reached -> pixel untouched
    float s;
};

// __float2int_rd: floor, then saturating convert with NaN -> 0 (v_floor_f32 + v_cvt_i32_f32).
// The fused v_cvt_flr_i32_f32 is NOT equivalent: measured on gfx950 it maps NaN to INT_MAX.
// The convert is inline asm because a C++ float->int cast is undefined outside int range.
__device__ __forceinline__ int floor_i32(float f)
{
    int r;
    const float fl = __builtin_floorf(f);
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(fl));
    return r;
}

__device__ __forceinline__ int add_wrap(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }

// floor(x) == -1  <=>  -1 <= x < 0   (NaN: false).  tree_eval.cu:101-102 / :186-187.
__device__ __forceinline__ uint32_t child_flags(float l, float r)
{
    return ((l >= -1.0f && l < 0.0f) ? 1u : 0u) | ((r >= -1.0f && r < 0.0f) ? 2u : 0u);
}

// XCD-aware bijective remap of a physical slot (slot % 8 = XCD group under round-robin dispatch)
// to a logical tile, so that each XCD group owns one contiguous run of tiles.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t slot, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, x = slot & 7u, o = slot >> 3;
    const uint32_t base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + o;
}

// Depth probe with per-axis bounds check, out of bounds -> 65535 (cu_utils.hpp:79-86).
__device__ __forceinline__ float probe(const uint16_t *depth, uint32_t img_off, int x, int y, int W, int H)
{
    const bool inb = (uint32_t)x < (uint32_t)W && (uint32_t)y < (uint32_t)H;
    const uint32_t off = inb ? img_off + (uint32_t)y * (uint32_t)W + (uint32_t)x : img_off;
    const uint32_t v = depth[off];
    return inb ? (float)v : 65535.0f;
}

template <bool PACKED>
__device__ __forceinline__ NodeRec load_global_node(const EvalArgs &a, int tree, uint32_t idx)
{
    NodeRec n;
    if (PACKED) {
        const NodeRec *p = a.packed + (size_t)tree * (size_t)a.nodes + idx;
        const float4 v = *reinterpret_cast<const float4 *>(p);
        const float2 w = *reinterpret_cast<const float2 *>(&p->thresh);
        n.sux = v.x; n.suy = v.y; n.svx = v.z; n.svy = v.w;
        n.thresh = w.x; n.flags = __float_as_uint(w.y);
    } else {
        const float *p = a.forest + ((size_t)tree * (size_t)a.nodes + idx) * (size_t)a.E;
        n.sux = a.s * p[0]; n.suy = a.s * p[1]; n.svx = a.s * p[2]; n.svy = a.s * p[3];
        n.thresh = p[4];
        n.flags = child_flags(p[5], p[6]);
    }
    n.pad0 = n.pad1 = 0;
    return n;
}

template <int BLOCK, bool PACKED, int CMAX, bool STATS>
__global__ __launch_bounds__(BLOCK) void k_eval_forest(const EvalArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    NodeRec *lds = reinterpret_cast<NodeRec *>(lds_raw);

    const int tid = threadIdx.x;
    const int K = a.lds_levels;
    const uint32_t nodes_lds = (1u << K) - 1u;

    // ---- stage the top K levels of every tree (level order => the first 2^K-1 records) ----
    for (uint32_t i = tid; i < (uint32_t)a.T * nodes_lds; i += BLOCK) {
        const uint32_t k = i / nodes_lds, n = i - k * nodes_lds;
        lds[i] = load_global_node<PACKED>(a, (int)k, n);
    }
    __syncthreads();

    unsigned long long st_px = 0, st_lv = 0, st_lf = 0;

    for (uint32_t slot = blockIdx.x; slot < a.n_tiles; slot += gridDim.x) {
        const uint32_t tile = xcd_tile(slot, a.n_tiles);
        const uint32_t i = tile * BLOCK + tid;
        if (i >= a.total) continue;

        const uint32_t img = i / a.per_img_l;
        const uint32_t rem = i - img * a.per_img_l;
        const int ly = (int)(rem / (uint32_t)a.Wl);
        const int lx = (int)(rem - (uint32_t)ly * (uint32_t)a.Wl);
        const int x = lx * a.r, y = ly * a.r;
        const uint32_t img_off = img * a.per_img_d;

        if (a.filter_class != -1) {
            if ((int)a.filter[i] != a.filter_class) continue;
        }
        const uint32_t d = a.depth[img_off + (uint32_t)y * (uint32_t)a.W + (uint32_t)x];
        if (d == 0u || d == kNoPixel) continue;
        const float df = (float)d;

        float best = 0.0f;
        int best_c = 0;
        bool any_leaf = false;

        for (int c0 = 0; c0 < a.C || c0 == 0; c0 += CMAX) {
            float pdf[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) pdf[c] = 0.0f;

            for (int kb = 0; kb < a.T; kb += kGroup) {
                uint32_t g[kGroup];
                int leaf[kGroup];
                bool act[kGroup];
#pragma unroll
                for (int k = 0; k < kGroup; ++k) {
                    g[k] = 0u;
                    leaf[k] = -1;
                    act[k] = (kb + k) < a.T;
                }

                for (int j = 0; j < a.D; ++j) {
                    bool any = false;
#pragma unroll
                    for (int k = 0; k < kGroup; ++k) any |= act[k];
                    if (!__any(any)) break;

                    const uint32_t lvl = (1u << j) - 1u;
                    NodeRec n[kGroup];
                    if (j < K) {
#pragma unroll
                        for (int k = 0; k < kGroup; ++k) {
                            const uint32_t t = act[k] ? (uint32_t)(kb + k) : 0u;
                            const uint32_t idx = act[k] ? lvl + g[k] : 0u;
                            const NodeRec *p = lds + t * nodes_lds + idx;
                            const float4 v = *reinterpret_cast<const float4 *>(p);
                            const float2 w = *reinterpret_cast<const float2 *>(&p->thresh);
                            n[k].sux = v.x; n[k].suy = v.y; n[k].svx = v.z; n[k].svy = v.w;
                            n[k].thresh = w.x; n[k].flags = __float_as_uint(w.y);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < kGroup; ++k) {
                            const int t = act[k] ? (kb + k) : 0;
                            const uint32_t idx = act[k] ? lvl + g[k] : 0u;
                            n[k] = load_global_node<PACKED>(a, t, idx);
                        }
                    }

                    float pu[kGroup], pv[kGroup];
#pragma unroll
                    for (int k = 0; k < kGroup; ++k) {
                        const int ux = add_wrap(x, floor_i32(n[k].sux / df));
                        const int uy = add_wrap(y, floor_i32(n[k].suy / df));
                        const int vx = add_wrap(x, floor_i32(n[k].svx / df));
                        const int vy = add_wrap(y, floor_i32(n[k].svy / df));
                        pu[k] = probe(a.depth, img_off, ux, uy, a.W, a.H);
                        pv[k] = probe(a.depth, img_off, vx, vy, a.W, a.H);
                    }

#pragma unroll
                    for (int k = 0; k < kGroup; ++k) {
                        if (act[k]) {
                            if (STATS && c0 == 0) st_lv++;
                            const float f = pu[k] - pv[k];
                            const bool left = f < n[k].thresh;
                            const bool cont = (n[k].flags & (left ? 1u : 2u)) != 0u;
                            const uint32_t side = left ? 0u : 1u;
                            if (cont) {
                                g[k] = g[k] * 2u + side;
                            } else {
                                leaf[k] = (int)(((lvl + g[k]) << 1) | side);
                                act[k] = false;
                            }
                        }
                    }
                }

                // leaf PDFs, strictly in tree order (canonical sum order)
#pragma unroll
                for (int k = 0; k < kGroup; ++k) {
                    if (leaf[k] >= 0) {
                        any_leaf = true;
                        if (STATS && c0 == 0) st_lf++;
                        const float *pp = a.forest +
                            ((size_t)(kb + k) * (size_t)a.nodes + (uint32_t)(leaf[k] >> 1)) * (size_t)a.E +
                            7 + (leaf[k] & 1) * a.C + c0;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) {
                            if (c0 + c < a.C) pdf[c] = pdf[c] + pp[c];
                        }
                    }
                }
            }

#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c0 + c < a.C && pdf[c] > best) {
                    best = pdf[c];
                    best_c = c0 + c;
                }
            }
        }

        if (STATS) st_px++;
        if (a.keep_if_no_leaf && !any_leaf) continue;
        a.labels[i] = (uint16_t)best_c;
    }

    if (STATS) {
        // wave reduction, one atomic per wave and counter
        for (int o = 32; o > 0; o >>= 1) {
            st_px += __shfl_down(st_px, o);
            st_lv += __shfl_down(st_lv, o);
            st_lf += __shfl_down(st_lf, o);
        }
        if ((tid & 63) == 0) {
            atomicAdd(a.stats + 0, st_px);
            atomicAdd(a.stats + 1, st_lv);
            atomicAdd(a.stats + 2, st_lf);
        }
    }
}

// ---- load-time repack: one thread per node ----
__global__ __launch_bounds__(256) void k_pack(const float *forest, NodeRec *packed, size_t total_nodes, int E, float s)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_nodes) return;
    const float *p = forest + i * (size_t)E;
    NodeRec n;
    n.sux = s * p[0]; n.suy = s * p[1]; n.svx = s * p[2]; n.svy = s * p[3];
    n.thresh = p[4];
    n.flags = child_flags(p[5], p[6]);
    n.pad0 = n.pad1 = 0;
    packed[i] = n;
}

// ---- composite (tree_eval.cu:214-248): one lane per label pixel ----
__global__ __launch_bounds__(256) void k_composite(const uint16_t *const *imgs, int n_images, uint32_t n_px,
                                                   const int2 *cond, int n_cond, uint16_t *out, int32_t *bad)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_px) return;
    long long off = 0;
    for (int i = 0; i < n_images; ++i) {
        const uint32_t l = imgs[i][p];
        if (l == 0u || l == kNoPixel) return;
        const long long e = off + (long long)l - 1;
        if (e < 0 || e >= n_cond) {
            if (bad) atomicAdd(bad, 1);
            return;
        }
        const int2 tv = cond[e];
        if (tv.x == 0) {
            out[p] = (uint16_t)tv.y;
            return;
        }
        off = tv.y;
    }
    if (bad) atomicAdd(bad, 1);
}

__global__ __launch_bounds__(256) void k_fill_u16(uint16_t *dst, size_t n, uint16_t v)
{
    // 8 elements (16 B) per lane where aligned, scalar head/tail otherwise
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(dst);
    size_t head = ((16 - (addr & 15)) & 15) / 2;
    if (head > n) head = n;
    const size_t nvec = (n - head) / 8;
    const uint32_t w = (uint32_t)v | ((uint32_t)v << 16);
    uint4 *vp = reinterpret_cast<uint4 *>(dst + head);
    for (size_t i = gid; i < nvec; i += stride) vp[i] = make_uint4(w, w, w, w);
    const size_t tail0 = head + nvec * 8;
    for (size_t i = gid; i < head; i += stride) dst[i] = v;
    for (size_t i = tail0 + gid; i < n; i += stride) dst[i] = v;
}

__global__ void k_debug_floor(const float *in, int32_t *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = floor_i32(in[i]);
}

__global__ void k_debug_div(const float *num, const float *den, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = num[i] / den[i];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int g_lds_budget = 0;
int g_block_threads = 0;

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int lds_budget()
{
    int b = g_lds_budget > 0 ? g_lds_budget : env_int("RDF_LDS_BUDGET", kDefaultLdsBudget);
    if (b > 160 * 1024) b = 160 * 1024;
    if (b < 0) b = 0;
    return b;
}

struct DeviceInfo {
    int cus = 0;
    bool ok = false;
};

int device_info(DeviceInfo *out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    out->cus = cus > 0 ? cus : 256;
    out->ok = true;
    return 0;
}

template <int BLOCK, bool PACKED, int CMAX, bool STATS>
int launch_variant(const EvalArgs &a, int lds_bytes, int grid, hipStream_t st)
{
    auto kern = k_eval_forest<BLOCK, PACKED, CMAX, STATS>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds_bytes, st, a);
    return (int)hipGetLastError();
}

template <int BLOCK, bool PACKED, bool STATS>
int launch_cmax(const EvalArgs &a, int lds_bytes, int grid, hipStream_t st)
{
    if (a.C <= 4) return launch_variant<BLOCK, PACKED, 4, STATS>(a, lds_bytes, grid, st);
    if (a.C <= 8) return launch_variant<BLOCK, PACKED, 8, STATS>(a, lds_bytes, grid, st);
    return launch_variant<BLOCK, PACKED, 16, STATS>(a, lds_bytes, grid, st);
}

template <bool PACKED, bool STATS>
int launch_block(int block, const EvalArgs &a, int lds_bytes, int grid, hipStream_t st)
{
    switch (block) {
    case 256: return launch_cmax<256, PACKED, STATS>(a, lds_bytes, grid, st);
    case 512: return launch_cmax<512, PACKED, STATS>(a, lds_bytes, grid, st);
    default: return launch_cmax<1024, PACKED, STATS>(a, lds_bytes, grid, st);
    }
}

int check_common(const void *depth, int n_img, int dim_x, int dim_y, const void *forest, int n_trees,
                 int max_depth, int n_classes, const void *labels, int r)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_trees < 0 || max_depth < 0 || max_depth > 30 ||
        n_classes < 0 || r < 1)
        return RDF_ERR_BAD_ARG;
    if ((long long)n_img * dim_x * dim_y >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    const long long total = (long long)n_img * (dim_x / r) * (dim_y / r);
    if (total == 0) return 1; // nothing to do
    if (!depth || !labels) return RDF_ERR_NULL_PTR;
    if (!forest && n_trees > 0 && max_depth > 0) return RDF_ERR_NULL_PTR;
    return 0;
}

int eval_common(const uint16_t *depth, int n_img, int dim_x, int dim_y, const NodeRec *packed,
                const float *forest, int n_trees, int max_depth, int n_classes, const uint16_t *filter,
                int filter_class, uint16_t *labels_out, int r, float s, int keep_if_no_leaf,
                unsigned long long *stats, void *stream)
{
    int rc = check_common(depth, n_img, dim_x, dim_y, forest, n_trees, max_depth, n_classes, labels_out, r);
    if (rc == 1) return RDF_OK;
    if (rc != 0) return rc;
    if (filter_class != -1 && !filter) return RDF_ERR_NULL_PTR;

    DeviceInfo di;
    rc = device_info(&di);
    if (rc != 0) return rc;

    EvalArgs a;
    memset(&a, 0, sizeof(a));
    a.depth = depth; a.forest = forest; a.packed = packed; a.filter = filter; a.labels = labels_out;
    a.stats = stats;
    a.W = dim_x; a.H = dim_y; a.Wl = dim_x / r; a.r = r;
    const int Hl = dim_y / r;
    a.per_img_l = (uint32_t)a.Wl * (uint32_t)Hl;
    a.per_img_d = (uint32_t)dim_x * (uint32_t)dim_y;
    a.total = a.per_img_l * (uint32_t)n_img;
    a.T = n_trees; a.D = max_depth; a.C = n_classes; a.E = 7 + 2 * n_classes;
    a.nodes = (int)((1ll << max_depth) - 1);
    a.filter_class = filter_class;
    a.keep_if_no_leaf = keep_if_no_leaf;
    a.s = s;

    // top levels that fit the LDS budget: T * (2^K - 1) * 32 B <= budget
    int K = 0;
    const long long budget = lds_budget();
    while (K < max_depth && (long long)n_trees * ((1ll << (K + 1)) - 1) * 32 <= budget) ++K;
    a.lds_levels = K;
    const int lds_bytes = (int)((long long)n_trees * ((1ll << K) - 1) * 32);

    int block = g_block_threads > 0 ? g_block_threads : env_int("RDF_BLOCK", 0);
    if (block != 256 && block != 512 && block != 1024) {
        // small launches: smaller workgroups spread over more CUs
        const long long t1024 = ((long long)a.total + 1023) / 1024;
        block = t1024 >= 2ll * di.cus ? 1024 : (t1024 * 2 >= 2ll * di.cus ? 512 : 256);
    }
    a.n_tiles = (uint32_t)(((long long)a.total + block - 1) / block);
    const int per_cu = block == 1024 ? 2 : (block == 512 ? 4 : 8);
    long long grid = (long long)di.cus * per_cu; // multiple of 8 on MI355X (256 CUs)
    grid -= grid % 8;
    if (grid < 8) grid = 8;
    if ((long long)a.n_tiles < grid) grid = a.n_tiles;

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (stats) {
        return packed ? launch_block<true, true>(block, a, lds_bytes, (int)grid, st)
                      : launch_block<false, true>(block, a, lds_bytes, (int)grid, st);
    }
    return packed ? launch_block<true, false>(block, a, lds_bytes, (int)grid, st)
                  : launch_block<false, false>(block, a, lds_bytes, (int)grid, st);
}

} // namespace

extern "C" {

int rdf_eval_forest(const uint16_t *depth, int n_img, int dim_x, int dim_y, const float *forest,
                    int n_trees, int max_depth, int n_classes, const uint16_t *filter, int filter_class,
                    uint16_t *labels_out, int labels_reduce, float scale_factor, void *stream)
{
    return eval_common(depth, n_img, dim_x, dim_y, nullptr, forest, n_trees, max_depth, n_classes, filter,
                       filter_class, labels_out, labels_reduce, scale_factor, 0, nullptr, stream);
}

int rdf_eval_forest_stats(const uint16_t *depth, int n_img, int dim_x, int dim_y, const float *forest,
                          int n_trees, int max_depth, int n_classes, const uint16_t *filter,
                          int filter_class, uint16_t *labels_out, int labels_reduce, float scale_factor,
                          unsigned long long *stats, void *stream)
{
    if (!stats) return RDF_ERR_NULL_PTR;
    return eval_common(depth, n_img, dim_x, dim_y, nullptr, forest, n_trees, max_depth, n_classes, filter,
                       filter_class, labels_out, labels_reduce, scale_factor, 0, stats, stream);
}

int rdf_eval_tree(const uint16_t *depth, int n_img, int dim_x, int dim_y, const float *tree, int max_depth,
                  int n_classes, uint16_t *labels_out, void *stream)
{
    return eval_common(depth, n_img, dim_x, dim_y, nullptr, tree, 1, max_depth, n_classes, nullptr, -1,
                       labels_out, 1, 1.0f, 1, nullptr, stream);
}

size_t rdf_forest_packed_bytes(int n_trees, int max_depth)
{
    if (n_trees < 0 || max_depth < 0 || max_depth > 30) return 0;
    return (size_t)n_trees * (size_t)((1ll << max_depth) - 1) * sizeof(NodeRec);
}

int rdf_forest_pack(const float *forest, int n_trees, int max_depth, int n_classes, float scale_factor,
                    void *packed, void *stream)
{
    if (n_trees < 0 || max_depth < 0 || max_depth > 30 || n_classes < 0) return RDF_ERR_BAD_ARG;
    const size_t total = (size_t)n_trees * (size_t)((1ll << max_depth) - 1);
    if (total == 0) return RDF_OK;
    if (!forest || !packed) return RDF_ERR_NULL_PTR;
    const size_t blocks = (total + 255) / 256;
    if (blocks >= (1ull << 31)) return RDF_ERR_TOO_LARGE;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       forest, reinterpret_cast<NodeRec *>(packed), total, 7 + 2 * n_classes, scale_factor);
    return (int)hipGetLastError();
}

int rdf_eval_forest_packed(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                           const float *forest, int n_trees, int max_depth, int n_classes,
                           const uint16_t *filter, int filter_class, uint16_t *labels_out,
                           int labels_reduce, void *stream)
{
    if (!packed && n_trees > 0 && max_depth > 0) return RDF_ERR_NULL_PTR;
    if (!packed) // degenerate forest: nothing to walk, the unpacked path handles it
        return eval_common(depth, n_img, dim_x, dim_y, nullptr, forest, n_trees, max_depth, n_classes, filter,
                           filter_class, labels_out, labels_reduce, 1.0f, 0, nullptr, stream);
    return eval_common(depth, n_img, dim_x, dim_y, reinterpret_cast<const NodeRec *>(packed), forest, n_trees,
                       max_depth, n_classes, filter, filter_class, labels_out, labels_reduce, 1.0f, 0, nullptr,
                       stream);
}

int rdf_composite(const uint16_t *const *label_images, int n_images, int dim_x, int dim_y, const int32_t *cond,
                  int n_cond, uint16_t *out, int32_t *bad_count, void *stream)
{
    if (n_images < 0 || dim_x < 0 || dim_y < 0 || n_cond < 0) return RDF_ERR_BAD_ARG;
    const long long n_px = (long long)dim_x * dim_y;
    if (n_px == 0) return RDF_OK;
    if (n_px >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    if (!out || (n_images > 0 && !label_images) || (n_cond > 0 && !cond)) return RDF_ERR_NULL_PTR;
    const unsigned blocks = (unsigned)((n_px + 255) / 256);
    hipLaunchKernelGGL(k_composite, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       label_images, n_images, (uint32_t)n_px, reinterpret_cast<const int2 *>(cond), n_cond, out,
                       bad_count);
    return (int)hipGetLastError();
}

int rdf_fill_u16(uint16_t *dst, size_t n, uint16_t value, void *stream)
{
    if (n == 0) return RDF_OK;
    if (!dst) return RDF_ERR_NULL_PTR;
    size_t blocks = (n / 8 + 255) / 256 + 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_u16, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       dst, n, value);
    return (int)hipGetLastError();
}

int rdf_debug_floor_i32(const float *in, int32_t *out, size_t n, void *stream)
{
    if (n == 0) return RDF_OK;
    if (!in || !out) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_debug_floor, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in, out, n);
    return (int)hipGetLastError();
}

int rdf_debug_div_f32(const float *num, const float *den, float *out, size_t n, void *stream)
{
    if (n == 0) return RDF_OK;
    if (!num || !den || !out) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_debug_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), num, den, out, n);
    return (int)hipGetLastError();
}

void rdf_set_lds_budget_bytes(int bytes) { g_lds_budget = bytes; }
void rdf_set_block_threads(int threads) { g_block_threads = threads; }

int rdf_event_create(void **event)
{
    if (!event) return RDF_ERR_NULL_PTR;
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return (int)rc;
    *event = e;
    return RDF_OK;
}
int rdf_event_record(void *event, void *stream)
{
    return (int)hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream));
}
int rdf_event_synchronize(void *event) { return (int)hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)); }
int rdf_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!ms) return RDF_ERR_NULL_PTR;
    return (int)hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
}
int rdf_event_destroy(void *event) { return (int)hipEventDestroy(reinterpret_cast<hipEvent_t>(event)); }
int rdf_stream_synchronize(void *stream) { return (int)hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)); }

int rdf_abi_version(void) { return RDF_ABI_VERSION; }

const char *rdf_error_string(int code)
{
    switch (code) {
    case RDF_OK: return "ok";
    case RDF_ERR_BAD_ARG: return "rdf: bad argument";
    case RDF_ERR_NULL_PTR: return "rdf: required pointer is NULL";
    case RDF_ERR_TOO_LARGE: return "rdf: call addresses >= 2^31 pixels, split the batch";
    case RDF_ERR_NO_DEVICE: return "rdf: no usable HIP device";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "rdf: unknown error";
}

} // extern "C"
