// tree_train_hip.hip -- gfx950 kernels for training one randomized decision tree.
//
// Reference: src/cuda/tree_train.cu (evaluate_random_features :4-64, gini helpers :66-97,
// pick_best_features :99-236, get_active_nodes_next_level :238-273, copy_pixel_groups :275-324) driven
// level by level from src/decision_tree.py:444-600.  Not a translation:
//   * the histogram kernel keeps one PIXEL per lane and loops over the proposals (wave-uniform, read
//     through the scalar cache) with the pixel's depth neighbourhood staged in LDS, where the
//     reference spends one thread and one 64-bit global atomic per (pixel, proposal);
//   * a wave covers an 8x8 block of pixels and counts with ballots and popcounts: one atomic per
//     (node, class) group the block touches instead of one per pixel -- and, on the fast path
//     (rdf_train_histogram_left_ws), only for the left children (the right ones follow from the
//     parents) and for two proposals at a time (32-bit halves of one 64-bit word), four where the
//     (node, class) holds at most 65535 pixels (16-bit fields).  The rate of
//     scattered global atomics (24e9/s, tools/ubench_atomic.hip) is what bounds this kernel;
//   * the next level's node list is built by an ordered scan (the reference appends with an atomic
//     counter, i.e. in scheduler order), so the whole training run is reproducible bit for bit;
//   * floor((u)/d) uses the shared-reciprocal divide verified exhaustively for inference
//     (tools/verify_fastdiv.hip) with the IEEE divide for numerators outside the verified range.
// Counts are integers, so they are exact whatever the order; the fp32 gain arithmetic follows the
// reference expression by expression (built with -ffp-contract=off).  Limit: the reference's kernels go through nvcc,
// whose default (--fmad=true) contracts p + p_i*p_i and the weighted impurity sum into FMAs; a gain can differ from the
// reference binary's in the last ulp and a strict `>` on a near-tie can then pick another proposal.  Trained trees are
// bit-identical to oracle/train_numpy.py (the same unfused arithmetic), not provably to what the reference binary would
// train.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/rdf_hip.h"
#include "rdf_device.hpp"

namespace {

typedef unsigned long long u64;
constexpr int kHalo = 16;
// A wave covers a kBW x kBH block of pixels rather than a 64 x 1 row segment: pixels of the same (node, class) are
// compact in 2-D, so a block holds fewer distinct groups -- and issues fewer atomics -- than a strip.  The
// workgroup's 4 waves sit side by side.
constexpr int kBW = 8, kBH = 8;
static_assert(kBW * kBH == 64, "one pixel per lane");
constexpr int kCols = 4 * kBW;                // pixels per tile row
constexpr int kRows = kBH;
constexpr int kTW = kCols + 2 * kHalo;        // staged tile columns
constexpr int kTH = kRows + 2 * kHalo;        // staged tile rows
constexpr int kMaxClasses = 64;
constexpr int kBatch = 4;                     // proposals evaluated together: 8 probes in flight per lane

// Tile queue of the histogram kernel: {next tile, workgroups finished}.  Zero at rest: the last workgroup of a launch
// puts it back (the host clears it before every launch all the same).  One queue per device: histogram launches of one device must not overlap
// (the trainer issues them on one stream).  Labelled pixels cluster (a hand covers ~15 % of a frame), so tiles handed
// out by static striding left the average SIMD with ONE resident wave while the unluckiest workgroup finished
// (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE, level 0: 3.2 ms against a 1.4 ms VALU floor).
__device__ unsigned int g_train_queue[2];

// ---- root counts + nodes_by_pixel initialisation (decision_tree.py:452-468, done on the host there) ----
__global__ __launch_bounds__(256) void k_train_init(const uint16_t *labels, size_t n_px, int C, int32_t *nodes, u64 *root)
{
    __shared__ unsigned int s_cnt[kMaxClasses];
    if (threadIdx.x < kMaxClasses) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_px; i += (size_t)gridDim.x * 256) {
        const uint32_t l = labels[i];
        nodes[i] = l > 0u ? 0 : -1;
        if (l > 0u && l < (uint32_t)C) atomicAdd(&s_cnt[l], 1u);
    }
    __syncthreads();
    if (threadIdx.x < C && s_cnt[threadIdx.x]) atomicAdd(root + threadIdx.x, (u64)s_cnt[threadIdx.x]);
}

struct HistArgs {
    const uint16_t *depth, *labels;
    const int32_t *nodes;
    const float *props;   // [P][5]
    u64 *counts;          // [P][NB][C]
    uint32_t n_tiles, tiles_x, tiles_y;
    int W, H, P, C, NB, node_start, node_end;
    int left_only;   // count the left children only (the right ones follow from the parents, k_train_right_counts)
    uint32_t *tmp;   // PAIRS: 32-bit counters [pair][bin][2], two proposals per 64-bit atomic (k_train_unpack_pairs),
                     // followed by 16-bit counters [quad][bin][4], four proposals per atomic (k_train_unpack_quads)
    int Ppad;
    const u64 *parent_counts;   // [node][class] of the live pixels, or null: which bins may use 16-bit counters
};

// evaluate_random_features (tree_train.cu:4-64).
// Counts go straight to the global histogram, aggregated per wave first (see "peers" below).  A workgroup-
// private LDS histogram for the upper levels was measured and dropped: with the per-wave aggregation it was
// no faster at any level (r01 notes in DESIGN.md).
// PAIRS: left children only, into a.tmp: the counters of proposals j and j+1 of one bin are the two halves of one
// 64-bit word, so ONE atomic serves two proposals (a bin's total is < 2^31 pixels: no carry between the halves).
template <bool PAIRS>
__global__ __launch_bounds__(256) void k_train_histogram(const HistArgs a)
{
    __shared__ uint16_t s_buf[8 + kTH * kTW];
    uint16_t *s_tile = s_buf + 8;   // s_tile[-1] = 65535: what a probe outside the tile reads (rdf_device.hpp)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per_img = (uint32_t)a.W * (uint32_t)a.H;

    // Are all numerators of all proposals inside the range the shared-reciprocal divide is verified for?  (They
    // are, for anything make_random_features draws.)  Decided once per workgroup: testing 4 numerators per
    // proposal per wave in the loop below made the kernel scalar-ALU bound.
    if (threadIdx.x == 0) s_buf[7] = (uint16_t)kNoPixel;   // (visible after the barrier below)
    bool mine_ok = true;
    for (int i = tid; i < a.P * 5; i += 256)
        if (i % 5 != 4) mine_ok = mine_ok && fast_divide_ok(a.props[i]);
    const bool all_fast = __syncthreads_and(mine_ok ? 1 : 0) != 0;

    const int j0 = 0, j1 = a.P;
    __shared__ uint32_t s_next;
    for (;;) {
        if (tid == 0) s_next = atomicAdd(&g_train_queue[0], 1u);
        __syncthreads();
        const uint32_t tile = s_next;
        __syncthreads();   // (s_next is rewritten in the next round)
        if (tile >= a.n_tiles) break;
        const uint32_t per = a.tiles_x * a.tiles_y;
        const uint32_t img = tile / per, rem = tile - img * per;
        const uint32_t ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int x = (int)(tx * kCols) + wave * kBW + (lane % kBW), y = (int)(ty * kRows) + lane / kBW;
        const int tx0 = (int)(tx * kCols) - kHalo, ty0 = (int)(ty * kRows) - kHalo;
        const size_t img_off = (size_t)img * per_img;

        // does this tile hold any pixel that counts?  (most of a frame is unlabelled background)
        int node = -1;
        uint32_t label = 0u;
        if (x < a.W && y < a.H) {
            const size_t i = img_off + (size_t)y * a.W + x;
            node = a.nodes[i];
            if (node >= 0 && (node * 2 < a.node_start || node * 2 + 1 >= a.node_end)) node = -1;
            if (node >= 0) label = a.labels[i];
            if (label >= (uint32_t)a.C) node = -1;   // (the reference would index out of bounds)
        }
        if (!__syncthreads_or(node >= 0 ? 1 : 0)) continue;   // (also: the previous tile's readers are done)

        for (int row = wave; row < kTH; row += 4) {
            const int gy = ty0 + row;
            const bool row_in = (uint32_t)gy < (uint32_t)a.H;
            for (int col = lane; col < kTW; col += 64) {
                const int gx = tx0 + col;
                uint32_t v = kNoPixel;
                if (row_in && (uint32_t)gx < (uint32_t)a.W) v = a.depth[img_off + (size_t)gy * a.W + gx];
                s_tile[row * kTW + col] = (uint16_t)v;
            }
        }
        __syncthreads();

        const bool live = node >= 0;
        if (!__any(live)) continue;   // wave-uniform; no barrier below this point in the iteration
        const int xl = x - tx0, yl = y - ty0;
        const uint32_t d = live ? s_tile[yl * kTW + xl] : 1u;
        const float df = (float)(d == 0u ? 1u : d);
        const float r0 = __builtin_amdgcn_rcpf(df);
        const float rcp = __builtin_fmaf(__builtin_fmaf(-df, r0, 1.0f), r0, r0);
        const bool zero_depth = d == 0u;              // compute_feature returns 0.f (decision_tree_common.hpp:12)

        // Lanes that share (node, class) form a group; per proposal each group splits into a left and a right
        // part and only the lowest lane of each part adds the part's size.  That is the fewest atomics a wave
        // can issue without knowing the bins in advance: one per (node, class, side) it touches instead of one
        // per pixel (deep levels were bound by ~1e9 global atomics per level before).
        const u64 live_mask = __ballot(live);
        const int key = node * a.C + (int)label;
        u64 peers = 0ull;
        for (u64 todo = live_mask; todo;) {               // wave-uniform loop, once per tile
            const int src = __ffsll((long long)todo) - 1;
            const int k0 = __builtin_amdgcn_readfirstlane(__shfl(key, src));
            const u64 m = __ballot(live && key == k0);
            if (key == k0) peers = m;
            todo &= ~m;
        }
        const int bin0 = (node * 2 - a.node_start) * a.C + (int)label;   // left child's bin; right child's is + C
        // a (node, class) with at most 65535 pixels can count four proposals in the 16-bit fields of one word
        const bool small = PAIRS && a.parent_counts != nullptr && live &&
                           a.parent_counts[(size_t)node * a.C + label] <= 65535ull;

        const ProbeCtx pc = {s_tile, reinterpret_cast<const char *>(a.depth) + img_off * 2, tx0, ty0, kTW, kTH, kTW,
                             a.W, a.H};
        const f2 r2 = {rcp, rcp}, nd = {-df, -df};

        // kBatch proposals at a time: all their probes are issued before any is consumed (a far probe is a
        // global round trip; waiting for each one inside its branch made this kernel 10x slower)
        for (int jb = j0; jb < j1; jb += kBatch) {
            Probe pu[kBatch], pv[kBatch];
            float thr[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int j = min(jb + k, j1 - 1);               // tail: repeat the last proposal, result unused
                const float *p = a.props + (size_t)j * 5;
                const float ux = p[0], uy = p[1], vx = p[2], vy = p[3];
                thr[k] = p[4];
                int cux, cuy, cvx, cvy;
                if (all_fast || (fast_divide_ok(ux) && fast_divide_ok(uy) && fast_divide_ok(vx) && fast_divide_ok(vy))) {   // wave-uniform
                    const f2 qu = fast_quotient2(f2{ux, uy}, r2, nd), qv = fast_quotient2(f2{vx, vy}, r2, nd);
                    cux = add_wrap(xl, floor_i32_not_nan(qu.x)); cuy = add_wrap(yl, floor_i32_not_nan(qu.y));
                    cvx = add_wrap(xl, floor_i32_not_nan(qv.x)); cvy = add_wrap(yl, floor_i32_not_nan(qv.y));
                } else {
                    cux = add_wrap(xl, floor_i32(ux / df)); cuy = add_wrap(yl, floor_i32(uy / df));
                    cvx = add_wrap(xl, floor_i32(vx / df)); cvy = add_wrap(yl, floor_i32(vy / df));
                }
                pu[k] = probe_issue(pc, cux, cuy);
                pv[k] = probe_issue(pc, cvx, cvy);
            }
            if (PAIRS) {
                unsigned n[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const float f = zero_depth ? 0.0f : (float)(probe_value(pu[k]) - probe_value(pv[k]));
                    const u64 left = __ballot(live && f < thr[k]);      // tree_train.cu:57-58
                    n[k] = (jb + k) < j1 ? (unsigned)__popcll(peers & left) : 0u;
                }
                // the lowest lane of each (node, class) group adds the group's left counts, two proposals at a time
                if (live && lane == __ffsll((long long)peers) - 1) {
                    // tmp[pair][bin][2]: like the reference's [proposal][bin] layout, a bin's pairs lie far apart, so the
                    // adds of one bin spread over the L2 channels (a [bin][proposal] layout ran 2x slower)
                    const size_t nbc = (size_t)a.NB * a.C;
                    if (small) {
                        static_assert(kBatch == 4, "one quad per batch");
                        const u64 v = (u64)n[0] | ((u64)n[1] << 16) | ((u64)n[2] << 32) | ((u64)n[3] << 48);
                        if (v) atomicAdd(reinterpret_cast<u64 *>(a.tmp) + (size_t)(a.Ppad >> 1) * nbc + (size_t)(jb >> 2) * nbc + bin0, v);
                    } else {
                        u64 *cell = reinterpret_cast<u64 *>(a.tmp) + (size_t)(jb >> 1) * nbc + bin0;
#pragma unroll
                        for (int k = 0; k < kBatch; k += 2) {
                            const u64 v = (u64)n[k] | ((u64)n[k + 1] << 32);
                            if (v) atomicAdd(cell + (size_t)(k >> 1) * nbc, v);
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int j = jb + k;
                if (j >= j1) break;                              // wave-uniform
                const float f = zero_depth ? 0.0f : (float)(probe_value(pu[k]) - probe_value(pv[k]));
                const bool is_left_px = f < thr[k];              // tree_train.cu:57-58
                const u64 left = __ballot(live && is_left_px);
                const u64 part = peers & (is_left_px ? left : ~left);
                if (live && lane == __ffsll((long long)part) - 1 && (is_left_px || !a.left_only)) {
                    const unsigned n = (unsigned)__popcll(part);
                    const int bin = bin0 + (is_left_px ? 0 : a.C);
                    atomicAdd(a.counts + (size_t)j * a.NB * a.C + bin, (u64)n);
                }
            }
        }
    }
    if (tid == 0) {   // every workgroup has made its final, failing pull before it gets here
        const unsigned int done = atomicAdd(&g_train_queue[1], 1u);
        if (done == gridDim.x - 1u) {
            atomicExch(&g_train_queue[0], 0u);
            atomicExch(&g_train_queue[1], 0u);
        }
    }
}

// tmp[pair][bin][2] (32-bit halves, written by k_train_histogram<true>) -> counts[j][bin] (64-bit, the layout of the
// reference); tmp is left zeroed for the next call.
__global__ __launch_bounds__(256) void k_train_unpack_pairs(uint32_t *tmp, int n_bins, int P, int Ppad, int NB, int C, u64 *counts)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // (pair, bin, half), bins < n_bins only
    if (t >= (long long)(Ppad / 2) * n_bins * 2) return;
    const int half = (int)(t & 1), bin = (int)((t >> 1) % n_bins), pair = (int)((t >> 1) / n_bins);
    uint32_t *cell = tmp + ((size_t)pair * NB * C + bin) * 2 + half;
    const uint32_t v = *cell;
    if (!v) return;
    *cell = 0u;
    const int j = pair * 2 + half;
    if (j < P) counts[(size_t)j * NB * C + bin] += v;
}

// the 16-bit part of the workspace: quads[quad][bin] (one 64-bit word = proposals 4q .. 4q+3) -> counts[j][bin]
__global__ __launch_bounds__(256) void k_train_unpack_quads(u64 *quads, int n_bins, int P, int Ppad, int NB, int C, u64 *counts)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // (quad, bin), bins < n_bins only
    if (t >= (long long)(Ppad / 4) * n_bins) return;
    const int bin = (int)(t % n_bins), quad = (int)(t / n_bins);
    u64 *cell = quads + (size_t)quad * NB * C + bin;
    const u64 v = *cell;
    if (!v) return;
    *cell = 0ull;
    for (int f = 0; f < 4; ++f) {
        const u64 c = (v >> (16 * f)) & 0xFFFFull;
        const int j = quad * 4 + f;
        if (c && j < P) counts[(size_t)j * NB * C + bin] += c;
    }
}

// Right-child counts from the parents: every live pixel of a node goes either left or right, so
// counts[j][right][c] = parent_counts[node][c] - counts[j][left][c].  Counting only the left side halves the
// atomics of the histogram kernel, which is what bounds it.  Valid once ALL images have been counted.
__global__ __launch_bounds__(256) void k_train_right_counts(int n_active, const int32_t *active, int P, int NB, int node_start,
                                                            int node_end, int C, const u64 *parent_counts, u64 *counts)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_active * P) return;
    const int j = (int)(t / n_active), i = (int)(t - (long long)j * n_active);   // neighbouring lanes: neighbouring nodes
    const int parent = active[i];
    const int l_child = parent * 2, r_child = parent * 2 + 1;
    if (l_child < node_start || r_child >= node_end) return;
    u64 *l = counts + ((size_t)j * NB + (l_child - node_start)) * C;
    for (int c = 0; c < C; ++c) l[C + c] = parent_counts[(size_t)parent * C + c] - l[c];
}

// ---- gini helpers (tree_train.cu:66-97), fp32 exactly as written ----
__device__ __forceinline__ u64 counts_sum(const u64 *p, int C)
{
    u64 s = 0;
    for (int i = 0; i < C; ++i) s += p[i];
    return s;
}
__device__ __forceinline__ float gini_impurity(const u64 *c, int C)
{
    const float s = (float)counts_sum(c, C) * 1.f;
    float p = 0.f;
    for (int i = 0; i < C; ++i) {
        const float p_i = (float)c[i] / s;
        p = p + p_i * p_i;
    }
    return 1 - p;
}
__device__ __forceinline__ float gini_gain(const u64 *pc, const u64 *lc, const u64 *rc, int C)
{
    const float p_sum = (float)counts_sum(pc, C);
    const float p_imp = gini_impurity(pc, C);
    const float rem = (((float)counts_sum(lc, C) / p_sum) * gini_impurity(lc, C)) +
                      (((float)counts_sum(rc, C) / p_sum) * gini_impurity(rc, C));
    return p_imp - rem;
}
__device__ __forceinline__ int count_above_cutoff(const u64 *c, int C, u64 sum, float cutoff)
{
    for (int i = 0; i < C; ++i)
        if ((float)c[i] * 1.f / (float)sum >= cutoff) return i;
    return -1;
}

// pick_best_features (tree_train.cu:99-236): one WAVE per active node.  The reference scans the proposals in order
// and keeps the first one that attains the largest gain (strict >, starting from -1 with proposal 0); here lane l
// scans proposals l, l+64, ... the same way and the wave then keeps the largest gain, the lowest proposal index
// among equals -- the same winner.  Lane 0 writes the node.
__global__ __launch_bounds__(256) void k_train_pick_best(int n_active, const int32_t *active, int P, int D, int NB,
                                                         int node_start, int node_end, int C, int level,
                                                         const u64 *parent_counts, const u64 *by_feature,
                                                         const float *props, float *tree, u64 *child_counts,
                                                         float *best_gain)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n_active) return;                       // wave-uniform
    const int E = 7 + 2 * C;
    const int parent = active[i];
    const int lchild = parent * 2, rchild = parent * 2 + 1;
    if (lchild < node_start || rchild >= node_end) return;
    const u64 *pc = parent_counts + (size_t)parent * C;
    const u64 p_sum = counts_sum(pc, C);
    const float prev = best_gain[i];

    float best_g = -1.f;
    int best_j = 0;
    for (int j = lane; j < P; j += 64) {
        const u64 *lc = by_feature + ((size_t)j * NB + (lchild - node_start)) * C;
        const u64 *rc = by_feature + ((size_t)j * NB + (rchild - node_start)) * C;
        const u64 ls = counts_sum(lc, C), rs = counts_sum(rc, C);
        const float g = (!ls || !rs) ? 0.f : gini_gain(pc, lc, rc, C);
        if (g > best_g) { best_g = g; best_j = j; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float og = __shfl_xor(best_g, o);
        const int oj = __shfl_xor(best_j, o);
        if (og > best_g || (og == best_g && oj < best_j)) { best_g = og; best_j = oj; }
    }
    if (lane != 0) return;
    if (!(best_g > prev)) return;
    best_gain[i] = best_g;

    const u64 *lc = by_feature + ((size_t)best_j * NB + (lchild - node_start)) * C;
    const u64 *rc = by_feature + ((size_t)best_j * NB + (rchild - node_start)) * C;
    const u64 ls = counts_sum(lc, C), rs = counts_sum(rc, C);
    float *rec = tree + ((size_t)((1 << level) - 1) + parent) * E;
    for (int k = 0; k < 5; ++k) rec[k] = props[(size_t)best_j * 5 + k];

    if (best_g <= 0.f) {   // no proposal separates anything: both sides end here with the parent's PDF
        rec[5] = 0.f; rec[6] = 0.f;
        for (int k = 0; k < C; ++k) {
            const float p = ((float)pc[k] * 1.f) / (float)p_sum;
            rec[7 + k] = p; rec[7 + C + k] = p;
        }
        return;
    }
    const float kCutoff = 0.999f;
    for (int side = 0; side < 2; ++side) {
        const u64 *cc = side ? rc : lc;
        const u64 cs = side ? rs : ls;
        const int cut = count_above_cutoff(cc, C, cs, kCutoff);
        if (cut > -1) {
            rec[5 + side] = 0.f;
            rec[7 + side * C + cut] = 1.f;       // other entries keep whatever an earlier proposal block left
        } else if (level == D - 1) {
            rec[5 + side] = 0.f;
            for (int n = 0; n < C; ++n) rec[7 + side * C + n] = ((float)cc[n] * 1.f) / (float)cs;
        } else {
            rec[5 + side] = -1.f;
            u64 *dst = child_counts + (size_t)(side ? rchild : lchild) * C;
            for (int n = 0; n < C; ++n) dst[n] = cc[n];
        }
    }
}

// get_active_nodes_next_level (tree_train.cu:238-273) as an ORDERED compaction: one workgroup, each thread a
// contiguous run of active nodes, exclusive scan of the per-thread child counts, then the writes.
__global__ __launch_bounds__(1024) void k_train_next_active(int level, int C, const float *tree, const int32_t *active,
                                                            int n_active, int32_t *next_active, int32_t *n_next)
{
    __shared__ int s_cnt[1024];
    const int E = 7 + 2 * C, t = threadIdx.x;
    const int per = (n_active + 1023) / 1024;
    const int b = t * per, e = min(b + per, n_active);
    const size_t base = (size_t)((1 << level) - 1);
    int cnt = 0;
    for (int i = b; i < e; ++i) {
        const float *rec = tree + (base + active[i]) * E;
        cnt += (rec[5] == -1.f) + (rec[6] == -1.f);
    }
    s_cnt[t] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan
        const int v = t >= o ? s_cnt[t - o] : 0;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    int pos = s_cnt[t] - cnt;
    for (int i = b; i < e; ++i) {
        const int node = active[i];
        const float *rec = tree + (base + node) * E;
        if (rec[5] == -1.f) next_active[pos++] = node * 2;
        if (rec[6] == -1.f) next_active[pos++] = node * 2 + 1;
    }
    if (t == 1023) *n_next = s_cnt[1023];
}

// copy_pixel_groups (tree_train.cu:275-324): route every live pixel through its node's chosen feature
__global__ __launch_bounds__(256) void k_train_update_pixels(const uint16_t *depth, size_t n_px, int W, int H, int level,
                                                             int C, int32_t *nodes, const float *tree)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_px) return;
    const int parent = nodes[i];
    if (parent == -1) return;
    const size_t per = (size_t)W * H;
    const size_t img_off = (i / per) * per;
    const int y = (int)((i % per) / W), x = (int)((i % per) % W);
    const float *rec = tree + ((size_t)((1 << level) - 1) + parent) * (7 + 2 * C);
    const uint32_t d = depth[i];
    float f = 0.f;
    if (d != 0u) {
        const float df = (float)d;
        auto at = [&](float ox, float oy) -> float {
            const int px = add_wrap(x, floor_i32(ox / df)), py = add_wrap(y, floor_i32(oy / df));
            if ((uint32_t)px < (uint32_t)W && (uint32_t)py < (uint32_t)H) return (float)depth[img_off + (size_t)py * W + px];
            return 65535.0f;
        };
        f = at(rec[0], rec[1]) - at(rec[2], rec[3]);
    }
    const bool left = f < rec[4];
    const int status = floor_i32(rec[left ? 5 : 6]);
    nodes[i] = status != -1 ? -1 : parent * 2 + (left ? 0 : 1);
}


// ------------------------------------------------------------------------------------------------------------------
// Round 3: counting without per-wave atomics.  The deep levels were bound by scattered global atomics -- a wave's 64
// pixels sit in up to 64 different (node, class) groups there, and every group costs an atomic per four proposals.
// Now, once per level, the live pixels are given ROW NUMBERS in (node, class) order (k_train_sort_*: a counting sort by
// the live group sizes); per proposal block every live pixel writes its P left/right decisions as one row of bits
// (k_train_bits: the same probes as before, no ballots, no atomics), and k_train_count_rows walks the rows in order with
// a proposal slice per lane, adding a group's bits up in registers and touching the count array once per group (or once
// per chunk of rows for the huge groups of the upper levels).  Integer sums: the same counts whatever the row order.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSortIters = 4;     // distinct keys a wave aggregates before its remaining lanes use an atomic each

struct SortArgs {
    const uint16_t *labels;
    const int32_t *nodes;
    size_t n_px;
    int C;
    int slot_shift;        // every key has 2^slot_shift counters, a wave uses the one of its number: the upper levels have a
                           // handful of keys and a million waves, and atomics on ONE line run at ~90 per microsecond
    uint32_t *sizes, *offsets, *cursor, *n_rows;   // [n_keys << slot_shift] each, + 1
    int32_t *pos, *rowkey;
};

// pass 1 (POSITIONS = false): sizes[key] += 1 for every live pixel; pass 2 (true): pos[pixel] = offsets[key] + its rank,
// rowkey[pos] = key.  Lanes of a wave that share a key go through ONE atomic (the first kSortIters keys of a wave; at
// the upper levels a wave holds one or two keys and a million pixels would otherwise queue on four counters).
template <bool POSITIONS>
__global__ __launch_bounds__(256) void k_train_sort(const SortArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t slot = ((blockIdx.x * 256u + threadIdx.x) >> 6) & ((1u << a.slot_shift) - 1u);
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) - lane; i0 < a.n_px; i0 += (size_t)gridDim.x * 256) {
        const size_t i = i0 + lane;
        int key = -1;
        if (i < a.n_px) {
            const int node = a.nodes[i];
            if (node >= 0) {
                const uint32_t label = a.labels[i];
                if (label < (uint32_t)a.C) key = node * a.C + (int)label;
            }
        }
        u64 todo = __ballot(key >= 0);
        if (!todo) {                                   // (most of a frame is unlabelled background)
            if (POSITIONS && i < a.n_px) a.pos[i] = -1;
            continue;
        }
        int my_pos = -1;
        bool mine_done = key < 0;
        for (int it = 0; it < kSortIters && todo; ++it) {
            const int src = __ffsll((long long)todo) - 1;
            const int k0 = __builtin_amdgcn_readfirstlane(__shfl(key, src));
            const u64 m = __ballot(key == k0);
            const unsigned n = (unsigned)__popcll(m);
            unsigned base = 0;
            const size_t c0 = ((size_t)k0 << a.slot_shift) + slot;
            if (lane == src) base = POSITIONS ? atomicAdd(a.cursor + c0, n) : atomicAdd(a.sizes + c0, n);
            if (POSITIONS) {
                base = (unsigned)__shfl((int)base, src);
                if (key == k0) my_pos = (int)(a.offsets[c0] + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull)));
            }
            if (key == k0) mine_done = true;
            todo &= ~m;
        }
        if (!mine_done) {                              // deep levels: every pixel its own group
            const size_t c = ((size_t)key << a.slot_shift) + slot;
            if (POSITIONS) my_pos = (int)(a.offsets[c] + atomicAdd(a.cursor + c, 1u));
            else atomicAdd(a.sizes + c, 1u);
        }
        if (POSITIONS && i < a.n_px) {
            a.pos[i] = my_pos;
            if (my_pos >= 0) a.rowkey[my_pos] = key;
        }
    }
}

// exclusive scan of sizes[n_keys] -> offsets, total -> *n_rows.  One workgroup: thread t takes a contiguous run.
__global__ __launch_bounds__(1024) void k_train_sort_scan(const uint32_t *sizes, uint32_t *offsets, uint32_t *n_rows, int n_keys)
{
    __shared__ uint32_t s_sum[1024];
    const int t = threadIdx.x;
    const int per = (n_keys + 1023) / 1024;
    const int b = min(t * per, n_keys), e = min(b + per, n_keys);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += sizes[i];
    s_sum[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan
        const uint32_t v = t >= o ? s_sum[t - o] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - sum;
    for (int i = b; i < e; ++i) {
        offsets[i] = run;
        run += sizes[i];
    }
    if (t == 1023) *n_rows = s_sum[1023];
}

struct BitsArgs {
    const uint16_t *depth;
    const int32_t *pos;    // row of every pixel, -1 = not live
    const float *props;    // [P][5] as drawn (the IEEE path), then [P][5] prepared (k_train_prepare_props), then the flag
    uint32_t *bits;        // [rows][words]
    uint32_t n_tiles, tiles_x, tiles_y;
    int W, H, P, words;    // words = 32-bit words per row (P padded to a power of two >= 64, / 32)
};

// Proposals in the form the one-fma divide-and-floor wants (rdf_hip.hip, NodeRec16): numerator n = 512 * (floor(u) + 1/2),
// exactly representable for |u| < 2^21; flag = 1.0f when every numerator of the block is in that range (anything
// make_random_features draws is), else the block takes the IEEE divide.  floor(IEEE u / d) = floor(floor(u) / d) for those
// u (tools/verify_intoffset.hip), and floor(x / d) is what the fma gives (tools/verify_magic.hip: payload byte 0).
__global__ __launch_bounds__(256) void k_train_prepare_props(const float *props, float *out, int P)
{
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    bool ok = true;
    for (int i = threadIdx.x; i < P * 5; i += 256) {
        const float v = props[i];
        if (i % 5 == 4) { out[i] = v; continue; }
        const uint32_t b = __float_as_uint(v);
        const bool fine = (b << 1) == 0u || (((b >> 23) & 0xFFu) - 40u) <= 107u;      // +-0 or 2^-87 <= |v| < 2^21
        ok = ok && fine;
        out[i] = fine ? (float)((int)__builtin_floorf(v) * 512 + 256) : 0.0f;
    }
    if (!ok) s_ok = 0;
    __syncthreads();
    if (threadIdx.x == 0) out[P * 5] = s_ok ? 1.0f : 0.0f;
}

// One row of P bits per live pixel: bit j = the pixel goes LEFT under proposal j (tree_train.cu:57-58).  The histogram
// kernel's tiles (a wave = an 8 x 8 block, four side by side) with a 32-pixel halo (16 there), and the forest kernel's
// probes: tile at LDS address 0, x doubled, one fma in round-down mode per coordinate (the kernel has no static LDS).  A lane
// collects its own decisions 32 at a time and stores them 16 bytes at a time straight into its row: LDS holds the tile only,
// so occupancy is set by registers (the first version staged the rows in LDS, 38 KB per workgroup, four workgroups per CU:
// TA 66 % busy, VALU 45 %, nothing at its limit).
constexpr int kHaloB = 32;      // (24, 40 and 48 measured the same on the 256-frame benchmark: 0.556-0.559 s)
constexpr int kTWb = kCols + 2 * kHaloB, kTHb = kRows + 2 * kHaloB;
constexpr int kBatchB = 4;      // proposals whose probes are in flight together (2: 0.567 s, 4: 0.556 s, 8: 0.552 s on the 256-frame benchmark)
static_assert(32 % kBatchB == 0, "a word of decisions is a whole number of batches");

__global__ __launch_bounds__(256) void k_train_bits(const BitsArgs a)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    uint16_t *s_tile = reinterpret_cast<uint16_t *>(s_dyn);                                      // at LDS address 0 (TileCtx)
    uint32_t *s_mail = reinterpret_cast<uint32_t *>(s_dyn + kTHb * kTWb * 2);                    // next tile, "any live" x 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per_img = (uint32_t)a.W * (uint32_t)a.H;
    const float *nprops = a.props + (size_t)a.P * 5;
    const bool all_fast = nprops[(size_t)a.P * 5] != 0.0f;          // (wave-uniform, scalar load)
    const int words_used = (a.P + 31) >> 5;
    if (tid == 0) s_mail[1] = s_mail[2] = 0u;

    for (uint32_t it = 0;; ++it) {
        if (tid == 0) s_mail[0] = atomicAdd(&g_train_queue[0], 1u);
        __syncthreads();
        const uint32_t tile = s_mail[0];
        if (tile >= a.n_tiles) break;
        const uint32_t per = a.tiles_x * a.tiles_y;
        const uint32_t img = tile / per, rem = tile - img * per;
        const uint32_t ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int x = (int)(tx * kCols) + wave * kBW + (lane % kBW), y = (int)(ty * kRows) + lane / kBW;
        const int tx0 = (int)(tx * kCols) - kHaloB, ty0 = (int)(ty * kRows) - kHaloB;
        const size_t img_off = (size_t)img * per_img;

        int row = -1;
        if (x < a.W && y < a.H) row = a.pos[img_off + (size_t)y * a.W + x];
        const bool live = row >= 0;
        // workgroup-wide OR through two alternating words (no static LDS: see the forest kernel's empty-tile test)
        if (__any(live) && lane == 0) s_mail[1u + (it & 1u)] = 1u;
        __syncthreads();                                               // (also: the previous tile's readers are done, s_mail[0] is read)
        const uint32_t any_live = s_mail[1u + (it & 1u)];
        if (tid == 0) s_mail[1u + ((it + 1u) & 1u)] = 0u;
        if (any_live == 0u) continue;

        for (int r = wave; r < kTHb; r += 4) {
            const int gy = ty0 + r;
            const bool row_in = (uint32_t)gy < (uint32_t)a.H;
            for (int col = lane; col < kTWb; col += 64) {
                const int gx = tx0 + col;
                uint32_t v = kNoPixel;
                if (row_in && (uint32_t)gx < (uint32_t)a.W) v = a.depth[img_off + (size_t)gy * a.W + gx];
                s_tile[r * kTWb + col] = (uint16_t)v;
            }
        }
        __syncthreads();

        if (!__any(live)) continue;   // wave-uniform; no barrier below this point in the iteration
        const int xl = x - tx0, yl = y - ty0;
        const uint32_t d = live ? s_tile[yl * kTWb + xl] : 1u;
        const float df = (float)(d == 0u ? 1u : d);
        const float r0 = __builtin_amdgcn_rcpf(df);
        const float rcp = __builtin_fmaf(__builtin_fmaf(-df, r0, 1.0f), r0, r0);
        float rcp_s = rcp * (1.0f / kNumScale);
        const bool zero_depth = d == 0u;              // compute_feature returns 0.f (decision_tree_common.hpp:12)
        const TileCtx pc = {reinterpret_cast<const char *>(a.depth) + img_off * 2, (uint32_t)kTWb * 2u, (uint32_t)kTHb,
                            (uint32_t)kTWb * 2u, (uint32_t)a.W * 2u, (uint32_t)a.H, (uint32_t)tx0 * 2u, (uint32_t)ty0};
        uint32_t kx2 = ((uint32_t)xl - kMagicBits) * 2u;
        const uint32_t ky = (uint32_t)yl - kMagicBits;
        pin(kx2);
        if (all_fast) set_round_down(rcp_s);          // (the loop below has no other rounding arithmetic)
        uint32_t *my_row = a.bits + (size_t)(live ? row : 0) * a.words;

        uint4 acc = make_uint4(0u, 0u, 0u, 0u);
        for (int w = 0; w < words_used; ++w) {
            uint32_t word = 0u;
            const int j_end = min(a.P, w * 32 + 32);
            for (int jb = w * 32; jb < j_end; jb += kBatchB) {
                TileProbe pu[kBatchB], pv[kBatchB];
                float thr[kBatchB];
#pragma unroll
                for (int k = 0; k < kBatchB; ++k) {
                    const int j = min(jb + k, a.P - 1);               // tail: repeat the last proposal, result unused
                    uint32_t cux, cuy, cvx, cvy;
                    if (all_fast) {                                   // wave-uniform
                        const float *p = nprops + (size_t)j * 5;
                        thr[k] = p[4];
                        const f2 nu = {p[0], p[1]}, nv = {p[2], p[3]}, r2 = {rcp_s, rcp_s}, m2 = {kMagic, kMagic};
                        const f2 tu = __builtin_elementwise_fma(nu, r2, m2), tv = __builtin_elementwise_fma(nv, r2, m2);
                        cux = (__float_as_uint(tu.x) << 1) + kx2; cuy = __float_as_uint(tu.y) + ky;
                        cvx = (__float_as_uint(tv.x) << 1) + kx2; cvy = __float_as_uint(tv.y) + ky;
                    } else {
                        const float *p = a.props + (size_t)j * 5;
                        thr[k] = p[4];
                        cux = (uint32_t)min(max(add_wrap(xl, floor_i32(p[0] / df)), -(1 << 30)), (1 << 30) - 1) * 2u;
                        cuy = (uint32_t)add_wrap(yl, floor_i32(p[1] / df));
                        cvx = (uint32_t)min(max(add_wrap(xl, floor_i32(p[2] / df)), -(1 << 30)), (1 << 30) - 1) * 2u;
                        cvy = (uint32_t)add_wrap(yl, floor_i32(p[3] / df));
                    }
                    pu[k] = tprobe_issue(pc, cux, cuy);
                    pv[k] = tprobe_issue(pc, cvx, cvy);
                }
#pragma unroll
                for (int k = 0; k < kBatchB; ++k) {
                    const float f = zero_depth ? 0.0f : (float)(tprobe_value(pu[k]) - tprobe_value(pv[k]));
                    const uint32_t left = (jb + k) < a.P && f < thr[k] ? 1u : 0u;      // tree_train.cu:57-58
                    word |= left << ((jb + k) & 31);
                }
            }
            if (a.words < 4) {                         // (wave-uniform) rows of 8 bytes: word by word
                if (live) my_row[w] = word;
                continue;
            }
            const int slot = w & 3;
            acc.x = slot == 0 ? word : acc.x; acc.y = slot == 1 ? word : acc.y;
            acc.z = slot == 2 ? word : acc.z; acc.w = slot == 3 ? word : acc.w;
            if (slot == 3 || w == words_used - 1) {    // 16 bytes of the row at a time (unused words of the last piece: zero)
                if (live) *reinterpret_cast<uint4 *>(my_row + (w & ~3)) = acc;
                acc = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (all_fast) { pin(acc.x); set_round_nearest(acc.x); }
        // words of the padding beyond P (rows are padded to a power of two)
        if (live) {
            if (a.words < 4) {
                for (int w = words_used; w < a.words; ++w) my_row[w] = 0u;
            } else {
                for (int w = (words_used + 3) & ~3; w < a.words; w += 4)
                    *reinterpret_cast<uint4 *>(my_row + w) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    if (tid == 0) {   // every workgroup has made its final, failing pull before it gets here
        const unsigned int done = atomicAdd(&g_train_queue[1], 1u);
        if (done == gridDim.x - 1u) {
            atomicExch(&g_train_queue[0], 0u);
            atomicExch(&g_train_queue[1], 0u);
        }
    }
}

struct CountArgs {
    const uint32_t *bits;
    const int32_t *rowkey;
    const uint32_t *n_rows;
    u64 *counts;          // [P][NB][C]
    int P, words, q;      // q = bits of a row per lane = 32 * words / 64 (a power of two, 1..32)
    int C, NB, node_start, node_end;
};

// Rows in (node, class) order; a wave takes a contiguous chunk of them.  Lane l owns proposals l*q .. l*q + q - 1: it adds
// the bits of a group's rows up in q registers and flushes them into counts[j][left child's bin] when the group (or the
// chunk) ends -- one atomic per (group, proposal) where the histogram kernel spent one per (wave, group, four proposals).
template <int Q>
__global__ __launch_bounds__(256) void k_train_count_rows(const CountArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t n_rows = *a.n_rows;
    const uint32_t n_waves = gridDim.x * 4u, wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t chunk = (n_rows + n_waves - 1u) / n_waves;
    chunk = (chunk + 63u) & ~63u;
    const uint32_t r0 = wave_id * chunk, r1 = min(r0 + chunk, n_rows);
    if (r0 >= r1) return;
    const int word_of_lane = (lane * Q) >> 5, shift = (lane * Q) & 31;
    uint32_t cnt[Q];
#pragma unroll
    for (int m = 0; m < Q; ++m) cnt[m] = 0u;
    int cur_key = -1;
    auto flush = [&](int key) {
        if (key < 0) return;
        const int node = key / a.C, label = key - node * a.C;
        if (node * 2 < a.node_start || node * 2 + 1 >= a.node_end) return;     // (children outside this node block)
        const size_t bin = (size_t)(node * 2 - a.node_start) * a.C + label;
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            const int j = lane * Q + m;
            if (cnt[m] && j < a.P) atomicAdd(a.counts + (size_t)j * a.NB * a.C + bin, (u64)cnt[m]);
        }
    };
    for (uint32_t rb = r0; rb < r1; rb += 64u) {
        const uint32_t r_mine = rb + (uint32_t)lane;
        const int key_mine = r_mine < r1 ? a.rowkey[r_mine] : -1;
        const int n_here = (int)min(64u, r1 - rb);
        for (int i = 0; i < n_here; ++i) {
            const int key = __builtin_amdgcn_readfirstlane(__shfl(key_mine, i));
            if (key != cur_key) {
                flush(cur_key);
#pragma unroll
                for (int m = 0; m < Q; ++m) cnt[m] = 0u;
                cur_key = key;
            }
            const uint32_t w = a.bits[(size_t)(rb + (uint32_t)i) * a.words + word_of_lane] >> shift;
#pragma unroll
            for (int m = 0; m < Q; ++m) cnt[m] += (w >> m) & 1u;
        }
    }
    flush(cur_key);
}

} // namespace

extern "C" {

int rdf_train_init(const uint16_t *labels, size_t n_px, int n_classes, int32_t *nodes_by_pixel,
                   unsigned long long *root_counts, void *stream)
{
    if (n_classes < 1 || n_classes > kMaxClasses) return RDF_ERR_BAD_ARG;
    if (n_px == 0) return RDF_OK;
    if (!labels || !nodes_by_pixel || !root_counts) return RDF_ERR_NULL_PTR;
    size_t blocks = (n_px + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_train_init, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), labels, n_px,
                       n_classes, nodes_by_pixel, root_counts);
    return (int)hipGetLastError();
}

static int pairs_ppad(int n_proposals) { return (n_proposals + kBatch - 1) / kBatch * kBatch; }

static int train_histogram(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                           int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes, int node_start,
                           int node_end, int nodes_per_block, unsigned long long *counts, int left_only, void *workspace,
                           const unsigned long long *parent_counts, void *stream)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_proposals < 0 || n_classes < 1 || n_classes > kMaxClasses ||
        nodes_per_block < 1 || node_end - node_start > nodes_per_block || node_start < 0)
        return RDF_ERR_BAD_ARG;
    if ((long long)n_img * dim_x * dim_y >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    if (n_img == 0 || dim_x == 0 || dim_y == 0 || n_proposals == 0) return RDF_OK;
    if (!depth || !labels || !nodes_by_pixel || !proposals || !counts) return RDF_ERR_NULL_PTR;
    HistArgs a;
    a.depth = depth; a.labels = labels; a.nodes = nodes_by_pixel; a.props = proposals; a.counts = counts;
    a.W = dim_x; a.H = dim_y; a.P = n_proposals; a.C = n_classes; a.NB = nodes_per_block;
    a.node_start = node_start; a.node_end = node_end;
    a.left_only = left_only;
    a.tmp = reinterpret_cast<uint32_t *>(workspace);
    a.Ppad = pairs_ppad(n_proposals);
    a.parent_counts = parent_counts;
    a.tiles_x = (uint32_t)(dim_x + kCols - 1) / kCols;
    a.tiles_y = (uint32_t)(dim_y + kRows - 1) / kRows;
    const long long n_tiles = (long long)n_img * a.tiles_x * a.tiles_y;
    if (n_tiles >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    a.n_tiles = (uint32_t)n_tiles;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long grid = (long long)cus * 8;
    if (grid > n_tiles) grid = n_tiles;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {   // the kernel leaves its tile queue at zero; clearing it here as well keeps a launch that died half-way from
        // silently shortening the next one
        void *q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_train_queue)) != hipSuccess || !q) return RDF_ERR_NO_DEVICE;
        const hipError_t eq = hipMemsetAsync(q, 0, sizeof(unsigned int) * 2, st);
        if (eq != hipSuccess) return (int)eq;
    }
    if (!workspace) {
        hipLaunchKernelGGL(k_train_histogram<false>, dim3((unsigned)grid), dim3(256), 0, st, a);
        return (int)hipGetLastError();
    }
    if (((uintptr_t)workspace & 7u) != 0) return RDF_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_train_histogram<true>, dim3((unsigned)grid), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int n_bins = (node_end - node_start) * n_classes;
    const long long n = (long long)n_bins * a.Ppad;
    if (n > 0) {
        hipLaunchKernelGGL(k_train_unpack_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.tmp, n_bins, n_proposals,
                           a.Ppad, nodes_per_block, n_classes, counts);
        e = hipGetLastError();
        if (e == hipSuccess && parent_counts) {
            u64 *quads = reinterpret_cast<u64 *>(a.tmp) + (size_t)(a.Ppad / 2) * nodes_per_block * n_classes;
            const long long nq = (long long)n_bins * (a.Ppad / 4);
            hipLaunchKernelGGL(k_train_unpack_quads, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, quads, n_bins,
                               n_proposals, a.Ppad, nodes_per_block, n_classes, counts);
            e = hipGetLastError();
        }
    }
    return (int)e;
}

int rdf_train_histogram(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                        int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes,
                        int node_start, int node_end, int nodes_per_block, unsigned long long *counts, void *stream)
{
    return train_histogram(depth, labels, nodes_by_pixel, n_img, dim_x, dim_y, proposals, n_proposals, n_classes,
                           node_start, node_end, nodes_per_block, counts, 0, nullptr, nullptr, stream);
}

int rdf_train_histogram_left(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                             int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes,
                             int node_start, int node_end, int nodes_per_block, unsigned long long *counts, void *stream)
{
    return train_histogram(depth, labels, nodes_by_pixel, n_img, dim_x, dim_y, proposals, n_proposals, n_classes,
                           node_start, node_end, nodes_per_block, counts, 1, nullptr, nullptr, stream);
}

size_t rdf_train_histogram_workspace_bytes(int n_proposals, int nodes_per_block, int n_classes)
{
    if (n_proposals < 0 || nodes_per_block < 0 || n_classes < 0) return 0;
    // 32-bit pair counters (4 bytes per bin and proposal) + 16-bit quad counters (2 bytes)
    return (size_t)nodes_per_block * (size_t)n_classes * (size_t)pairs_ppad(n_proposals) * 6u;
}

int rdf_train_histogram_left_ws(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                                int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes,
                                int node_start, int node_end, int nodes_per_block, unsigned long long *counts,
                                void *workspace, const unsigned long long *parent_counts, void *stream)
{
    if (!workspace) return RDF_ERR_NULL_PTR;
    return train_histogram(depth, labels, nodes_by_pixel, n_img, dim_x, dim_y, proposals, n_proposals, n_classes,
                           node_start, node_end, nodes_per_block, counts, 1, workspace, parent_counts, stream);
}


constexpr int kSortCounters = 16384;     // counters a level with few keys spreads them over (2^slot_shift per key)

static int sort_slot_shift(long long n_keys)
{
    int sh = 0;
    while (sh < 8 && (n_keys << (sh + 1)) <= kSortCounters) ++sh;
    return sh;
}

size_t rdf_train_sort_workspace_bytes(int n_nodes, int n_classes)
{
    if (n_nodes < 0 || n_classes < 0) return 0;
    size_t n = (size_t)n_nodes * (size_t)n_classes;
    if (n < (size_t)kSortCounters) n = kSortCounters;
    return (n * 3u + 4u) * sizeof(uint32_t);
}

static int bits_words(int n_proposals)     // 32-bit words of a row: P padded to a power of two >= 64
{
    int p = 64;
    while (p < n_proposals) p <<= 1;
    return p / 32;
}

size_t rdf_train_bits_row_bytes(int n_proposals)
{
    if (n_proposals < 1 || n_proposals > 1024) return 0;
    return (size_t)bits_words(n_proposals) * 4u;
}

size_t rdf_train_bits_workspace_bytes(int n_proposals)
{
    if (n_proposals < 0) return 0;
    return ((size_t)n_proposals * 10u + 4u) * sizeof(float);
}

int rdf_train_sort_pixels(const uint16_t *labels, const int32_t *nodes_by_pixel, size_t n_px, int n_classes, int n_nodes,
                          int32_t *pos, int32_t *rowkey, void *workspace, void *stream)
{
    if (n_classes < 1 || n_classes > kMaxClasses || n_nodes < 1 || (long long)n_nodes * n_classes >= (1ll << 30)) return RDF_ERR_BAD_ARG;
    if (n_px >= ((size_t)1 << 31)) return RDF_ERR_TOO_LARGE;
    if (!workspace) return RDF_ERR_NULL_PTR;
    const int shift = sort_slot_shift((long long)n_nodes * n_classes);
    const int n_keys = (n_nodes * n_classes) << shift;        // counters
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(workspace, 0, ((size_t)n_keys * 3u + 4u) * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    if (n_px == 0) return RDF_OK;
    if (!labels || !nodes_by_pixel || !pos || !rowkey) return RDF_ERR_NULL_PTR;
    SortArgs a;
    a.labels = labels; a.nodes = nodes_by_pixel; a.n_px = n_px; a.C = n_classes; a.slot_shift = shift;
    a.sizes = reinterpret_cast<uint32_t *>(workspace);
    a.offsets = a.sizes + n_keys;
    a.cursor = a.offsets + n_keys;
    a.n_rows = a.cursor + n_keys;
    a.pos = pos; a.rowkey = rowkey;
    size_t blocks = (n_px + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_train_sort<false>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_train_sort_scan, dim3(1), dim3(1024), 0, st, a.sizes, a.offsets, a.n_rows, n_keys);
    hipLaunchKernelGGL(k_train_sort<true>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int rdf_train_decision_bits(const uint16_t *depth, const int32_t *pos, int n_img, int dim_x, int dim_y,
                            const float *proposals, int n_proposals, void *bits, void *workspace, void *stream)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_proposals < 0 || n_proposals > 1024) return RDF_ERR_BAD_ARG;
    if ((long long)n_img * dim_x * dim_y >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    if (n_img == 0 || dim_x == 0 || dim_y == 0 || n_proposals == 0) return RDF_OK;
    if (!depth || !pos || !proposals || !bits || !workspace) return RDF_ERR_NULL_PTR;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the proposals as drawn, then in the form the one-fma divide wants, then the "all in range" flag
    float *props2 = reinterpret_cast<float *>(workspace);
    hipError_t ec = hipMemcpyAsync(props2, proposals, (size_t)n_proposals * 5 * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (ec != hipSuccess) return (int)ec;
    hipLaunchKernelGGL(k_train_prepare_props, dim3(1), dim3(256), 0, st, proposals, props2 + (size_t)n_proposals * 5, n_proposals);
    BitsArgs a;
    a.depth = depth; a.pos = pos; a.props = props2; a.bits = reinterpret_cast<uint32_t *>(bits);
    a.W = dim_x; a.H = dim_y; a.P = n_proposals; a.words = bits_words(n_proposals);
    a.tiles_x = (uint32_t)(dim_x + kCols - 1) / kCols;
    a.tiles_y = (uint32_t)(dim_y + kRows - 1) / kRows;
    const long long n_tiles = (long long)n_img * a.tiles_x * a.tiles_y;
    if (n_tiles >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    a.n_tiles = (uint32_t)n_tiles;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    void *q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_train_queue)) != hipSuccess || !q) return RDF_ERR_NO_DEVICE;
    const hipError_t eq = hipMemsetAsync(q, 0, sizeof(unsigned int) * 2, st);
    if (eq != hipSuccess) return (int)eq;
    static_assert((kTHb * kTWb * 2) % 16 == 0, "the mailbox words sit aligned behind the tile");
    hipFuncAttributes fa;      // the probes address the tile as LDS address 0 + offset (TileCtx): no static LDS allowed
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_train_bits)) != hipSuccess || fa.sharedSizeBytes != 0)
        return RDF_ERR_NO_DEVICE;
    const int lds = kTHb * kTWb * 2 + 16;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_train_bits, 256, (size_t)lds) != hipSuccess || per_cu < 1) per_cu = 4;
    long long grid = (long long)cus * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(k_train_bits, dim3((unsigned)grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

int rdf_train_count_rows(const void *bits, const int32_t *rowkey, const void *sort_workspace, int n_nodes, int n_proposals,
                         int n_classes, int node_start, int node_end, int nodes_per_block, unsigned long long *counts,
                         void *stream)
{
    if (n_proposals < 0 || n_proposals > 1024 || n_classes < 1 || n_classes > kMaxClasses || n_nodes < 1 || nodes_per_block < 1 ||
        node_end - node_start > nodes_per_block || node_start < 0)
        return RDF_ERR_BAD_ARG;
    if (n_proposals == 0) return RDF_OK;
    if (!bits || !rowkey || !sort_workspace || !counts) return RDF_ERR_NULL_PTR;
    CountArgs a;
    a.bits = reinterpret_cast<const uint32_t *>(bits);
    a.rowkey = rowkey;
    a.n_rows = reinterpret_cast<const uint32_t *>(sort_workspace) +
               ((size_t)(n_nodes * n_classes) << sort_slot_shift((long long)n_nodes * n_classes)) * 3u;
    a.counts = counts;
    a.P = n_proposals; a.words = bits_words(n_proposals); a.q = a.words * 32 / 64;
    a.C = n_classes; a.NB = nodes_per_block; a.node_start = node_start; a.node_end = node_end;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const dim3 grid((unsigned)cus * 8u), block(256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (a.q) {
    case 1: hipLaunchKernelGGL(k_train_count_rows<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_train_count_rows<2>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_train_count_rows<4>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(k_train_count_rows<8>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(k_train_count_rows<16>, grid, block, 0, st, a); break;
    }
    return (int)hipGetLastError();
}

int rdf_train_right_counts(int n_active, const int32_t *active_nodes, int n_proposals, int nodes_per_block,
                           int node_start, int node_end, int n_classes, const unsigned long long *parent_counts,
                           unsigned long long *counts, void *stream)
{
    if (n_active < 0 || n_proposals < 0 || n_classes < 1 || n_classes > kMaxClasses || nodes_per_block < 1 ||
        node_end - node_start > nodes_per_block || node_start < 0)
        return RDF_ERR_BAD_ARG;
    if (n_active == 0 || n_proposals == 0) return RDF_OK;
    if (!active_nodes || !parent_counts || !counts) return RDF_ERR_NULL_PTR;
    const long long n = (long long)n_active * n_proposals;
    if (n >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    hipLaunchKernelGGL(k_train_right_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), n_active, active_nodes, n_proposals, nodes_per_block,
                       node_start, node_end, n_classes, parent_counts, counts);
    return (int)hipGetLastError();
}

int rdf_train_pick_best(int n_active, const int32_t *active_nodes, int n_proposals, int max_depth, int nodes_per_block,
                        int node_start, int node_end, int n_classes, int level,
                        const unsigned long long *parent_counts, const unsigned long long *counts_by_feature,
                        const float *proposals, float *tree_out, unsigned long long *child_counts,
                        float *best_gain_per_node, void *stream)
{
    if (n_active < 0 || n_proposals < 0 || n_classes < 1 || n_classes > kMaxClasses || level < 0 || level >= max_depth ||
        max_depth > 30)
        return RDF_ERR_BAD_ARG;
    if (n_active == 0 || n_proposals == 0) return RDF_OK;
    if (!active_nodes || !parent_counts || !counts_by_feature || !proposals || !tree_out || !child_counts || !best_gain_per_node)
        return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_train_pick_best, dim3((n_active + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       n_active, active_nodes, n_proposals, max_depth, nodes_per_block, node_start, node_end, n_classes, level,
                       parent_counts, counts_by_feature, proposals, tree_out, child_counts, best_gain_per_node);
    return (int)hipGetLastError();
}

int rdf_train_next_active(int level, int max_depth, int n_classes, const float *tree, const int32_t *active_nodes,
                          int n_active, int32_t *next_active_nodes, int32_t *n_next_active, void *stream)
{
    if (level < 0 || level >= max_depth || max_depth > 30 || n_classes < 1 || n_active < 0) return RDF_ERR_BAD_ARG;
    if (!n_next_active) return RDF_ERR_NULL_PTR;
    if (n_active > 0 && (!tree || !active_nodes || !next_active_nodes)) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_train_next_active, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), level, n_classes,
                       tree, active_nodes, n_active, next_active_nodes, n_next_active);
    return (int)hipGetLastError();
}

int rdf_train_update_pixels(const uint16_t *depth, int n_img, int dim_x, int dim_y, int level, int max_depth,
                            int n_classes, int32_t *nodes_by_pixel, const float *tree, void *stream)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || level < 0 || level >= max_depth || max_depth > 30 || n_classes < 1)
        return RDF_ERR_BAD_ARG;
    const size_t n_px = (size_t)n_img * dim_x * dim_y;
    if (n_px == 0) return RDF_OK;
    if (n_px >= ((size_t)1 << 39)) return RDF_ERR_TOO_LARGE;
    if (!depth || !nodes_by_pixel || !tree) return RDF_ERR_NULL_PTR;
    hipLaunchKernelGGL(k_train_update_pixels, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), depth, n_px, dim_x, dim_y, level, n_classes, nodes_by_pixel, tree);
    return (int)hipGetLastError();
}

} // extern "C"
