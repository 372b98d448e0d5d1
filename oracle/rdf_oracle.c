/*
 * rdf_oracle.c -- CPU restatement of 3d-beats' randomized-decision-forest inference.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path in
 * 3d-beats_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the shipped package never does.
 *
 * PARITY STATUS: *parity unpinned* by reference fixtures.  The reference
 * (carsonswope/3d-beats) has no CPU path, no tests and no recorded outputs for the
 * forest kernels, and its CUDA sources cannot be built or run here (no nvcc, no
 * NVIDIA device).  This restatement is pinned only by
 *   (a) the one worked example the reference holds (composite conditions table,
 *       src/decision_tree.py:214-220), and
 *   (b) hand-derived known-answer cases in tests/test_oracle_kat.py, each citing
 *       the .cu lines it was derived from,
 *   (c) agreement with an independent numpy restatement (oracle/rdf_numpy.py).
 *
 * Reference lines followed (paths relative to /root/reference):
 *   forest kernel        src/cuda/tree_eval.cu:24-137
 *   single-tree kernel   src/cuda/tree_eval.cu:140-212
 *   composite kernel     src/cuda/tree_eval.cu:214-248
 *   argmax               src/cuda/tree_eval.cu:7-21
 *   feature              src/cuda/decision_tree_common.hpp:8-28
 *   OOB-default arrays   src/cuda/cu_utils.hpp:43-130
 *   level-order nodes    src/cuda/cu_utils.hpp:19-40
 *
 * Canonical summation order (the reference sums per-tree leaf PDFs with shared-memory
 * float atomicAdd, tree_eval.cu:125, whose order is scheduler dependent): trees are
 * added in index order k = 0..T-1, fp32, starting from +0.0f.
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RDF_NO_PIXEL 65535u /* MAX_UINT16, decision_tree_common.hpp:4 */

/* __float2int_rd: round toward -inf, saturate to int32, NaN -> 0 (CUDA math API). */
static inline int32_t f2i_floor_sat(float x)
{
    if (x != x) return 0;
    float f = floorf(x);
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

/* int + int with two's-complement wrap (what the device add does on overflow). */
static inline int32_t add_wrap(int32_t a, int32_t b)
{
    return (int32_t)((uint32_t)a + (uint32_t)b);
}

/* Array3d<uint16>::get with default 65535: per-axis bounds check, cu_utils.hpp:58-62,79-86.
 * `img` is always in range on this path. */
static inline uint16_t depth_at(const uint16_t *frame, int dim_x, int dim_y, int32_t y, int32_t x)
{
    if (x < 0 || x >= dim_x || y < 0 || y >= dim_y) return RDF_NO_PIXEL;
    return frame[(size_t)y * (size_t)dim_x + (size_t)x];
}

/* decision_tree_common.hpp:8-28.  d is the (already non-zero) centre depth. */
static inline float feature(const uint16_t *frame, int dim_x, int dim_y, int32_t x, int32_t y,
                            uint16_t d, const float *node, float s)
{
    if (d == 0) return 0.f;
    const float df = (float)d;
    /* `uv_scale * u.x / d_f` parses as (uv_scale*u.x)/d_f: one RN multiply, one RN divide */
    const float qux = (float)(s * node[0]) / df;
    const float quy = (float)(s * node[1]) / df;
    const float qvx = (float)(s * node[2]) / df;
    const float qvy = (float)(s * node[3]) / df;
    const int32_t ux = add_wrap(x, f2i_floor_sat(qux));
    const int32_t uy = add_wrap(y, f2i_floor_sat(quy));
    const int32_t vx = add_wrap(x, f2i_floor_sat(qvx));
    const int32_t vy = add_wrap(y, f2i_floor_sat(qvy));
    const float du = (float)depth_at(frame, dim_x, dim_y, uy, ux);
    const float dv = (float)depth_at(frame, dim_x, dim_y, vy, vx);
    return du - dv;
}

/* Walk one tree for one pixel.  Returns pointer to the leaf PDF reached (C floats) or NULL
 * when the walk falls off level D-1 with a "continue" flag (tree_eval.cu:95-128).
 * *levels receives the number of node records read. */
static inline const float *walk(const float *tree, int max_depth, int n_classes,
                                const uint16_t *frame, int dim_x, int dim_y,
                                int32_t x, int32_t y, uint16_t d, float s, int *levels)
{
    const int E = 7 + 2 * n_classes;
    int64_t g = 0;
    int visited = 0;
    const float *leaf = NULL;
    for (int j = 0; j < max_depth; j++) {
        /* BinaryTree::get_ptr, cu_utils.hpp:32-39 */
        const float *node = tree + (((int64_t)1 << j) - 1 + g) * (int64_t)E;
        visited++;
        const int l_next = f2i_floor_sat(node[5]);
        const int r_next = f2i_floor_sat(node[6]);
        const float f = feature(frame, dim_x, dim_y, x, y, d, node, s);
        if (f < node[4]) {
            if (l_next == -1) g = g * 2;
            else { leaf = node + 7; break; }
        } else {
            if (r_next == -1) g = g * 2 + 1;
            else { leaf = node + 7 + n_classes; break; }
        }
    }
    *levels = visited;
    return leaf;
}

/* get_best_pdf_chance, tree_eval.cu:7-21 */
static inline int argmax_pdf(const float *pdf, int n)
{
    float best = 0.f;
    int best_c = 0;
    for (int j = 0; j < n; j++) {
        if (pdf[j] > best) { best = pdf[j]; best_c = j; }
    }
    return best_c;
}

#define RDF_ORACLE_MAX_CLASSES 1024

/*
 * Forest evaluation, tree_eval.cu:24-137.  Pixels the reference kernel returns early on
 * (filter mismatch, depth 0 or 65535) are NOT written.
 * stats (nullable, 3 x uint64): [0] += evaluated label-pixels, [1] += node records read,
 * [2] += leaves reached.  These feed bench.py's algorithmic-byte count.
 * Returns 0, or a negative value for bad arguments.
 */
int rdf_oracle_eval_forest(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                           const float *forest, int n_trees, int max_depth, int n_classes,
                           const uint16_t *filter, int filter_class,
                           uint16_t *labels_out, int labels_reduce, float scale_factor,
                           uint64_t *stats, int n_threads)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_trees < 0 || max_depth < 0 || max_depth > 30 ||
        n_classes < 0 || n_classes > RDF_ORACLE_MAX_CLASSES || labels_reduce < 1)
        return -1;
    const int lw = dim_x / labels_reduce, lh = dim_y / labels_reduce; /* tree_eval.cu:45 */
    const int64_t per_img = (int64_t)lw * lh;
    const int64_t total = per_img * n_img;
    const int64_t nodes = ((int64_t)1 << max_depth) - 1;
    const int E = 7 + 2 * n_classes;
    uint64_t s_px = 0, s_lv = 0, s_lf = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 4096) num_threads(n_threads) reduction(+ : s_px, s_lv, s_lf)
    for (int64_t i = 0; i < total; i++) {
        const int img = (int)(i / per_img);
        const int64_t rem = i % per_img;
        const int ly = (int)(rem / lw), lx = (int)(rem % lw);
        const int32_t y = ly * labels_reduce, x = lx * labels_reduce;
        if (filter_class != -1) { /* tree_eval.cu:81-85 */
            const uint16_t fl = filter[(size_t)img * per_img + (size_t)ly * lw + lx];
            if ((int)fl != filter_class) continue;
        }
        const uint16_t *frame = depth + (size_t)img * dim_x * dim_y;
        const uint16_t d = depth_at(frame, dim_x, dim_y, y, x);
        if (d == 0 || d == RDF_NO_PIXEL) continue; /* tree_eval.cu:88-89 */
        float pdf[RDF_ORACLE_MAX_CLASSES];
        for (int c = 0; c < n_classes; c++) pdf[c] = 0.f; /* tree_eval.cu:57-60 */
        for (int k = 0; k < n_trees; k++) {               /* canonical order */
            int lv;
            const float *leaf = walk(forest + (int64_t)k * nodes * E, max_depth, n_classes,
                                     frame, dim_x, dim_y, x, y, d, scale_factor, &lv);
            s_lv += (uint64_t)lv;
            if (leaf) {
                s_lf++;
                for (int c = 0; c < n_classes; c++) pdf[c] = pdf[c] + leaf[c];
            }
        }
        s_px++;
        labels_out[(size_t)img * per_img + (size_t)ly * lw + lx] = (uint16_t)argmax_pdf(pdf, n_classes);
    }
    if (stats) { stats[0] += s_px; stats[1] += s_lv; stats[2] += s_lf; }
    return 0;
}

/*
 * Single tree, tree_eval.cu:140-212: no filter, no reduce, scale 1.  A pixel whose walk
 * never reaches a leaf is left untouched (the kernel's loop simply ends).
 */
int rdf_oracle_eval_tree(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                         const float *tree, int max_depth, int n_classes,
                         uint16_t *labels_out, int n_threads)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || max_depth < 0 || max_depth > 30 || n_classes < 0)
        return -1;
    const int64_t per_img = (int64_t)dim_x * dim_y;
    const int64_t total = per_img * n_img;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 4096) num_threads(n_threads)
    for (int64_t i = 0; i < total; i++) {
        const int img = (int)(i / per_img);
        const int64_t rem = i % per_img;
        const int32_t y = (int32_t)(rem / dim_x), x = (int32_t)(rem % dim_x);
        const uint16_t *frame = depth + (size_t)img * per_img;
        const uint16_t d = frame[(size_t)y * dim_x + x];
        if (d == 0 || d == RDF_NO_PIXEL) continue; /* tree_eval.cu:170-171 */
        int lv;
        const float *leaf = walk(tree, max_depth, n_classes, frame, dim_x, dim_y, x, y, d, 1.0f, &lv);
        if (leaf) labels_out[(size_t)i] = (uint16_t)argmax_pdf(leaf, n_classes);
    }
    return 0;
}

/*
 * Composite of per-layer label images, tree_eval.cu:214-248.
 * cond is int32[n_cond][2] = (type, value).  The reference does not bounds-check the table and
 * device-asserts when the walk falls off the last layer; here both cases leave the pixel
 * untouched and are counted in *n_bad (nullable).
 */
int rdf_oracle_composite(const uint16_t *const *label_images, int n_images, int dim_x, int dim_y,
                         const int32_t *cond, int n_cond, uint16_t *out, int64_t *n_bad)
{
    if (n_images < 0 || dim_x < 0 || dim_y < 0 || n_cond < 0) return -1;
    int64_t bad = 0;
    for (int y = 0; y < dim_y; y++) {
        for (int x = 0; x < dim_x; x++) {
            const size_t p = (size_t)y * dim_x + x;
            int64_t off = 0;
            int done = 0;
            for (int i = 0; i < n_images && !done; i++) {
                const uint16_t l = label_images[i][p];
                if (l == 0 || l == RDF_NO_PIXEL) { done = 1; break; } /* :235 */
                const int64_t e = off + (int64_t)l - 1;
                if (e < 0 || e >= n_cond) { bad++; done = 1; break; }
                const int32_t t = cond[2 * e], v = cond[2 * e + 1];
                if (t == 0) { out[p] = (uint16_t)v; done = 1; break; } /* :237-239 */
                off = v;                                             /* :242 */
            }
            if (!done) bad++; /* :246-247 fell off the end */
        }
    }
    if (n_bad) *n_bad = bad;
    return 0;
}

/*
 * How many evaluated pixels have an argmax that depends on the order in which the T leaf
 * PDFs are added (all T! orders, T <= 4)?  Documents SURVEY R5: the reference's atomicAdd
 * order is not deterministic, so only order-insensitive pixels are comparable against *any*
 * execution of the reference kernel.  Returns the count, or -1 for bad arguments.
 */
int64_t rdf_oracle_order_sensitive(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                                   const float *forest, int n_trees, int max_depth, int n_classes,
                                   int labels_reduce, float scale_factor)
{
    if (n_trees < 1 || n_trees > 4 || labels_reduce < 1 || n_classes > RDF_ORACLE_MAX_CLASSES) return -1;
    static const int perms4[24][4] = {
        {0,1,2,3},{0,1,3,2},{0,2,1,3},{0,2,3,1},{0,3,1,2},{0,3,2,1},{1,0,2,3},{1,0,3,2},
        {1,2,0,3},{1,2,3,0},{1,3,0,2},{1,3,2,0},{2,0,1,3},{2,0,3,1},{2,1,0,3},{2,1,3,0},
        {2,3,0,1},{2,3,1,0},{3,0,1,2},{3,0,2,1},{3,1,0,2},{3,1,2,0},{3,2,0,1},{3,2,1,0}};
    const int lw = dim_x / labels_reduce, lh = dim_y / labels_reduce;
    const int64_t per_img = (int64_t)lw * lh, total = per_img * n_img;
    const int64_t nodes = ((int64_t)1 << max_depth) - 1;
    const int E = 7 + 2 * n_classes;
    int64_t sensitive = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : sensitive)
    for (int64_t i = 0; i < total; i++) {
        const int img = (int)(i / per_img);
        const int64_t rem = i % per_img;
        const int32_t y = (int32_t)(rem / lw) * labels_reduce, x = (int32_t)(rem % lw) * labels_reduce;
        const uint16_t *frame = depth + (size_t)img * dim_x * dim_y;
        const uint16_t d = depth_at(frame, dim_x, dim_y, y, x);
        if (d == 0 || d == RDF_NO_PIXEL) continue;
        const float *leaf[4] = {0, 0, 0, 0};
        for (int k = 0; k < n_trees; k++) {
            int lv;
            leaf[k] = walk(forest + (int64_t)k * nodes * E, max_depth, n_classes, frame, dim_x, dim_y,
                           x, y, d, scale_factor, &lv);
        }
        int first = -1, differs = 0;
        for (int p = 0; p < 24 && !differs; p++) {
            int ok = 1; /* keep only permutations of the first n_trees indices */
            for (int q = 0; q < 4; q++) {
                if (q < n_trees ? perms4[p][q] >= n_trees : perms4[p][q] != q) ok = 0;
            }
            if (!ok) continue;
            float pdf[RDF_ORACLE_MAX_CLASSES];
            for (int c = 0; c < n_classes; c++) pdf[c] = 0.f;
            for (int q = 0; q < n_trees; q++) {
                const float *l = leaf[perms4[p][q]];
                if (l) for (int c = 0; c < n_classes; c++) pdf[c] = pdf[c] + l[c];
            }
            const int a = argmax_pdf(pdf, n_classes);
            if (first < 0) first = a;
            else if (a != first) differs = 1;
        }
        sensitive += differs;
    }
    return sensitive;
}

/*
 * Which nodes does a batch visit?  `visited` (uint8 [T][2^D - 1], level order like the forest) gets a 1 for every node
 * record some evaluated pixel's walk reads -- the working set behind bench.py's per-level "distinct nodes" figures
 * (how much of a level the workload really touches decides which cache serves it).  Same walk as
 * rdf_oracle_eval_forest (labels_reduce, scale_factor, no filter); writes no labels.  Threads store the same value
 * into a byte, so the races are benign.  Returns 0, or a negative value for bad arguments.
 */
int rdf_oracle_visit_map(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                         const float *forest, int n_trees, int max_depth, int n_classes,
                         int labels_reduce, float scale_factor, uint8_t *visited, int n_threads)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_trees < 0 || max_depth < 0 || max_depth > 30 ||
        n_classes < 0 || labels_reduce < 1 || !visited)
        return -1;
    const int lw = dim_x / labels_reduce, lh = dim_y / labels_reduce;
    const int64_t per_img = (int64_t)lw * lh;
    const int64_t total = per_img * n_img;
    const int64_t nodes = ((int64_t)1 << max_depth) - 1;
    const int E = 7 + 2 * n_classes;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 4096) num_threads(n_threads)
    for (int64_t i = 0; i < total; i++) {
        const int img = (int)(i / per_img);
        const int64_t rem = i % per_img;
        const int32_t y = (int32_t)(rem / lw) * labels_reduce, x = (int32_t)(rem % lw) * labels_reduce;
        const uint16_t *frame = depth + (size_t)img * dim_x * dim_y;
        const uint16_t d = depth_at(frame, dim_x, dim_y, y, x);
        if (d == 0 || d == RDF_NO_PIXEL) continue;
        for (int k = 0; k < n_trees; k++) {
            const float *tree = forest + (int64_t)k * nodes * E;
            uint8_t *vis = visited + (int64_t)k * nodes;
            int64_t g = 0;
            for (int j = 0; j < max_depth; j++) {      /* the loop of walk(), with the node's index kept */
                const int64_t at = (((int64_t)1 << j) - 1 + g);
                const float *node = tree + at * (int64_t)E;
                if (!vis[at]) vis[at] = 1;
                const float f = feature(frame, dim_x, dim_y, x, y, d, node, scale_factor);
                if (f < node[4]) {
                    if (f2i_floor_sat(node[5]) == -1) g = g * 2; else break;
                } else {
                    if (f2i_floor_sat(node[6]) == -1) g = g * 2 + 1; else break;
                }
            }
        }
    }
    return 0;
}

/*
 * How many node records does every walk read?  levels_out (uint8 [n_img][dim_y/r][dim_x/r][n_trees]) gets the number of
 * records the walk of that label pixel in that tree reads (0 for pixels the kernel returns early on).  Same walk as
 * rdf_oracle_eval_forest without a filter; feeds tools/refill_bound.py (what refilling lanes could save at most).
 */
int rdf_oracle_walk_lengths(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                            const float *forest, int n_trees, int max_depth, int n_classes,
                            int labels_reduce, float scale_factor, uint8_t *levels_out, int n_threads)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_trees < 0 || max_depth < 0 || max_depth > 30 ||
        n_classes < 0 || labels_reduce < 1 || !levels_out)
        return -1;
    const int lw = dim_x / labels_reduce, lh = dim_y / labels_reduce;
    const int64_t per_img = (int64_t)lw * lh;
    const int64_t total = per_img * n_img;
    const int64_t nodes = ((int64_t)1 << max_depth) - 1;
    const int E = 7 + 2 * n_classes;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 4096) num_threads(n_threads)
    for (int64_t i = 0; i < total; i++) {
        const int img = (int)(i / per_img);
        const int64_t rem = i % per_img;
        const int32_t y = (int32_t)(rem / lw) * labels_reduce, x = (int32_t)(rem % lw) * labels_reduce;
        const uint16_t *frame = depth + (size_t)img * dim_x * dim_y;
        const uint16_t d = depth_at(frame, dim_x, dim_y, y, x);
        for (int k = 0; k < n_trees; k++) {
            int lv = 0;
            if (d != 0 && d != RDF_NO_PIXEL)
                (void)walk(forest + (int64_t)k * nodes * E, max_depth, n_classes, frame, dim_x, dim_y, x, y, d, scale_factor, &lv);
            levels_out[(size_t)i * n_trees + k] = (uint8_t)lv;
        }
    }
    return 0;
}

int rdf_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
