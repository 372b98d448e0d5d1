/*
 * rdf_hip.h -- C ABI of librdf_hip.so: per-pixel randomized-decision-forest inference on
 * MI355X (gfx950).  Drop-in boundary for the three kernels of 3d-beats' tree_eval.cu; a host
 * language binds these symbols where the reference binds PyCUDA's
 * `SourceModule.get_function(...)` callables (src/decision_tree.py:269-272,
 * src/cuda/py_nvcc_utils.py:25-37).
 *
 * Conventions
 *   - every pointer is caller-owned DEVICE memory unless stated; the library allocates nothing
 *     and frees nothing (same ownership as the reference: forest_cu / label buffers belong to
 *     the caller, src/decision_tree.py:167, 203-207);
 *   - launches are asynchronous on `stream` (a hipStream_t; NULL = the default stream), as the
 *     reference's launches are on the CUDA default stream;
 *   - return value: 0 on success, a negative RDF_ERR_* for rejected arguments, or a positive
 *     hipError_t reported by the HIP runtime.  rdf_error_string() names either kind;
 *   - pixels the reference kernels `return` early on are never written ("untouched" rule).
 */
#ifndef RDF_HIP_H
#define RDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rdf_forest_packed_bytes grew (last-level table behind the three per-slot tables: re-pack with this library's
 * rdf_forest_pack); new: rdf_set_last_level_table, rdf_eval_forest_packed_filled.  Nothing was removed or re-typed.
 * 3: rdf_forest_packed_bytes grew again (deep blocks behind the last-level table; `packed` must be 128-byte aligned);
 * new: rdf_set_deep_from, rdf_forest_set_deep_from, rdf_forest_tune, rdf_eval_forest_packed_stats.  Nothing was removed or re-typed.
 * 4: the choice of rdf_forest_set_deep_from / rdf_forest_tune is kept IN the packed table (its info block) and a table nobody
 * chose for is walked from the heap-order records (version 3 chose by forest size); new: rdf_forest_info,
 * rdf_forest_forget, rdf_build_id, RDF_ERR_CAPTURE (was RDF_ERR_BAD_ARG).  Nothing was removed or re-typed; table sizes unchanged.
 * 5: a packed table's info block carries the shape it was packed for and a generation number (tables of version 4 are refused:
 * re-pack); new: RDF_ERR_STALE, rdf_eval_forest_packed_split.  Nothing was removed or re-typed; table sizes unchanged. */
#define RDF_ABI_VERSION 5

#define RDF_OK 0
#define RDF_ERR_BAD_ARG (-1)     /* negative size, labels_reduce < 1, max_depth outside [0,30] ... */
#define RDF_ERR_NULL_PTR (-2)    /* a required pointer is NULL */
#define RDF_ERR_TOO_LARGE (-3)   /* one call addresses >= 2^31 depth pixels (split the batch), or dim_x >= 2^23 / dim_y >= 2^24 */
#define RDF_ERR_NO_DEVICE (-4)   /* no HIP device / not a gfx950 code object */
#define RDF_ERR_BUILD (-5)       /* the library was built in a way its own kernels do not allow (a forest kernel got static LDS:
                                    its depth tile must sit at LDS address 0) */
#define RDF_ERR_CAPTURE (-6)     /* `stream` is being captured into a hipGraph and the call needs a synchronous step: the FIRST
                                    evaluation of a packed table this process has not seen at this address reads the table's info
                                    block back (evaluate once, or rdf_forest_pack, before capturing) */
#define RDF_ERR_STALE (-7)       /* a kernel of an EARLIER call on this device found, at a packed table's address, another table than
                                    the one this process remembered there -- a different table was written over a known address by
                                    other means than rdf_forest_pack and rdf_forest_forget was not called.  Every packing carries a
                                    generation number in its info block; every launch carries the number the host remembers and
                                    compares the two on the device (no synchronous read).  That earlier call's labels are right
                                    unless the new table holds nodes that need the exact numerators and the call passed no `forest`
                                    (the kernel reads the scale from the table itself, never from the host's memory).  The library
                                    has dropped everything it knew about this device's tables when it returns this code: call
                                    again, and the info blocks are read afresh. */

/*
 * Forest evaluation.  Replaces `evaluate_image_using_forest`
 * (src/cuda/tree_eval.cu:24-137; launched from src/decision_tree.py:298-330).
 *   depth       uint16 [n_img][dim_y][dim_x]; 0 and 65535 mean "no pixel"
 *   forest      float32 [n_trees][2^max_depth - 1][7 + 2*n_classes], the reference's .npy layout
 *   filter      uint16 [n_img][dim_y/r][dim_x/r], or NULL; used only when filter_class != -1
 *   labels_out  uint16 [n_img][dim_y/r][dim_x/r]
 * Per-tree leaf PDFs are summed in tree order 0..n_trees-1 in fp32 from +0.0f.
 */
int rdf_eval_forest(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                    const float *forest, int n_trees, int max_depth, int n_classes,
                    const uint16_t *filter, int filter_class,
                    uint16_t *labels_out, int labels_reduce, float scale_factor, void *stream);

/*
 * Single-tree evaluation.  Replaces `evaluate_image_using_tree`
 * (src/cuda/tree_eval.cu:140-212; launched from src/decision_tree.py:277-294).
 * No filter, labels_reduce 1, scale 1; a walk that reaches no leaf leaves the pixel untouched.
 */
int rdf_eval_tree(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                  const float *tree, int max_depth, int n_classes,
                  uint16_t *labels_out, void *stream);

/*
 * Composite label image.  Replaces `make_composite_labels_image`
 * (src/cuda/tree_eval.cu:214-248; launched from src/decision_tree.py:333-347).
 *   label_images  DEVICE array of n_images device pointers to uint16 [dim_y][dim_x]
 *                 (the int64 pointer table of src/decision_tree.py:205-207)
 *   cond          int32 [n_cond][2] = (type, value) rows (src/decision_tree.py:209-223)
 *   bad_count     optional device int32, incremented once per pixel whose walk leaves the table
 *                 or falls off the last image (the reference device-asserts there, :246-247);
 *                 such pixels are left untouched.  May be NULL.
 */
int rdf_composite(const uint16_t *const *label_images, int n_images, int dim_x, int dim_y,
                  const int32_t *cond, int n_cond, uint16_t *out, int32_t *bad_count, void *stream);

/*
 * One call for LayeredDecisionForest.run (src/decision_tree.py:233-264): every layer's forest
 * evaluation (layer i optionally filtered on layer filter_layer[i]'s labels == filter_class[i])
 * followed by the composite.  Equivalent to: fill composite_out and every layer_labels[i] with
 * 65535; rdf_eval_forest[_packed] per layer in order; rdf_composite -- but the fills are fused
 * into the kernels (they store 65535 wherever they write nothing), so a 2-layer run is 3
 * launches instead of 6.  One image per call, as in the reference.  A launch too small to fill the chip (one frame) of a
 * packed stack of two or three layers evaluates the layers unfiltered in ONE launch (workgroup b takes layer
 * b % n_layers) and applies the filters in the composite kernel: same label images and composite, one ramp and drain
 * instead of n_layers (rdf_set_layers_one_launch(0) turns it off).
 *   packed, forests, n_trees, max_depth, n_classes, filter_layer (-1 = none), filter_class,
 *   layer_labels are HOST arrays of length n_layers (pointers inside them are device pointers;
 *   packed may be NULL, or hold NULL entries, to evaluate from the reference-layout forest);
 *   layer_labels_dev_table is the DEVICE pointer table the composite reads
 *   (src/decision_tree.py:205-207).  A packed table must have been built for scale_factor.
 */
int rdf_layered_run(const uint16_t *depth, int dim_x, int dim_y, int n_layers,
                    const void *const *packed, const float *const *forests,
                    const int *n_trees, const int *max_depth, const int *n_classes,
                    const int *filter_layer, const int *filter_class,
                    uint16_t *const *layer_labels, const uint16_t *const *layer_labels_dev_table,
                    const int32_t *cond, int n_cond, uint16_t *composite_out, int32_t *bad_count,
                    int labels_reduce, float scale_factor, void *stream);

/*
 * rdf_layered_run plus what the app does with the composite of one hand (src/3d_bz.py:440-456), folded into the composite
 * kernel's store instead of three more launches: flip_x != 0 writes the composite mirrored in x (the left hand's frame
 * was flipped on the way in, :402-404, and its labels are flipped back, :441-447) and image_rgba, if not NULL, receives
 * colors_rgba[label - 1] for every pixel that gets a label in 1..num_colors, at the place the label is written
 * (make_rgba_from_labels, src/cuda/points_ops.cu:258-281; other texels are left alone).  The per-layer label images
 * stay unflipped.
 */
int rdf_layered_run_hand(const uint16_t *depth, int dim_x, int dim_y, int n_layers,
                         const void *const *packed, const float *const *forests,
                         const int *n_trees, const int *max_depth, const int *n_classes,
                         const int *filter_layer, const int *filter_class,
                         uint16_t *const *layer_labels, const uint16_t *const *layer_labels_dev_table,
                         const int32_t *cond, int n_cond, uint16_t *composite_out, int32_t *bad_count,
                         int labels_reduce, float scale_factor, int flip_x, const uint8_t *colors_rgba,
                         int num_colors, uint8_t *image_rgba, void *stream);

/*
 * Load-time repack of a forest into a table of 16-byte hot records {23-bit floor(s*u), 23-bit floor(s*v), integer threshold,
 * leaf flags} followed by the leaf PDFs as 16-byte aligned rows [left: C padded to a multiple of 4][right: ...] per node.
 * The reference has no counterpart: its "load" is the plain upload at src/decision_tree.py:148-158.  The packed tables depend
 * on scale_factor and must be rebuilt when the forest changes.  `packed` is caller-owned, 128-byte aligned,
 * rdf_forest_packed_bytes() bytes; packed tables support max_depth <= 27 (32-bit byte offsets inside one tree).
 * A node whose numerators s*u, s*v the integer record cannot hold (|a| >= 2^21, denormal, inf, NaN) is flagged and takes the
 * IEEE divide on the fp32 numerators, which the kernel recomputes from the node's record in the caller's reference-layout
 * forest: rdf_forest_pack counts such nodes, and a packed table that has any needs the `forest` argument at evaluation
 * (RDF_ERR_NULL_PTR otherwise); the count, the scale and a mark sit in the table's last 128 bytes, and rdf_forest_pack waits
 * for them (it is load-time work), so that evaluations -- also those recorded into a hipGraph -- never ask the device.
 * Forests of up to four classes and two or more levels carry a table of one 64-byte record per node of the deepest level,
 * {hot record, left PDF, right PDF}, and a 64-byte trailer: word 0 counts the records that are not ordinary nodes with two
 * leaves, word 1 the records whose parent continues to them.  When word 0 is zero and word 1 is at least half the level, a walk
 * takes its last node and its leaf PDF from one cache line (rdf_set_last_level_table); otherwise the table is ignored.
 * Forests of up to eight classes and five to twenty-four levels carry the deep blocks, 128-byte aligned: the hot records once more,
 * grouped into three-level subtrees of seven records per 128-byte line, the last two levels together with their four leaf
 * PDFs in one line (five to eight classes: the last level's node with its two PDFs), one all-zero line and a 128-byte
 * trailer {1 + deepest level holding a flagged node, nodes of the last level that are not plain two-leaf nodes}.  Launches
 * walk the levels no cache holds from it, a third of the line fetches per walk (rdf_set_deep_from).
 * Footprint, per heap slot (2^max_depth slots per tree): 16 B hot + 8 B x classes (padded to 4) of leaf PDFs, + 32 B (half a
 * 64-byte record per slot of the deepest level; up to four classes) + ~37 B of deep blocks (128 B per seven nodes plus 128 B
 * per node of level D - 2): a T4/D20/C4 forest (240 MiB as .npy) packs into 64 + 128 + 128 + 146 MiB = 466 MiB, T8/D22/C4
 * (1.9 GiB) into 3.6 GiB -- 1.9 x the model, 1.3 % of one MI355X's HBM; every table exists so that a walk touches ONE cache
 * line where the .npy layout touches two or three.
 */
size_t rdf_forest_packed_bytes(int n_trees, int max_depth, int n_classes);
int rdf_forest_pack(const float *forest, int n_trees, int max_depth, int n_classes,
                    float scale_factor, void *packed, void *stream);

/*
 * The launch rdf_eval_forest_packed would make for these arguments (same workgroup size, halo, LDS levels, table choice),
 * with counters on; forests of up to four classes, no filter.  stats8 (8 x uint64, device memory, accumulated into):
 *   [0] label pixels evaluated   [1] node records read   [2] leaves reached   (as rdf_eval_forest_stats)
 *   [3] node records read from LDS
 *   128-byte lines the loads that serve WALKING slots touch, per wave instruction, neighbouring lanes on one line merged
 *   (what the L1 merges): [4] node records from global memory (last-level records included), [5] leaf rows of the PDF table,
 *   [6] far probes that load (outside the staged tile, inside the image), [7] deep blocks.
 * [4] + [5] + [6] + [7] is the least the launch needs from the L1 / texture-addresser pipeline with this geometry: the
 * "useful" numerator of bench.py's roofline (tools/roofline.py), next to the issued accesses the hardware counts.
 */
int rdf_eval_forest_packed_stats(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                                 const float *forest, int n_trees, int max_depth, int n_classes, uint16_t *labels_out,
                                 int labels_reduce, unsigned long long *stats8, void *stream);

/*
 * Which table serves a packed forest's deep levels is a property of the forest AND of the frames: a forest whose deep levels
 * are occupied (a trained forest) walks them fastest from the deep blocks, one that sends most pixels down a few paths from
 * the heap-order records, and nothing in the records tells the two apart.  A table nobody chose for is walked from the
 * heap-order records.  rdf_forest_set_deep_from makes the choice for one packed table (level > 0: deep blocks from that level
 * on, 0: never, -1: no choice) and WRITES IT INTO THE TABLE's info block (a synchronous 4-byte copy), so it stays with the
 * table: later evaluations, a copy of the table, another process that maps it all find it (the host keeps what it knows of a
 * table per device and address and reads the info block once); rdf_forest_pack into the same memory starts over.
 * rdf_forest_info reports what a table's info block says -- the choice (-1: none made), how many nodes need the exact
 * numerators (> 0: evaluations need the caller's forest), the scale it was packed for -- reading the block back first if this
 * process has not seen the table at this address yet (synchronous then; RDF_ERR_BAD_ARG for memory that is not a table of
 * this shape written by this library version's rdf_forest_pack; all three outputs nullable).
 * rdf_forest_tune makes the choice by
 * measurement: it evaluates the caller's sample frames (device memory; results go to `labels_scratch`, uint16
 * [n_img][dim_y/r][dim_x/r]) with every candidate -- never, and each block root level -- four to thirteen launches each, keeps the
 * fastest (the deep blocks must beat the heap-order records by 2 %: a tie goes to the default) and reports what it tried
 * (up to 12 entries in levels_tried / ms_tried, all three outputs nullable).  Synchronous; tune once per table, at load time.
 * Labels do not depend on the choice.  The process-wide knob rdf_set_deep_from (>= 0) overrides both.
 * rdf_forest_forget drops what the host remembers about the table at `packed` (exact-node count, choice, generation): call it
 * before freeing a packed table, or after writing a DIFFERENT packed table to an address this process has evaluated from by
 * any means other than rdf_forest_pack (device-to-device copy, IPC mapping) -- the next evaluation then reads the info block
 * again.  A consumer that forgets to is told (round 6): the kernels compare the table's generation word with the one the
 * host remembers and the call after such a launch returns RDF_ERR_STALE (see there); the scale is read by the kernel from the
 * table itself, so even that launch's labels are the new table's.  The life of a table in a C program:
 *     hipMalloc(&packed, rdf_forest_packed_bytes(T, D, C));  rdf_forest_pack(forest, T, D, C, s, packed, stream);
 *     [rdf_forest_tune(...) once]  rdf_eval_forest_packed(...) ...  rdf_forest_forget(packed);  hipFree(packed);
 * (examples/eval_forest.c).  rdf_forest_set_deep_from and rdf_forest_tune WRITE to the table (`packed` is const for the
 * evaluation calls only): a synchronous 4-byte copy, device-wide -- load-time calls, not to be made while another thread
 * captures a hipGraph in global mode.  All four calls find the table's device from the pointer, not from the current device.
 */
int rdf_forest_set_deep_from(const void *packed, int level);
int rdf_forest_info(const void *packed, int n_trees, int max_depth, int n_classes, void *stream, int *deep_from,
                    int *exact_nodes, float *scale);
int rdf_forest_forget(const void *packed);
int rdf_forest_tune(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed, const float *forest,
                    int n_trees, int max_depth, int n_classes, uint16_t *labels_scratch, int labels_reduce, void *stream,
                    int *chosen_level, int *n_tried, int *levels_tried, float *ms_tried);

/* rdf_eval_forest on a packed table (hot records and leaf PDFs are read from `packed`); `forest` (original layout, the
 * array the table was packed from) is read for the nodes that need the exact numerators -- it may be NULL for a table that
 * has none (RDF_ERR_NULL_PTR otherwise) -- and stands in when `packed` is NULL for a degenerate forest (no tree or depth 0).
 * A buffer that rdf_forest_pack did not write (for these trees / depth / classes) is refused with RDF_ERR_BAD_ARG. */
int rdf_eval_forest_packed(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                           const void *packed, const float *forest,
                           int n_trees, int max_depth, int n_classes,
                           const uint16_t *filter, int filter_class,
                           uint16_t *labels_out, int labels_reduce, void *stream);

/* The same with the caller's pre-fill folded in: every label pixel the evaluation leaves alone (no depth, filtered out)
 * is written 65535, so labels_out needs no fill before the call (the fills of src/decision_tree.py:237-240 and the
 * `labels.fill(MAX_UINT16)` every caller of get_labels_forest does first).  One pass over the labels instead of two. */
int rdf_eval_forest_packed_filled(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                                  const void *packed, const float *forest,
                                  int n_trees, int max_depth, int n_classes,
                                  const uint16_t *filter, int filter_class,
                                  uint16_t *labels_out, int labels_reduce, void *stream);

/*
 * One evaluation as TWO launches that share one tile queue: a main launch on `stream` and a helper launch on `helper_stream`.
 * For multi-GPU steps whose label gather is an RCCL kernel (no reference counterpart: the reference is single-GPU,
 * src/engine/window.py:45-46): RCCL's send/recv kernel cannot start next to the forest kernel's persistent workgroups unless some
 * compute units are left to it (rdf_stream_create_with_reserved_cus), and those units are idle for the part of the step the
 * gather does not need.  Here `stream` is the CU-masked stream -- the main launch starts at once on the units it may use -- and
 * the caller makes `helper_stream` (an ordinary stream) wait for the previous step's gather before this call: the helper's
 * workgroups then start on the units the gather has just left and pull tiles from the same queue until it is empty, however
 * early or late they arrive (a helper that arrives after the main launch has finished finds the queue empty and retires).
 * Both streams are the caller's to order: `helper_stream` must ALSO come after whatever wrote `depth`, `filter` and the
 * pre-fill of `labels_out` (a helper that runs ahead of a fill sees its labels overwritten), and whoever reads `labels_out`
 * must wait for both streams.
 *   helper_cus    compute units the helper may count on (its grid is this many times the workgroups one unit holds); 0: one
 *                 ordinary launch
 *   queue_tag     0..3: which of `stream`'s split queue slots the two launches share.  A slot must not be used by a later
 *                 call before BOTH launches of this one have finished: alternate the tag from step to step, or order the next
 *                 call on `stream` after `helper_stream`
 *   fill_untouched  as rdf_eval_forest_packed_filled
 *   helper_workgroups (out, nullable)  workgroups the helper launch got; 0: the launch was not split (it fits one round of
 *                 the main launch, or the scheduler is not the dynamic one)
 * Labels are the same as rdf_eval_forest_packed's.  Not recordable into a hipGraph (RDF_ERR_CAPTURE).
 */
int rdf_eval_forest_packed_split(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                                 const void *packed, const float *forest,
                                 int n_trees, int max_depth, int n_classes,
                                 const uint16_t *filter, int filter_class,
                                 uint16_t *labels_out, int labels_reduce, int fill_untouched,
                                 void *stream, void *helper_stream, int helper_cus, int queue_tag, int *helper_workgroups);

/*
 * Visit counters for the roofline figure: same walk as rdf_eval_forest, labels_out
 * written identically; stats (device uint64[3]) += {evaluated label-pixels, node records read,
 * leaves reached}.
 */
int rdf_eval_forest_stats(const uint16_t *depth, int n_img, int dim_x, int dim_y,
                          const float *forest, int n_trees, int max_depth, int n_classes,
                          const uint16_t *filter, int filter_class,
                          uint16_t *labels_out, int labels_reduce, float scale_factor,
                          unsigned long long *stats, void *stream);

/*
 * ---- consumer of the composite label map ----
 * Per-class 2-D mean-shift mode finding.  Replaces the kernel `run` of src/cuda/mean_shift.cu:3-48
 * AND the host loop of src/cuda/mean_shift.py:35-59 (per round: zero the sums, launch, copy sums and
 * means to the host, divide, add, copy back): all `num_rounds` rounds of all classes run on the device
 * in ONE launch (one workgroup per class), no host round trip, bitwise-reproducible sums (no atomics).
 *   labels     uint16 [dim_y][dim_x]; 0, 65535 and values > num_classes are ignored
 *   variances  float32 [num_classes] (device): ALL num_classes entries are read, also those of classes without pixels
 *              (the reference only reads the entries of labels that occur)
 *   means_out  float64 [num_classes][2] = (x, y) per class (device); NaN for a class without pixels,
 *              as the reference's 0/0
 *   workspace  unused since round 2 (rdf_mean_shift_workspace_bytes() returns 0); may be NULL
 * num_classes <= 64, dim_x and dim_y <= 65535.
 */
size_t rdf_mean_shift_workspace_bytes(int num_classes, int num_rounds);
int rdf_mean_shift(const uint16_t *labels, int dim_x, int dim_y, int num_classes, const float *variances,
                   int num_rounds, double *means_out, void *workspace, void *stream);

/*
 * Height of each requested class's mode above the calibrated plane.  Replaces the host code of
 * src/3d_bz.py:503-522: px,py = int(mean) * labels_reduce; z = depth[py][px];
 * pt = rs2_deproject_pixel_to_point (librealsense2, distortion-free intrinsics: z*((px-ppx)/fx,
 * (py-ppy)/fy, 1) in fp32); height = -(plane @ [pt,1]).z.  NaN where the reference "resets" the
 * fingertip (mode off-frame or NaN).
 *   means float64 [num_classes][2] (device); class_ids int32 [n_ids] (device, 1-based labels);
 *   plane float32 [4][4] row-major (device); heights_out float64 [n_ids] (device)
 */
int rdf_fingertip_heights(const double *means, int num_classes, const int *class_ids, int n_ids,
                          const uint16_t *depth, int dim_x, int dim_y, int labels_reduce, float fx, float fy,
                          float ppx, float ppy, const float *plane, double *heights_out, void *stream);

/*
 * rdf_mean_shift followed by rdf_fingertip_heights in ONE launch: the workgroup that finds a class's mode also writes
 * the heights of the ids that name that class (ids that name no class: NaN).  Same means and heights, bit for bit, as the
 * two calls; what the app's per-hand chain uses (src/3d_bz.py:461-465 then :503-522).  n_ids <= 1024.  means_out and
 * heights_out may be device memory or device-accessible (pinned, mapped) host memory.
 */
int rdf_mean_shift_heights(const uint16_t *labels, int dim_x, int dim_y, int num_classes, const float *variances,
                           int num_rounds, double *means_out, const int *class_ids, int n_ids, const uint16_t *depth,
                           int depth_dim_x, int depth_dim_y, int labels_reduce, float fx, float fy, float ppx, float ppy,
                           const float *plane, double *heights_out, void *stream);

/*
 * ---- element-wise kernels either side of the forest; all in place / byte exact ----
 * rdf_convert_0s_to_maxuint           src/cuda/points_ops.cu:117-127   depth[i] == 0 -> 65535
 * rdf_setup_depth_image_for_forest    :149-165   depth[i] == 0 or pts[i].w == 0 -> 65535 (pts = float4 per pixel)
 * rdf_stencil_depth_image_by_group    :440-463   d_out[y][x] = d_in[y][x] where groups[y>>level][x>>level] == group
 * rdf_flip_x                          :466-483   out[y][W-1-x] = in[y][x]
 * rdf_make_rgba_from_labels           :258-281   image[y][x] = colors[l-1] for labels l not in {0, 65535};
 *                                                labels > num_colors are skipped (the reference reads from nullptr)
 */
int rdf_convert_0s_to_maxuint(uint16_t *depth, size_t num_pixels, void *stream);
int rdf_setup_depth_image_for_forest(const float *pts_xyzw, uint16_t *depth, size_t num_pixels, void *stream);
int rdf_stencil_depth_image_by_group(int dim_x, int dim_y, int mipmap_level, int group, const uint16_t *groups_in,
                                     const uint16_t *depth_in, uint16_t *depth_out, void *stream);
int rdf_flip_x(int dim_x, int dim_y, const uint16_t *in, uint16_t *out, void *stream);
/* The app's per-hand input chain (src/3d_bz.py:396-420) in one pass: depth_out[y][x'] = depth_in[y][x] where the pixel's
 * hand group (groups_in at mip level mipmap_level, as rdf_stencil_depth_image_by_group reads it) equals `group` and the
 * depth is not 0, else 65535; x' = dim_x - 1 - x if flip_x.  Equals fill(0) + rdf_stencil_depth_image_by_group +
 * rdf_flip_x (or a copy) + rdf_convert_0s_to_maxuint.  depth_in and depth_out may be the same buffer only when flip_x == 0. */
int rdf_prepare_hand_depth(int dim_x, int dim_y, int mipmap_level, int group, const uint16_t *groups_in,
                           const uint16_t *depth_in, uint16_t *depth_out, int flip_x, void *stream);
int rdf_make_rgba_from_labels(int dim_x, int dim_y, int num_colors, const uint16_t *labels,
                              const uint8_t *colors_rgba, uint8_t *image_rgba, void *stream);

/*
 * ---- training of one tree; kernel parameters as in src/cuda/tree_train.cu ----
 * rdf_train_init            root class counts + nodes_by_pixel = (label > 0 ? 0 : -1): the host-side set-up of
 *                           src/decision_tree.py:452-468, on the device; root_counts[n_classes] must be zero
 * rdf_train_histogram       evaluate_random_features, tree_train.cu:4-64: counts[j][child - node_start][label] += 1 for
 *                           every live pixel whose children fall in [node_start, node_end); counts is uint64
 *                           [n_proposals][nodes_per_block][n_classes], zeroed by the caller; proposals float32 [P][5]
 * rdf_train_histogram_left  the same, but only the LEFT children are counted (half the atomics, which bound the kernel);
 * rdf_train_histogram_left_ws  the same result as rdf_train_histogram_left, faster: it counts into a caller-owned
 *                           workspace of rdf_train_histogram_workspace_bytes() bytes (8-byte aligned, ZERO before the first
 *                           call; every call leaves it zero) where one 64-bit atomic serves two proposals -- or four, for a
 *                           (node, class) with at most 65535 pixels according to parent_counts ([node][class] counts of the
 *                           live pixels, as kept by the trainer; may be NULL) -- then adds the workspace into `counts`
 * rdf_train_right_counts    then fills counts[j][right][c] = parent_counts[node][c] - counts[j][left][c] for the children of
 *                           the active nodes that fall in [node_start, node_end).  Call it once, after every image has
 *                           been counted; the pair leaves `counts` exactly as rdf_train_histogram does
 * rdf_train_pick_best       pick_best_features, tree_train.cu:99-236 (same arguments)
 * rdf_train_next_active     get_active_nodes_next_level, tree_train.cu:238-273, but in ascending order of the parents
 *                           (the reference appends in scheduler order); *n_next_active is written on the device
 * rdf_train_update_pixels   copy_pixel_groups, tree_train.cu:275-324
 * All images of the training set are addressed in one call (no image blocks): n_img*dim_x*dim_y < 2^31.
 */
int rdf_train_init(const uint16_t *labels, size_t n_px, int n_classes, int32_t *nodes_by_pixel,
                   unsigned long long *root_counts, void *stream);
int rdf_train_histogram(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                        int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes,
                        int node_start, int node_end, int nodes_per_block, unsigned long long *counts, void *stream);
int rdf_train_histogram_left(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                             int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes,
                             int node_start, int node_end, int nodes_per_block, unsigned long long *counts, void *stream);
size_t rdf_train_histogram_workspace_bytes(int n_proposals, int nodes_per_block, int n_classes);
int rdf_train_histogram_left_ws(const uint16_t *depth, const uint16_t *labels, const int32_t *nodes_by_pixel, int n_img,
                                int dim_x, int dim_y, const float *proposals, int n_proposals, int n_classes,
                                int node_start, int node_end, int nodes_per_block, unsigned long long *counts,
                                void *workspace, const unsigned long long *parent_counts, void *stream);
/*
 * The same left-child counts without an atomic per (wave, group): once per level rdf_train_sort_pixels gives every live
 * pixel (nodes_by_pixel >= 0, label < n_classes) a row number in (node, class) order (pos[pixel], -1 for the others;
 * rowkey[row] = node * n_classes + label; the number of rows is kept in the workspace, which the call zeroes itself:
 * rdf_train_sort_workspace_bytes(n_nodes, n_classes) bytes, n_nodes = nodes of the current level);
 * per proposal block rdf_train_decision_bits writes one row of bits per live pixel (bit j = the pixel goes left under
 * proposal j; rdf_train_bits_row_bytes(n_proposals) bytes per row, n_proposals <= 1024, rows for every labelled pixel
 * must fit) and rdf_train_count_rows adds the rows of each (node, class) group up into
 * counts[j][left child - node_start][class], exactly what rdf_train_histogram_left leaves there (counts is zeroed by the
 * caller; rdf_train_right_counts then fills the right children).  The decision bits do not depend on the node block:
 * one rdf_train_decision_bits serves every rdf_train_count_rows of a proposal block.
 */
size_t rdf_train_sort_workspace_bytes(int n_nodes, int n_classes);
size_t rdf_train_bits_row_bytes(int n_proposals);
size_t rdf_train_bits_workspace_bytes(int n_proposals);   /* scratch of rdf_train_decision_bits (the proposals, prepared) */
int rdf_train_sort_pixels(const uint16_t *labels, const int32_t *nodes_by_pixel, size_t n_px, int n_classes, int n_nodes,
                          int32_t *pos, int32_t *rowkey, void *workspace, void *stream);
int rdf_train_decision_bits(const uint16_t *depth, const int32_t *pos, int n_img, int dim_x, int dim_y,
                            const float *proposals, int n_proposals, void *bits, void *workspace, void *stream);
int rdf_train_count_rows(const void *bits, const int32_t *rowkey, const void *sort_workspace, int n_nodes, int n_proposals,
                         int n_classes, int node_start, int node_end, int nodes_per_block, unsigned long long *counts,
                         void *stream);
int rdf_train_right_counts(int n_active, const int32_t *active_nodes, int n_proposals, int nodes_per_block,
                           int node_start, int node_end, int n_classes, const unsigned long long *parent_counts,
                           unsigned long long *counts, void *stream);
int rdf_train_pick_best(int n_active, const int32_t *active_nodes, int n_proposals, int max_depth, int nodes_per_block,
                        int node_start, int node_end, int n_classes, int level,
                        const unsigned long long *parent_counts, const unsigned long long *counts_by_feature,
                        const float *proposals, float *tree_out, unsigned long long *child_counts,
                        float *best_gain_per_node, void *stream);
int rdf_train_next_active(int level, int max_depth, int n_classes, const float *tree, const int32_t *active_nodes,
                          int n_active, int32_t *next_active_nodes, int32_t *n_next_active, void *stream);
int rdf_train_update_pixels(const uint16_t *depth, int n_img, int dim_x, int dim_y, int level, int max_depth,
                            int n_classes, int32_t *nodes_by_pixel, const float *tree, void *stream);

/* GPUArray.fill(65535) of src/decision_tree.py:237-240 for uint16 buffers. */
int rdf_fill_u16(uint16_t *dst, size_t n, uint16_t value, void *stream);

/* A stream whose kernels leave the first n_reserved CUs (hipExtStreamCreateWithCUMask numbering; use a multiple of 32 =
 * one CU per shader engine on MI355X) to other streams, for running the forest kernel next to RCCL: DESIGN.md section 6.
 * rdf_stream_destroy waits for the stream's work, gives its tile-queue slot back and destroys it (call it for any stream
 * that launched forest kernels and is going away: a device has 128 stream slots; beyond them launches fall back to
 * static tiles). */
int rdf_stream_create_with_reserved_cus(void **stream, int n_reserved);
int rdf_stream_destroy(void *stream);
/* A forest launch recorded into a hipGraph (stream capture) takes a tile-queue slot of its own -- a graph replays on any
 * stream -- out of 384 per device; when they are all taken, further captured launches use static tiles (correct, slower on
 * uneven batches; the library says so once on stderr).  The owner of a graph gives its slots back when the executable graph
 * is gone: rdf_stream_capture_id(stream, &id) during the capture names it, rdf_graph_slots_release(id) afterwards returns
 * the number of slots released (0 for an unknown id).  Releasing the slots of a graph that is still replayed is an error
 * the library cannot see. */
int rdf_stream_capture_id(void *stream, unsigned long long *capture_id);
int rdf_graph_slots_release(unsigned long long capture_id);
/* test hook: tile-queue slots of the current device held by streams / by launches recorded into hipGraphs right now */
int rdf_debug_sched_slots(int *stream_slots_in_use, int *graph_slots_used);
/* Peer-to-peer plumbing for the multi-GPU gather (no reference counterpart): a raw device allocation
 * whose 64-byte IPC handle another process of the node opens to get a pointer it can copy into.  rdf_memcpy_device_async
 * is hipMemcpyAsync(hipMemcpyDefault): device-to-device copies between GPUs run on the copy engines, not on CUs. */
int rdf_device_malloc(void **ptr, size_t bytes);
int rdf_device_free(void *ptr);
int rdf_ipc_export(void *ptr, unsigned char handle_out[64]);
int rdf_ipc_open(const unsigned char handle[64], void **ptr_out);
int rdf_ipc_close(void *ptr);
int rdf_memcpy_device_async(void *dst, const void *src, size_t bytes, void *stream);
/* Test hook: n_workgroups workgroups with RCCL's send/recv kernel's footprint (256 threads, >250 VGPRs, 19.7 KB LDS) that
 * record their start time (wall_clock64 ticks, 100 MHz) in t_start[workgroup] and spin for spin_ticks. */
int rdf_debug_fat_kernel(int n_workgroups, unsigned long long spin_ticks, unsigned long long *t_start, void *stream);
/* Test hook: out[i] = __float2int_rd(in[i]) as the kernels compute it (floor, saturate, NaN -> 0). */
int rdf_debug_floor_i32(const float *in, int32_t *out, size_t n, void *stream);
/* Test hook: out[i] = num[i] / den[i] as the kernels compute it (IEEE fp32 divide). */
int rdf_debug_div_f32(const float *num, const float *den, float *out, size_t n, void *stream);

/* Measurement hook: host-side cost of the calls since the last reset.  out[0] nanoseconds spent inside this library before a
 * forest launch is handed to the HIP runtime, out[1] nanoseconds inside the runtime's launch call, out[2] forest launches;
 * out[3] nanoseconds inside rdf_layered_run[_hand] from entry to return (runtime included), out[4] such calls.  `out` nullable. */
int rdf_debug_host_overhead(unsigned long long out[5], int reset);

/* Tuning knobs for measurements and tests.  Not part of the reference surface, and NOT part of the re-entrancy promise the
 * evaluation calls make: every knob is ONE process-wide atomic int, read once per call, so a change is seen by every
 * thread, stream and device of the process from its next call on (a call that overlaps a change sees the old or the new
 * value, never a torn one).  Labels never depend on a knob -- only launch geometry and which table is walked -- so a program
 * that evaluates on several threads stays correct whatever a knob is set to, but it cannot give two of its threads two
 * settings: leave the knobs alone in such a program (the per-table choice, rdf_forest_set_deep_from / rdf_forest_tune, is the
 * production interface for the one choice that matters).  0 or -1 restores the default as noted.  The RDF_* environment
 * variables that name the same choices are read once per process, at the first call that looks at them; a knob set through
 * these functions wins. */
void rdf_set_lds_budget_bytes(int bytes);
void rdf_set_block_threads(int threads); /* 256 or 512; anything else: the default (512 for launches that fill the chip) */
void rdf_set_compaction(int mode);       /* -1 (default): filtered launches list their pixels first; 0: never */
void rdf_set_scheduler(int mode);        /* 1 dynamic tile queue (default), 0 static round-robin, 2 one tile per
                                           workgroup (non-persistent), -1 env RDF_SCHED = static | tile */
void rdf_set_rows_per_wave(int rows);    /* label rows per wave in a tile: 1, 2 or 4; 0 = choose by launch size */
void rdf_set_halo(int pixels);           /* depth pixels staged in LDS around a tile; -1 = default (56 with 512-thread
                                            workgroups at labels_reduce 1, 32 otherwise, 40 for eight trees and more) */
void rdf_set_tree_waves(int mode);     /* small packed launches of 2-4 trees: one wave per tree and pixel row (k_eval_forest<..., TW>);
                                        * -1 = default (on, or RDF_TREE_WAVES), 0 = off, 1 = on.  Same labels either way. */
void rdf_set_lds_levels(int levels);     /* top levels of every tree pinned in LDS (the depth tile then gets the rest of
                                            the LDS budget instead of half of it); -1 = fill what the tile leaves */
void rdf_set_group(int trees);           /* trees a lane walks interleaved: 1..4, 0 = by forest size (reference-layout forests: 1 or 4) */
void rdf_set_layers_one_launch(int on);  /* rdf_layered_run on a small launch evaluates the layers of a packed 2- or 3-layer
                                            stack unfiltered in ONE launch and filters in the composite kernel: 1/-1 (default) on, 0 off */
void rdf_set_stage_vec(int on);          /* tile staging with 16-byte loads where alignment allows: 1/-1 (default) on, 0 off */
void rdf_set_force_exact(int on);        /* test knob: rdf_forest_pack flags every node for the IEEE-divide path */
void rdf_set_last_level_table(int on);   /* packed forests of up to four classes can walk level D-1 from the table that holds
                                            a node and both its leaf PDFs in one half line.  -1: when every node of that level
                                            is an ordinary one with two leaves AND the forest uses at least half of the level
                                            (walks that mostly end higher up gain nothing); 1: whenever the nodes allow it;
                                            0: never */

void rdf_set_deep_from(int level);       /* packed forests of up to eight classes can walk their deep levels from the deep blocks
                                            (three levels per 128-byte line, one tree after the other, the wave fetching its
                                            lanes' blocks together): -1 = each table's own choice (rdf_forest_set_deep_from /
                                            rdf_forest_tune; none: never), 0 = never, > 0 = from this level on (rounded up to a
                                            block root; never inside the levels held in LDS).  Same labels either way. */
void rdf_set_fold(int on);               /* a label map whose width leaves at most 32 columns beyond a multiple of 64 (848 = 13 x 64 + 16)
                                            gets narrow tiles for them, in which a wave covers 2 rows x 32 or 4 rows x 16 pixels,
                                            instead of a tile column whose waves run mostly empty: 1/-1 (default) on, 0 off */
/* hipEvent timing on the caller's stream (bench.py times the stream the kernels run on). */
int rdf_event_create(void **event);
int rdf_event_record(void *event, void *stream);
int rdf_event_synchronize(void *event);
int rdf_event_elapsed_ms(void *start, void *stop, float *ms);
int rdf_event_destroy(void *event);
int rdf_stream_synchronize(void *stream);

int rdf_abi_version(void);
/* Identity of the build: 16 hex digits of a SHA-256 over the library's sources (the four .hip files, rdf_device.hpp, this
 * header) and its compiler flags, baked in at compile time (3d-beats_amd/_build.py).  The Python binding recomputes it from
 * the sources next to the library and refuses a library built from other sources (an ABI number cannot tell yesterday's
 * kernels from today's).  "unknown" for a build that did not define it. */
const char *rdf_build_id(void);
const char *rdf_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* RDF_HIP_H */
