// rdf_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for per-pixel randomized-decision-forest
// inference.  Written from the semantics of 3d-beats' tree_eval.cu (see include/rdf_hip.h for the
// reference lines each entry point replaces); not a translation of it:
//
//   * one LANE owns one label pixel and walks kGroup trees interleaved, level-synchronously, so
//     that kGroup node fetches and 2*kGroup depth probes are in flight per lane (the reference
//     uses one thread per (pixel,tree), shared-memory float atomics and two block barriers);
//   * a wave is 64 consecutive label pixels of one row; a workgroup owns a 2-D tile (64 columns x
//     1..4 rows per wave, rows of the waves interleaved) whose depth neighbourhood it stages in LDS;
//     the label store is one 128-byte line.  Launches that fill the chip: 512 threads, three
//     workgroups per CU (24 waves), 56-pixel halo; small launches: 256 threads, one row per wave;
//   * per-tree leaf PDFs are added in registers in tree order (canonical order, no atomics);
//   * node records are 16 bytes {4 x 23-bit floor(s*u), floor(s*v), integer threshold, flags}, decoded with one convert per numerator:
//     the top levels of every tree live in LDS, deeper levels are ONE 128-bit load each from the
//     packed table (rdf_forest_pack) or, for the unpacked entry point, the reference's own
//     7+2C-float records.  An earlier version with two loads per node was bound by the L1 tag rate
//     (one cache line per clock per CU, profiles/): what counts is lines touched per wave
//     instruction, not bytes.  What bounds the current version: DESIGN.md section 4;
//   * all probes of a level are issued before any is consumed: nothing waits inside a divergent
//     branch (an earlier version serialised eight global round trips per level that way);
//   * the workgroup's depth tile plus a halo is staged in LDS (out-of-image cells = 65535), so
//     most probes are LDS reads with no bounds check; far probes go to global memory; the tile sits at LDS
//     address 0 and x coordinates are kept doubled, so a probe's LDS address is one multiply-add;
//   * tiles are handed to persistent workgroups by a device-side queue (one atomic per tile), so
//     empty (background) tiles cost almost nothing and frames of unequal cost balance out;
//   * the layers of a small layered run share ONE launch (workgroup b walks layer b % NL, unfiltered;
//     the composite kernel applies the filters): one ramp and drain instead of one per layer.
//
// Bit-exactness: the reference computes floor((s*u)/d) with one fp32 multiply, one IEEE-correct
// fp32 divide and __float2int_rd.  The fast path here (integer numerator with a guard bit, the pixel's
// refined reciprocal, ONE fma in round-toward-minus-infinity mode whose mantissa then holds the floor:
// NodeRec16) is proven equal to that for every depth value and every numerator it is used for by
// exhaustive GPU enumeration (tools/verify_magic.hip, tools/verify_intoffset.hip, tools/verify_fastdiv.hip);
// all other numerators take hipcc's IEEE divide (never build this file with -ffast-math) followed by
// v_floor_f32 + v_cvt_i32_f32.  Coordinate adds wrap, bounds are checked per axis.  The level loop runs with
// the wave's rounding mode switched (s_setreg) and is written so that no other rounding arithmetic sits in it.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <tuple>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/rdf_hip.h"
#include "rdf_device.hpp"

namespace {

constexpr int kGroup = 4;          // trees walked interleaved by one lane (template GROUP: 1 and 2 for smaller forests)
constexpr int kMaxRowsPerWave = 4; // label rows each wave owns in a tile (fewer for small launches)
constexpr int kDefaultLdsBudget = 32700;   // node table + depth tile per 256-thread workgroup (five per CU)
constexpr int kDefaultLdsBudget512 = 54600;   // ... per 512-thread workgroup (three per CU)
constexpr int kDefaultHalo = 32;   // depth pixels staged around a tile's centres
constexpr int kMinLdsLevels = 6;   // top levels of every tree that stay in LDS when the depth tile competes for it
constexpr uint32_t kFlagLeftLeaf = 1u, kFlagRightLeaf = 2u, kFlagExact = 4u;
constexpr int kSchedSlots = 128;        // one per (device, stream) that launches directly
constexpr int kGraphSlots = 384;        // one per launch recorded into a hipGraph (stream capture)

// Dynamic tile queue state.  A workgroup's first tile is its own index; the queues hand out the tiles beyond the grid size
// (a launch with no more tiles than workgroups never touches its slot).  One slot per (device, stream) that launches directly
// (launches on one stream run in order, so they never meet in a slot) and a slot of its own for every launch recorded
// into a hipGraph: a graph replays on whatever stream it is launched on, so its launch must not share the slot of the
// stream it happened to be captured on (an executable graph never runs concurrently with itself).
// Zero at rest: the last workgroup of a launch resets its slot, so launches need no memset and replay correctly.
// A slot holds one queue head per XCD and the count of finished workgroups, each on a 128-byte line of its own (word
// 32 q, q = 0..7; word 256): the tiles of a launch are cut into eight contiguous ranges, a workgroup pulls from the range
// of the XCD it runs on (s_getreg XCC_ID) until it is empty and then helps the next two XCDs -- the workgroups that
// share an L2 work on the same frames (4.74 -> 4.54 ms on the bench batch, 11.45 -> 10.74 ms on config 5's shard).
// Helping is for balance and is kept short (one workgroup in sixteen goes round all eight queues, so that a range
// whose XCD runs no workgroup of the launch -- CU masks -- is drained all the same): a failing pull is an atomic on a contended line (~90 per us per line), and with all heads on one line
// and every workgroup trying all eight at the end a single frame took 188 instead of 97 us.
constexpr int kSchedStride = 32;                 // words between two counters
constexpr int kSchedWords = 9 * kSchedStride;
constexpr uint32_t kStealFrom = 2;               // neighbours a workgroup helps once its own range is empty
constexpr uint32_t kXcdQueuesFrom = 64;          // launches with fewer workgroups keep one queue
__device__ unsigned int g_sched[kSchedSlots + kGraphSlots][kSchedWords];

// ---- node records --------------------------------------------------------------------------
// Hot record, 16 bytes, one 128-bit load per node.  Each word carries a numerator in its high 23 bits, a guard
// bit and one byte of unrelated payload:
//   w0 = floor(s*u.x) << 9 | g0 << 8 | T[7:0]       T = integer threshold (see thresh_to_int), 24-bit two's complement
//   w1 = floor(s*u.y) << 9 | g1 << 8 | T[15:8]
//   w2 = floor(s*v.x) << 9 | g2 << 8 | T[23:16]
//   w3 = floor(s*v.y) << 9 | g3 << 8 | flags        flags: kFlagLeftLeaf/RightLeaf = that side is a leaf,
//                                                          kFlagExact = use the exact record
//   g = NOT bit 7 of the payload byte, so the nine low bits are a value in [128, 383].
// The numerator is decoded with ONE instruction, v_cvt_f32_i32 of the whole word = 512*(x + e), 1/4 <= e <= 3/4
// (the payload and the convert's rounding only move e inside that range): for integers x and d, (x + e)/d lies at
// least 1/(4d) away from the integers on either side of floor(x/d), which is more than the error of ONE multiply by the
// pixel's refined reciprocal -- so the whole divide-floor-add of decision_tree_common.hpp:15-18 is
//     t = fma(n, r/512, 1.5*2^23)   in round-toward-minus-infinity mode:   bits(t) - bits(1.5*2^23) = floor(x/d)
//     coordinate = bits(t) + (pixel coordinate - bits(1.5*2^23))            one v_add_u32
// (the level loop runs with the wave's fp32 rounding mode set to round-down; everything outside it in round-to-nearest).
// Checked exhaustively on gfx950 by tools/verify_magic.hip: every x in [-2^21, 2^21) x every payload byte x every
// depth 1..65535 against integer floor division (logs under profiles/); tools/verify_intoffset.hip and
// tools/verify_fastdiv.hip tie integer floor division to the reference's floor(IEEE (s*u)/d) for every fp32 numerator
// that is +-0 or has a biased exponent in [40,147] (|a| < 2^21).  Any other numerator (huge, denormal, inf, NaN) flags
// the node kFlagExact and the kernel takes the IEEE divide on the fp32 numerators s*u, s*v, which it recomputes -- the same
// one multiply rdf_forest_pack did -- from the node's record in the caller's reference-layout forest (rounds 1-3 kept them in
// a dense 32-byte table of their own, 128 MB for a T4/D20 forest that practically never has such a node).  rdf_forest_pack
// counts the flagged nodes and notes the scale in the table's info block; a packed forest that has any needs the `forest`
// argument at evaluation (RDF_ERR_NULL_PTR otherwise).
struct alignas(16) NodeRec16 {
    uint32_t w[4];
};
// What rdf_forest_pack leaves in the last 128 bytes of a packed table.
struct PackInfo {
    uint32_t exact_nodes;   // nodes flagged kFlagExact
    float scale;            // the scale_factor the table was packed for
    uint32_t magic;         // kPackMagic once k_pack has run
    uint32_t deep_choice;   // which table serves the deep levels, chosen for THIS table (rdf_forest_set_deep_from / rdf_forest_tune):
                            // 0 = no choice made, 1 + level otherwise (1 = heap-order records).  It travels with the table: a process that
                            // maps or copies a tuned table (and every later evaluation in this one) finds the choice without tuning again
    uint32_t generation;    // (round 6) which packing this is: a non-zero number rdf_forest_pack draws per call, unlike any other this
                            // process drew and, with 31 random bits, practically unlike another process's.  The host remembers it
                            // with what else it knows about the table at this address and hands it to every launch; the KERNEL compares
                            // it with the word in the table and raises the device's stale flag when they differ (a different table was
                            // copied over a known address without rdf_forest_forget): the next call returns RDF_ERR_STALE
    uint32_t n_trees, max_depth, n_classes;     // the shape the table was packed for
    uint32_t pad[24];
};
constexpr uint32_t kPackMagic = 0x52444635u;   // "RDF5" (round 6: generation and shape words)
// Last-level record, 64 bytes (forests of up to four classes): the hot record of a node of level D-1 and the PDFs of
// its two leaves in ONE half cache line.  A walk that reaches level D-1 ends there (tree_eval.cu:95-128), so its last
// node fetch and its leaf fetch are the same 128-byte line: fetched together they are one L1 fill instead of two
// (leaf loads were 6.5 % of the batch's time, profiles/r03_*).  The table is used when every record of it is an
// ordinary node with two leaves (flags == both leaves, no kFlagExact; untrained all-zero nodes are such nodes);
// k_pack counts the others behind the table and the kernel takes the general path when the count is not zero.
struct alignas(64) LastLevelRec {
    NodeRec16 node;
    float pdf_left[4], pdf_right[4];
    uint32_t pad[4];
};
static_assert(sizeof(NodeRec16) == 16 && sizeof(PackInfo) == 128 && sizeof(LastLevelRec) == 64, "record sizes");

// ---- deep blocks: the levels whose records no cache holds, three levels to a 128-byte line ---------------------------
// A forest whose deep levels are OCCUPIED (a trained forest; synth's "balanced" topology) visits practically every node of
// them, so below the levels that fit an XCD's 4-MB L2 every node fetch of the heap-order table is a 128-byte line from
// the Infinity Cache or HBM for 16 useful bytes: the bench batch on a balanced T4/D20 forest moves 0.67e9 such lines =
// 86 GB per launch at 7.7 TB/s -- the fabric's whole bandwidth -- and takes 11.2 ms against 3.9 ms on the cache-resident
// "full" topology (profiles/r04_*).  The packed forest therefore holds the hot records a second time, grouped by SUBTREE:
//   * block of three levels, root on level R: records {root, left, right, LL, LR, RL, RR} = 112 bytes of one 128-byte line
//     (the eighth slot is zero);
//   * last block, root on level D - 2 (forests of up to four classes): {root, left, right} and the four leaf PDFs of
//     level D - 1 {L.left, L.right, R.left, R.right}, 16 bytes each: a walk's last two node fetches AND its leaf fetch;
//     forests of five to eight classes: root on level D - 1: {node, pad, left PDF (32 B), right PDF (32 B)}.
// Block roots sit on levels R0 = D - 2 (or D - 1), R0 - 3, R0 - 6, ... >= 0; the blocks of one root level are contiguous,
// [tree][root's index on its level], smaller root levels first; one all-zero line and a 64-byte trailer follow.
// The kernel (k_eval_forest<..., DEEP>) walks the levels below `deep_from` (a root level, chosen per table) from this
// table, ONE TREE AFTER THE OTHER, and the WAVE fetches its lanes' blocks together: eight LDS-DMA loads of eight whole lines
// each into an 8-KB slab per wave, from which every lane reads the three records its walk takes (see "deep blocks" in the
// kernel; round 4's walk had every lane fetch its own block with seven 16-byte loads -- seven L1 look-ups per line -- and
// was bound by the L1's line accesses: 7.9 against 6.7 ms on the bench batch, 23.4 against 18.4 ms on config 5's shard,
// profiles/r05_deep_coop.txt) -- a third of the lines from beyond L2 per walk, and the leaf comes with the last of them.
// Usable from root level R on iff no node of a level >= R is flagged kFlagExact and (for the last block) every node
// of level D - 1 has two leaves: k_pack leaves both facts in the trailer {1 + deepest level with an exact node,
// nodes of level D - 1 that are not plain two-leaf nodes}.
__host__ __device__ inline int deep_last_levels(int cpad) { return cpad == 4 ? 2 : 1; }
// lines of every block level whose root level is below R (R, Lmin members of the series R0 - 3 i)
__host__ __device__ inline size_t deep_lines_before(int n_trees, int R, int Lmin)
{
    return (size_t)n_trees * ((((size_t)1 << R) - ((size_t)1 << Lmin)) / 7u);
}
__host__ __device__ inline size_t deep_total_lines(int n_trees, int max_depth, int cpad)   // without the zero line and the trailer
{
    const int R0 = max_depth - deep_last_levels(cpad);
    return deep_lines_before(n_trees, R0, R0 % 3) + ((size_t)n_trees << R0);
}
// A lane's block is addressed as {wave-uniform 64-bit base of its tree's blocks of the root level} + {32-bit byte offset =
// heap index x 128}: root levels up to 23, i.e. forests of up to 24 levels.
__host__ __device__ inline bool deep_possible(int n_trees, int max_depth, int cpad)
{
    return n_trees >= 1 && (cpad == 4 || cpad == 8) && max_depth >= 5 && max_depth <= 24;
}

struct EvalArgs {
    const uint16_t *depth;
    const float *forest;
    const NodeRec16 *packed16;   // nullptr on the unpacked path
    const uint16_t *filter;
    uint16_t *labels;
    unsigned long long *stats;
    int stats_wide;        // stats holds 8 counters (rdf_eval_forest_packed_stats), not 3
    unsigned int *sched;   // queue slot, or nullptr for static round-robin tiles
    uint32_t n_tiles;      // n_img * (tiles_x * tiles_y + tiles_y_n)
    uint32_t tiles_x;      // ceil(Wl / 64); with a folded last column: floor(Wl / 64)
    // (round 6) The label map's last columns when Wl is not a multiple of 64 and at most 32 are left over (848 = 13 x 64 + 16):
    // instead of a tile column whose waves run with 16 of 64 lanes, NARROW tiles of 64 / fold columns x fold x tile_rows rows in
    // which a wave covers `fold` rows of 64 / fold pixels -- every lane has a pixel.  tiles_y_n of them per image, behind the
    // image's ordinary tiles; their staged depth tile has its own shape.
    uint32_t fold;         // 0: no narrow tiles; 2 or 4
    uint32_t tiles_y_n;
    int tw_n, th_n, twp_n;
    int halo_xn;           // their halo in x (<= halo): chosen so that the `fold` rows of a wave do not share LDS banks
    uint32_t stage_tw8_n, stage_magic_n;
    uint32_t tiles_y;      // ceil(Hl / (waves per block * rows_per_wave))
    int rows_per_wave;     // 1..kMaxRowsPerWave
    uint32_t per_img_l;    // Wl*Hl
    uint32_t per_img_d;    // W*H
    int W, H, Wl, Hl, r;
    int T, D, C, E;
    int nodes;             // 2^D - 1
    int lds_levels;        // top levels held in LDS
    int halo;              // depth pixels around the tile's centres held in LDS
    int tw, th, twp;       // staged depth tile: width, height, row pitch (0: no staged tile)
    uint32_t stage_tw8;    // > 0: stage with 16-byte loads, tw / 8 vectors per row (W, tx0 and twp are multiples of 8)
    uint32_t stage_magic;  // floor(2^32 / stage_tw8) + 1: i / stage_tw8 == umulhi(i, magic) for i < 2^32 / stage_tw8
    uint32_t lds_xchg_off; // (tree waves) where the trees of a pixel row meet
    uint32_t lds_nodes_off; // byte offsets inside the dynamic LDS allocation (the depth tile is at 0)
    uint32_t lds_mail_off;
    uint32_t lds_list_off;
    uint32_t lds_slab_off; // (DEEP) the waves' block slabs: 8 KB each, 1-KB aligned
    const float *packed_pdf;   // leaf PDFs [T][2^D][2][cpad], 16-byte aligned rows (packed path), or null
    const uint4 *last_level;   // LastLevelRec [T][2^(D-1)] followed by the trailer (see k_pack), or null
    uint32_t last_level_min;   // deepest-level nodes in use from which the table is taken
    const uint4 *deep;         // deep blocks (see above), or null
    int deep_from;             // root level from which the walk takes the deep blocks (a member of the series; DEEP kernels)
    int cpad;                  // classes rounded up to a multiple of 4
    int filter_class;
    int check_empty;       // look at a tile's centre depths before staging it (throughput shape)
    int keep_if_no_leaf;   // single-tree semantics: no leaf reached -> pixel untouched
    int fill_untouched;    // fused pre-fill: write 65535 to every label pixel that is not evaluated
    float s;
    // (round 6) the packed table's info block, the generation the host believes the table at this address has, and where a
    // kernel that finds another one says so (pinned host memory, one word per device): see PackInfo.generation
    const PackInfo *info;
    uint32_t expect_gen;
    unsigned int *stale_flag;
    // (round 6) two launches that share ONE tile queue (rdf_eval_forest_packed_split: a main launch on a CU-masked stream and
    // a helper launch that starts when the CUs left to another kernel are free again): q_static = tiles handed out
    // statically (the main launch's grid; 0: this launch's own grid), q_blocks = workgroups of both launches together (the
    // last of them puts the slot back to zero; 0: this launch's own), q_helper: this launch's workgroups have no static tile
    uint32_t q_static, q_blocks;
    int q_helper;
};

// A side continues iff floor(x) == -1  <=>  -1 <= x < 0   (NaN: false); anything else makes it a leaf.
// tree_eval.cu:101-102 / :186-187.
__device__ __forceinline__ uint32_t child_flags(float l, float r)
{
    return ((l >= -1.0f && l < 0.0f) ? 0u : kFlagLeftLeaf) | ((r >= -1.0f && r < 0.0f) ? 0u : kFlagRightLeaf);
}

// Numerators the integer record represents exactly (see NodeRec16): +-0, or biased exponent 40..147 (|a| < 2^21).
__device__ __forceinline__ bool int_offset_ok(float a)
{
    const uint32_t b = __float_as_uint(a);
    return (b << 1) == 0u || (((b >> 23) & 0xFFu) - 40u) <= 107u;
}

// payload byte -> the nine low bits of a record word: bit 8 = NOT bit 7 (a value in [128, 383], see NodeRec16)
__device__ __forceinline__ uint32_t guarded_payload(uint32_t byte)
{
    return (byte & 0xFFu) | ((~byte & 0x80u) << 1);
}

// f = depth[u] - depth[v] is an integer in [-65535, 65535], so `f < thresh` (tree_eval.cu:107)
// equals `f < T` with T = ceil(thresh) clamped to [-65535, 65536]; NaN never compares true.
__device__ __forceinline__ int thresh_to_int(float t)
{
    if (t != t) return -65535;
    const float c = fminf(fmaxf(__builtin_ceilf(t), -65535.0f), 65536.0f);
    return (int)c;
}

__device__ __forceinline__ NodeRec16 encode_node(float sux, float suy, float svx, float svy, float thresh,
                                                 float l_next, float r_next)
{
    uint32_t flags = child_flags(l_next, r_next);
    const bool ok = int_offset_ok(sux) && int_offset_ok(suy) && int_offset_ok(svx) && int_offset_ok(svy);
    int nx = 0, ny = 0, mx = 0, my = 0;
    if (ok) {
        nx = (int)__builtin_floorf(sux); ny = (int)__builtin_floorf(suy);
        mx = (int)__builtin_floorf(svx); my = (int)__builtin_floorf(svy);
    } else {
        flags |= kFlagExact;
    }
    const uint32_t t = (uint32_t)thresh_to_int(thresh);   // two's complement, |T| <= 65536
    NodeRec16 r;
    r.w[0] = ((uint32_t)nx << 9) | guarded_payload(t);
    r.w[1] = ((uint32_t)ny << 9) | guarded_payload(t >> 8);
    r.w[2] = ((uint32_t)mx << 9) | guarded_payload(t >> 16);
    r.w[3] = ((uint32_t)my << 9) | guarded_payload(flags);
    return r;
}

__device__ __forceinline__ uint4 select4(bool c, const uint4 a, const uint4 b)
{
    return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

// The reference's records are 7+2C floats long, so their fields are only 4-byte aligned; gfx950 global loads do
// not need more, and two wide loads (4 + 3 floats) replace seven scalar ones on the unpacked path.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));


struct Node {
    float ax, ay, bx, by;   // numerators of the u and v offsets: kNumScale * (integer + e) from a hot record,
                            // the fp32 values s*u, s*v themselves on a kFlagExact node
    uint32_t lo16;          // T[15:0], T = integer threshold
    uint32_t w2;            // byte 0 = T[23:16] (two's complement: -1, 0 or 1); the bits above are not defined
    uint32_t flags;         // low 3 bits; the bits above are not defined
};

__device__ __forceinline__ Node decode_node(const uint4 w)
{
    Node n;
    n.ax = (float)(int)w.x;
    n.ay = (float)(int)w.y;
    n.bx = (float)(int)w.z;
    n.by = (float)(int)w.w;
    // T[15:0] = {w1.b0, w0.b0} in one byte permute (v_perm_b32: selectors 0-3 pick bytes of the second operand, 4-7 of
    // the first, 12 is 0x00); T[23:16] stays where it is and is read in place by the decision's compare (walk_step)
    n.lo16 = __builtin_amdgcn_perm(w.y, w.x, 0x0c0c0400u);
    n.w2 = w.z;
    n.flags = w.w;
    return n;
}

// The same three fields from an integer threshold (nodes read from the reference's records).
__device__ __forceinline__ void set_threshold(Node &n, int t)
{
    n.lo16 = (uint32_t)t & 0xFFFFu;
    n.w2 = (uint32_t)t >> 16;
}

// One step of the walk (tree_eval.cu:107-121).  g = depth[u] - depth[v] - T[15:0]; the feature is below the threshold
// iff g < T[23:16] * 65536 iff (g >> 16) < T[23:16] -- one SDWA compare that sign-extends the high word of g and byte 0
// of w2 in place.  side = 0 (left) when below, else 1; next = 2 h + side by an add with carry-in; the leaf flag of that
// side (flags bit `side`) is shifted to bit 31 and or-ed in.  Result: the next 1-based heap index while walking,
// 0x80000000 | (2 h + side) once a leaf is reached.
__device__ __forceinline__ uint32_t walk_step(uint32_t h, int g, uint32_t w2, uint32_t flags)
{
    uint32_t next, t;
    asm("v_cmp_ge_i32_sdwa vcc, sext(%2), sext(%3) src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
        "v_cndmask_b32_e64 %1, 31, 30, vcc\n\t"
        "v_lshlrev_b32_e32 %1, %1, %4\n\t"
        "v_addc_co_u32_e32 %0, vcc, %5, %5, vcc\n\t"
        "v_and_or_b32 %0, %1, %6, %0"
        : "=&v"(next), "=&v"(t)
        : "v"(g), "v"(w2), "v"(flags), "v"(h), "s"(0x80000000u)
        : "vcc");
    return next;
}

// Every wave owns a.rows_per_wave label rows of the tile (up to kMaxRowsPerWave in the throughput shape, one for small
// launches such as one live frame).
// NL > 1 ("layers in one launch", small launches of a layered stack): workgroup b evaluates forest b % NL of the kernel's
// NL argument sets on the same frame -- NL independent forest evaluations that share ONE launch, hence one ramp and one
// drain (a small launch costs one wave-row's dependent chain however little it evaluates), with a tile queue per role.
template <int NL>
struct EvalArgsN {
    EvalArgs l[NL];
};

// ---- a reader's map of k_eval_forest (one kernel, ~900 lines; the template parameters select code, never semantics) ----
//   template parameters   BLOCK threads per workgroup | PACKED tables or the reference's records | CMAX classes summed in
//                         registers at a time | STATS visit / line counters | GROUP trees a lane walks interleaved | COMPACT
//                         pixel list (filtered launches) | NL layers of a stack in one launch | TW one wave per tree (small
//                         launches) | DEEP walk the deep levels from the deep blocks
//   per workgroup, once   "stage the top K levels": the node table's top levels into LDS; which tables serve the forest
//                         (last-level table, deep blocks: their trailers)
//   per tile              "take the next tile" (static first tile, then the per-XCD queues; a helper launch's workgroups have no
//                         static tile: rdf_eval_forest_packed_split) -> which pixels the lanes of a wave stand on (one row of 64, or
//                         -- narrow tiles for a label map's last columns -- `fold` rows of 64 / fold) -> "empty tile?" -> "stage
//                         depth" into LDS at address 0 -> "compact" (COMPACT only)
//   per pixel group       early-outs (tree_eval.cu:81-89; DEEP: a lane without a pixel stays in the loop with idle tree slots, `live`,
//                         because the wave fetches its deep blocks together), the pixel's reciprocal, then per CMAX classes and per GROUP trees:
//       the level loop    round-down mode; node fetch (LDS / packed table / reference records) -> probe coordinates (one fma per
//                         coordinate, or the IEEE divide for flagged nodes) -> probes issued -> decisions (walk_step)
//       then ONE of       "deep blocks" (DEEP: tree after tree, the WAVE fetches its lanes' 128-byte blocks with eight LDS-DMA loads into
//                         its 8-KB slab, every lane reads the three records it takes; leaf PDFs with the last block)
//                         "level D-1 from the last-level table" (node and both PDFs in one 64-byte record, tree after tree)
//                         the general leaf fetch (PDF rows, in tree order)
//       argmax (tree_eval.cu:7-21) and the label store
//   TW only               "the T waves of a pixel row hand their trees' results to the row's first wave"
//   epilogue              (last workgroup) the table's generation against the host's (PackInfo.generation); queue slot back to zero; STATS reduction
// Bit-exactness rests on three things the sections keep apart: the order of the PDF adds (tree order, round-to-nearest), the
// mode the fmas run in (round-down, between set_round_down / set_round_nearest: tools/check_rounding_isa.py), and the per-axis
// bounds of the probes (TileCtx).
template <int BLOCK, bool PACKED, int CMAX, bool STATS, int GROUP, bool COMPACT, int NL = 1, bool TW = false, bool DEEP = false>
// second launch bound = waves per SIMD the register allocation must allow: three 512-thread workgroups per CU are six
// waves per SIMD (80 VGPRs; the 4-wide walk needs 86 without the bound and spills two dwords with it)
// (DEEP: the block slabs leave room for two 512-thread or four 256-thread workgroups per CU = four waves per SIMD)
__global__ __launch_bounds__(BLOCK, TW ? 8 : DEEP ? 4 : BLOCK == 512 ? 6 : BLOCK == 256 ? 5 : 4) void k_eval_forest(const EvalArgsN<NL> ka)
{
    const uint32_t role = NL > 1 ? blockIdx.x % NL : 0u;           // workgroup-uniform
    const uint32_t block_id = NL > 1 ? blockIdx.x / NL : blockIdx.x, n_blocks = NL > 1 ? gridDim.x / NL : gridDim.x;
    const EvalArgs &a = ka.l[role];
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // the depth tile sits at LDS address 0, so that a probe's byte offset inside the tile IS its LDS address
    uint16_t *lds_tile = reinterpret_cast<uint16_t *>(lds_raw);
    uint4 *lds_nodes = reinterpret_cast<uint4 *>(lds_raw + a.lds_nodes_off);
    uint32_t *s_tile = reinterpret_cast<uint32_t *>(lds_raw + a.lds_mail_off);
    uint16_t *px_list = reinterpret_cast<uint16_t *>(lds_raw + a.lds_list_off);   // the tile's pixels to evaluate, compacted

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int K = a.lds_levels;
    const uint32_t nodes_lds = (1u << K) - 1u;
    // (round 6) the table's own words, not the host's memory of them, read where they are needed and nowhere near the
    // prologue (a load in front of the first tile is a round trip beyond the L2 that every workgroup of a one-frame launch
    // would wait for): the scale inside the rare kFlagExact branch, the generation in the epilogue of the launch's last workgroup (asked by
    // workgroup 0 it cost the smallest launches 0.4 us, measured with the check compiled out).
    constexpr uint32_t kWaves = BLOCK / 64;
    const int rows_per_wave = a.rows_per_wave;
    const uint32_t wave = (uint32_t)tid >> 6;
    // TW ("tree waves", small launches): the T trees of a forest are walked by T different waves -- wave w takes tree
    // w % T for pixel row w / T of the tile -- and meet in LDS.  A wave alone on its SIMD issues an instruction every
    // ~8 cycles, and a launch of one live frame is a few hundred such waves: its duration is the instruction count of ONE
    // wave's walk (1.3 us per level with four trees interleaved in a lane, 0.3 us with one; measured), not throughput.
    const uint32_t tw_T = TW ? (uint32_t)a.T : 1u;
    const uint32_t tw_row = TW ? wave / tw_T : 0u, tw_tree = TW ? wave - tw_row * tw_T : 0u;
    const uint32_t tile_rows = TW ? (kWaves / tw_T) * (uint32_t)rows_per_wave : kWaves * (uint32_t)rows_per_wave;

    // ---- stage the top K levels of every tree.  Tables use 1-based heap numbering (slot 0 unused):
    // node h has children 2h and 2h+1, which therefore share one aligned 32-byte pair ----
    const uint32_t lds_pitch = 1u << K;   // records per tree in LDS
    // (LastLevelRec) level D-1 is walked from the last-level table when the forest has a usable one
    constexpr bool kLastLevelTable = PACKED && CMAX == 4 && !TW;
    bool last_from_table = false;
    if (kLastLevelTable && a.last_level) {
        // the trailer's words: every record usable, and the forest uses at least a.last_level_min of its deepest nodes (a
        // forest whose walks mostly end higher up is better off with the general path: its leaf rows come in one round
        // trip for the group, the table walks level D-1 tree after tree)
        const uint2 t = *reinterpret_cast<const uint2 *>(a.last_level + (((size_t)a.T << (a.D - 1)) << 2));
        last_from_table = t.x == 0u && t.y >= a.last_level_min;
    }
    // (deep blocks) the levels from a.deep_from on are walked block by block, one tree after the other, when the forest's
    // deep table can serve them: the trailer's words, see "deep blocks" above
    constexpr int kDeepLast = CMAX == 4 ? 2 : 1;           // levels in the last block
    bool deep_on = false;
    if (DEEP && a.deep) {
        const uint2 t = *reinterpret_cast<const uint2 *>(a.deep + ((deep_total_lines(a.T, a.D, a.cpad) + 1u) << 3));
        deep_on = (uint32_t)a.deep_from >= t.x && t.y == 0u;
    }
    if (DEEP && deep_on) last_from_table = false;
    const int walk_levels = (DEEP && deep_on) ? a.deep_from : last_from_table ? a.D - 1 : a.D;
    for (uint32_t i = tid; i < (uint32_t)a.T * lds_pitch; i += BLOCK) {
        const uint32_t k = i >> K, h = i & (lds_pitch - 1u);
        if (h == 0u) {      // slot 0: the all-zero node that finished tree slots fetch (see the level loop)
            if (K > 0) lds_nodes[i] = make_uint4(0u, 0u, 0u, 0u);
            continue;
        }
        if (PACKED) {
            lds_nodes[i] = reinterpret_cast<const uint4 *>(a.packed16)[((size_t)k << a.D) + h];
        } else {
            const float *p = a.forest + ((size_t)k * (size_t)a.nodes + (h - 1u)) * (size_t)a.E;
            const NodeRec16 r = encode_node(a.s * p[0], a.s * p[1], a.s * p[2], a.s * p[3], p[4], p[5], p[6]);
            lds_nodes[i] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
        }
    }
    if (tid == 0) s_tile[2] = s_tile[3] = 0u;   // (made visible by the first tile's barrier)
    // (STATS) per-lane visit counters: pixels, node records read, leaves reached; and what a launch with THIS geometry needs
    // from the vector memory pipeline at the least: node records read from LDS, and 128-byte lines touched by the loads that
    // serve walking slots -- records and leaf rows from global memory, far probes that load, deep blocks -- counted per wave
    // instruction with neighbouring lanes on one line merged, as the L1 merges them (tools/roofline.py: useful accesses)
    uint32_t st_px = 0, st_lv = 0, st_lf = 0, st_lds = 0, st_rec = 0, st_leaf = 0, st_far = 0, st_blk = 0;
    // lanes of this wave instruction that touch a line their left neighbour does not touch as well
    // a probe of a walking slot that leaves the staged tile and lies inside the image loads from global memory (TileCtx)
    auto far_lines = [&](const TileCtx &c, uint32_t cx2, uint32_t cy, bool walking, uint32_t img_byte_off) -> uint32_t {
        const bool in_tile = cx2 < c.tw2 && cy < c.th;
        const uint32_t x2 = cx2 + c.tx0_2, y = cy + c.ty0;
        const bool loads = walking && !in_tile && x2 < c.W2 && y < c.H;
        const uint32_t line = (img_byte_off + __umul24(y, c.W2) + x2) >> 7;
        const uint32_t prev = (uint32_t)__shfl_up((int)line, 1);
        const unsigned long long m = __ballot(loads);        // (a lane the pixel loop has switched off counts as idle)
        const bool prev_loads = lane > 0 && ((m >> (lane - 1)) & 1ull) != 0ull;
        return (loads && (!prev_loads || prev != line)) ? 1u : 0u;
    };
    auto lines_of = [&](bool active, uint32_t byte_off) -> uint32_t {
        const uint32_t line = byte_off >> 7;
        const uint32_t prev = (uint32_t)__shfl_up((int)line, 1);
        const unsigned long long m = __ballot(active);
        const bool prev_active = lane > 0 && ((m >> (lane - 1)) & 1ull) != 0ull;
        return (active && (!prev_active || prev != line)) ? 1u : 0u;
    };
    const char *depth_b = reinterpret_cast<const char *>(a.depth);
    uint32_t static_tile = block_id;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 7u;   // HW_REG_XCC_ID: the XCD this workgroup runs on
    uint32_t q_empty = 0u;                   // (thread 0) queues found empty so far
    const uint32_t q_static = a.q_static ? a.q_static : n_blocks;      // tiles handed out statically (split launches: the main launch's grid)
    const uint32_t q_blocks = a.q_blocks ? a.q_blocks : n_blocks;      // workgroups that finish on this slot
    const uint32_t queued = a.n_tiles > q_static ? a.n_tiles - q_static : 0u;   // tiles the queues hand out (the first q_static are static)

    for (uint32_t it = 0;; ++it) {
        // ---- take the next tile: 64 label columns x tile_rows label rows of one image ----
        uint32_t tile;
        if (a.sched) {
            // A workgroup's FIRST tile is its own index: no atomic round trip in front of the first staging (2-3 us
            // of every launch), and a launch with no more tiles than workgroups -- one live frame -- touches no queue at
            // all: no pull, no failing pulls at the end, no finished-workgroup count.  The queues hand out the rest.
            if (it == 0u && !a.q_helper) {
                __syncthreads();
                tile = block_id;     // (a permutation that hands each XCD a contiguous run of first tiles was tried: a
                                     // four-frame batch 0.24 -> 0.29 ms, larger batches unchanged)
            } else {
                if (queued == 0u) break;             // (workgroup-uniform)
                if (tid == 0) {
                    uint32_t t = a.n_tiles;      // nothing left anywhere
                    if (q_static < kXcdQueuesFrom) {            // a small launch: one queue (head 0) over the rest
                        const uint32_t got = atomicAdd(a.sched, 1u);
                        if (got < queued) t = q_static + got;
                    } else {
                        // nothing guarantees that every XCD runs a workgroup of this launch (CU masks), so some workgroups
                        // -- eight consecutive ones in every 128, workgroups 0-7 always among them -- go round all queues
                        // (and every workgroup does when the queues hold several rounds of tiles: a few frames of uneven
                        // cost are uneven ranges, and the failing pulls at the end are a small share of such a launch)
                        const uint32_t reach = (((block_id >> 3) & 15u) == 0u || queued >= 2u * q_static || a.q_helper) ? 7u : kStealFrom;
                        for (uint32_t k = 0; k <= reach; ++k) {
                            const uint32_t q = (xcc + k) & 7u;
                            if ((q_empty >> q) & 1u) continue;
                            const uint32_t lo = (uint32_t)(((unsigned long long)queued * q) >> 3);
                            const uint32_t hi = (uint32_t)(((unsigned long long)queued * (q + 1u)) >> 3);
                            const uint32_t got = lo < hi ? atomicAdd(a.sched + q * kSchedStride, 1u) : hi;
                            if (lo < hi && got < hi - lo) { t = q_static + lo + got; break; }
                            q_empty |= 1u << q;
                        }
                    }
                    s_tile[it & 1u] = t;
                }
                __syncthreads();   // also: every wave is done with the previous tile's LDS image
                tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tile[it & 1u]);   // scalar: the tile's geometry and image base stay in SGPRs
            }
        } else {
            __syncthreads();
            tile = static_tile;
            static_tile += n_blocks;
        }
        if (tile >= a.n_tiles) break;
        const uint32_t tiles_wide = a.tiles_x * a.tiles_y;          // an image's ordinary tiles, then its narrow ones
        const uint32_t tiles_per_img = tiles_wide + a.tiles_y_n;
        const uint32_t img = tile / tiles_per_img;
        const uint32_t trem = tile - img * tiles_per_img;
        const bool narrow = trem >= tiles_wide;                     // (scalar)
        const uint32_t ty = narrow ? trem - tiles_wide : trem / a.tiles_x;
        const uint32_t tx = narrow ? a.tiles_x : trem - ty * a.tiles_x;
        // where a wave's 64 lanes sit: one row of 64 pixels, or (narrow tiles) `fold` rows of 64 / fold
        const uint32_t fold = narrow ? a.fold : 1u;
        const uint32_t col_shift = narrow ? (a.fold == 4u ? 4u : 5u) : 6u;
        const uint32_t lane_col = (uint32_t)lane & ((1u << col_shift) - 1u), lane_row = (uint32_t)lane >> col_shift;
        const uint32_t x0 = tx * 64u, y0 = ty * tile_rows * fold;   // the tile's first label pixel
        const int lx = (int)(x0 + lane_col);
        const uint32_t img_boff = (img * a.per_img_d) << 1;
        const uint32_t img_loff = img * a.per_img_l;
        // depth coordinates of the staged tile's first cell
        const int tx0 = (int)x0 * a.r - (narrow ? a.halo_xn : a.halo);
        const int ty0 = (int)y0 * a.r - a.halo;
        const int tw = narrow ? a.tw_n : a.tw, th = narrow ? a.th_n : a.th, twp = narrow ? a.twp_n : a.twp;
        const uint32_t stage_tw8 = narrow ? a.stage_tw8_n : a.stage_tw8, stage_magic = narrow ? a.stage_magic_n : a.stage_magic;

        const TileCtx pc = {depth_b + img_boff,
                            (uint32_t)tw * 2u, (uint32_t)th, (uint32_t)twp * 2u, (uint32_t)a.W * 2u, (uint32_t)a.H,
                            (uint32_t)tx0 * 2u, (uint32_t)ty0};

        // ---- empty tile?  (live frames are mostly background.)  Every wave looks at the centre depths of its
        // own rows straight from global memory; a tile without a single pixel to evaluate is not staged.
        // (a.check_empty: launches that fill the chip, and small ones at labels_reduce > 1 -- eval_common) ----
        if (a.check_empty && tw > 0) {
            bool mine = false;
            for (int sub = 0; sub < rows_per_wave; ++sub) {
                const uint32_t trow = (uint32_t)sub * kWaves + wave;
                if (trow >= tile_rows) break;
                const int ly = (int)(y0 + trow * fold + lane_row);
                if (ly < a.Hl && lx < a.Wl) {
                    const uint32_t i = img_loff + (uint32_t)ly * (uint32_t)a.Wl + (uint32_t)lx;
                    const uint32_t d = *reinterpret_cast<const uint16_t *>(
                        depth_b + (img_boff + ((__umul24((uint32_t)(ly * a.r), (uint32_t)a.W) + (uint32_t)(lx * a.r)) << 1)));
                    bool ok = d != 0u && d != kNoPixel;
                    if (ok && a.filter_class != -1) ok = (int)a.filter[i] == a.filter_class;
                    mine |= ok;
                    if (!ok && a.fill_untouched) a.labels[i] = (uint16_t)kNoPixel;
                }
            }
            // workgroup-wide OR through two alternating mailbox words (__syncthreads_or would bring the runtime's static
            // LDS scratch, which moves the depth tile away from LDS address 0): the word of tile `it` is cleared again
            // by thread 0 during tile it + 1, after that tile's barrier -- every wave has read it by then, and nobody
            // writes it before the barrier at the top of tile it + 2
            if (__any(mine) && lane == 0) s_tile[2u + (it & 1u)] = 1u;
            __syncthreads();
            const uint32_t any_mine = s_tile[2u + (it & 1u)];
            if (tid == 0) s_tile[2u + ((it + 1u) & 1u)] = 0u;
            if (any_mine == 0u) continue;   // block-uniform
        }

        // ---- stage depth [ty0, ty0+th) x [tx0, tx0+tw) into LDS; outside the image = 65535 ----
        if (stage_tw8 > 0u) {
            // eight pixels per lane (one 128-bit load, one ds_write_b128): W, tx0 and the row pitch are multiples of 8, so
            // a vector lies inside or outside the image as a whole and every address is 16-byte aligned
            const uint32_t total = (uint32_t)th * stage_tw8;
            for (uint32_t i = (uint32_t)tid; i < total; i += BLOCK) {
                const uint32_t row = __umulhi(i, stage_magic);
                const uint32_t c = (i - row * stage_tw8) << 3;
                const int gy = ty0 + (int)row, gx = tx0 + (int)c;
                uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                if ((uint32_t)gy < (uint32_t)a.H && (uint32_t)gx < (uint32_t)a.W)
                    v = *reinterpret_cast<const uint4 *>(
                        depth_b + (img_boff + ((__umul24((uint32_t)gy, (uint32_t)a.W) + (uint32_t)gx) << 1)));
                *reinterpret_cast<uint4 *>(lds_tile + row * (uint32_t)twp + c) = v;
            }
            __syncthreads();
        } else if (tw > 0) {
            for (int row = (int)wave; row < th; row += (int)kWaves) {
                const int gy = ty0 + row;
                const bool row_in = (uint32_t)gy < (uint32_t)a.H;
                for (int col = lane; col < tw; col += 64) {
                    const int gx = tx0 + col;
                    uint32_t v = kNoPixel;
                    if (row_in && (uint32_t)gx < (uint32_t)a.W)
                        v = *reinterpret_cast<const uint16_t *>(
                            depth_b + (img_boff + ((__umul24((uint32_t)gy, (uint32_t)a.W) + (uint32_t)gx) << 1)));
                    lds_tile[row * twp + col] = (uint16_t)v;
                }
            }
            __syncthreads();
        }

        // ---- compact the tile's pixels that have something to evaluate.  Live frames are mostly background and a
        // filtered layer only looks at one class: with one fixed pixel per lane, waves along the edge of the hand
        // ran mostly empty (a live-only batch reached 72 % of the dense rate per valid pixel).  The pixels that
        // pass the early-outs of tree_eval.cu:81-89 are listed in row-major order and dealt to the waves 64 at a
        // time, so a wave is idle only in the tile's last partial group.  Pixels that do not pass are left
        // untouched -- unless the caller asked for the fused pre-fill, which stores what its fill(65535) would. ----
        // COMPACT is instantiated for filtered launches only: without a filter the memory pipeline, which idle lanes
        // do not load, is the limit (a live-only batch did not get faster) and the two extra barriers cost a single
        // dense frame 8 % (measured); then lane = column, as the list would have it for a full tile, and the early-outs
        // are taken in the loop below.
        constexpr bool compact = COMPACT;
        uint32_t *s_cnt = reinterpret_cast<uint32_t *>(px_list + BLOCK * rows_per_wave);   // valid pixels per tile row
        uint32_t my_rank[kMaxRowsPerWave];
        uint32_t my_valid = 0u;
#pragma unroll
        for (int sub = 0; sub < kMaxRowsPerWave; ++sub) {
            if (sub >= rows_per_wave || !compact) break;
            // rows of the workgroup's waves are interleaved: at any time they cover adjacent rows
            const uint32_t trow = (uint32_t)sub * kWaves + wave;
            const int ly = (int)(y0 + trow * fold + lane_row);
            bool ok = trow < tile_rows && ly < a.Hl && lx < a.Wl;
            if (ok) {
                const uint32_t i = img_loff + (uint32_t)ly * (uint32_t)a.Wl + (uint32_t)lx;
                if (a.filter_class != -1) ok = (int)a.filter[i] == a.filter_class;
                if (ok) {
                    const uint32_t d = (uint32_t)tprobe_value(tprobe_issue(pc, (uint32_t)(lx * a.r - tx0) * 2u, (uint32_t)(ly * a.r - ty0)));
                    ok = d != 0u && d != kNoPixel;
                }
                if (!ok && a.fill_untouched) a.labels[i] = (uint16_t)kNoPixel;
            }
            const unsigned long long b = __ballot(ok);
            if (lane == 0 && trow < tile_rows) s_cnt[trow] = (uint32_t)__popcll(b);
            my_rank[sub] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            my_valid |= ok ? 1u << sub : 0u;
        }
        uint32_t n_valid = 64u * tile_rows;
        if (compact) {
            __syncthreads();
            n_valid = 0u;
            uint32_t my_base[kMaxRowsPerWave];
            for (uint32_t row = 0; row < tile_rows; ++row) {   // (LDS broadcast reads; tile_rows <= 64)
#pragma unroll
                for (int sub = 0; sub < kMaxRowsPerWave; ++sub)
                    if (row == (uint32_t)sub * kWaves + wave) my_base[sub] = n_valid;
                n_valid += s_cnt[row];
            }
#pragma unroll
            for (int sub = 0; sub < kMaxRowsPerWave; ++sub) {
                if (sub >= rows_per_wave) break;
                if ((my_valid >> sub) & 1u)
                    px_list[my_base[sub] + my_rank[sub]] = (uint16_t)((((uint32_t)sub * kWaves + wave) << 6) | (uint32_t)lane);
            }
            __syncthreads();
        }

        // pixel slots of the tile, row-major (row << 6 | column), 64 per wave step; uncompacted, the rows of the waves
        // are interleaved
        constexpr uint32_t kDone = 0x80000000u, kIdle = 0xFFFFFFFFu;   // walk state of a tree slot: see the level loop
        // adds the PDF of the leaf a finished walk (state word hk, see the level loop) reached in tree `tree` to pdf[],
        // classes c0 .. c0 + CMAX - 1; false when the walk reached no leaf (tree_eval.cu:95-128: level D-1 says "continue")
        auto add_leaf_pdf = [&](uint32_t hk, int tree, int c0, float (&pdf)[CMAX]) -> bool {
            if (STATS && c0 == 0)       // (the reference layout's leaf sits inside its node's record: another line more often than not)
                st_leaf += lines_of((int)hk < 0 && hk != kIdle, PACKED ? (hk & ~kDone) * (uint32_t)(a.cpad * 4) : ((hk & ~kDone) >> 1) * (uint32_t)a.E * 4u);
            if (!((int)hk < 0 && hk != kIdle)) return false;
            const uint32_t leaf = (hk & ~kDone) - 2u;   // (node - 1) * 2 + side
            if (PACKED) {
                // one aligned 16-byte load per four classes from the packed PDF table (zero-padded to
                // cpad), instead of C scalar loads at odd offsets inside the 7+2C-float record: the
                // scattered leaf reads were a sixth of all L1 accesses
                const float4 *pp = reinterpret_cast<const float4 *>(a.packed_pdf) +
                    (((((size_t)tree) << a.D) + (leaf >> 1) + 1u) * 2u + (leaf & 1u)) * (size_t)(a.cpad >> 2) +
                    (size_t)(c0 >> 2);
#pragma unroll
                for (int c = 0; c < CMAX; c += 4) {
                    if (c0 + c < a.cpad) {
                        const float4 v = pp[c >> 2];
                        pdf[c] = pdf[c] + v.x; pdf[c + 1] = pdf[c + 1] + v.y;
                        pdf[c + 2] = pdf[c + 2] + v.z; pdf[c + 3] = pdf[c + 3] + v.w;
                    }
                }
            } else {
                const float *pp = a.forest +
                    ((size_t)tree * (size_t)a.nodes + (leaf >> 1)) * (size_t)a.E +
                    7 + (leaf & 1u) * a.C + c0;
#pragma unroll
                for (int c = 0; c < CMAX; c += 4) {
                    if (c0 + c + 3 < a.C) {           // four classes with one (4-byte aligned) wide load
                        const f4u v = *reinterpret_cast<const f4u *>(pp + c);
                        pdf[c] = pdf[c] + v.x; pdf[c + 1] = pdf[c + 1] + v.y;
                        pdf[c + 2] = pdf[c + 2] + v.z; pdf[c + 3] = pdf[c + 3] + v.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + c + e < a.C) pdf[c + e] = pdf[c + e] + pp[c + e];
                    }
                }
            }
            return true;
        };
        uint32_t tw_h = kIdle, tw_i = 0u;       // (TW) this wave's tree: where its walk ended for this lane's pixel
        bool tw_valid = false;
        for (uint32_t first = TW ? (tw_row < tile_rows ? tw_row * 64u : n_valid) : wave * 64u; first < n_valid; first += BLOCK) {
            const uint32_t slot = first + (uint32_t)lane;
            uint32_t entry = slot;
            // DEEP: the wave fetches its deep blocks TOGETHER (every lane loads slots of other lanes' blocks), so a lane without a
            // pixel to evaluate cannot leave the loop body: it stays as a lane whose tree slots are all idle (`live` false) and
            // skips only the stores
            bool live = true;
            if (compact) {
                if (slot >= n_valid) {
                    if (!DEEP) continue;
                    live = false;
                }
                entry = px_list[live ? slot : first];
            }
            // (entry = wave-row of the tile << 6 | lane)
            const int ly = (int)(y0 + (entry >> 6) * fold + ((entry & 63u) >> col_shift));
            const int px = (int)(x0 + (entry & ((1u << col_shift) - 1u)));
            if (!compact && (ly >= a.Hl || px >= a.Wl)) {
                if (!DEEP) continue;
                live = false;
            }
            const uint32_t i = img_loff + (uint32_t)ly * (uint32_t)a.Wl + (uint32_t)px;
            const int x = px * a.r, y = ly * a.r;
            const int xl = x - tx0, yl = y - ty0;   // this pixel, relative to the staged tile
            // tree_eval.cu:81-89 (a listed pixel passed these already)
            bool skip = false;
            if (!compact && live && a.filter_class != -1) skip = (int)a.filter[i] != a.filter_class;
            uint32_t d = 0u;
            if (!skip && live) d = (uint32_t)tprobe_value(tprobe_issue(pc, (uint32_t)xl * 2u, (uint32_t)yl));
            if (!compact && live && (skip || d == 0u || d == kNoPixel)) {
                if (a.fill_untouched && (!TW || tw_tree == 0u)) a.labels[i] = (uint16_t)kNoPixel;
                if (!DEEP) continue;
                live = false;
            }
            if (DEEP && !live) d = 1u;      // (finite arithmetic for a lane that only helps to fetch)
            const float df = (float)d;
            // refined reciprocal shared by every divide of this pixel (fast path only), in round-to-nearest
            const float r0 = __builtin_amdgcn_rcpf(df);
            const float rcp = __builtin_fmaf(__builtin_fmaf(-df, r0, 1.0f), r0, r0);
            float rcp_s = rcp * (1.0f / kNumScale);                           // exact scalings
            const float df_s = df * kNumScale;
            // coordinate = bits(fma(n, rcp_s, kMagic)) + (pixel coordinate - kMagicBits), x doubled (NodeRec16, TileCtx)
            uint32_t kx2 = ((uint32_t)xl - kMagicBits) * 2u;
            const uint32_t ky = (uint32_t)yl - kMagicBits;
            pin(kx2);     // (opaque: bits * 2 + kx2 stays one v_lshl_add_u32)

            float best = 0.0f;
            int best_c = 0;
            bool any_leaf = false;

            for (int c0 = 0; c0 < a.C || c0 == 0; c0 += CMAX) {
                float pdf[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) pdf[c] = 0.0f;

                for (int kb = TW ? (int)tw_tree : 0; kb < (TW ? (int)tw_tree + 1 : a.T); kb += GROUP) {
                    // Walk state, one word per tree: while walking, the 1-based heap index of the current node
                    // (< 2^31); once a leaf is reached, kDone | ((node - 1) * 2 + side); kIdle for tree slots
                    // beyond T.  "Still walking" is simply (int)h > 0.
                    uint32_t h[GROUP];
#pragma unroll
                    for (int k = 0; k < GROUP; ++k) h[k] = (live && (kb + k) < a.T) ? 1u : kIdle;

                    // The level loop runs in round-down mode (NodeRec16).  The sums of the previous group's leaf PDFs
                    // are pinned in front of the switch; the reference-layout kernel switches back for good at its
                    // first level below the staged ones, where every node takes the IEEE divide.
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) pin(pdf[c]);
                    const bool fast_levels = PACKED || K > 0;
                    if (fast_levels) set_round_down(rcp_s);

                    for (int j = 0; j < walk_levels; ++j) {
                        bool any = false;
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) any |= (int)h[k] > 0;
                        if (!__any(any)) break;

                        const bool in_lds = j < K;
                        float df_e = df;     // (a copy the mode switches tie)
                        if (!PACKED && fast_levels && j == K) set_round_nearest(df_e);
                        Node n[GROUP];
                        uint32_t hn[GROUP];   // node to fetch.  A finished (negative) or idle slot reads slot 0 of its tree's table, the
                                              // all-zero node: offsets 0, so its two (discarded) probes are the pixel itself -- inside
                                              // the staged tile, never a far load (on the trained-like topology 44 % of the slot-levels
                                              // are such slots).  The reference layout has no slot 0: there it is the level's first node.
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) hn[k] = (uint32_t)max((int)h[k], (PACKED || in_lds) ? 0 : 1 << j);
                        if (in_lds) {
#pragma unroll
                            for (int k = 0; k < GROUP; ++k) {
                                const uint32_t tk = (uint32_t)min(kb + k, a.T - 1);   // wave-uniform
                                n[k] = decode_node(lds_nodes[tk * lds_pitch + hn[k]]);
                                if (STATS && c0 == 0) st_lds += (int)h[k] > 0 ? 1u : 0u;
                            }
                        } else if (PACKED) {
#pragma unroll
                            for (int k = 0; k < GROUP; ++k) {
                                const int tk = min(kb + k, a.T - 1);
                                const char *base = reinterpret_cast<const char *>(a.packed16 + ((size_t)tk << a.D));
                                n[k] = decode_node(*reinterpret_cast<const uint4 *>(base + hn[k] * 16u));
                                if (STATS && c0 == 0) st_rec += lines_of((int)h[k] > 0, hn[k] * 16u);
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < GROUP; ++k) {
                                const int tk = min(kb + k, a.T - 1);
                                const float *p = a.forest + ((size_t)tk * (size_t)a.nodes + (hn[k] - 1u)) * (size_t)a.E;
                                if (STATS && c0 == 0) st_rec += lines_of((int)h[k] > 0, (hn[k] - 1u) * (uint32_t)a.E * 4u);
                                const f4u uv = *reinterpret_cast<const f4u *>(p);
                                const f3u tf = *reinterpret_cast<const f3u *>(p + 4);
                                n[k].ax = a.s * uv.x; n[k].ay = a.s * uv.y; n[k].bx = a.s * uv.z; n[k].by = a.s * uv.w;
                                set_threshold(n[k], thresh_to_int(tf.x));
                                n[k].flags = child_flags(tf.y, tf.z) | kFlagExact;   // fp32 numerators: IEEE divide
                            }
                        }

                        // ---- probe coordinates x + floor((s*u.x)/d) ... (decision_tree_common.hpp:15-22), relative to the
                        // staged tile, x doubled ----
                        uint32_t ux[GROUP], uy[GROUP], vx[GROUP], vy[GROUP];
                        uint32_t fl = 0u;
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) fl |= n[k].flags;
                        if (__any((fl & kFlagExact) != 0u)) {
                            // some lane holds a node whose numerators are not integer-representable: fetch the
                            // fp32 numerators for those lanes and divide IEEE, in round-to-nearest (every lane: same results)
                            const bool switch_mode = PACKED || in_lds;
                            if (switch_mode) set_round_nearest(df_e);
                            if (PACKED || in_lds) {
#pragma unroll
                                for (int k = 0; k < GROUP; ++k) {
                                    if (n[k].flags & kFlagExact) {
                                        const int tk = min(kb + k, a.T - 1);
                                        if (a.forest) {
                                            // (packed tables too: the numerators are recomputed from the caller's forest with the
                                            // scale the table was packed for -- read from the table's own info block -- one
                                            // multiply, as rdf_forest_pack did)
                                            const float s_exact = (PACKED && a.info) ? a.info->scale : a.s;
                                            const float *p = a.forest +
                                                ((size_t)tk * (size_t)a.nodes + (hn[k] - 1u)) * (size_t)a.E;
                                            n[k].ax = s_exact * p[0]; n[k].ay = s_exact * p[1];
                                            n[k].bx = s_exact * p[2]; n[k].by = s_exact * p[3];
                                        } else {
                                            // a table with such nodes evaluated without the forest: the host checks its count and refuses
                                            // (RDF_ERR_NULL_PTR), so this is a table the host does not know -- no fault, the stale flag
                                            n[k].ax = n[k].ay = n[k].bx = n[k].by = 0.0f;
                                            if (a.stale_flag) __hip_atomic_store(a.stale_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                        }
                                    }
                                }
                            }
#pragma unroll
                            for (int k = 0; k < GROUP; ++k) {
                                // lanes on an ordinary node still hold kNumScale * (x + e): same floor (NodeRec16)
                                const float den = (n[k].flags & kFlagExact) ? df_e : df_s;
                                // (x is doubled afterwards: a wrapped x + INT_MAX stays far outside every tile and image
                                // when held to +-2^30 first)
                                ux[k] = (uint32_t)min(max(add_wrap(xl, floor_i32(n[k].ax / den)), -(1 << 30)), (1 << 30) - 1) * 2u;
                                uy[k] = (uint32_t)add_wrap(yl, floor_i32(n[k].ay / den));
                                vx[k] = (uint32_t)min(max(add_wrap(xl, floor_i32(n[k].bx / den)), -(1 << 30)), (1 << 30) - 1) * 2u;
                                vy[k] = (uint32_t)add_wrap(yl, floor_i32(n[k].by / den));
                            }
                            if (switch_mode) {
#pragma unroll
                                for (int k = 0; k < GROUP; ++k) { pin(ux[k]); pin(uy[k]); pin(vx[k]); pin(vy[k]); }
                                set_round_down(ux[0]);
                            }
                        } else {
                            // divide, floor and add in two instructions per coordinate: t = fma(n, r/512, 1.5 * 2^23) rounded down
                            // holds floor(x/d) in its mantissa (NodeRec16; tools/verify_magic.hip)
#pragma unroll
                            for (int k = 0; k < GROUP; ++k) {
                                const f2 nu = {n[k].ax, n[k].ay};
                                const f2 nv = {n[k].bx, n[k].by};
                                const f2 r2 = {rcp_s, rcp_s};
                                const f2 m2 = {kMagic, kMagic};
                                const f2 tu = __builtin_elementwise_fma(nu, r2, m2);
                                const f2 tv = __builtin_elementwise_fma(nv, r2, m2);
                                ux[k] = (__float_as_uint(tu.x) << 1) + kx2;
                                uy[k] = __float_as_uint(tu.y) + ky;
                                vx[k] = (__float_as_uint(tv.x) << 1) + kx2;
                                vy[k] = __float_as_uint(tv.y) + ky;
                            }
                        }

                        TileProbe qu[GROUP], qv[GROUP];
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) {
                            qu[k] = tprobe_issue(pc, ux[k], uy[k]);
                            qv[k] = tprobe_issue(pc, vx[k], vy[k]);
                            if (STATS && c0 == 0) {
                                st_far += far_lines(pc, ux[k], uy[k], (int)h[k] > 0, img_boff);
                                st_far += far_lines(pc, vx[k], vy[k], (int)h[k] > 0, img_boff);
                            }
                        }

                        // ---- decide (tree_eval.cu:107-121) ----
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) {
                            const bool walking = (int)h[k] > 0;
                            if (STATS && c0 == 0) st_lv += walking ? 1u : 0u;
                            const int g = tprobe_value(qu[k]) - tprobe_value(qv[k]) - (int)n[k].lo16;
                            const uint32_t next = walk_step(h[k], g, n[k].w2, n[k].flags);   // low bits: (node-1)*2 + side + 2
                            h[k] = walking ? next : h[k];
                        }
                    }
                    if constexpr (DEEP) {
                      if (deep_on) {
                        // ---- deep blocks: the levels from a.deep_from on, one tree after the other, three levels per block (the last
                        // block: two levels and their four leaf PDFs; five to eight classes: one level and its two PDFs).  A block is
                        // one 128-byte line, and the WAVE fetches the 64 blocks its lanes stand on with eight LDS-DMA loads of 1 KB: in
                        // load k the eight lanes 8 i .. 8 i + 7 fetch the eight 16-byte slots of the block lane 8 k + i stands on (its
                        // offset comes over by ds_bpermute; a lane whose walk has ended names the level's first block) --
                        // eight whole lines per instruction, 64 line look-ups per block step where a lane fetching its own block with
                        // seven 16-byte loads made 448 (round 4: TA busy 94-97 %) -- straight into the wave's 8-KB slab
                        // [lane's block][slot]; then each lane reads the three records its walk takes (root, the child and the
                        // grandchild it goes to) with three ds_read_b128.  The slot a record sits in is XOR-ed with bits 1-3 of the
                        // owning lane, so that the 16 lanes of a ds_read_b128 group hit 16 different bank quads when they read the same
                        // record (an LDS-DMA's destination is lane-linear: the swizzle is applied to the SOURCE address).  Every lane of
                        // the wave takes part in the fetch, which is why lanes without a pixel stay in the loop (`live`).  The PDFs are
                        // added tree after tree (the canonical order) in round-to-nearest: the mode goes back and forth per tree.  A
                        // lane whose walk has ended decodes an all-zero record (offsets 0: its discarded probes are the pixel itself);
                        // one that ended above level D-1 takes its leaf from the PDF table. ----
                        const int R0 = a.D - kDeepLast, Lmin = R0 % 3;
                        const f2 r2 = {rcp_s, rcp_s};
                        const f2 m2 = {kMagic, kMagic};
                        unsigned char *slab = lds_raw + a.lds_slab_off + wave * 8192u;
                        const unsigned char *my_block = slab + ((uint32_t)lane << 7);
                        const uint32_t my_swz = ((uint32_t)lane >> 1) & 7u;
                        // the record the lane fetches in an even load (owner 8 k + lane / 8, swizzle (4 k + lane / 16) & 7): slot ^ swizzle
                        const uint32_t src_rec = (((uint32_t)lane & 7u) ^ ((uint32_t)lane >> 4)) << 4;
#pragma unroll
                        for (int k0 = 0; k0 < GROUP; ++k0) {
                            uint32_t hk = h[k0];
                            const int tk = min(kb + k0, a.T - 1);
                            auto fetch_blocks = [&](int R) {
                                const size_t base_at = (deep_lines_before(a.T, R, Lmin) + ((size_t)tk << R) - ((size_t)1 << R)) << 7;
                                // (a lane whose walk has ended names the level's first block of the tree: one line the whole chip shares,
                                // so every lane issues all eight loads and no load sits behind a branch -- fetching nothing for such
                                // lanes, 8 x s_cbranch_execz, cost 3 % on a forest where every lane walks and gained nothing on one where
                                // half of them have ended; what such a lane reads from its slab is replaced by an all-zero record)
                                const uint32_t off = ((int)hk > 0 ? hk : (1u << R)) << 7;
                                const char *tb = reinterpret_cast<const char *>(a.deep) + base_at;
#pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    const uint32_t owner_off = (uint32_t)__shfl((int)off, k * 8 + (lane >> 3));
                                    const uint32_t byte = owner_off + (src_rec ^ ((k & 1) ? 64u : 0u));
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tb + byte),
                                                                     (__attribute__((address_space(3))) void *)(slab + k * 1024), 16, 0, 0);
                                }
                                if (STATS && c0 == 0) st_blk += (int)hk > 0 ? 1u : 0u;
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            };
                            auto slot_of = [&](bool mine, uint32_t rec) -> uint4 {       // a record of this lane's block, if it has one
                                const uint4 v = *reinterpret_cast<const uint4 *>(my_block + ((rec ^ my_swz) << 4));
                                return select4(mine, v, make_uint4(0u, 0u, 0u, 0u));
                            };
                            auto slot = [&](uint32_t rec) -> uint4 { return slot_of((int)hk > 0, rec); };
                            auto level1 = [&](const uint4 rec) {
                                const Node n = decode_node(rec);
                                const f2 nu = {n.ax, n.ay};
                                const f2 nv = {n.bx, n.by};
                                const f2 tu = __builtin_elementwise_fma(nu, r2, m2);
                                const f2 tv = __builtin_elementwise_fma(nv, r2, m2);
                                const TileProbe qu = tprobe_issue(pc, (__float_as_uint(tu.x) << 1) + kx2, __float_as_uint(tu.y) + ky);
                                const TileProbe qv = tprobe_issue(pc, (__float_as_uint(tv.x) << 1) + kx2, __float_as_uint(tv.y) + ky);
                                if (STATS && c0 == 0) {
                                    st_lv += (int)hk > 0 ? 1u : 0u;
                                    st_far += far_lines(pc, (__float_as_uint(tu.x) << 1) + kx2, __float_as_uint(tu.y) + ky, (int)hk > 0, img_boff);
                                    st_far += far_lines(pc, (__float_as_uint(tv.x) << 1) + kx2, __float_as_uint(tv.y) + ky, (int)hk > 0, img_boff);
                                }
                                const int g = tprobe_value(qu) - tprobe_value(qv) - (int)n.lo16;
                                const uint32_t next = walk_step(hk, g, n.w2, n.flags);
                                hk = (int)hk > 0 ? next : hk;
                            };
                            for (int R = a.deep_from; R < R0; R += 3) {
                                if (!__any((int)hk > 0)) break;
                                fetch_blocks(R);
                                level1(slot(0u));
                                const uint32_t sa = hk & 1u;
                                level1(slot(1u + sa));
                                level1(slot(3u + 2u * sa + (hk & 1u)));
                            }
                            bool have = false;                  // this lane's walk reached level D-1 (every node there has two leaves)
                            uint4 p0 = make_uint4(0u, 0u, 0u, 0u), p1 = make_uint4(0u, 0u, 0u, 0u);
                            if (__any((int)hk > 0)) {
                                fetch_blocks(R0);
                                if (CMAX == 4) {                // {root, left, right, four leaf PDFs}
                                    level1(slot(0u));
                                    const uint32_t sa = hk & 1u;
                                    have = (int)hk > 0;
                                    level1(slot(1u + sa));
                                    p0 = slot_of(have, 3u + 2u * sa + (hk & 1u));
                                } else {                        // {node, pad, left PDF (32 bytes), right PDF (32 bytes)}
                                    have = (int)hk > 0;
                                    level1(slot(0u));
                                    const uint32_t sb = hk & 1u;
                                    p0 = slot_of(have, 2u + 2u * sb);
                                    p1 = slot_of(have, 3u + 2u * sb);
                                }
                            }
                            const bool early = (int)hk < 0 && hk != kIdle && !have;     // ended above level D-1: the PDF table has its leaf
                            if (__any(early)) {
                                if (STATS && c0 == 0) st_leaf += lines_of(early, (hk & ~kDone) * (uint32_t)(a.cpad * 4));
                                const uint4 *row = reinterpret_cast<const uint4 *>(a.packed_pdf) +
                                                   (early ? (((size_t)tk << (a.D + 1)) + (hk & ~kDone)) * (size_t)(a.cpad >> 2) : (size_t)0);
                                const uint4 e0 = row[0];
                                p0 = select4(early, e0, p0);
                                if (CMAX == 8) {
                                    const uint4 e1 = row[1];
                                    p1 = select4(early, e1, p1);
                                }
                            }
                            pin(p0.x); pin(p0.y); pin(p0.z); pin(p0.w);
                            if (CMAX == 8) { pin(p1.x); pin(p1.y); pin(p1.z); pin(p1.w); }
                            set_round_nearest(p0.x);
                            pin(p0.y); pin(p0.z); pin(p0.w);
                            if (CMAX == 8) { pin(p1.x); pin(p1.y); pin(p1.z); pin(p1.w); }
                            if (have || early) {
                                if (STATS && c0 == 0) st_lf++;
                                pdf[0] = pdf[0] + __uint_as_float(p0.x); pdf[1] = pdf[1] + __uint_as_float(p0.y);
                                pdf[2] = pdf[2] + __uint_as_float(p0.z); pdf[3] = pdf[3] + __uint_as_float(p0.w);
                                if constexpr (CMAX == 8) {
                                    pdf[4] = pdf[4] + __uint_as_float(p1.x); pdf[5] = pdf[5] + __uint_as_float(p1.y);
                                    pdf[6] = pdf[6] + __uint_as_float(p1.z); pdf[7] = pdf[7] + __uint_as_float(p1.w);
                                }
                                any_leaf = true;
                            }
                            if (k0 + 1 < GROUP) {
#pragma unroll
                                for (int c = 0; c < CMAX; ++c) pin(pdf[c]);
                                set_round_down(pdf[0]);
#pragma unroll
                                for (int c = 1; c < CMAX; ++c) pin(pdf[c]);
                            }
                        }
                        continue;       // (next group of trees; the mode is round-to-nearest)
                      }
                    }
                    // ---- level D-1 from the last-level table, one tree after the other: the node and both leaf PDFs of a
                    // tree come with one line fill and the side taken picks the PDF (every record of a usable table is an
                    // ordinary node with two leaves: no IEEE divide, no "continue").  A lane whose walk ended higher up
                    // points the same two PDF loads at its leaf in the PDF table instead, so the PDFs of a tree arrive together
                    // and are added at once, in tree order (the canonical order), in round-to-nearest: the mode goes back and
                    // forth per tree.  (A wave with no lane at level D-1 takes the general path below.) ----
                    if (kLastLevelTable && last_from_table) {
                        bool any_w = false;
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) any_w |= (int)h[k] > 0;
                        if (__any(any_w)) {
                            const uint32_t first = 1u << (a.D - 1);
#pragma unroll
                            for (int k = 0; k < GROUP; ++k) {
                                const bool walking = (int)h[k] > 0;
                                const bool ended = (int)h[k] < 0 && h[k] != kIdle;      // at a leaf above level D-1
                                const uint32_t hn = (uint32_t)max((int)h[k], (int)first);
                                const int tk = min(kb + k, a.T - 1);
                                const uint4 *rec = a.last_level + ((((size_t)tk << (a.D - 1)) + (hn - first)) << 2);
                                // PDF table, 16-byte rows: ((tree << D) + node) * 2 + side = (tree << (D + 1)) + (h & ~kDone), see add_leaf_pdf
                                const uint4 *leaf = reinterpret_cast<const uint4 *>(a.packed_pdf) + (((size_t)tk << (a.D + 1)) + (h[k] & ~kDone));
                                const uint4 *pp = ended ? leaf : rec + 1;
                                // (a lane that is not walking takes the all-zero node of slot 0: probes inside the tile)
                                const uint4 *np = walking ? rec : reinterpret_cast<const uint4 *>(a.packed16) + ((size_t)tk << a.D);
                                uint4 w = np[0];
                                float4 pa = *reinterpret_cast<const float4 *>(pp);
                                float4 pb = *reinterpret_cast<const float4 *>(pp + 1);
                                // (the three loads are issued together -- a record's line is filled once -- and nothing is
                                // scheduled between the node's arrival and the PDFs' issue)
                                asm volatile("" : "+v"(w.x), "+v"(pa.x), "+v"(pb.x));
                                const Node n = decode_node(w);
                                const f2 nu = {n.ax, n.ay};
                                const f2 nv = {n.bx, n.by};
                                const f2 r2 = {rcp_s, rcp_s};
                                const f2 m2 = {kMagic, kMagic};
                                const f2 tu = __builtin_elementwise_fma(nu, r2, m2);
                                const f2 tv = __builtin_elementwise_fma(nv, r2, m2);
                                const TileProbe qu = tprobe_issue(pc, (__float_as_uint(tu.x) << 1) + kx2, __float_as_uint(tu.y) + ky);
                                const TileProbe qv = tprobe_issue(pc, (__float_as_uint(tv.x) << 1) + kx2, __float_as_uint(tv.y) + ky);
                                if (STATS && c0 == 0) {
                                    st_lv += walking ? 1u : 0u;
                                    st_rec += lines_of(walking, (hn - first) * 64u);          // (node and leaf PDFs: one 64-byte record)
                                    st_leaf += lines_of(ended, (h[k] & ~kDone) * 16u);
                                    st_far += far_lines(pc, (__float_as_uint(tu.x) << 1) + kx2, __float_as_uint(tu.y) + ky, walking, img_boff);
                                    st_far += far_lines(pc, (__float_as_uint(tv.x) << 1) + kx2, __float_as_uint(tv.y) + ky, walking, img_boff);
                                }
                                const int g = tprobe_value(qu) - tprobe_value(qv) - (int)n.lo16;
                                const uint32_t next = walk_step(h[k], g, n.w2, n.flags);
                                const bool right = walking && (next & 1u) != 0u;
                                float4 sel = make_float4(right ? pb.x : pa.x, right ? pb.y : pa.y, right ? pb.z : pa.z, right ? pb.w : pa.w);
                                pin(sel.x); pin(sel.y); pin(sel.z); pin(sel.w);
                                set_round_nearest(sel.x);
                                pin(sel.y); pin(sel.z); pin(sel.w);
                                if (walking || ended) {
                                    pdf[0] = pdf[0] + sel.x; pdf[1] = pdf[1] + sel.y;
                                    pdf[2] = pdf[2] + sel.z; pdf[3] = pdf[3] + sel.w;
                                    any_leaf = true;
                                    if (STATS && c0 == 0) st_lf++;
                                }
                                if (k + 1 < GROUP) {
#pragma unroll
                                    for (int c = 0; c < CMAX; ++c) pin(pdf[c]);
                                    set_round_down(pdf[0]);
#pragma unroll
                                    for (int c = 1; c < CMAX; ++c) pin(pdf[c]);
                                }
                            }
                            continue;       // (next group of trees; the mode is round-to-nearest)
                        }
                    }
                    // back to round-to-nearest for the sums of leaf PDFs (tied to the walk's results)
                    if (fast_levels) {
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) pin(h[k]);
                        set_round_nearest(h[0]);
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) pin(h[k]);
                    }

                    if (TW) { tw_h = h[0]; break; }       // the trees meet in LDS, below the pixel loop
                    // leaf PDFs, strictly in tree order (canonical sum order)
                    if (PACKED && CMAX == 4) {
                        // the group's PDF rows are fetched together (a lane without a leaf in a tree reads the table's
                        // first row and drops it) and then added in order: one round trip, not one per tree
                        float4 row[GROUP];
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) {
                            const bool ended = (int)h[k] < 0 && h[k] != kIdle;
                            const int tk = min(kb + k, a.T - 1);
                            // 16-byte rows: (((tree << D) + node) * 2 + side) * (cpad / 4) + c0 / 4, node * 2 + side = h & ~kDone (add_leaf_pdf)
                            const size_t at = ((((size_t)tk << (a.D + 1)) + (h[k] & ~kDone)) * (size_t)(a.cpad >> 2)) + (size_t)(c0 >> 2);
                            row[k] = reinterpret_cast<const float4 *>(a.packed_pdf)[ended ? at : 0];
                            if (STATS && c0 == 0) st_leaf += lines_of(ended, (uint32_t)at * 16u);
                        }
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) {
                            if ((int)h[k] < 0 && h[k] != kIdle) {
                                pdf[0] = pdf[0] + row[k].x; pdf[1] = pdf[1] + row[k].y;
                                pdf[2] = pdf[2] + row[k].z; pdf[3] = pdf[3] + row[k].w;
                                any_leaf = true;
                                if (STATS && c0 == 0) st_lf++;
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < GROUP; ++k) {
                            if (add_leaf_pdf(h[k], kb + k, c0, pdf)) {
                                any_leaf = true;
                                if (STATS && c0 == 0) st_lf++;
                            }
                        }
                    }
                }
                if (TW) break;

#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    if (c0 + c < a.C && pdf[c] > best) {
                        best = pdf[c];
                        best_c = c0 + c;
                    }
                }
            }

            if (TW) { tw_valid = true; tw_i = i; continue; }
            if (DEEP && !live) continue;
            if (STATS) st_px++;
            if (a.keep_if_no_leaf && !any_leaf) {
                if (a.fill_untouched) a.labels[i] = (uint16_t)kNoPixel;
                continue;
            }
            a.labels[i] = (uint16_t)best_c;
        }
        if (TW) {
            // ---- the T waves of a pixel row hand their trees' results to the row's first wave, which sums the leaf PDFs
            // in tree order (the canonical order) and writes the label ----
            uint32_t *xch = reinterpret_cast<uint32_t *>(lds_raw + a.lds_xchg_off);     // [tile row][tree][lane]
            if (tw_row < tile_rows) xch[(tw_row * tw_T + tw_tree) * 64u + (uint32_t)lane] = tw_valid ? tw_h : kIdle;
            __syncthreads();       // (the barrier at the top of the next tile keeps the next round's writers away)
            if (tw_row < tile_rows && tw_tree == 0u && tw_valid) {
                float best = 0.0f;
                int best_c = 0;
                bool any_leaf = false;
                for (int c0 = 0; c0 < a.C || c0 == 0; c0 += CMAX) {
                    float pdf[CMAX];
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) pdf[c] = 0.0f;
                    for (uint32_t t = 0; t < tw_T; ++t)
                        any_leaf |= add_leaf_pdf(xch[(tw_row * tw_T + t) * 64u + (uint32_t)lane], (int)t, c0, pdf);
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (c0 + c < a.C && pdf[c] > best) {
                            best = pdf[c];
                            best_c = c0 + c;
                        }
                    }
                }
                if (a.keep_if_no_leaf && !any_leaf) {
                    if (a.fill_untouched) a.labels[tw_i] = (uint16_t)kNoPixel;
                } else {
                    a.labels[tw_i] = (uint16_t)best_c;
                }
            }
        }
    }

    // ---- (round 6) is the table the one the host remembers at this address?  Every packing carries a generation; a launch
    // that finds another one raises the device's stale flag (pinned host memory): the library then forgets what it knew and
    // the next call says RDF_ERR_STALE ----
    // (the LAST workgroup of the launch -- of every role of a multi-layer launch -- asks: its tile is a frame's bottom-right
    // corner, often partial or empty, so the load's round trip tends to end before the launch's other workgroups do)
    if (PACKED && a.info && a.stale_flag && block_id == n_blocks - 1u && tid == 0 && !a.q_helper) {
        if (a.info->generation != a.expect_gen)
            __hip_atomic_store(a.stale_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    // ---- queue epilogue: the last workgroup to finish puts the slot back to zero (every
    // workgroup has made its final, failing pull before it gets here) ----
    if (a.sched && queued != 0u && tid == 0) {
        const unsigned int done = atomicAdd(a.sched + 8 * kSchedStride, 1u);
        if (done == q_blocks - 1u) {
            for (int q = 0; q < 9; ++q) atomicExch(a.sched + q * kSchedStride, 0u);
        }
    }

    if (STATS) {
        // wave reduction, one atomic per wave and counter
        uint32_t c[8] = {st_px, st_lv, st_lf, st_lds, st_rec, st_leaf, st_far, st_blk};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned long long v = c[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0 && (i < 3 || a.stats_wide)) atomicAdd(a.stats + i, v);
        }
    }
}

// ---- load-time repack: one thread per node; writes the hot records, the PDF rows, the last-level records, the deep blocks ----
__global__ __launch_bounds__(256) void k_pack(const float *forest, NodeRec16 *packed16, PackInfo *info,
                                              float *packed_pdf, LastLevelRec *last_level, unsigned int *last_level_unusable,
                                              uint4 *deep, int n_trees, int C, int cpad,
                                              size_t total_slots, int D, int E, float s, int force_exact, uint32_t generation)
{
    // slot = tree * 2^D + h, h = 1-based heap index (slot h == 0 of each tree is unused and zeroed)
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= total_slots) return;
    const size_t tree = slot >> D, h = slot & (((size_t)1 << D) - 1);
    if (h == 0) {
        NodeRec16 z16 = {{0u, 0u, 0u, 0u}};
        packed16[slot] = z16;
        for (int c = 0; c < 2 * cpad; ++c) packed_pdf[slot * 2 * (size_t)cpad + c] = 0.f;
        if (tree == 0) {
            info->scale = s; info->magic = kPackMagic; info->generation = generation;
            info->n_trees = (uint32_t)n_trees; info->max_depth = (uint32_t)D; info->n_classes = (uint32_t)C;
        }
        if (deep && tree == 0) {     // the all-zero line behind the blocks
            uint4 *z = deep + (deep_total_lines(n_trees, D, cpad) << 3);
            for (int i = 0; i < 8; ++i) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    const size_t i = tree * ((((size_t)1) << D) - 1) + (h - 1);
    const float *p = forest + i * (size_t)E;
    struct { uint32_t flags; } n;
    NodeRec16 h16 = encode_node(s * p[0], s * p[1], s * p[2], s * p[3], p[4], p[5], p[6]);
    if (force_exact) h16.w[3] |= kFlagExact;   // test knob: exercise the IEEE branch everywhere
    n.flags = h16.w[3] & 0xFFu;
    packed16[slot] = h16;
    if (n.flags & kFlagExact) atomicAdd(&info->exact_nodes, 1u);     // (practically never)
    float *q = packed_pdf + slot * 2 * (size_t)cpad;   // [left: cpad][right: cpad], the PDFs as stored (row N of SURVEY 8a)
    for (int c = 0; c < cpad; ++c) {
        q[c] = c < C ? p[7 + c] : 0.f;
        q[cpad + c] = c < C ? p[7 + C + c] : 0.f;
    }
    if (deep) {     // the node's place in the deep blocks (see "deep blocks" above)
        const int R0 = D - deep_last_levels(cpad), Lmin = R0 % 3;
        const int j = 63 - __clzll((unsigned long long)h);      // the node's level
        if (j >= Lmin) {
            const int R = j >= R0 ? R0 : R0 - 3 * ((R0 - j + 2) / 3);   // its block's root level
            const int t = j - R;
            const size_t line = deep_lines_before(n_trees, R, Lmin) + (tree << R) + ((h >> t) - ((size_t)1 << R));
            uint4 *blk = deep + (line << 3);
            const uint4 rec = make_uint4(h16.w[0], h16.w[1], h16.w[2], h16.w[3]);
            const float4 *pq = reinterpret_cast<const float4 *>(q);
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            if (R == R0 && cpad == 8) {          // {node, pad, left PDF, right PDF}
                blk[0] = rec; blk[1] = z; blk[6] = z; blk[7] = z;
                for (int i = 0; i < 4; ++i) reinterpret_cast<float4 *>(blk)[2 + i] = pq[i];
            } else {
                blk[((1u << t) - 1u) + (uint32_t)(h & ((1u << t) - 1u))] = rec;
                if (t == 0) blk[7] = z;
                if (R == R0 && t == 1) {         // a node of level D-1 (cpad == 4): its two leaf PDFs
                    const uint32_t c = (uint32_t)(h & 1u);
                    reinterpret_cast<float4 *>(blk)[3 + 2 * c] = pq[0];
                    reinterpret_cast<float4 *>(blk)[4 + 2 * c] = pq[1];
                }
            }
            // trailer: word 0 = 1 + the deepest level that holds a kFlagExact node, word 1 = nodes of level D-1 that are not
            // plain two-leaf nodes
            unsigned int *tr = reinterpret_cast<unsigned int *>(deep + ((deep_total_lines(n_trees, D, cpad) + 1u) << 3));
            if (n.flags & kFlagExact) atomicMax(tr, (unsigned)j + 1u);
            if (j == D - 1 && (n.flags & 7u) != (kFlagLeftLeaf | kFlagRightLeaf)) atomicAdd(tr + 1, 1u);
        }
    }
    const size_t first = (size_t)1 << (D - 1);
    if (last_level && h >= first) {     // (LastLevelRec; cpad == 4)
        LastLevelRec r;
        r.node = h16;
        for (int c = 0; c < 4; ++c) {
            r.pdf_left[c] = q[c];
            r.pdf_right[c] = q[4 + c];
            r.pad[c] = 0u;
        }
        last_level[(tree << (D - 1)) + (h - first)] = r;
        // trailer word 0: records the table cannot serve; word 1: records whose parent says "continue" on their side (how
        // much of level D-1 the forest uses at all).  One atomic per wave and word.
        const bool unusable = (n.flags & 7u) != (kFlagLeftLeaf | kFlagRightLeaf);
        const float *pp = forest + (tree * ((((size_t)1) << D) - 1) + ((h >> 1) - 1)) * (size_t)E;
        const float to_me = pp[5 + (h & 1)];
        const bool in_use = to_me >= -1.0f && to_me < 0.0f;
        const unsigned long long bu = __ballot(unusable), bi = __ballot(in_use);
        const unsigned long long mine = __ballot(true);
        if ((unsigned)__lane_id() == (unsigned)__ffsll((long long)mine) - 1u) {
            if (bu) atomicAdd(last_level_unusable, (unsigned)__popcll(bu));
            if (bi) atomicAdd(last_level_unusable + 1, (unsigned)__popcll(bi));
        }
    }
}

// ---- composite (tree_eval.cu:214-248): one lane per label pixel ----
// Optional extras of the app's per-hand chain folded into the store (3d_bz.py:440-456): the composite is written
// mirrored in x (`flip_w` = label row width, 0 = as is; the app flips the left hand's labels back) and the label's
// colour goes to an RGBA image at the same place (make_rgba_from_labels, points_ops.cu:258-281: labels 0, 65535 and
// > num_colors leave the texel alone).
// `spec.n` > 0: the layers were evaluated UNFILTERED in one launch (layered_run on a small launch); this kernel then
// applies each layer's filter first -- layer i keeps its label only where the (already filtered) label of layer
// spec.fl[i] equals spec.fc[i], else it becomes 65535, exactly what the filtered evaluation leaves behind its fill
// (tree_eval.cu:81-85, decision_tree.py:237-240) -- and stores what changed.
constexpr int kMaxSpecLayers = 4;
struct SpecFilters {
    int n;
    int fl[kMaxSpecLayers], fc[kMaxSpecLayers];
};

__global__ __launch_bounds__(256) void k_composite(const uint16_t *const *imgs, int n_images, uint32_t n_px,
                                                   const int2 *cond, int n_cond, uint16_t *out, int32_t *bad,
                                                   int fill_untouched, uint32_t flip_w, const uint32_t *colors,
                                                   int num_colors, uint32_t *rgba, const SpecFilters spec)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_px) return;
    uint32_t lab[kMaxSpecLayers];
    if (spec.n > 0) {
#pragma unroll
        for (int i = 0; i < kMaxSpecLayers; ++i) {
            lab[i] = kNoPixel;
            if (i < spec.n) {
                uint32_t l = imgs[i][p];
                if (spec.fl[i] >= 0) {
                    uint32_t f = kNoPixel;
#pragma unroll
                    for (int j = 0; j < kMaxSpecLayers; ++j)
                        if (j < i && j == spec.fl[i]) f = lab[j];
                    if ((int)f != spec.fc[i] && l != kNoPixel) {
                        l = kNoPixel;
                        const_cast<uint16_t *>(imgs[i])[p] = (uint16_t)kNoPixel;
                    }
                }
                lab[i] = l;
            }
        }
    }
    uint32_t q = p;
    if (flip_w) {
        const uint32_t y = p / flip_w, x = p - y * flip_w;
        q = y * flip_w + (flip_w - 1u - x);
    }
    long long off = 0;
    bool invalid = true;   // fell off the last image (tree_eval.cu:246-247)
    for (int i = 0; i < n_images; ++i) {
        uint32_t l;
        if (spec.n > 0) {
            l = kNoPixel;
#pragma unroll
            for (int j = 0; j < kMaxSpecLayers; ++j)
                if (j == i) l = lab[j];
        } else {
            l = imgs[i][p];
        }
        if (l == 0u || l == kNoPixel) { invalid = false; break; }   // :235 -- pixel keeps its pre-fill
        const long long e = off + (long long)l - 1;
        if (e < 0 || e >= n_cond) break;
        const int2 tv = cond[e];
        if (tv.x == 0) {
            const uint32_t v = (uint32_t)(uint16_t)tv.y;
            out[q] = (uint16_t)v;
            if (rgba && v != 0u && v != kNoPixel && v <= (uint32_t)num_colors) rgba[q] = colors[v - 1u];
            return;
        }
        off = tv.y;
    }
    if (invalid && bad) atomicAdd(bad, 1);
    if (fill_untouched) out[q] = (uint16_t)kNoPixel;
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

// A kernel with the resource footprint of RCCL's send/recv kernel on gfx950 (rcclGenericKernel in librccl.so:
// 256 threads, ~280 VGPRs, 19.7 KB of LDS -- one workgroup fills the register files of a whole CU), spinning for a
// given time.  tools/ubench_overlap.py uses it to show when such a kernel can start next to the forest kernel.
__global__ __launch_bounds__(256) void k_debug_fat(unsigned long long spin_ticks, unsigned long long *t_start)
{
    __shared__ unsigned int pad[19744 / 4];
    asm volatile("v_mov_b32 v250, 0" ::: "v250");           // allocates > 250 VGPRs per wave
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && t_start) t_start[blockIdx.x] = t0;
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (pad[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu && t_start) t_start[0] = 0;   // keeps pad alive
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
// Tuning / test knobs: process-wide, each read once per call (atomic: a library two threads use must not tear them; a
// call that overlaps a change sees the old or the new value)
typedef std::atomic<int> Knob;
Knob g_lds_budget{0};
Knob g_block_threads{0};

// The RDF_* environment is read ONCE per variable (the first call that looks at it; thread-safe function-local statics): a
// launch used to pay ten getenv scans of the environment, microseconds on the critical path of a sync-per-frame live loop.
// The rdf_set_* knobs are the way to change a choice at run time.
struct EnvVal {
    bool set;
    int val;
};
inline EnvVal read_env(const char *name)
{
    const char *v = getenv(name);
    const bool set = v && *v;
    return {set, set ? atoi(v) : 0};
}
#define env_int(name, dflt) ([&]() -> int { static const EnvVal e_ = read_env(name); return e_.set ? e_.val : (dflt); }())

// C-side cost of a call (rdf_debug_host_overhead): nanoseconds eval_common spends before it hands the launch to the HIP
// runtime, and inside the runtime's launch call
std::atomic<unsigned long long> g_host_ns_plan{0}, g_host_ns_launch{0}, g_host_calls{0}, g_host_ns_layered{0}, g_host_layered_calls{0};
inline unsigned long long now_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec;
}

// Launches that walk deep blocks (their waves fetch the blocks through LDS slabs): what is left for the depth tile, the node
// table, the mailbox and the pixel list next to 8 KB of slab per wave -- two 512-thread or four 256-thread workgroups per CU.
int coop_lds_budget(int block)
{
    const int knob = g_lds_budget;
    if (knob > 0) return knob;
    const int v = env_int("RDF_COOP_LDS_BUDGET", 0);
    if (v > 0) return v;
    return block == 512 ? 81920 - 65536 - 1024 : 40960 - 32768 - 1024;
}

int lds_budget(int block)
{
    // per workgroup: five 256-thread or three 512-thread workgroups share a CU's 160 KB
    const int knob = g_lds_budget;
    int b = knob > 0 ? knob : env_int("RDF_LDS_BUDGET", block == 512 ? kDefaultLdsBudget512 : kDefaultLdsBudget);
    if (b > 160 * 1024) b = 160 * 1024;
    if (b < 0) b = 0;
    return b;
}

struct DeviceInfo {
    int cus = 0, dev = 0;
    bool ok = false;
};

std::atomic<int> g_device_cus[64];      // compute units per device ordinal, asked once (0: not yet)

int device_info(DeviceInfo *out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    int cus = (dev >= 0 && dev < 64) ? g_device_cus[dev].load(std::memory_order_relaxed) : 0;
    if (cus == 0) {
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        cus = cus > 0 ? cus : 256;
        if (dev >= 0 && dev < 64) g_device_cus[dev].store(cus, std::memory_order_relaxed);
    }
    out->cus = cus;
    out->dev = dev;
    out->ok = true;
    return 0;
}

// ---- scheduler slots: one per (device, stream) that has launched; beyond kSchedSlots distinct
// streams the kernel falls back to static round-robin chunks (sched == nullptr) ----
std::mutex g_sched_mu;
std::map<std::tuple<int, void *, int>, int> g_sched_slot;   // (device, stream, role of a multi-forest launch) -> slot
std::map<int, unsigned int *> g_sched_base;     // device -> address of g_sched on that device
Knob g_compaction{-1};                          // -1: filtered launches compact their pixels; 0: never
Knob g_group{0};                                // 0: trees per lane chosen by forest size; 1..4: forced (rdf_set_group)
Knob g_layers_one_launch{-1};                   // -1/1: small packed layered runs evaluate their layers in one launch; 0: never
Knob g_sched_mode{-1};                          // -1: env RDF_SCHED (default dynamic), 0 static, 1 dynamic, 2 one tile per workgroup

int sched_mode()
{
    int mode = g_sched_mode;
    if (mode < 0) {
        static const int from_env = []() {
            const char *v = getenv("RDF_SCHED");
            return (v && strcmp(v, "static") == 0) ? 0 : (v && strcmp(v, "tile") == 0) ? 2 : 1;
        }();
        mode = from_env;
    }
    return mode;
}

std::map<int, std::vector<int>> g_sched_free;   // device -> stream slots given back by rdf_stream_destroy
std::map<int, int> g_sched_next;                // device -> stream slots handed out so far
std::map<int, int> g_graph_next;                // device -> graph slots handed out so far (never handed out twice at once)
std::map<int, std::vector<int>> g_graph_free;   // device -> graph slots given back by rdf_graph_slots_release
std::map<std::pair<int, unsigned long long>, std::vector<int>> g_graph_by_capture;   // (device, capture id) -> its slots
bool g_graph_exhausted_logged = false;

unsigned int *sched_slot(void *stream, int role = 0)
{
    if (sched_mode() != 1) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_sched_mu);
    auto bit = g_sched_base.find(dev);
    if (bit == g_sched_base.end()) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sched)) != hipSuccess || !p) return nullptr;
        bit = g_sched_base.emplace(dev, reinterpret_cast<unsigned int *>(p)).first;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    unsigned long long cap_id = 0;
    if (stream && hipStreamGetCaptureInfo(reinterpret_cast<hipStream_t>(stream), &cap, &cap_id) == hipSuccess &&
        cap == hipStreamCaptureStatusActive) {
        // a slot of its own for every launch recorded into a graph, remembered under the capture's id so that the owner of
        // the graph can give the slots back (rdf_graph_slots_release) when the executable graph is gone
        int slot = -1;
        auto &fl = g_graph_free[dev];
        if (!fl.empty()) {
            slot = fl.back();
            fl.pop_back();
        } else if (g_graph_next[dev] < kGraphSlots) {
            slot = g_graph_next[dev]++;
        }
        if (slot < 0) {   // static tiles: slower on uneven batches, never wrong -- but say so once
            if (!g_graph_exhausted_logged) {
                g_graph_exhausted_logged = true;
                fprintf(stderr, "librdf_hip: all %d tile-queue slots for graph-captured launches are taken; further captured "
                                "launches use static tiles (release graphs' slots with rdf_graph_slots_release)\n", kGraphSlots);
            }
            return nullptr;
        }
        g_graph_by_capture[std::make_pair(dev, cap_id)].push_back(slot);
        return bit->second + kSchedWords * (kSchedSlots + slot);
    }
    const auto key = std::make_tuple(dev, stream, role);
    auto it = g_sched_slot.find(key);
    if (it == g_sched_slot.end()) {
        int slot = -1;
        auto &fl = g_sched_free[dev];
        if (!fl.empty()) {
            slot = fl.back();
            fl.pop_back();
        } else if (g_sched_next[dev] < kSchedSlots) {
            slot = g_sched_next[dev]++;
        }
        if (slot < 0) return nullptr;
        it = g_sched_slot.emplace(key, slot).first;
    }
    return bit->second + kSchedWords * it->second;
}

// CUs a stream's kernels may run on (hipExtStreamCreateWithCUMask), cached per stream handle.
std::map<std::pair<int, void *>, int> g_stream_cus;
int usable_cus(hipStream_t st, int device_cus)
{
    if (!st) return device_cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return device_cus;
    const auto key = std::make_pair(dev, reinterpret_cast<void *>(st));
    std::lock_guard<std::mutex> lock(g_sched_mu);
    auto it = g_stream_cus.find(key);
    if (it != g_stream_cus.end()) return it->second;
    uint32_t mask[32] = {0};
    int n = 0;
    if (hipExtStreamGetCUMask(st, 32, mask) == hipSuccess)
        for (int i = 0; i < 32; ++i) n += __builtin_popcount(mask[i]);
    if (n < 1 || n > device_cus) n = device_cus;
    g_stream_cus.emplace(key, n);
    return n;
}

struct LaunchGeom {
    int blocks_per_cu;
};
std::map<std::tuple<int, const void *, int>, int> g_occ_cache;   // (device, kernel, LDS bytes) -> workgroups per CU

// The forest kernel addresses its depth tile as LDS address 0 + byte offset (TileCtx): true as long as the kernel has no
// static LDS in front of the dynamic allocation.  Checked once per kernel; a build that breaks it fails loudly.
bool tile_at_lds_zero(const void *kernel)
{
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, kernel) == hipSuccess && fa.sharedSizeBytes == 0;
}

// A launch in two parts that share one tile queue (rdf_eval_forest_packed_split): the entry point leaves its wishes here for
// the launch_one at the end of its eval_common (thread-local: the plan in between is the ordinary one).
struct SplitSpec {
    hipStream_t helper_stream;
    int helper_cus;     // compute units the helper launch may count on (its grid: this many times the workgroups a CU holds)
    int tag;            // which of the stream's split queue slots (callers alternate it from step to step)
    int launched;       // (out) workgroups of the helper launch; 0: the launch was not split
};
thread_local SplitSpec *tl_split = nullptr;

template <int BLOCK, bool PACKED, int CMAX, bool STATS, int GROUP, bool COMPACT, bool DEEP = false>
int launch_one(const EvalArgs &a, int lds_bytes, int cus, hipStream_t st)
{
    auto kern = k_eval_forest<BLOCK, PACKED, CMAX, STATS, GROUP, COMPACT, 1, false, DEEP>;
    const void *kp = reinterpret_cast<const void *>(kern);
    int per_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RDF_ERR_NO_DEVICE;
    const auto key = std::make_tuple(dev, kp, lds_bytes);   // the function attribute below is per device
    {
        std::lock_guard<std::mutex> lock(g_sched_mu);
        auto it = g_occ_cache.find(key);
        if (it != g_occ_cache.end()) per_cu = it->second;
    }
    if (per_cu == 0) {
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return (int)e;
        }
        if (!tile_at_lds_zero(kp)) return RDF_ERR_BUILD;
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, BLOCK, (size_t)lds_bytes) != hipSuccess || n < 1)
            n = 1;
        per_cu = n;
        std::lock_guard<std::mutex> lock(g_sched_mu);
        g_occ_cache[key] = per_cu;
    }
    // persistent workgroups: as many as are resident at once, never more than there are tiles.  Mode 2 launches one
    // workgroup per tile instead (8 % slower alone: the node table is staged per tile) -- workgroups then retire all
    // the time, so a concurrent kernel on another stream (RCCL's) finds free slots at once.
    long long grid = sched_mode() == 2 ? (long long)a.n_tiles : (long long)cus * per_cu;
    if (grid > (long long)a.n_tiles) grid = a.n_tiles;
    if (grid < 1) grid = 1;
    EvalArgsN<1> ka;
    ka.l[0] = a;
    if (tl_split && a.sched && sched_mode() == 1 && (long long)a.n_tiles > grid) {
        // main launch: `grid` workgroups with a static first tile each; helper launch: workgroups that only pull from the
        // queues.  Either order of arrival is correct: the queues hand out the tiles beyond `grid` whoever asks, and the
        // workgroup that finishes last -- of both launches together -- puts the slot back to zero.
        long long hgrid = (long long)tl_split->helper_cus * per_cu;
        if (hgrid > (long long)a.n_tiles - grid) hgrid = (long long)a.n_tiles - grid;
        if (hgrid >= 1) {
            ka.l[0].q_static = (uint32_t)grid;
            ka.l[0].q_blocks = (uint32_t)(grid + hgrid);
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, st, ka);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return (int)e;
            ka.l[0].q_helper = 1;
            hipLaunchKernelGGL(kern, dim3((unsigned)hgrid), dim3(BLOCK), lds_bytes, tl_split->helper_stream, ka);
            tl_split->launched = (int)hgrid;
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, st, ka);
    return (int)hipGetLastError();
}

// NL forests (the layers of a small layered run) in one launch: 256 threads, runtime rows, four trees per lane, no
// pixel list -- the one geometry every small packed launch can take.
template <int CMAX, int NL, bool TW = false>
int launch_multi(const EvalArgsN<NL> &ka, int lds_bytes, int cus, hipStream_t st)
{
    constexpr int kBlock = TW ? 512 : 256;      // tree waves: 8 waves = 8 / T pixel rows x T trees
    auto kern = k_eval_forest<kBlock, true, CMAX, false, TW ? 1 : kGroup, false, NL, TW>;
    const void *kp = reinterpret_cast<const void *>(kern);
    int per_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RDF_ERR_NO_DEVICE;
    const auto key = std::make_tuple(dev, kp, lds_bytes);
    {
        std::lock_guard<std::mutex> lock(g_sched_mu);
        auto it = g_occ_cache.find(key);
        if (it != g_occ_cache.end()) per_cu = it->second;
    }
    if (per_cu == 0) {
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return (int)e;
        }
        if (!tile_at_lds_zero(kp)) return RDF_ERR_BUILD;
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kBlock, (size_t)lds_bytes) != hipSuccess || n < 1) n = 1;
        per_cu = n;
        std::lock_guard<std::mutex> lock(g_sched_mu);
        g_occ_cache[key] = per_cu;
    }
    // every role gets the same number of persistent workgroups (all roles have the same tiles: same frame, same reduce)
    long long per_role = (long long)cus * per_cu / NL;
    uint32_t most_tiles = 0;     // (tree waves: a role's tile holds 4 / T rows, so the roles' tile counts may differ)
    for (int l = 0; l < NL; ++l) most_tiles = ka.l[l].n_tiles > most_tiles ? ka.l[l].n_tiles : most_tiles;
    if (per_role > (long long)most_tiles) per_role = most_tiles;
    if (per_role < 1) per_role = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_role * NL)), dim3(kBlock), lds_bytes, st, ka);
    return (int)hipGetLastError();
}

// Which instantiations exist (65 kernels; every one costs half a second of compile time and ~20 KB of code object, so the list is
// what launches really take -- rows per wave, halo, LDS levels, scheduler are run-time arguments):
//   packed table, no pixel list:         256 / 512 threads x 4 / 8 classes in registers x 1 / 2 / 3 / 4 trees in a lane, 16 classes x 1 / 4 trees
//   packed table, filtered (pixel list): 256 / 512 threads x 4 / 8 / 16 classes x 4 trees in a lane
//   reference-layout forest:             256 / 512 threads x 4 / 16 classes x 1 or 4 trees in a lane (a filtered launch
//                                        takes the early-outs per lane instead of listing pixels)
//   visit counters (rdf_eval_forest_stats): reference layout, 256 threads, classes four at a time
//   visit and line counters of a packed launch (rdf_eval_forest_packed_stats): 256 / 512 threads, four classes, heap-order / deep
//   deep blocks (launch_deep):           256 / 512 threads x 4 / 8 classes x 2 / 3 / 4 trees in a lane + the pixel-list variant
//   layers of a stack / tree waves in one launch: launch_multi, 4 / 8 classes
// A lane walks GROUP trees interleaved: forests of one, two, three or six trees (and `rdf_eval_tree`) get kernels without the
// idle slots of the 4-wide one (measured: T = 1 costs 54 % of T = 4 with idle slots).  rdf_set_group overrides the choice by
// forest size (any width is correct for any forest: slots beyond the last tree idle).
int group_for(const EvalArgs &a, bool packed)
{
    const int knob = g_group;
    const int g = knob > 0 ? knob : (a.T == 1 ? 1 : a.T == 2 ? 2 : (a.T == 3 || a.T == 6) ? 3 : 4);
    return packed ? g : (g == 1 ? 1 : 4);
}

template <int BLOCK, bool PACKED, int CMAX>
int launch_group(bool compact, const EvalArgs &a, int lds_bytes, int cus, hipStream_t st)
{
    const int g = group_for(a, PACKED);
    if constexpr (PACKED) {
        if (compact) return launch_one<BLOCK, true, CMAX, false, 4, true>(a, lds_bytes, cus, st);
        if constexpr (CMAX <= 8) {      // (forests of more than eight classes: one or four trees in a lane only -- round 6 took the two- and
                                        // three-wide sixteen-class kernels out to pay for the narrow tiles' code: slots beyond the last tree idle)
            if (g == 2) return launch_one<BLOCK, true, CMAX, false, 2, false>(a, lds_bytes, cus, st);
            if (g == 3) return launch_one<BLOCK, true, CMAX, false, 3, false>(a, lds_bytes, cus, st);
        }
        if (g == 1) return launch_one<BLOCK, true, CMAX, false, 1, false>(a, lds_bytes, cus, st);
        return launch_one<BLOCK, true, CMAX, false, 4, false>(a, lds_bytes, cus, st);
    } else {
        return g == 1 ? launch_one<BLOCK, false, CMAX, false, 1, false>(a, lds_bytes, cus, st)
                      : launch_one<BLOCK, false, CMAX, false, 4, false>(a, lds_bytes, cus, st);
    }
}

template <int BLOCK>
int launch_block(bool packed, bool compact, const EvalArgs &a, int lds_bytes, int cus, hipStream_t st)
{
    if (packed) {
        if (a.C <= 4) return launch_group<BLOCK, true, 4>(compact, a, lds_bytes, cus, st);
        if (a.C <= 8) return launch_group<BLOCK, true, 8>(compact, a, lds_bytes, cus, st);
        return launch_group<BLOCK, true, 16>(compact, a, lds_bytes, cus, st);
    }
    if (a.C <= 4) return launch_group<BLOCK, false, 4>(false, a, lds_bytes, cus, st);
    return launch_group<BLOCK, false, 16>(false, a, lds_bytes, cus, st);
}

// Launches that walk their deep levels from the deep blocks (packed forests of up to eight classes): the same choices of
// workgroup size, classes in registers, trees per lane and pixel list as launch_block.
template <int BLOCK, int CMAX>
int launch_deep_c(bool compact, const EvalArgs &a, int lds_bytes, int cus, hipStream_t st)
{
    if (compact) return launch_one<BLOCK, true, CMAX, false, 4, true, true>(a, lds_bytes, cus, st);
    switch (group_for(a, true)) {
    case 1:         // (a single tree walks the two-wide kernel with one slot idle: forests that big have more trees)
    case 2: return launch_one<BLOCK, true, CMAX, false, 2, false, true>(a, lds_bytes, cus, st);
    case 3: return launch_one<BLOCK, true, CMAX, false, 3, false, true>(a, lds_bytes, cus, st);
    default: return launch_one<BLOCK, true, CMAX, false, 4, false, true>(a, lds_bytes, cus, st);
    }
}
template <int BLOCK>
int launch_deep(bool compact, const EvalArgs &a, int lds_bytes, int cus, hipStream_t st)
{
    return a.cpad == 4 ? launch_deep_c<BLOCK, 4>(compact, a, lds_bytes, cus, st) : launch_deep_c<BLOCK, 8>(compact, a, lds_bytes, cus, st);
}

// Where the deep blocks take over when nothing was chosen (no knob, no rdf_forest_set_deep_from / rdf_forest_tune for the table):
// NEVER -- the heap-order records, round 3's walk.  Round 4 switched forests of 32 MB of hot records and more to the blocks by
// size alone; that default was slower on two of the three kinds of forest measured (synth's "full" topology 5.9 against 3.9 ms,
// the trainer's own forest 0.83 against 0.70 ms: profiles/r04_trained_forest_probe.txt) and faster only where the deep levels
// are occupied (balanced topology: 7.8 against 11.2 ms).  Nothing in the records tells the kinds apart, so the blocks are taken
// only where a measurement on the caller's frames chose them: rdf_forest_tune, or rdf_forest_set_deep_from with a level found
// that way (a good first guess for an occupied forest: the deepest root level that still holds <= 512 KB of heap-order records).
constexpr int kUntunedDeepFrom = 0;

// What the host knows about a packed table, per (device, table): the per-forest choice of the deep-level table
// (rdf_forest_set_deep_from / rdf_forest_tune; level, 0 = never, -1 = none made) and what rdf_forest_pack found -- how many
// nodes need the exact numerators (then evaluations need the caller's forest) and the scale the table was packed for.
struct PackedState {
    int deep_from = -1;
    int exact_nodes = -1;      // -1: not read back yet
    float scale = 1.0f;
    const void *info_dev = nullptr;     // the table's info block in device memory, once known (the choice is written through to it)
    uint32_t generation = 0;   // PackInfo.generation as read back: every launch carries it, the kernel compares
    // the deep blocks' trailer (k_pack): 1 + the deepest level that holds a kFlagExact node, and the nodes of level D-1 that are
    // not plain two-leaf nodes -- the blocks serve a launch only from a root level >= deep_min_root and only if deep_bad_last == 0
    uint32_t deep_min_root = 0, deep_bad_last = 0;
};
std::map<std::pair<int, const void *>, PackedState> g_packed;

// The device a table lives on: the (device, address) keys must not depend on which device happens to be current when a
// finalizer or another thread calls rdf_forest_forget / rdf_forest_set_deep_from (falls back to the current device).
int device_of(const void *p)
{
    hipPointerAttribute_t at;
    if (p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice) return at.device;
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

int forest_deep_choice(const void *packed)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(g_sched_mu);
    const auto it = g_packed.find(std::make_pair(dev, packed));
    return it == g_packed.end() ? -1 : it->second.deep_from;
}

// ---- the stale flag: one word of pinned host memory per device that kernels can write (PackInfo.generation) ----
struct StaleFlag {
    volatile unsigned int *host = nullptr;
    unsigned int *dev = nullptr;
};
std::map<int, StaleFlag> g_stale;       // (under g_sched_mu)
// allocated when a device's first packed table is looked at (never during a stream capture: packed_info refuses those)
StaleFlag stale_flag_for(int dev, bool create)
{
    std::lock_guard<std::mutex> lock(g_sched_mu);
    auto it = g_stale.find(dev);
    if (it != g_stale.end()) return it->second;
    StaleFlag f;
    if (!create) return f;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && h) {
        memset(h, 0, 64);
        if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
            f.host = reinterpret_cast<volatile unsigned int *>(h);
            f.dev = reinterpret_cast<unsigned int *>(d);
        } else {
            (void)hipHostFree(h);
        }
    }
    (void)hipGetLastError();
    g_stale.emplace(dev, f);            // (a failed allocation is remembered too: launches then carry no flag)
    return f;
}
// true once after a kernel on this device found a table that was not the one the host remembered: everything the host knew
// about this device's tables is dropped (the flag does not say which), so the next evaluation of each reads its info block again
bool stale_flag_raised(int dev)
{
    std::lock_guard<std::mutex> lock(g_sched_mu);
    auto it = g_stale.find(dev);
    if (it == g_stale.end() || !it->second.host || *it->second.host == 0u) return false;
    *it->second.host = 0u;
    for (auto p = g_packed.begin(); p != g_packed.end();) p = p->first.first == dev ? g_packed.erase(p) : std::next(p);
    return true;
}

std::atomic<uint32_t> g_generation{0};
uint32_t next_generation()      // non-zero, unlike any other this process drew; 31 random bits set it apart from other processes
{
    uint32_t seed = g_generation.load(std::memory_order_relaxed);
    if (seed == 0u) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        uint32_t s = (uint32_t)ts.tv_nsec * 2654435761u ^ (uint32_t)ts.tv_sec * 40503u ^ ((uint32_t)getpid() << 16);
        s = (s | 1u) & 0x7FFFFFFFu;
        uint32_t expect = 0u;
        g_generation.compare_exchange_strong(expect, s);
    }
    uint32_t g = g_generation.fetch_add(2u, std::memory_order_relaxed) + 2u;   // (stays odd: never zero)
    return g;
}

static size_t info_offset(int n_trees, int max_depth, int n_classes);
static size_t deep_offset(int n_trees, int max_depth, int n_classes);
static size_t deep_bytes(int n_trees, int max_depth, int n_classes);

// The table's info block as the host knows it; read back from the device once (a synchronous 128-byte copy behind whatever the
// stream holds: rdf_forest_pack does it, and so does the first evaluation of a table this process did not pack at this
// address), together with the deep blocks' trailer.  Returns 0 and fills `out`, or an error code.
int packed_info(const void *packed, int n_trees, int max_depth, int n_classes, void *stream, PackedState *out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RDF_ERR_NO_DEVICE;
    const auto key = std::make_pair(dev, packed);
    {
        std::lock_guard<std::mutex> lock(g_sched_mu);
        const auto it = g_packed.find(key);
        if (it != g_packed.end() && it->second.exact_nodes >= 0) { *out = it->second; return RDF_OK; }
    }
    const void *info_dev = reinterpret_cast<const char *>(packed) + info_offset(n_trees, max_depth, n_classes);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cap) == hipSuccess && cap == hipStreamCaptureStatusActive)
        return RDF_ERR_CAPTURE;     // a table's first evaluation cannot be recorded into a graph: evaluate it once before capturing
    PackInfo host;
    uint32_t trailer[2] = {0u, 0u};
    hipError_t e = hipMemcpyAsync(&host, info_dev, sizeof(host), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream));
    const size_t db = deep_bytes(n_trees, max_depth, n_classes);
    if (e == hipSuccess && db != 0)
        e = hipMemcpyAsync(trailer, reinterpret_cast<const char *>(packed) + deep_offset(n_trees, max_depth, n_classes) + db - 128, sizeof(trailer),
                           hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return (int)e;
    // not a table this library version's rdf_forest_pack wrote, or one of another shape
    if (host.magic != kPackMagic || host.generation == 0u || host.n_trees != (uint32_t)n_trees || host.max_depth != (uint32_t)max_depth ||
        host.n_classes != (uint32_t)n_classes)
        return RDF_ERR_BAD_ARG;
    (void)stale_flag_for(dev, true);
    uint32_t write_word = 0u;
    {
        std::lock_guard<std::mutex> lock(g_sched_mu);
        PackedState &st = g_packed[key];
        st.exact_nodes = (int)(host.exact_nodes > 0x7FFFFFFFu ? 0x7FFFFFFFu : host.exact_nodes);
        st.scale = host.scale;
        st.info_dev = info_dev;
        st.generation = host.generation;
        st.deep_min_root = trailer[0];
        st.deep_bad_last = trailer[1];
        // the choice the table carries (made by whoever tuned it, in this process or another), unless this process made one since
        // (rdf_forest_set_deep_from on a table it had not looked at yet): that one goes into the table now
        if (st.deep_from < 0 && host.deep_choice != 0u && host.deep_choice <= 31u) st.deep_from = (int)host.deep_choice - 1;
        else if (st.deep_from >= 0 && host.deep_choice != (uint32_t)st.deep_from + 1u) write_word = (uint32_t)(st.deep_from > 30 ? 30 : st.deep_from) + 1u;
        *out = st;
    }
    if (write_word != 0u)       // (outside the lock: a synchronous copy must not hold up other threads' launches)
        (void)hipMemcpy(reinterpret_cast<char *>(const_cast<void *>(info_dev)) + offsetof(PackInfo, deep_choice), &write_word, sizeof(write_word),
                        hipMemcpyHostToDevice);
    return RDF_OK;
}

// The per-table choice in the host's map only (rdf_forest_tune's candidates), and written through to the table's info block
// (rdf_forest_set_deep_from: a synchronous 4-byte copy -- choosing is a load-time action).
int set_deep_choice(const void *packed, int level, bool write_through)
{
    if (!packed) return RDF_ERR_NULL_PTR;
    const int dev = device_of(packed);
    if (dev < 0) return RDF_ERR_NO_DEVICE;
    const void *info_dev = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_sched_mu);
        PackedState &st = g_packed[std::make_pair(dev, packed)];
        st.deep_from = level < 0 ? -1 : level;
        info_dev = st.info_dev;
    }
    if (write_through && info_dev) {
        const uint32_t word = level < 0 ? 0u : (uint32_t)(level > 30 ? 30 : level) + 1u;
        const hipError_t e = hipMemcpy(reinterpret_cast<char *>(const_cast<void *>(info_dev)) + offsetof(PackInfo, deep_choice), &word,
                                       sizeof(word), hipMemcpyHostToDevice);
        if (e != hipSuccess) return (int)e;
    }
    return RDF_OK;
}

int check_common(const void *depth, int n_img, int dim_x, int dim_y, const void *forest, int n_trees,
                 int max_depth, int n_classes, const void *labels, int r)
{
    if (n_img < 0 || dim_x < 0 || dim_y < 0 || n_trees < 0 || max_depth < 0 || max_depth > 30 ||
        n_classes < 0 || r < 1)
        return RDF_ERR_BAD_ARG;
    if ((long long)n_img * dim_x * dim_y >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    // (a far probe's byte offset is one 24-bit multiply-add, TileCtx: y * 2W + 2x with y < 2^24 and 2W < 2^24)
    if (dim_x >= (1 << 23) || dim_y >= (1 << 24)) return RDF_ERR_TOO_LARGE;
    const long long total = (long long)n_img * (dim_x / r) * (dim_y / r);
    if (total == 0) return 1; // nothing to do
    if (!depth || !labels) return RDF_ERR_NULL_PTR;
    if (!forest && n_trees > 0 && max_depth > 0) return RDF_ERR_NULL_PTR;
    return 0;
}

static int classes_padded(int n_classes) { return (n_classes + 3) & ~3; }

// the last-level table (LastLevelRec) behind the three per-slot tables, and the 64 bytes that hold its count of unusable
// records: forests of one to four classes and two or more levels
static size_t last_level_bytes(int n_trees, int max_depth, int n_classes)
{
    if (classes_padded(n_classes) != 4 || max_depth < 2 || n_trees < 1) return 0;
    return ((size_t)n_trees << (max_depth - 1)) * sizeof(LastLevelRec) + 64;
}

// the deep blocks behind the last-level table (128-byte aligned), one all-zero line and a 128-byte trailer
static size_t slot_tables_bytes(int n_trees, int max_depth, int n_classes)      // hot records + PDF rows
{
    return ((size_t)n_trees << max_depth) * (sizeof(NodeRec16) + 2 * (size_t)classes_padded(n_classes) * sizeof(float));
}
static size_t deep_offset(int n_trees, int max_depth, int n_classes)
{
    const size_t before = slot_tables_bytes(n_trees, max_depth, n_classes) + last_level_bytes(n_trees, max_depth, n_classes);
    return (before + 127u) & ~(size_t)127u;
}
static size_t deep_bytes(int n_trees, int max_depth, int n_classes)
{
    if (!deep_possible(n_trees, max_depth, classes_padded(n_classes))) return 0;
    return (deep_total_lines(n_trees, max_depth, classes_padded(n_classes)) + 2u) * 128u;
}
// the info block (PackInfo) is the table's last 128 bytes
static size_t info_offset(int n_trees, int max_depth, int n_classes)
{
    return deep_offset(n_trees, max_depth, n_classes) + deep_bytes(n_trees, max_depth, n_classes);
}


Knob g_fold{-1};                                // -1/1: narrow tiles for a label map's last <= 32 columns (EvalArgs.fold); 0: never
Knob g_halo{-1};
Knob g_lds_levels{-1};
Knob g_tree_waves{-1};
Knob g_stage_vec{-1};
Knob g_rows_per_wave{0};
Knob g_force_exact{0};
Knob g_deep_from{-1};                           // -1: the table's own choice (eval_common); 0: never; > 0: from this level on (rounded up to a block root)
Knob g_last_level_table{-1};                    // -1: level D-1 from the last-level table when it is usable and the forest uses half of that level; 1: whenever usable; 0: never

// What eval_common works out before it launches: the kernel arguments (without a queue slot), the dynamic LDS and the
// geometry.  layered_run asks for the plan only (plan_only) to put several layers into one launch.
struct Plan {
    EvalArgs a;
    int lds_bytes = 0, block = 0;
    bool big = false, empty = false, tw = false, deep = false;
};

int eval_common(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                const float *forest, int n_trees, int max_depth, int n_classes, const uint16_t *filter,
                int filter_class, uint16_t *labels_out, int r, float s, int keep_if_no_leaf,
                unsigned long long *stats, void *stream, int fill_untouched = 0, Plan *plan_only = nullptr, bool allow_tw = true)
{
    const unsigned long long t_entry = now_ns();
    // (a packed table stands in for the forest: the forest itself is only needed for nodes that take the exact numerators, below)
    int rc = check_common(depth, n_img, dim_x, dim_y, forest ? (const void *)forest : packed, n_trees, max_depth, n_classes, labels_out, r);
    if (rc == 1) {
        if (plan_only) plan_only->empty = true;
        return RDF_OK;
    }
    if (rc != 0) return rc;
    if (filter_class != -1 && !filter) return RDF_ERR_NULL_PTR;

    DeviceInfo di;
    rc = device_info(&di);
    if (rc != 0) return rc;

    EvalArgs a;
    PackedState ps;
    memset(&a, 0, sizeof(a));
    a.depth = depth; a.forest = forest; a.filter = filter; a.labels = labels_out;
    a.stats = stats;
    a.W = dim_x; a.H = dim_y; a.Wl = dim_x / r; a.Hl = dim_y / r; a.r = r;
    a.per_img_l = (uint32_t)a.Wl * (uint32_t)a.Hl;
    a.per_img_d = (uint32_t)dim_x * (uint32_t)dim_y;
    a.T = n_trees; a.D = max_depth; a.C = n_classes; a.E = 7 + 2 * n_classes;
    a.nodes = (int)((1ll << max_depth) - 1);
    a.filter_class = filter_class;
    a.keep_if_no_leaf = keep_if_no_leaf;
    a.fill_untouched = fill_untouched;
    a.s = s;
    if (packed) {
        const size_t slots = (size_t)n_trees << max_depth;   // 1-based heap slots, 2^D per tree
        a.packed16 = reinterpret_cast<const NodeRec16 *>(packed);
        a.packed_pdf = reinterpret_cast<const float *>(a.packed16 + slots);
        a.cpad = (n_classes + 3) & ~3;
        // nodes that need the exact numerators read them from the caller's forest, with the scale the table was packed for
        // a kernel of an earlier call found a table that was not the one the host remembered at its address (a copy over a
        // known address without rdf_forest_forget): what the host knew is gone, this call says so, the next one reads afresh
        if (stale_flag_raised(di.dev)) return RDF_ERR_STALE;
        rc = packed_info(packed, n_trees, max_depth, n_classes, stream, &ps);
        if (rc != RDF_OK) return rc;
        if (ps.exact_nodes > 0 && !forest) return RDF_ERR_NULL_PTR;
        a.s = ps.scale;
        a.info = reinterpret_cast<const PackInfo *>(ps.info_dev);
        a.expect_gen = ps.generation;
        a.stale_flag = stale_flag_for(di.dev, false).dev;
    }

    a.tiles_x = ((uint32_t)a.Wl + 63u) / 64u;
    // (round 6) a label map whose width leaves at most 32 columns beyond a multiple of 64 (848 = 13 x 64 + 16: every frame of the
    // metric) gets NARROW tiles for those columns -- a wave covers 2 rows x 32 or 4 rows x 16 pixels there -- instead of a tile
    // column whose waves run three quarters empty: 5.4 % fewer wave-rows on an 848-wide frame (EvalArgs.fold)
    const int fold_knob = g_fold;
    const int want_fold = fold_knob >= 0 ? fold_knob : env_int("RDF_FOLD", 1);
    {
        const int w_rem = a.Wl % 64;
        if (want_fold && a.Wl > 64 && w_rem != 0 && w_rem <= 32) {
            a.fold = w_rem <= 16 ? 4u : 2u;
            a.tiles_x = (uint32_t)a.Wl / 64u;
        }
    }
    // Throughput shape: the launch fills the chip with four-row waves.  Small launches (a single live frame) take fewer
    // rows per wave so that every CU still gets several waves.
    const long long waves_wanted = 24ll * di.cus;
    const bool big = (long long)n_img * a.tiles_x * ((a.Hl + kMaxRowsPerWave - 1) / kMaxRowsPerWave) >= waves_wanted;
    // the empty-tile check (see the kernel): always on launches that fill the chip; on small launches only at labels_reduce > 1,
    // where a staged tile covers r x r times the image per label pixel and a live frame is mostly background (config 3's
    // two-layer frame 30.5 -> 29.2 us; at labels_reduce 1 the extra round trip costs more than the skipped staging saves:
    // one live 848x480 frame 41.6 -> 43.5 us, a dense one 76 -> 77 us, tools/latency.py with RDF_CHECK_EMPTY_SMALL=1)
    a.check_empty = (big || r > 1) ? 1 : env_int("RDF_CHECK_EMPTY_SMALL", 0);
    // Workgroup size.  Big unfiltered launches: 512 threads, three workgroups per CU = 24 waves with a 48-pixel halo
    // (54 KB of LDS each) instead of five 256-thread workgroups = 20 waves with 32 pixels: 4.77 vs 5.14 ms on the bench
    // batch, 11.45 vs 12.60 ms on config 5's shard (profiles/r02_sweep_512.txt).  Small launches keep 256 threads.
    const int block_knob = g_block_threads;
    int block = block_knob > 0 ? block_knob : env_int("RDF_BLOCK", 0);
    // (filtered launches at labels_reduce 1 gain nothing from 512 threads: 1.36 ms either way, 1.63 with this default's
    // halo; at labels_reduce 2 they do: 0.46 vs 0.55 ms)
    const bool compaction = g_compaction != 0;
    const bool filtered_r1 = packed && filter_class != -1 && compaction && r == 1;
    if (block != 256 && block != 512) block = (big && !filtered_r1) ? 512 : 256;   // (other requests: the default)
    if (stats && !packed) block = 256;      // (the packed visit-counter kernels take the geometry of the launch they describe)
    // deep blocks (k_eval_forest<..., DEEP>): from which root level on.  The knob or RDF_DEEP_FROM names a level (rounded up
    // to a block root below, and never inside the levels LDS holds); nothing chosen: kUntunedDeepFrom.  (Tree-wave launches
    // never walk them: see below.)
    int deep_from_wanted = 0;
    if (packed && deep_bytes(n_trees, max_depth, n_classes) != 0) {
        const int knob = g_deep_from;
        deep_from_wanted = knob >= 0 ? knob : env_int("RDF_DEEP_FROM", -1);      // process-wide knob first,
        if (deep_from_wanted < 0) deep_from_wanted = forest_deep_choice(packed);   // then what was chosen for this packed forest,
        if (deep_from_wanted < 0) deep_from_wanted = kUntunedDeepFrom;             // else the heap-order records
        const char *dp = reinterpret_cast<const char *>(packed) + deep_offset(n_trees, max_depth, n_classes);
        if ((reinterpret_cast<uintptr_t>(dp) & 127u) != 0) deep_from_wanted = 0;
        // what the kernel would find in the blocks' trailer, known to the host since the table was first seen: blocks that cannot
        // serve the forest (a level D-1 node that is not a plain two-leaf node; exact nodes down to the last block) are not planned
        // for at all -- the launch takes the ordinary geometry instead of the deep one with heap-order records in it
        if (deep_from_wanted > 0) {
            const int R0 = max_depth - deep_last_levels((n_classes + 3) & ~3);
            if (ps.deep_bad_last != 0u || (int)ps.deep_min_root > R0) deep_from_wanted = 0;
            else if (deep_from_wanted < (int)ps.deep_min_root) deep_from_wanted = (int)ps.deep_min_root;      // (rounded up to a root below)
        }
    }
    const int rpw_knob = g_rows_per_wave;
    int rpw = rpw_knob > 0 ? rpw_knob : env_int("RDF_ROWS_PER_WAVE", 0);
    if (rpw < 1 || rpw > kMaxRowsPerWave) {
        rpw = kMaxRowsPerWave;
        if (big && block == 512) {
            // 8 waves x 2 rows: the 64 x 16 tile of four waves x 4 rows -- when that gives every workgroup slot (three per
            // CU) ten tiles or more; a smaller batch takes 8-row tiles, so that its last round is not half empty
            // (848x480 frames: 4 in a launch 0.237 -> 0.212 ms, 8: 0.378 -> 0.362, 16: 0.696 -> 0.677; 24: 0.937 vs 0.952)
            const long long tiles2 = (long long)n_img * a.tiles_x * ((a.Hl + 15) / 16);
            rpw = tiles2 >= 30ll * di.cus ? 2 : 1;
            // (deep blocks, eight trees and more: 8-row tiles leave the 15 KB beside the slabs to a 24-pixel halo -- config 5's
            // shard on a balanced forest 18.5 -> 18.0 ms, profiles/r05_deep_coop.txt section 7)
            if (deep_from_wanted > 0 && n_trees >= 8) rpw = 1;
        }
        while (rpw > 1 && (long long)n_img * a.tiles_x * ((a.Hl + rpw - 1) / rpw) < waves_wanted) rpw >>= 1;
    }
    // Tree waves (k_eval_forest<..., TW>): a small unfiltered packed launch of a forest of 2-4 trees gives every tree
    // of a pixel row a wave of its own -- 512 threads = 8 waves = 8 / T rows of 64 pixels x T trees (two rows for three or
    // four trees: half the tile of four waves x four trees in a lane, on twice the CUs).  One live 848x480 frame at
    // labels_reduce 2 (T4/D20): 31 -> 19 us; a dense one 37 -> 31 us (with 1 024 threads = four rows x four trees: 20 and
    // 35 us; with 256 = one row: 32 and 44 us, more tiles than workgroup slots).  Only for label maps of up to 128 K pixels:
    // a tile now needs four times the wave slots, and a dense 848x480 frame at full resolution got slower (82 -> 95 us with
    // the 1 024-thread shape; live: 48 -> 42).  The layers of a stack evaluated in one launch (rdf_layered_run) take the
    // same shape when every layer can.
    const int tw_knob = g_tree_waves;
    const int want_tw = tw_knob >= 0 ? tw_knob : env_int("RDF_TREE_WAVES", 1);
    const bool tw = allow_tw && want_tw != 0 && !big && !stats && packed && filter_class == -1 && block == 256 && n_trees >= 2 && n_trees <= 4 && n_classes <= 8 &&
                    max_depth >= 1 && sched_mode() != 2 && (long long)n_img * a.Wl * a.Hl <= 131072;
    if (tw) { rpw = 1; block = 512; }
    a.rows_per_wave = rpw;
    const uint32_t tile_rows = tw ? (uint32_t)(8 / n_trees) : (uint32_t)(block / 64) * (uint32_t)rpw;
    a.tiles_y = ((uint32_t)a.Hl + tile_rows - 1u) / tile_rows;
    a.tiles_y_n = a.fold ? ((uint32_t)a.Hl + tile_rows * a.fold - 1u) / (tile_rows * a.fold) : 0u;
    const long long n_tiles = (long long)n_img * ((long long)a.tiles_x * a.tiles_y + a.tiles_y_n);
    if (n_tiles >= (1ll << 31)) return RDF_ERR_TOO_LARGE;
    a.n_tiles = (uint32_t)n_tiles;

    if (tw) deep_from_wanted = 0;
    // A launch that walks deep blocks fetches them by the WAVE, through a slab of 8 KB per wave in LDS (k_eval_forest<..., DEEP>): the
    // slabs take 64 KB of a 512-thread workgroup (32 KB of a 256-thread one), so such a launch keeps a smaller depth tile and
    // fewer levels in LDS and runs two (four) workgroups per CU instead of three (five).
    const bool coop = deep_from_wanted > 0;
    const long long slab_bytes = coop ? (long long)(block / 64) * 8192 : 0;

    // ---- LDS plan: [depth tile: th*twp*2 B, at address 0][node table: T*2^K*16 B][queue mailbox 32 B][pixel list][slabs] ----
    // (whatever the budget knob says: a workgroup's tile, node table and slabs together fit a CU's 160 KB)
    const long long budget = coop ? std::min<long long>(coop_lds_budget(block), 163840 - slab_bytes - 2048) : lds_budget(tw ? 256 : block);
    // filtered launches of the default geometry carry the pixel list in LDS (k_eval_forest<..., COMPACT>)
    const bool compact_launch = packed && !stats && filter_class != -1 && compaction;
    // Halo and the levels that must stay in LDS, by measurement (profiles/r02_sweep_*.txt).  256-thread workgroups (32.7 KB):
    // four trees 7 levels + 32 px (5.17 ms; 8 + 24: 5.26, 6 + 40: 5.61), eight trees 5 levels + 40 px (config 5's shape:
    // 12.68 ms; 6 + 32: 12.90).  512-thread workgroups (54.6 KB): 56 px with 7 levels (4.74 ms; 8 + 48: 4.76) or, for eight
    // trees, 5 levels (11.45 ms; 6 + 48: 11.66).  With labels_reduce > 1 the halo shrinks until the tile fits.
    const bool many_trees = n_trees >= 8;
    // (labels_reduce 2, 64 frames, 512 threads: 6 levels + 32 px 0.81 ms, 7 + 40 0.89 ms -- a tile spans r times the pixels)
    const int halo_default = coop ? (block == 512 ? (many_trees ? 24 : 16) : 8)
                                  : block == 512 ? (r > 1 ? 32 : 56) : (many_trees ? kDefaultHalo + 8 : kDefaultHalo);
    const int halo_knob = g_halo;
    int halo = halo_knob >= 0 ? halo_knob : env_int("RDF_HALO", halo_default);
    long long tile_bytes = 0;
    // The staged tile may take half the budget.  With labels_reduce > 1 a tile spans r times the pixels per label, so
    // the halo shrinks (by twos) until the tile fits -- a narrow tile still beats none: 64 frames at r = 2 take 1.08 ms
    // with an 8-pixel halo against 1.49 ms with every probe going to global memory; only when not even the bare
    // centres fit is the tile dropped.
    // (Throughput shape only: for a single small frame staging a narrow tile cost more than it saved, 91 vs 87 us.)
    const int h_min = big ? 0 : halo;
    // pixel list: one uint16 per tile pixel, then one uint32 per tile row
    // (tree waves use the list region for the words the trees of a row exchange: tile_rows x T x 64)
    const long long list_bytes = tw ? (long long)tile_rows * n_trees * 256 :
                                 compact_launch ? (((long long)block * rpw * 2 + (long long)tile_rows * 4) + 15) & ~15ll : 0;
    // levels of the forest the caller pinned into LDS (rdf_set_lds_levels): the tile then gets all that is left of the
    // budget instead of half of it
    const int levels_knob = g_lds_levels;
    int k_forced = levels_knob >= 0 ? levels_knob : env_int("RDF_LDS_LEVELS", -1);
    if (k_forced > max_depth) k_forced = max_depth;
    while (k_forced > 0 && (long long)n_trees * (1ll << k_forced) * 16 + 32 + list_bytes > budget) --k_forced;
    // otherwise the tile may take what kMinLdsLevels levels of the forest leave: level for level, a level moved from LDS
    // to L1-resident global records costs less than the far probes a wider halo saves (measured: T4, 8 levels + 24 px
    // 5.26 ms, 7 levels + 32 px 5.17 ms, 6 levels + 40 px 5.6 ms on the bench batch)
    const int k_min = coop ? (many_trees ? kMinLdsLevels - 2 : kMinLdsLevels - 1)       // (4 and 5 levels)
                           : many_trees ? kMinLdsLevels - 1 : (block == 512 && r == 1 ? kMinLdsLevels + 1 : kMinLdsLevels);
    const int k_floor = k_forced >= 0 ? k_forced : (max_depth < k_min ? max_depth : k_min);
    const long long tile_budget = budget - 32 - list_bytes - (k_floor > 0 ? (long long)n_trees * (1ll << k_floor) * 16 : 0);
    // 16-byte staging needs every row start 16-byte aligned in the image and in LDS
    const int vec_knob = g_stage_vec;
    const int want_vec = vec_knob >= 0 ? vec_knob : env_int("RDF_STAGE_VEC", 1);
    const bool vec_ok = want_vec && dim_x % 8 == 0 && ((64 * r) % 8 == 0) && (reinterpret_cast<uintptr_t>(depth) & 15u) == 0;
    for (int h = halo; h >= h_min && h >= 0; h -= 2) {
        const bool vec = vec_ok && h % 8 == 0;
        long long tw = 63ll * r + 1 + 2ll * h;
        const long long th = ((long long)tile_rows - 1) * r + 1 + 2ll * h;
        if (vec) tw = (tw + 7) & ~7ll;                      // a few more columns than the probes' reach: harmless
        long long twp = vec ? tw : (tw + 1) & ~1ll;
        if (vec && (twp * 2) % 512 == 0) twp += 8;          // not a whole number of LDS bank sweeps per row
        long long bytes = (th * twp * 2 + 15) & ~15ll;
        // (narrow tiles: 64 / fold columns x fold x tile_rows rows of centres, the same halo; the allocation holds either shape)
        long long tw_n = 0, th_n = 0, twp_n = 0, hx_n = h;
        if (a.fold) {
            // A wave of a narrow tile reads `fold` rows of the staged tile at once: with a row pitch of a whole number of LDS bank
            // sweeps (256 bytes: 16 columns + 2 x 56 of halo) the rows fall into the SAME banks -- a four-way conflict on every probe
            // (measured: the narrow tiles gave 2 % where their 5 % fewer wave-rows promised more).  The pitch modulo 256 bytes must
            // leave room for one row segment on either side: padding when it fits the budget, else a few columns less halo in x.
            th_n = ((long long)tile_rows * a.fold - 1) * r + 1 + 2ll * h;
            const long long seg = (64ll / a.fold) * 2 * r;          // bytes one row of a wave spans
            auto shape = [&](long long hx, long long pad) {
                tw_n = (64ll / a.fold - 1) * r + 1 + 2ll * hx;
                if (vec) tw_n = (tw_n + 7) & ~7ll;
                twp_n = (vec ? tw_n : (tw_n + 1) & ~1ll) + pad;
                const long long m = (twp_n * 2) % 256;
                return seg > 128 || (m >= seg && m <= 256 - seg);
            };
            bool ok_banks = shape(h, 0);
            for (long long pad = 8; !ok_banks && pad <= 64; pad += 8)
                ok_banks = shape(h, pad) && ((th_n * twp_n * 2 + 15) & ~15ll) <= std::max(bytes, (long long)0) ;   // (padding that costs no halo)
            for (long long hx = h - (vec ? 8 : 2); !ok_banks && hx >= 0 && hx >= h - 24; hx -= (vec ? 8 : 2)) {
                ok_banks = shape(hx, 0);
                if (ok_banks) hx_n = hx;
            }
            if (!ok_banks) { hx_n = h; (void)shape(h, 0); }
            bytes = std::max(bytes, (th_n * twp_n * 2 + 15) & ~15ll);
        }
        if (bytes <= tile_budget) {
            tile_bytes = bytes;
            a.halo = h; a.tw = (int)tw; a.th = (int)th; a.twp = (int)twp;
            a.tw_n = (int)tw_n; a.th_n = (int)th_n; a.twp_n = (int)twp_n; a.halo_xn = (int)hx_n;
            if (vec) {
                a.stage_tw8 = (uint32_t)(tw >> 3);
                a.stage_magic = (uint32_t)((1ull << 32) / a.stage_tw8) + 1u;
                if (a.fold) {
                    a.stage_tw8_n = (uint32_t)(tw_n >> 3);
                    a.stage_magic_n = (uint32_t)((1ull << 32) / a.stage_tw8_n) + 1u;
                }
            }
            break;
        }
    }
    int K = 0;
    while (K < max_depth && (k_forced < 0 || K < k_forced) &&
           (long long)n_trees * (1ll << (K + 1)) * 16 + tile_bytes + 32 + list_bytes <= budget) ++K;
    a.lds_levels = K;
    // level D-1 from the last-level table (LastLevelRec) when the forest was packed with one and that level is not in LDS
    const int llt_knob = g_last_level_table;
    const int want_llt = llt_knob >= 0 ? llt_knob : env_int("RDF_LAST_LEVEL_TABLE", 1);
    if (packed && want_llt != 0 && !tw && max_depth - 1 >= K && last_level_bytes(n_trees, max_depth, n_classes) != 0) {
        a.last_level = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.packed_pdf) +
                                                       ((size_t)n_trees << max_depth) * 2 * (size_t)a.cpad * sizeof(float));
        // from which share of level D-1 in use (see k_pack) on: by default half; knob 1 = whenever the table is usable
        const int pct = llt_knob == 1 ? 0 : env_int("RDF_LAST_LEVEL_MIN_PCT", 50);
        a.last_level_min = (uint32_t)((((unsigned long long)n_trees << (max_depth - 1)) * (unsigned long long)(pct < 0 ? 0 : pct > 100 ? 100 : pct)) / 100u);
    }
    bool deep_launch = false;
    if (deep_from_wanted > 0) {
        const int R0 = max_depth - deep_last_levels(a.cpad);
        int from = deep_from_wanted;
        if (from < K) from = K;
        if (from > R0) from = R0;
        from = R0 - 3 * ((R0 - from) / 3);          // a root level: R0 - 3 i, rounded up
        a.deep = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(packed) + deep_offset(n_trees, max_depth, n_classes));
        a.deep_from = from;
        deep_launch = true;
    }
    const long long node_bytes = K > 0 ? (long long)n_trees * (1ll << K) * 16 : 0;
    a.lds_nodes_off = (uint32_t)tile_bytes;
    a.lds_mail_off = (uint32_t)(tile_bytes + node_bytes);
    a.lds_list_off = (uint32_t)(tile_bytes + node_bytes + 32);
    a.lds_xchg_off = a.lds_list_off;
    // (the slabs start on a 1-KB boundary: an LDS-DMA piece is 1 KB)
    const long long slab_at = (tile_bytes + node_bytes + 32 + list_bytes + 1023) & ~1023ll;
    a.lds_slab_off = (uint32_t)slab_at;
    const int lds_bytes = coop ? (int)(slab_at + slab_bytes) : (int)(node_bytes + tile_bytes + 32 + list_bytes);
    if (plan_only) {
        plan_only->a = a;
        plan_only->lds_bytes = lds_bytes;
        plan_only->block = block;
        plan_only->big = big;
        plan_only->tw = tw;
        plan_only->deep = deep_launch;
        return RDF_OK;
    }

    a.sched = tl_split ? sched_slot(stream, 16 + (tl_split->tag & 3)) : sched_slot(stream);

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int cus = usable_cus(st, di.cus);   // a CU-masked stream holds fewer persistent workgroups
    const unsigned long long t_launch = now_ns();
    if (tw) {
        EvalArgsN<1> ka;
        ka.l[0] = a;
        rc = a.C <= 4 ? launch_multi<4, 1, true>(ka, lds_bytes, cus, st) : launch_multi<8, 1, true>(ka, lds_bytes, cus, st);
    } else if (stats && packed) {
        // the launch rdf_eval_forest_packed would make, with the visit and line counters on (four classes, no pixel list)
        if (n_classes > 4 || compact_launch) return RDF_ERR_BAD_ARG;
        a.stats_wide = 1;
        if (deep_launch)
            rc = block == 512 ? launch_one<512, true, 4, true, 4, false, true>(a, lds_bytes, cus, st)
                              : launch_one<256, true, 4, true, 4, false, true>(a, lds_bytes, cus, st);
        else
            rc = block == 512 ? launch_one<512, true, 4, true, 4, false, false>(a, lds_bytes, cus, st)
                              : launch_one<256, true, 4, true, 4, false, false>(a, lds_bytes, cus, st);
    } else if (stats) {   // (reference layout, 256 threads: eval_common chose both)
        rc = launch_one<256, false, 4, true, 4, false>(a, lds_bytes, cus, st);
    } else if (deep_launch) {
        rc = block == 512 ? launch_deep<512>(compact_launch, a, lds_bytes, cus, st) : launch_deep<256>(compact_launch, a, lds_bytes, cus, st);
    } else {
        rc = block == 512 ? launch_block<512>(packed != nullptr, compact_launch, a, lds_bytes, cus, st)
                          : launch_block<256>(packed != nullptr, compact_launch, a, lds_bytes, cus, st);
    }
    const unsigned long long t_done = now_ns();
    g_host_ns_plan.fetch_add(t_launch - t_entry, std::memory_order_relaxed);
    g_host_ns_launch.fetch_add(t_done - t_launch, std::memory_order_relaxed);
    g_host_calls.fetch_add(1, std::memory_order_relaxed);
    return rc;
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

size_t rdf_forest_packed_bytes(int n_trees, int max_depth, int n_classes)
{
    if (n_trees < 0 || max_depth < 0 || max_depth > 30 || n_classes < 0) return 0;
    if (((size_t)n_trees << max_depth) == 0 || max_depth == 0) return 0;
    return info_offset(n_trees, max_depth, n_classes) + sizeof(PackInfo);
}

int rdf_forest_pack(const float *forest, int n_trees, int max_depth, int n_classes, float scale_factor,
                    void *packed, void *stream)
{
    if (n_trees < 0 || max_depth < 0 || max_depth > 30 || n_classes < 0) return RDF_ERR_BAD_ARG;
    if (max_depth > 27) return RDF_ERR_BAD_ARG;
    const size_t total = (size_t)n_trees << max_depth;   // slots
    if (total == 0 || max_depth == 0) return RDF_OK;
    if (!forest || !packed) return RDF_ERR_NULL_PTR;
    const size_t blocks = (total + 255) / 256;
    if (blocks >= (1ull << 31)) return RDF_ERR_TOO_LARGE;
    if ((reinterpret_cast<uintptr_t>(packed) & 127u) != 0) return RDF_ERR_BAD_ARG;     // `packed` must be 128-byte aligned
    char *tail = reinterpret_cast<char *>(packed) + slot_tables_bytes(n_trees, max_depth, n_classes);
    PackInfo *info = reinterpret_cast<PackInfo *>(reinterpret_cast<char *>(packed) + info_offset(n_trees, max_depth, n_classes));
    {
        const hipError_t e = hipMemsetAsync(info, 0, sizeof(PackInfo), reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    LastLevelRec *last_level = nullptr;
    unsigned int *unusable = nullptr;
    if (last_level_bytes(n_trees, max_depth, n_classes) != 0) {
        last_level = reinterpret_cast<LastLevelRec *>(tail);
        unusable = reinterpret_cast<unsigned int *>(last_level + ((size_t)n_trees << (max_depth - 1)));
        const hipError_t e = hipMemsetAsync(unusable, 0, 64, reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    {   // a re-packed table is another forest: what the host knew of the old one goes
        const int dev = device_of(packed);
        if (dev >= 0) {
            std::lock_guard<std::mutex> lock(g_sched_mu);
            g_packed.erase(std::make_pair(dev, (const void *)packed));
        }
    }
    uint4 *deep = nullptr;
    if (deep_bytes(n_trees, max_depth, n_classes) != 0) {
        deep = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(packed) + deep_offset(n_trees, max_depth, n_classes));
        if ((reinterpret_cast<uintptr_t>(deep) & 127u) != 0) return RDF_ERR_BAD_ARG;     // `packed` must be 128-byte aligned
        const hipError_t e = hipMemsetAsync(deep + ((deep_total_lines(n_trees, max_depth, classes_padded(n_classes)) + 1u) << 3), 0, 128,
                                            reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_pack, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       forest, reinterpret_cast<NodeRec16 *>(packed), info,
                       reinterpret_cast<float *>(reinterpret_cast<char *>(packed) + total * sizeof(NodeRec16)),
                       last_level, unusable, deep, n_trees, n_classes, classes_padded(n_classes), total,
                       max_depth, 7 + 2 * n_classes, scale_factor, (int)g_force_exact, next_generation());
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return (int)le;
    // what the kernel found (nodes that need the exact numerators) comes back now: packing is load time, and evaluations -- also
    // those recorded into a hipGraph -- then never have to ask the device
    PackedState ps;
    return packed_info(packed, n_trees, max_depth, n_classes, stream, &ps);
}

int rdf_eval_forest_packed(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                           const float *forest, int n_trees, int max_depth, int n_classes,
                           const uint16_t *filter, int filter_class, uint16_t *labels_out,
                           int labels_reduce, void *stream)
{
    if (!packed && n_trees > 0 && max_depth > 0) return RDF_ERR_NULL_PTR;
    if (max_depth > 27) return RDF_ERR_BAD_ARG; // packed records are addressed with 32-bit byte offsets
    if (!packed) // degenerate forest: nothing to walk, the unpacked path handles it
        return eval_common(depth, n_img, dim_x, dim_y, nullptr, forest, n_trees, max_depth, n_classes, filter,
                           filter_class, labels_out, labels_reduce, 1.0f, 0, nullptr, stream);
    return eval_common(depth, n_img, dim_x, dim_y, packed, forest, n_trees,
                       max_depth, n_classes, filter, filter_class, labels_out, labels_reduce, 1.0f, 0, nullptr,
                       stream);
}

int rdf_forest_set_deep_from(const void *packed, int level)
{
    return set_deep_choice(packed, level, true);
}

int rdf_forest_info(const void *packed, int n_trees, int max_depth, int n_classes, void *stream, int *deep_from, int *exact_nodes,
                    float *scale)
{
    if (!packed) return RDF_ERR_NULL_PTR;
    if (n_trees < 1 || max_depth < 1 || max_depth > 27 || n_classes < 0) return RDF_ERR_BAD_ARG;
    PackedState ps;
    const int rc = packed_info(packed, n_trees, max_depth, n_classes, stream, &ps);
    if (rc != RDF_OK) return rc;
    if (deep_from) *deep_from = ps.deep_from;
    if (exact_nodes) *exact_nodes = ps.exact_nodes;
    if (scale) *scale = ps.scale;
    return RDF_OK;
}

int rdf_forest_forget(const void *packed)
{
    if (!packed) return RDF_ERR_NULL_PTR;
    // (the table's own device, not the current one: finalizers call this from wherever they run)
    const int dev = device_of(packed);
    if (dev < 0) return RDF_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_sched_mu);
    g_packed.erase(std::make_pair(dev, packed));
    return RDF_OK;
}

int rdf_forest_tune(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed, const float *forest,
                    int n_trees, int max_depth, int n_classes, uint16_t *labels_scratch, int labels_reduce, void *stream,
                    int *chosen_level, int *n_tried, int *levels_tried, float *ms_tried)
{
    if (!packed) return RDF_ERR_NULL_PTR;
    if (max_depth > 27) return RDF_ERR_BAD_ARG;
    const int cpad = classes_padded(n_classes);
    // candidates: never, and every block root level from the first below the staged levels (about 6) down to the last block's
    int cand[12], n_cand = 0;
    cand[n_cand++] = 0;
    if (deep_bytes(n_trees, max_depth, n_classes) != 0) {
        const int R0 = max_depth - deep_last_levels(cpad);
        for (int r = R0 % 3 + 6; r <= R0 && n_cand < 12; r += 3) cand[n_cand++] = r;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return (int)e;
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
    int best = 0, rc = RDF_OK;
    float best_ms = 0.f, heap_ms = 0.f;
    {   // the table is looked at BEFORE the first candidate goes into the host's map: packed_info writes a choice it finds there
        // into a table it sees for the first time (rdf_forest_set_deep_from before the first evaluation), and a trial candidate
        // is not a choice -- a failing tune must leave the table, its copies and other processes without one
        PackedState seen;
        rc = packed_info(packed, n_trees, max_depth, n_classes, stream, &seen);
        if (rc != RDF_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    }
    const int before = forest_deep_choice(packed);      // (what a failing tune leaves in place)
    for (int c = 0; c < n_cand && rc == RDF_OK; ++c) {
        rc = set_deep_choice(packed, cand[c], false);
        float ms_min = 0.f;
        int reps = 4;                                            // one warm-up, then the fastest of three ... twelve launches:
        for (int rep = 0; rep < reps && rc == RDF_OK; ++rep) {   // short launches are repeated until ~10 ms have been timed
            (void)hipEventRecord(e0, st);
            rc = eval_common(depth, n_img, dim_x, dim_y, packed, forest, n_trees, max_depth, n_classes, nullptr, -1,
                             labels_scratch, labels_reduce, 1.0f, 0, nullptr, stream);
            (void)hipEventRecord(e1, st);
            if (rc != RDF_OK) break;
            e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { rc = (int)e; break; }
            if (rep >= 1 && (rep == 1 || ms < ms_min)) ms_min = ms;
            if (rep == 1 && ms > 0.f) {
                const int want = 1 + (int)(10.0f / ms + 0.999f);
                reps = want < 4 ? 4 : want > 13 ? 13 : want;
            }
        }
        if (levels_tried) levels_tried[c] = cand[c];
        if (ms_tried) ms_tried[c] = ms_min;
        if (rc == RDF_OK && (c == 0 || ms_min < best_ms)) { best = cand[c]; best_ms = ms_min; }
        if (c == 0) heap_ms = ms_min;
    }
    // the heap-order table is the default a tie goes to: the blocks must win by 2 % (launch-to-launch noise is a fraction of that)
    if (rc == RDF_OK && best != 0 && best_ms > 0.98f * heap_ms) best = 0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != RDF_OK) {
        set_deep_choice(packed, before, false);
        return rc;
    }
    if (n_tried) *n_tried = n_cand;
    if (chosen_level) *chosen_level = best;
    return rdf_forest_set_deep_from(packed, best);
}

int rdf_eval_forest_packed_stats(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                                 const float *forest, int n_trees, int max_depth, int n_classes, uint16_t *labels_out,
                                 int labels_reduce, unsigned long long *stats8, void *stream)
{
    if (!stats8 || !packed) return RDF_ERR_NULL_PTR;
    if (max_depth > 27) return RDF_ERR_BAD_ARG;
    return eval_common(depth, n_img, dim_x, dim_y, packed, forest, n_trees, max_depth, n_classes, nullptr, -1, labels_out,
                       labels_reduce, 1.0f, 0, stats8, stream);
}

int rdf_eval_forest_packed_filled(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                                  const float *forest, int n_trees, int max_depth, int n_classes,
                                  const uint16_t *filter, int filter_class, uint16_t *labels_out,
                                  int labels_reduce, void *stream)
{
    if (max_depth > 27) return RDF_ERR_BAD_ARG;
    if (!packed) {      // degenerate forest (no tree or depth 0): nothing is evaluated, every label pixel is 65535
        if (n_trees > 0 && max_depth > 0) return RDF_ERR_NULL_PTR;
        if (n_img < 0 || dim_x < 0 || dim_y < 0 || labels_reduce < 1) return RDF_ERR_BAD_ARG;
        const size_t n = (size_t)n_img * (size_t)(dim_x / labels_reduce) * (size_t)(dim_y / labels_reduce);
        return rdf_fill_u16(labels_out, n, (uint16_t)kNoPixel, stream);
    }
    return eval_common(depth, n_img, dim_x, dim_y, packed, forest, n_trees, max_depth, n_classes, filter, filter_class,
                       labels_out, labels_reduce, 1.0f, 0, nullptr, stream, /*fill_untouched=*/1);
}

int rdf_eval_forest_packed_split(const uint16_t *depth, int n_img, int dim_x, int dim_y, const void *packed,
                                 const float *forest, int n_trees, int max_depth, int n_classes,
                                 const uint16_t *filter, int filter_class, uint16_t *labels_out,
                                 int labels_reduce, int fill_untouched, void *stream, void *helper_stream, int helper_cus,
                                 int queue_tag, int *helper_workgroups)
{
    if (helper_workgroups) *helper_workgroups = 0;
    if (!packed) return RDF_ERR_NULL_PTR;
    if (max_depth > 27 || helper_cus < 0) return RDF_ERR_BAD_ARG;
    if (helper_cus == 0 || helper_stream == stream)
        return eval_common(depth, n_img, dim_x, dim_y, packed, forest, n_trees, max_depth, n_classes, filter, filter_class,
                           labels_out, labels_reduce, 1.0f, 0, nullptr, stream, fill_untouched ? 1 : 0);
    for (void *s : {stream, helper_stream}) {       // (a graph replays on one stream; two launches on two streams are not recorded)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(reinterpret_cast<hipStream_t>(s), &cap) == hipSuccess && cap == hipStreamCaptureStatusActive)
            return RDF_ERR_CAPTURE;
    }
    SplitSpec spec = {reinterpret_cast<hipStream_t>(helper_stream), helper_cus, queue_tag, 0};
    tl_split = &spec;
    const int rc = eval_common(depth, n_img, dim_x, dim_y, packed, forest, n_trees, max_depth, n_classes, filter, filter_class,
                               labels_out, labels_reduce, 1.0f, 0, nullptr, stream, fill_untouched ? 1 : 0);
    tl_split = nullptr;
    if (helper_workgroups) *helper_workgroups = spec.launched;
    return rc;
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
                       bad_count, 0, 0u, nullptr, 0, nullptr, SpecFilters{});
    return (int)hipGetLastError();
}

static int layered_run(const uint16_t *depth, int dim_x, int dim_y, int n_layers, const void *const *packed,
                       const float *const *forests, const int *n_trees, const int *max_depth, const int *n_classes,
                       const int *filter_layer, const int *filter_class, uint16_t *const *layer_labels,
                       const uint16_t *const *layer_labels_dev_table, const int32_t *cond, int n_cond,
                       uint16_t *composite_out, int32_t *bad_count, int labels_reduce, float scale_factor,
                       int flip_x, const uint8_t *colors_rgba, int num_colors, uint8_t *image_rgba, void *stream)
{
    if (n_layers < 0 || dim_x < 0 || dim_y < 0 || labels_reduce < 1 || n_cond < 0) return RDF_ERR_BAD_ARG;
    if (n_layers > 0 && (!forests || !n_trees || !max_depth || !n_classes || !filter_layer || !filter_class ||
                         !layer_labels || !layer_labels_dev_table))
        return RDF_ERR_NULL_PTR;
    if (!composite_out) return RDF_ERR_NULL_PTR;
    for (int i = 0; i < n_layers; ++i)
        // a layer may only filter on an EARLIER layer here: the fills of LayeredDecisionForest.run are folded into the
        // kernels, so a later layer's buffer still holds the previous frame (the host wrapper takes the step-by-step path)
        if (filter_layer[i] >= i) return RDF_ERR_BAD_ARG;

    // Small launches (one frame) cost one wave-row's dependent chain (~40 us) each, however few pixels they evaluate, and
    // layer i+1 waits for layer i only because of its filter.  So the layers of a small packed stack are evaluated
    // UNFILTERED in ONE launch (workgroup b takes layer b % n_layers: k_eval_forest<..., NL>) and the composite kernel
    // applies the filters afterwards (SpecFilters): same label images, same composite, one ramp and drain instead of
    // n_layers.  Two launches on two streams do not get there: a cross-queue event join costs more than it saves
    // (DESIGN.md section 4).  Big launches keep the filtered evaluation, which does less work.
    SpecFilters spec = {};
    const int multi_knob = g_layers_one_launch;
    const int want_multi = multi_knob >= 0 ? multi_knob : env_int("RDF_LAYERS_ONE_LAUNCH", 1);
    if (want_multi && (n_layers == 2 || n_layers == 3) && packed) {
        Plan plans[3];
        bool ok = true;
        int cmax = 4, lds = 0;
        // Every layer as tree waves (512 threads: two pixel rows x T trees a workgroup) or none: planned with them first,
        // again without if a layer cannot (one tree, more than four, a big label map).  Config 3's frame: 37 -> 31 us.
        bool tw_all = false;
        for (int pass = 0; pass < 2; ++pass) {
            ok = true; cmax = 4; lds = 0;
            int n_tw = 0;
            for (int i = 0; i < n_layers && ok; ++i) {
                ok = packed[i] != nullptr && max_depth[i] <= 27 && n_trees[i] > 0 && max_depth[i] > 0 && n_classes[i] <= 8;
                if (!ok) break;
                const int rc = eval_common(depth, 1, dim_x, dim_y, packed[i], forests[i], n_trees[i], max_depth[i], n_classes[i],
                                           nullptr, -1, layer_labels[i], labels_reduce, 1.0f, 0, nullptr, stream,
                                           /*fill_untouched=*/1, &plans[i], /*allow_tw=*/pass == 0);
                if (rc != RDF_OK) return rc;
                const Plan &pl = plans[i];
                // (a layer whose table walks deep blocks takes a launch of its own: the kernel of several layers has no slabs)
                ok = !pl.empty && !pl.big && !pl.deep && (pl.block == 256 || pl.tw) && pl.a.rows_per_wave < kMaxRowsPerWave &&
                     pl.a.rows_per_wave == plans[0].a.rows_per_wave && (pl.tw || pl.a.n_tiles == plans[0].a.n_tiles);
                n_tw += pl.tw ? 1 : 0;
                cmax = n_classes[i] > 4 ? 8 : cmax;
                lds = pl.lds_bytes > lds ? pl.lds_bytes : lds;
            }
            tw_all = ok && n_tw == n_layers;
            if (!ok || n_tw == 0 || tw_all) break;
        }
        if (ok) {
            DeviceInfo di;
            int rc = device_info(&di);
            if (rc != 0) return rc;
            hipStream_t st = reinterpret_cast<hipStream_t>(stream);
            const int cus = usable_cus(st, di.cus);
            for (int i = 0; i < n_layers; ++i) {
                plans[i].a.sched = sched_slot(stream, i);
                // (a filter class of -1 means "no filter", as tree_eval.cu:81 and the filtered launches have it)
                spec.fl[i] = filter_class[i] != -1 ? filter_layer[i] : -1;
                spec.fc[i] = filter_layer[i] >= 0 ? filter_class[i] : -1;
            }
            // (all roles or none on the dynamic queue: a role without a slot would stride over tiles the others pull)
            bool all_sched = true;
            for (int i = 0; i < n_layers; ++i) all_sched = all_sched && plans[i].a.sched != nullptr;
            if (!all_sched) for (int i = 0; i < n_layers; ++i) plans[i].a.sched = nullptr;
            if (n_layers == 2) {
                EvalArgsN<2> ka;
                ka.l[0] = plans[0].a; ka.l[1] = plans[1].a;
                if (tw_all)
                    rc = cmax == 4 ? launch_multi<4, 2, true>(ka, lds, cus, st) : launch_multi<8, 2, true>(ka, lds, cus, st);
                else
                    rc = cmax == 4 ? launch_multi<4, 2>(ka, lds, cus, st) : launch_multi<8, 2>(ka, lds, cus, st);
            } else {
                EvalArgsN<3> ka;
                ka.l[0] = plans[0].a; ka.l[1] = plans[1].a; ka.l[2] = plans[2].a;
                if (tw_all)
                    rc = cmax == 4 ? launch_multi<4, 3, true>(ka, lds, cus, st) : launch_multi<8, 3, true>(ka, lds, cus, st);
                else
                    rc = cmax == 4 ? launch_multi<4, 3>(ka, lds, cus, st) : launch_multi<8, 3>(ka, lds, cus, st);
            }
            if (rc != RDF_OK) return rc;
            spec.n = n_layers;
        }
    }
    for (int i = 0; i < n_layers && spec.n == 0; ++i) {
        const int fl = filter_layer[i];
        const uint16_t *filt = fl >= 0 ? layer_labels[fl] : nullptr;
        const void *pk = packed ? packed[i] : nullptr;
        if (pk && max_depth[i] > 27) pk = nullptr;
        const int rc = eval_common(depth, 1, dim_x, dim_y, pk, forests[i], n_trees[i], max_depth[i], n_classes[i],
                                   filt, fl >= 0 ? filter_class[i] : -1, layer_labels[i], labels_reduce,
                                   pk ? 1.0f : scale_factor, 0, nullptr, stream, /*fill_untouched=*/1);
        if (rc != RDF_OK) return rc;
    }
    const int lw = dim_x / labels_reduce, lh = dim_y / labels_reduce;
    const long long n_px = (long long)lw * lh;
    if (n_px == 0) return RDF_OK;
    if (n_cond > 0 && !cond) return RDF_ERR_NULL_PTR;
    const unsigned blocks = (unsigned)((n_px + 255) / 256);
    hipLaunchKernelGGL(k_composite, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       layer_labels_dev_table, n_layers, (uint32_t)n_px, reinterpret_cast<const int2 *>(cond), n_cond,
                       composite_out, bad_count, 1, flip_x ? (uint32_t)lw : 0u,
                       reinterpret_cast<const uint32_t *>(colors_rgba), num_colors, reinterpret_cast<uint32_t *>(image_rgba), spec);
    return (int)hipGetLastError();
}

int rdf_layered_run(const uint16_t *depth, int dim_x, int dim_y, int n_layers, const void *const *packed,
                    const float *const *forests, const int *n_trees, const int *max_depth, const int *n_classes,
                    const int *filter_layer, const int *filter_class, uint16_t *const *layer_labels,
                    const uint16_t *const *layer_labels_dev_table, const int32_t *cond, int n_cond,
                    uint16_t *composite_out, int32_t *bad_count, int labels_reduce, float scale_factor,
                    void *stream)
{
    const unsigned long long t0 = now_ns();
    const int rc = layered_run(depth, dim_x, dim_y, n_layers, packed, forests, n_trees, max_depth, n_classes, filter_layer,
                               filter_class, layer_labels, layer_labels_dev_table, cond, n_cond, composite_out, bad_count,
                               labels_reduce, scale_factor, 0, nullptr, 0, nullptr, stream);
    g_host_ns_layered.fetch_add(now_ns() - t0, std::memory_order_relaxed);
    g_host_layered_calls.fetch_add(1, std::memory_order_relaxed);
    return rc;
}

int rdf_layered_run_hand(const uint16_t *depth, int dim_x, int dim_y, int n_layers, const void *const *packed,
                         const float *const *forests, const int *n_trees, const int *max_depth, const int *n_classes,
                         const int *filter_layer, const int *filter_class, uint16_t *const *layer_labels,
                         const uint16_t *const *layer_labels_dev_table, const int32_t *cond, int n_cond,
                         uint16_t *composite_out, int32_t *bad_count, int labels_reduce, float scale_factor,
                         int flip_x, const uint8_t *colors_rgba, int num_colors, uint8_t *image_rgba, void *stream)
{
    if (num_colors < 0 || (image_rgba && num_colors > 0 && !colors_rgba)) return RDF_ERR_BAD_ARG;
    const unsigned long long t0 = now_ns();
    const int rc = layered_run(depth, dim_x, dim_y, n_layers, packed, forests, n_trees, max_depth, n_classes, filter_layer,
                               filter_class, layer_labels, layer_labels_dev_table, cond, n_cond, composite_out, bad_count,
                               labels_reduce, scale_factor, flip_x, colors_rgba, num_colors, image_rgba, stream);
    g_host_ns_layered.fetch_add(now_ns() - t0, std::memory_order_relaxed);
    g_host_layered_calls.fetch_add(1, std::memory_order_relaxed);
    return rc;
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
void rdf_set_scheduler(int mode) { g_sched_mode = mode; }
void rdf_set_compaction(int mode) { g_compaction = mode; }
void rdf_set_halo(int pixels) { g_halo = pixels; }
void rdf_set_group(int trees) { g_group = trees; }
void rdf_set_layers_one_launch(int on) { g_layers_one_launch = on; }
void rdf_set_lds_levels(int levels) { g_lds_levels = levels; }
void rdf_set_tree_waves(int mode) { g_tree_waves = mode; }
void rdf_set_stage_vec(int on) { g_stage_vec = on; }
void rdf_set_rows_per_wave(int rows) { g_rows_per_wave = rows; }
void rdf_set_force_exact(int on) { g_force_exact = on; }
void rdf_set_last_level_table(int on) { g_last_level_table = on; }
void rdf_set_deep_from(int level) { g_deep_from = level; }
void rdf_set_fold(int on) { g_fold = on; }

int rdf_stream_create_with_reserved_cus(void **stream, int n_reserved)
{
    // A stream whose kernels never run on the first `n_reserved` CUs of hipExtStreamCreateWithCUMask's numbering.
    // The forest kernel's persistent workgroups fill every CU they may use, and a kernel as fat as RCCL's send/recv
    // kernel cannot squeeze in beside them (one of its workgroups needs the register files of a whole CU: measured,
    // tools/ubench_overlap.py -- it starts only when the forest launch drains).  CUs that the compute stream never
    // touches are where such kernels run while a forest launch is in flight.  The numbering is shader-engine-minor on
    // gfx950 (bit i = CU i/32 of shader engine i%32, measured with the same tool), and workgroups are dealt to shader
    // engines without regard to free CUs, so the useful amounts are multiples of 32: one CU in EVERY shader engine.
    if (!stream || n_reserved < 1) return RDF_ERR_BAD_ARG;
    DeviceInfo di;
    const int rc = device_info(&di);
    if (rc != 0) return rc;
    if (n_reserved >= di.cus) return RDF_ERR_BAD_ARG;
    const int words = (di.cus + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int cu = n_reserved; cu < di.cus; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
    hipStream_t st;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data());
    if (e != hipSuccess) return (int)e;
    *stream = st;
    return RDF_OK;
}
int rdf_stream_destroy(void *stream)
{
    // the stream's work is over once hipStreamDestroy returns: its queue slot (zero at rest) can serve another stream
    const hipError_t e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
    {
        std::lock_guard<std::mutex> lock(g_sched_mu);
        for (auto it = g_stream_cus.begin(); it != g_stream_cus.end();)
            it = it->first.second == stream ? g_stream_cus.erase(it) : std::next(it);
        for (auto it = g_sched_slot.begin(); it != g_sched_slot.end();) {
            if (std::get<1>(it->first) == stream) {
                g_sched_free[std::get<0>(it->first)].push_back(it->second);
                it = g_sched_slot.erase(it);
            } else {
                ++it;
            }
        }
    }
    const hipError_t d = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
    return (int)(d != hipSuccess ? d : e);
}

int rdf_stream_capture_id(void *stream, unsigned long long *capture_id)
{
    if (!capture_id) return RDF_ERR_NULL_PTR;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    const hipError_t e = hipStreamGetCaptureInfo(reinterpret_cast<hipStream_t>(stream), &cap, &id);
    if (e != hipSuccess) return (int)e;
    if (cap != hipStreamCaptureStatusActive) return RDF_ERR_BAD_ARG;
    *capture_id = id;
    return RDF_OK;
}

int rdf_graph_slots_release(unsigned long long capture_id)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RDF_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_sched_mu);
    auto it = g_graph_by_capture.find(std::make_pair(dev, capture_id));
    if (it == g_graph_by_capture.end()) return 0;
    const int n = (int)it->second.size();
    auto &fl = g_graph_free[dev];
    fl.insert(fl.end(), it->second.begin(), it->second.end());
    g_graph_by_capture.erase(it);
    return n;
}

// measurement hook: what the calls cost on the host since the last reset.  out[0] nanoseconds inside this library before a
// forest launch is handed to the HIP runtime (argument checks, LDS plan, queue slot), out[1] nanoseconds inside the runtime's
// launch call, out[2] forest launches counted; out[3] nanoseconds inside rdf_layered_run[_hand] from entry to return (plans,
// the forest launch(es) and the composite launch, runtime included), out[4] such calls
int rdf_debug_host_overhead(unsigned long long out[5], int reset)
{
    if (out) {
        out[0] = g_host_ns_plan.load(); out[1] = g_host_ns_launch.load(); out[2] = g_host_calls.load();
        out[3] = g_host_ns_layered.load(); out[4] = g_host_layered_calls.load();
    }
    if (reset) { g_host_ns_plan = 0; g_host_ns_launch = 0; g_host_calls = 0; g_host_ns_layered = 0; g_host_layered_calls = 0; }
    return RDF_OK;
}

// test hook: how many stream slots / graph slots the current device holds at the moment
int rdf_debug_sched_slots(int *stream_slots_in_use, int *graph_slots_used)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RDF_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_sched_mu);
    int used = 0;
    for (const auto &kv : g_sched_slot) used += std::get<0>(kv.first) == dev;
    if (stream_slots_in_use) *stream_slots_in_use = used;
    if (graph_slots_used) *graph_slots_used = g_graph_next[dev] - (int)g_graph_free[dev].size();
    return RDF_OK;
}

// ---- peer-to-peer plumbing for the multi-GPU gather (3d-beats_amd/distributed.py): the root
// exports its receive buffer, every other process maps it and copies its shard in with the copy engines ----
int rdf_device_malloc(void **ptr, size_t bytes)
{
    if (!ptr) return RDF_ERR_NULL_PTR;
    return (int)hipMalloc(ptr, bytes);     // a raw allocation: its base address is what hipIpcGetMemHandle needs
}
int rdf_device_free(void *ptr) { return (int)hipFree(ptr); }
int rdf_ipc_export(void *ptr, unsigned char handle_out[64])
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    if (!ptr || !handle_out) return RDF_ERR_NULL_PTR;
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return (int)e;
    memcpy(handle_out, &h, 64);
    return RDF_OK;
}
int rdf_ipc_open(const unsigned char handle[64], void **ptr_out)
{
    if (!handle || !ptr_out) return RDF_ERR_NULL_PTR;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    return (int)hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess);
}
int rdf_ipc_close(void *ptr) { return (int)hipIpcCloseMemHandle(ptr); }
int rdf_memcpy_device_async(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes == 0) return RDF_OK;
    if (!dst || !src) return RDF_ERR_NULL_PTR;
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, reinterpret_cast<hipStream_t>(stream));
}

int rdf_debug_fat_kernel(int n_workgroups, unsigned long long spin_ticks, unsigned long long *t_start, void *stream)
{
    if (n_workgroups < 1) return RDF_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_debug_fat, dim3((unsigned)n_workgroups), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       spin_ticks, t_start);
    return (int)hipGetLastError();
}

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

#ifndef RDF_BUILD_ID
#define RDF_BUILD_ID "unknown"
#endif
// (the marker in front lets a build script find the id in the file without loading it)
static const char kBuildIdMarker[] = "rdf-build-id:" RDF_BUILD_ID;
const char *rdf_build_id(void) { return kBuildIdMarker + 13; }

const char *rdf_error_string(int code)
{
    switch (code) {
    case RDF_OK: return "ok";
    case RDF_ERR_BAD_ARG: return "rdf: bad argument";
    case RDF_ERR_NULL_PTR: return "rdf: required pointer is NULL";
    case RDF_ERR_TOO_LARGE: return "rdf: call addresses >= 2^31 pixels (or a frame is 2^23 pixels wide / 2^24 high or more), split the batch";
    case RDF_ERR_NO_DEVICE: return "rdf: no usable HIP device";
    case RDF_ERR_CAPTURE: return "rdf: the stream is being captured into a hipGraph and this call needs a synchronous step (a packed table's first evaluation reads its info block back: evaluate it once, or pack it, before capturing)";
    case RDF_ERR_STALE: return "rdf: a kernel of an earlier call found another packed table at an address than the one this process remembered there (a table was copied over a known address without rdf_forest_forget); what the library knew about this device's tables has been dropped: call again";
    case RDF_ERR_BUILD: return "rdf: this build of the library breaks an assumption of its own kernels (static LDS in a forest kernel); rebuild it";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "rdf: unknown error";
}

} // extern "C"
