// rdf_device.hpp -- device helpers shared by the forest-evaluation and the training kernels:
// the reference's __float2int_rd, wrapping coordinate adds, the verified shared-reciprocal divide, the
// two-source depth probe (staged LDS tile / global memory) in its round-1 form (ProbeCtx) and in round 3's (TileCtx: tile at
// LDS address 0, x doubled), the rounding-mode switches of the one-fma divide-and-floor.
#ifndef RDF_DEVICE_HPP
#define RDF_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr uint32_t kNoPixel = 65535u;
typedef float f2 __attribute__((ext_vector_type(2)));

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

// Same result for every non-NaN input (checked on gfx950, tests/test_gpu_parity.py); one VALU op.
__device__ __forceinline__ int floor_i32_not_nan(float f)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

__device__ __forceinline__ int add_wrap(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }

// floor((a.x)/d), floor((a.y)/d) for a pair of numerators with the pixel's refined reciprocal rcp2 = (r, r),
// ndf2 = (-d, -d): q0 = a*r; rem = a - d*q0 (exact, fma); q = q0 + rem*r.  Equal to the IEEE divide in
// floor-to-int for every depth 1..65535 and every numerator that is +-0 or has a biased exponent in
// [40, 230] (tools/verify_fastdiv.hip, exhaustive on gfx950).  Packed f32: two quotients per instruction.
__device__ __forceinline__ f2 fast_quotient2(f2 a, f2 rcp2, f2 ndf2)
{
    const f2 q0 = a * rcp2;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(ndf2, q0, a), rcp2, q0);
}

__device__ __forceinline__ bool fast_divide_ok(float a)   // the verified numerator range
{
    const uint32_t b = __float_as_uint(a);
    return (b << 1) == 0u || (((b >> 23) & 0xFFu) - 40u) <= 190u;
}

// A depth probe at (x, y) of the current image is answered by the staged LDS tile when the coordinate
// falls inside it (cells outside the image already hold 65535), else by global memory with the
// per-axis bounds check of cu_utils.hpp:79-86.  The loads are only ISSUED here; their values are
// consumed after all probes of the level have been issued, so every probe of a level is in flight
// together and nothing waits inside a divergent branch (an earlier version waited vmcnt(0) per probe).
// `img_b` is the wave-uniform base of the current image (a scalar register pair, so a far probe's address is
// one 32-bit byte offset: a call addresses < 2^31 pixels); multiplies are 24-bit (full rate; every factor is
// < 2^24 when used).  tile[-1] must hold 65535: lanes that leave the tile read that cell, so a probe outside
// the image needs no separate constant.
struct ProbeCtx {
    const uint16_t *tile;   // LDS; tile[-1] == 65535
    const char *img_b;
    int tx0, ty0, tw, th, twp, W, H;
};

struct Probe {
    uint32_t lds_v, glb_v;   // the two candidate values
    bool from_global;
};

// (cx, cy) = probe position RELATIVE TO THE STAGED TILE (the caller adds the offsets to the pixel's own
// tile-relative position); image coordinates are only rebuilt for lanes that leave the tile.
__device__ __forceinline__ Probe probe_issue(const ProbeCtx &c, int cx, int cy)
{
    const bool in_tile = (uint32_t)cx < (uint32_t)c.tw && (uint32_t)cy < (uint32_t)c.th;
    const int li = in_tile ? (int)(__umul24((uint32_t)cy, (uint32_t)c.twp) + (uint32_t)cx) : -1;
    uint32_t undefined;
    asm("" : "=v"(undefined));   // glb_v is only read where from_global is set
    Probe p = {c.tile[li], undefined, false};
    if (!in_tile) {
        const int x = add_wrap(cx, c.tx0), y = add_wrap(cy, c.ty0);
        if ((uint32_t)x < (uint32_t)c.W && (uint32_t)y < (uint32_t)c.H) {
            // only far lanes touch global memory; the value is consumed after the branch
            const uint32_t go = (__umul24((uint32_t)y, (uint32_t)c.W) + (uint32_t)x) << 1;
            p.glb_v = *reinterpret_cast<const uint16_t *>(c.img_b + go);
            p.from_global = true;
        }
    }
    return p;
}

__device__ __forceinline__ int probe_value(const Probe &p)
{
    const uint32_t g = p.glb_v, l = p.lds_v;
    return (int)(p.from_global ? g : l);
}

// ---- round 3: divide, floor and add in one fma (rdf_hip.hip, NodeRec16; tools/verify_magic.hip) ----
constexpr float kNumScale = 512.0f;            // a decoded numerator is 512 * (x + e), 1/4 <= e <= 3/4
constexpr float kMagic = 12582912.0f;          // 1.5 * 2^23: an fp32 in [2^23, 2^24) counts in units of one
constexpr uint32_t kMagicBits = 0x4B400000u;   // its bit pattern

// The wave's fp32 rounding mode (MODE.FP_ROUND bits 1:0: 0 = to nearest even, 2 = toward minus infinity).  The level
// loop's divide-and-floor is one fma in round-down mode (NodeRec16); everything else -- the reciprocal's refinement,
// the IEEE divides of kFlagExact nodes, the sums of leaf PDFs -- needs round-to-nearest.  The operand ties a switch to
// the arithmetic around it: what produced `dep` runs before the switch, what uses it afterwards runs after.
template <typename T>
__device__ __forceinline__ void set_round_down(T &dep)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\ts_nop 0" : "+v"(dep));
}
template <typename T>
__device__ __forceinline__ void set_round_nearest(T &dep)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 0" : "+v"(dep));
}
template <typename T>
__device__ __forceinline__ void pin(T &v)   // v is computed before this point and used only after it
{
    asm volatile("" : "+v"(v));
}

// A depth probe is answered by the staged LDS tile when it falls inside it (cells outside the image hold 65535 there),
// else by global memory with the per-axis bounds check of cu_utils.hpp:79-86; a probe outside the image is 65535.
// Coordinates are RELATIVE TO THE STAGED TILE, x doubled (a byte offset): cx2 = 2 (x - tx0), cy = y - ty0.  The loads
// are only ISSUED here; their values are consumed after all probes of the level have been issued, so every probe of a
// level is in flight together and nothing waits inside a divergent branch.  A lane outside the tile reads LDS at an
// address that means nothing (inside the allocation: some cell; beyond it: the hardware returns 0) and discards it.
// `img_b` is the wave-uniform base of the current image (a scalar register pair), so a far probe's address is one 32-bit
// byte offset (a call addresses < 2^31 pixels); multiplies are 24-bit (full rate; every factor is < 2^24 when used).
struct TileCtx {
    const char *img_b;
    uint32_t tw2, th, twp2;  // staged tile: 2 x width, height, row pitch in bytes (0, 0, 0: no staged tile)
    uint32_t W2, H;          // image: 2 x width, height
    uint32_t tx0_2, ty0;     // image position of the tile's first cell (x doubled), two's complement
};

struct TileProbe {
    uint32_t lds_v, glb_v;
    bool in_tile;
};

__device__ __forceinline__ TileProbe tprobe_issue(const TileCtx &c, uint32_t cx2, uint32_t cy)
{
    TileProbe p;
    p.in_tile = cx2 < c.tw2 && cy < c.th;
    // the tile starts at LDS address 0 (the kernel has no static LDS; the launcher checks): byte offset = address
    p.lds_v = *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)(__umul24(cy, c.twp2) + cx2);
    asm("" : "=v"(p.glb_v));   // only read where in_tile is false
    if (!p.in_tile) {
        const uint32_t x2 = cx2 + c.tx0_2, y = cy + c.ty0;
        p.glb_v = kNoPixel;
        if (x2 < c.W2 && y < c.H)   // only far lanes touch global memory; the value is consumed after the branch
            p.glb_v = *reinterpret_cast<const uint16_t *>(c.img_b + (__umul24(y, c.W2) + x2));
    }
    return p;
}

__device__ __forceinline__ int tprobe_value(const TileProbe &p)
{
    const uint32_t g = p.glb_v, l = p.lds_v;
    return (int)(p.in_tile ? l : g);
}

} // namespace

#endif // RDF_DEVICE_HPP
