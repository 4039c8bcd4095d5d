// common.h -- constants and small device helpers shared by every translation unit of libgi2d.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <new>

// the library is built with hidden visibility; the C ABI is what it exports
#pragma GCC visibility push(default)
#include "../../include/gi2d.h"
#pragma GCC visibility pop

// sine / cosine as a fixed IEEE operation sequence shared with the CPU checker
#define GI2D_SINCOS_FN __device__ __forceinline__
#include "sincos_f32.h"

#define WAVE 64
// Workgroup size of the per-Gaussian kernels that issue scattered atomics: one wave, so that
// the launch spreads over every CU.
#ifndef PG_THREADS
#define PG_THREADS 64
#endif
// Per-tile counters live one per 128-byte line: device-scope atomics on one line serialise at
// the memory side, and ~48 Gaussians per tile hammering 32 neighbouring counters of one line
// made the atomic pass three times slower.
#define CNT_STRIDE 32
// ... two per line, in fact: the counters of tiles 2j and 2j+1 (of one frame) are neighbouring
// words, so that a projection that reserves places for both can do it with ONE 64-bit atomic
// (returning atomics are what the aggregated list filling waits for: about 2e10 per second on
// this part whatever their scope).  The words in between stay unused.
#define CNT_AT(t) (((size_t)((t) & ~1)) * CNT_STRIDE + (size_t)((t) & 1))
#define LOG2E 1.4426950408889634f
#define PART_STRIDE 9  // backward partial sums per (tile, Gaussian): v_xy 2, v_conic 3, v_rgb 3, v_op 1
// Backward pass 1 leaves the partial sums of a Gaussian's first GSLOTS tiles (row-major in its
// tile box) in the Gaussian's OWN slots, GSLOT_STRIDE floats each: the 9 sums and the stamp of
// the backward pass that wrote them.  Pass 2 reads the first GSLOTS_EAGER of them before it knows
// anything about the Gaussian (boxes at initialisation are mostly <= 2x2), the others once the
// box is known (fitted Gaussians cover 3x3 .. 4x4 tiles) -- no look-up either way; only tiles
// beyond GSLOTS go through the per-tile slab and a search in the tile's list.
#define GSLOTS 16
#define GSLOTS_EAGER 4
#define GSLOT_STRIDE 10

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------

// device-side status of a workspace
struct DevStatus {
    int max_count;            // largest per-tile count seen so far (monotone; capacity policy)
    int pad;
    long long overflow_gen;   // generation of the last frame in which a segment overflowed
};

__device__ __forceinline__ int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
    return (int)v;
}

// Conic (inverse covariance) and integer 3-sigma radius; false = early-out.
__device__ __forceinline__ bool cov_to_conic_radius(float sxx, float sxy, float syy, float &ca,
                                                    float &cb, float &cc, float &radius)
{
    float det = sxx * syy - sxy * sxy;
    if (det == 0.0f || !(fabsf(det) <= 3.0e38f)) return false;
    float inv_det = 1.0f / det;
    ca = syy * inv_det;
    cb = -sxy * inv_det;
    cc = sxx * inv_det;
    float b = 0.5f * (sxx + syy);
    float disc = fmaxf(0.1f, b * b - det);
    float s = sqrtf(disc);
    float v1 = b + s;
    float v2 = b - s;
    float r = ceilf(3.0f * sqrtf(fmaxf(v1, v2)));
    if (!(r <= 3.0e38f)) return false;
    radius = r;
    return true;
}

struct TileBox {
    int x0, x1, y0, y1;  // [x0,x1) x [y0,y1)
};

__device__ __forceinline__ TileBox tile_box(float cx, float cy, float radius, int tiles_x,
                                            int tiles_y)
{
    float tcx = cx / (float)GI2D_BLOCK, tcy = cy / (float)GI2D_BLOCK;
    float tr = radius / (float)GI2D_BLOCK;
    TileBox b;
    b.x0 = min(max(0, f2i_sat(tcx - tr)), tiles_x);
    b.x1 = min(max(0, f2i_sat(tcx + tr + 1.0f)), tiles_x);
    b.y0 = min(max(0, f2i_sat(tcy - tr)), tiles_y);
    b.y1 = min(max(0, f2i_sat(tcy + tr + 1.0f)), tiles_y);
    return b;
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 4 lanes of a quad, result in every lane of the quad
__device__ __forceinline__ float quad_sum(float v)
{
    v = v + dpp_mov<0xB1>(v);  // quad_perm:[1,0,3,2]
    v = v + dpp_mov<0x4E>(v);  // quad_perm:[2,3,0,1]
    return v;
}

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share one).  This maps the
// launch index to a tile index so that each XCD works on a contiguous run of tiles: neighbouring
// tiles share Gaussians (L2 hits on the gathers) and image cache lines.  Bijective for any T.
__device__ __forceinline__ int xcd_tile(int bid, int T)
{
    const int q = T >> 3, r = T & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// Sum over the 64 lanes of a wave, in the pairing order of an xor butterfly (1, 2, 4, 8, 16, 32)
// as lane 0 sees it, valid in lane 0: quads and rows by DPP adds (a lane's partner under
// row_half_mirror / row_mirror is another lane of the partner group, which holds the same sum),
// the four rows through read-lanes, (r0 + r1) + (r2 + r3).  The same bits as six __shfl_xor
// steps, without their six dependent LDS round trips.
__device__ __forceinline__ float wave_sum_lane0(float v)
{
    v = quad_sum(v);
    v = v + dpp_mov<0x141>(v);  // row_half_mirror: lane i <-> 7-i of each 8
    v = v + dpp_mov<0x140>(v);  // row_mirror: lane i <-> 15-i of each 16
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Inclusive prefix sum over the 64 lanes of a wave (every lane active) by DPP adds: shifts by
// 1, 2, 4, 8 inside each row of 16, then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31
// into rows 2 and 3.  Six 4-cycle instructions where six __shfl_up steps are six dependent
// ds_bpermute round trips with their address arithmetic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_shift_int(int v)
{
    // (a lane with no source -- shifted in from outside its row, or in a row the mask leaves
    // out -- receives the 0 given as the old value)
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += dpp_shift_int<0x111, 0xf>(v);  // row_shr:1
    v += dpp_shift_int<0x112, 0xf>(v);  // row_shr:2
    v += dpp_shift_int<0x114, 0xf>(v);  // row_shr:4
    v += dpp_shift_int<0x118, 0xf>(v);  // row_shr:8
    v += dpp_shift_int<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_shift_int<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- selects made of 2-cycle instructions ------------------------------------------------------
// clamp(a * b + c) to [0, 1] in one instruction
__device__ __forceinline__ float fma_clamp01(float a, float b, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// clamp(a - b) to [0, 1] in one instruction
__device__ __forceinline__ float sub_clamp01(float a, float b)
{
    float r;
    asm("v_sub_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// What the loops need of the conventions, in registers (a constant read from an SGPR would make
// its instruction a 4-cycle one): 1 where alpha >= alpha_min, 0 below, is
// clamp(alpha * step_scale + step_bias).
struct StepConsts {
    float step_scale, step_bias;
};

__device__ __forceinline__ StepConsts step_consts(const gi2d_conv &cv)
{
    StepConsts k;
    k.step_scale = 1.152921504606846976e18f;  // 2^60: one ulp of alpha_min becomes >= 1
    // (alpha - below(alpha_min)) * 2^60 is <= 0 for alpha < alpha_min and >= 2^28 otherwise
    k.step_bias = cv.alpha_min > 0.f
                      ? -__int_as_float(__float_as_int(cv.alpha_min) - 1) * k.step_scale
                      : 1.0f;
    return k;
}
