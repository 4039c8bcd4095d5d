// support.h -- the support rectangle of a Gaussian on a tile (forward and backward rasteriser).
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------
// support rectangle shared by the forward and the backward rasterizer
// ------------------------------------------------------------------------------------------

// Tile-local pixel range [lo,hi] (inclusive) whose sample coordinate o+l lies within h of c.
__device__ __forceinline__ void axis_range(float c, float o, float h, int lim_lo, int lim_hi,
                                           int &lo, int &hi)
{
    float flo = ceilf(c - o - h), fhi = floorf(c - o + h);
    flo = fminf(fmaxf(flo, -1.0f), 32.0f);
    fhi = fminf(fmaxf(fhi, -2.0f), 32.0f);
    lo = max(lim_lo, (int)flo);
    hi = min(lim_hi, (int)fhi);
}

// Half-widths of the bounding box of a Gaussian's support.  A pair contributes only if
// alpha >= alpha_min, i.e. sigma <= tau = ln(opacity/alpha_min); inside that ellipse
// |dx| <= sqrt(2 tau c/det) and |dy| <= sqrt(2 tau a/det).  The bound is padded against rounding
// (0.1 % + 1e-3 px, twice over: hardware-approximate rcp / log / sqrt are enough, and the exact
// alpha test still decides every pixel inside it) and given only for a well-conditioned
// positive-definite conic.  Returns 0 = no pixel can reach alpha_min, 1 = hx, hy are valid,
// 2 = no usable bound (every pixel has to be evaluated).
__device__ __forceinline__ int support_halfwidths(float a, float b, float c, float op,
                                                  float alpha_min, float &hx, float &hy)
{
    const float det = a * c - b * b;
    const float ratio = op * __builtin_amdgcn_rcpf(alpha_min);
    if (!(ratio >= 1.0005f)) {
        // opacity at or below the cut (within the rcp's rounding): leave the decision to the
        // exact per-pixel test unless it is clearly below
        return ratio >= 0.9995f ? 2 : 0;
    }
    if (a > 0.f && c > 0.f && det > 1.0e-3f * a * c && ratio < 3.0e38f) {
        const float tau = 1.001f * __logf(ratio) + 1.0e-3f;
        const float k2 = 2.0f * tau * __builtin_amdgcn_rcpf(det);
        hx = 1.001f * __builtin_amdgcn_sqrtf(k2 * c) + 1.0e-3f;
        hy = 1.001f * __builtin_amdgcn_sqrtf(k2 * a) + 1.0e-3f;
        if (hx < 1.0e6f && hy < 1.0e6f) return 1;
    }
    return 2;
}

// Pixels of a tile on which a Gaussian can contribute, as inclusive tile-local ranges clipped
// to [x0,x1] x [y0,y1] (in/out), from the bound support_halfwidths gave (kind, hx, hy); (ox, oy)
// is the sample coordinate of the tile's first pixel.  Without a usable bound the ranges are
// left as given.  An empty result has x1 < x0 or y1 < y0.
__device__ __forceinline__ void support_rect_from(int kind, float hx, float hy, float gx, float gy,
                                                  float ox, float oy, int &x0, int &x1, int &y0,
                                                  int &y1)
{
    if (kind == 0) {
        x1 = x0 - 1;
    } else if (kind == 1) {
        axis_range(gx, ox, hx, x0, x1, x0, x1);
        axis_range(gy, oy, hy, y0, y1, y0, y1);
    }
}

__device__ __forceinline__ void support_rect(float gx, float gy, float a, float b, float c,
                                             float op, float alpha_min, float ox, float oy,
                                             int &x0, int &x1, int &y0, int &y1)
{
    float hx = 0.f, hy = 0.f;
    const int kind = support_halfwidths(a, b, c, op, alpha_min, hx, hy);
    support_rect_from(kind, hx, hy, gx, gy, ox, oy, x0, x1, y0, y1);
}

// Tiles [lo, hi) of one axis whose pixels (sample coordinates j + pixel_offset) can lie within
// h of c; the caller intersects with the Gaussian's tile box.
__device__ __forceinline__ void axis_tiles(float c, float pixel_offset, float h, int &lo, int &hi)
{
    float jlo = ceilf(c - pixel_offset - h), jhi = floorf(c - pixel_offset + h);
    jlo = fminf(fmaxf(jlo, -1.0e6f), 1.0e6f);
    jhi = fminf(fmaxf(jhi, -1.0e6f), 1.0e6f);
    lo = (int)floorf(jlo * (1.0f / GI2D_BLOCK));
    hi = (int)floorf(jhi * (1.0f / GI2D_BLOCK)) + 1;
}

// Shrinks a Gaussian's tile box to the tiles its support can touch (the bound of
// support_halfwidths, evaluated in frame coordinates): a tile dropped here holds no pixel on
// which the Gaussian could reach alpha_min, so listing it would only cost list traffic.
__device__ __forceinline__ void cull_tile_box(TileBox &box, float cx, float cy, int kind, float hx,
                                              float hy, float pixel_offset)
{
    if (kind == 0) {
        box.x1 = box.x0;
    } else if (kind == 1) {
        int lo, hi;
        axis_tiles(cx, pixel_offset, hx, lo, hi);
        box.x0 = max(box.x0, lo);
        box.x1 = min(box.x1, hi);
        axis_tiles(cy, pixel_offset, hy, lo, hi);
        box.y0 = max(box.y0, lo);
        box.y1 = min(box.y1, hi);
    }
}
