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

// Pixels of a tile on which a Gaussian can contribute, as inclusive tile-local ranges clipped
// to [x0,x1] x [y0,y1] (in/out).  A pair contributes only if alpha >= alpha_min, i.e.
// sigma <= tau = ln(opacity/alpha_min); inside that ellipse |dx| <= sqrt(2 tau c/det) and
// |dy| <= sqrt(2 tau a/det).  The bound is padded against rounding and trusted only for a
// well-conditioned positive-definite conic; otherwise the ranges are left as given (every
// pixel is evaluated).  An empty result has x1 < x0 or y1 < y0.
__device__ __forceinline__ void support_rect(float gx, float gy, float a, float b, float c,
                                             float op, float alpha_min, float ox, float oy,
                                             int &x0, int &x1, int &y0, int &y1)
{
    // hardware-approximate rcp / log / sqrt (1 ulp) are enough here: the bound is padded by
    // 0.1 % + 1e-3 px twice over, and the exact alpha test still decides every pixel inside it
    const float det = a * c - b * b;
    const float ratio = op * __builtin_amdgcn_rcpf(alpha_min);
    if (!(ratio >= 1.0005f)) {
        // opacity at or below the cut (within the rcp's rounding): leave the decision to the
        // exact per-pixel test unless it is clearly below
        if (!(ratio >= 0.9995f)) x1 = x0 - 1;  // no pixel can reach alpha_min
    } else if (a > 0.f && c > 0.f && det > 1.0e-3f * a * c && ratio < 3.0e38f) {
        const float tau = 1.001f * __logf(ratio) + 1.0e-3f;
        const float k2 = 2.0f * tau * __builtin_amdgcn_rcpf(det);
        const float hx = 1.001f * __builtin_amdgcn_sqrtf(k2 * c) + 1.0e-3f;
        const float hy = 1.001f * __builtin_amdgcn_sqrtf(k2 * a) + 1.0e-3f;
        if (hx < 1.0e6f && hy < 1.0e6f) {
            axis_range(gx, ox, hx, x0, x1, x0, x1);
            axis_range(gy, oy, hy, y0, y1, y0, y1);
        }
    }
}
