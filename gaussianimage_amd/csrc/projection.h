// projection.h -- per-Gaussian device code shared by the projection kernels and the fused
// fitting tail: outputs + tile-list filling, and the chain rule from conic to covariance.
#pragma once
#include "support.h"
#include "tiles.h"

// Optional culling of the LISTS (never of the outputs): with the rasteriser's opacities known,
// a Gaussian enters only the tiles of its box that its support can touch.
struct ListCull {
    const float *opacity;  // [N]; nullptr = list every tile of the box
    float alpha_min, pixel_offset;
};

__device__ __forceinline__ void project_tail(int g, float mx, float my, float sxx, float sxy,
                                             float syy, int H, int W, int tiles_x, int tiles_y,
                                             float center_bias, float *xys, float *depths,
                                             int *radii, float *conics, int *num_tiles_hit,
                                             int *tile_count, int *seg, int C, int row0, int row1,
                                             const ListCull &cull)
{
    float ca = 0.f, cb = 0.f, cc = 0.f, radius = 0.f, cx = 0.f, cy = 0.f;
    int irad = 0, area = 0;
    TileBox box = {0, 0, 0, 0};
    if (cov_to_conic_radius(sxx, sxy, syy, ca, cb, cc, radius)) {
        cx = 0.5f * (float)W * mx + 0.5f * (float)W + center_bias;
        cy = 0.5f * (float)H * my + 0.5f * (float)H + center_bias;
        irad = f2i_sat(radius);
        box = tile_box(cx, cy, radius, tiles_x, tiles_y);
        area = (box.x1 - box.x0) * (box.y1 - box.y0);
        if (area < 0) area = 0;
    } else {
        ca = cb = cc = 0.f;
    }
    xys[2 * g] = cx;
    xys[2 * g + 1] = cy;
    depths[g] = 0.0f;
    radii[g] = irad;
    conics[3 * g] = ca;
    conics[3 * g + 1] = cb;
    conics[3 * g + 2] = cc;
    num_tiles_hit[g] = area;
    if (tile_count != nullptr && irad > 0 && area > 0) {
        if (cull.opacity != nullptr)
            cull_tile_box(box, cx, cy, ca, cb, cc, cull.opacity[g], cull.alpha_min,
                          cull.pixel_offset);
        emit_tiles(g, box, row0, row1, tiles_x, tile_count, seg, C);
    }
}

// dL/dSigma from dL/dconic: with C = Sigma^-1 and V = [[va, vb/2],[vb/2, vc]], G = -C V C;
// returned as (gxx, gxy, gyy), gxy being the derivative w.r.t. the single parameter sxy.
__device__ __forceinline__ void conic_grad_to_cov_grad(float a, float b, float c, float va,
                                                       float vb2, float vc, float &gxx, float &gxy,
                                                       float &gyy)
{
    float vb = 0.5f * vb2;
    float m00 = a * va + b * vb, m01 = a * vb + b * vc;
    float m10 = b * va + c * vb, m11 = b * vb + c * vc;
    gxx = -(m00 * a + m01 * b);
    float g01 = -(m00 * b + m01 * c);
    float g10 = -(m10 * a + m11 * b);
    gyy = -(m10 * b + m11 * c);
    gxy = g01 + g10;
}
