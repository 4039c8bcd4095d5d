// projection.h -- per-Gaussian device code shared by the projection kernels and the fused
// fitting tail: outputs + tile-list filling + the rasteriser's per-Gaussian record, and the
// chain rule from conic to covariance.
#pragma once
#include "support.h"
#include "tiles.h"

// Optional culling of the LISTS (never of the outputs): with the rasteriser's opacities known,
// a Gaussian enters only the tiles of its box that its support can touch.
struct ListCull {
    const float *opacity;  // [N]; nullptr = list every tile of the box
    float alpha_min, pixel_offset;
};

// The five tensors the projection operators return.
struct ProjOut {
    float *xys, *depths;
    int *radii;
    float *conics;
    int *num_tiles_hit;
};

// Where a projection that also fills the tile lists puts them (tile_count == nullptr: nowhere).
struct TileLists {
    int *tile_count, *seg;
    int C, row0, row1;
};

// The rasteriser's per-Gaussian RECORD: everything the forward rasteriser needs of a Gaussian in
// one 48-byte line, written once per Gaussian by a projection that already knows the colours and
// opacities the rasteriser will use (the frame, fitting and decode entry points), so that a tile
// gathers one line per list entry instead of four arrays, and nothing per-Gaussian is recomputed
// per (tile, Gaussian) pair:
//   [0..3]  x, y, -0.5*log2e*a, -log2e*b
//   [4..7]  -0.5*log2e*c, log2(opacity) (plain) or opacity (general, see the pair loop), red, green
//   [8..11] blue, flags (bit 0: general; bits 1-2: kind of the support bound), hx, hy
// (hx, hy: half-widths of the bounding box of the ellipse alpha >= alpha_min, support.h).
#define REC_FLOATS 12
struct RecordSink {
    float *rec;            // [N][REC_FLOATS]; nullptr = no records
    const float *opacity;  // [N]
    float alpha_cap;
};

struct Record {
    float4 r0, r1, r2;
};

// An entry is "plain" when its exponent can never be positive (a well-conditioned
// positive-definite conic: rounding cannot turn sigma >= 0 into a positive exponent) and its
// alpha can never reach the cap (0 <= opacity <= cap, and alpha <= opacity).  Plain entries carry
// log2(opacity), which the pair loop folds into the exponent; the others carry the opacity itself
// and take the loop's general form.
__device__ __forceinline__ bool plain_entry(float a, float b, float c, float op, float alpha_cap)
{
    return a > 0.f && c > 0.f && a * c - b * b > 1.0e-3f * (a * c) && op >= 0.f && op <= alpha_cap;
}

// The one place a record is made: the projection (per Gaussian) and the rasteriser's staging of
// separately passed tensors (per list entry) run exactly this, so both give the same bits.
__device__ __forceinline__ Record make_record(float cx, float cy, float ca, float cb, float cc,
                                              float op, float alpha_cap, float alpha_min, float red,
                                              float green, float blue)
{
    float hx = 0.f, hy = 0.f;
    const int kind = support_halfwidths(ca, cb, cc, op, alpha_min, hx, hy);
    const bool plain = plain_entry(ca, cb, cc, op, alpha_cap);
    const int flags = (plain ? 0 : 1) | (kind << 1);
    Record r;
    r.r0 = make_float4(cx, cy, (-0.5f * LOG2E) * ca, (-LOG2E) * cb);
    r.r1 = make_float4((-0.5f * LOG2E) * cc, plain ? __log2f(op) : op, red, green);
    r.r2 = make_float4(blue, __int_as_float(flags), hx, hy);
    return r;
}

// Everything of a projection but the list filling: the five outputs, the record, and the tile
// box the Gaussian is to be listed in (culled to its support when the opacities are known).
// Returns false when it is listed nowhere.  (cr, cg, cb_): the colour that goes into the record
// (unused without records).  want_lists: false skips the box culling.  [row0, row1): the tile
// rows the lists cover -- a Gaussian whose box lies outside them (seven of eight, for a rank that
// owns an eighth of the frame) gets its five outputs and nothing else: no support bound, no
// record (which only a list entry could ever refer to).
__device__ __forceinline__ bool project_body(int g, float mx, float my, float sxx, float sxy,
                                             float syy, int H, int W, int tiles_x, int tiles_y,
                                             float center_bias, const ProjOut &o, bool want_lists,
                                             const ListCull &cull, const RecordSink &rs, float cr,
                                             float cg, float cb_, TileBox &box, int row0 = 0,
                                             int row1 = INT_MAX)
{
    float ca = 0.f, cb = 0.f, cc = 0.f, radius = 0.f, cx = 0.f, cy = 0.f;
    int irad = 0, area = 0;
    box = TileBox{0, 0, 0, 0};
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
    o.xys[2 * g] = cx;
    o.xys[2 * g + 1] = cy;
    o.depths[g] = 0.0f;
    o.radii[g] = irad;
    o.conics[3 * g] = ca;
    o.conics[3 * g + 1] = cb;
    o.conics[3 * g + 2] = cc;
    o.num_tiles_hit[g] = area;
    const bool listed = irad > 0 && area > 0;
    if (listed && (box.y1 <= row0 || box.y0 >= row1)) return false;  // outside the band
    // the support bound serves the list culling and the record alike
    const float *opacity = rs.rec != nullptr ? rs.opacity : cull.opacity;
    Record rec = Record();
    int kind = 2;
    if (opacity != nullptr && listed) {
        rec = make_record(cx, cy, ca, cb, cc, opacity[g], rs.alpha_cap, cull.alpha_min, cr, cg,
                          cb_);
        kind = (__float_as_int(rec.r2.y) >> 1) & 3;
    }
    if (rs.rec != nullptr) {
        float4 *dst = reinterpret_cast<float4 *>(rs.rec + (size_t)g * REC_FLOATS);
        dst[0] = rec.r0;
        dst[1] = rec.r1;
        dst[2] = rec.r2;
    }
    if (want_lists && listed && cull.opacity != nullptr)
        cull_tile_box(box, cx, cy, kind, rec.r2.z, rec.r2.w, cull.pixel_offset);
    return listed;
}

__device__ __forceinline__ void project_tail(int g, float mx, float my, float sxx, float sxy,
                                             float syy, int H, int W, int tiles_x, int tiles_y,
                                             float center_bias, const ProjOut &o,
                                             const TileLists &tl, const ListCull &cull,
                                             const RecordSink &rs, float cr, float cg, float cb_)
{
    TileBox box;
    const bool fill = tl.tile_count != nullptr;
    const bool listed = project_body(g, mx, my, sxx, sxy, syy, H, W, tiles_x, tiles_y, center_bias,
                                     o, fill, cull, rs, cr, cg, cb_, box, fill ? tl.row0 : 0,
                                     fill ? tl.row1 : INT_MAX);
    if (fill && listed) emit_tiles(g, box, tl.row0, tl.row1, tiles_x, tl.tile_count, tl.seg, tl.C);
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
