/*
 * gi2d_oracle.c -- CPU restatement of the accumulated-summation 2D Gaussian
 * rasterizer path (project -> tile binning -> rasterize_sum, forward+backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link
 * or call this file; it is the checker for the HIP kernels and the CPU timing
 * baseline of bench.py, nothing else.
 *
 * PARITY UNPINNED.  The reference calls these operators from an un-vendored
 * `gsplat` submodule (/root/reference/.gitmodules:1-3; the directory is empty)
 * and ships no tests, fixtures or golden vectors for them.  This file therefore
 * follows the reference's call sites and in-repo conventions:
 *   - operator signatures and outputs: gaussianimage_cholesky.py:77,81-85,
 *     gaussianimage_rs.py:75-79
 *   - lower-triangular L = [[l11,0],[l21,l22]], Sigma = L L^T: utils.py:123-128
 *   - R S convention, Sigma = R S S^T R^T:                      utils.py:96-121
 *   - xys in pixel units, x = column, y = row:
 *     gaussianimage_cholesky_wMask.py:342-347, train.py:184,197
 *   - accumulated summation C = sum_n c_n * o_n * exp(-sigma_n): README.md:38
 * and fixes every convention the reference cannot settle as an explicit field
 * of gi2d_conv (see oracle_default_conv).  Equality with the fork's CUDA output
 * cannot be demonstrated offline.
 *
 * Arithmetic is written with an explicit operation order and no fused
 * multiply-add (build with -ffp-contract=off) so that the integer outputs
 * (radii, tile boxes, tile lists) can be matched bit for bit by the GPU code.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Sine and cosine of the rotation-scale projection: the product's own header, included as text.
 * The integer outputs (radius, tile box) hinge on the last bits of sin/cos, and libm's differ
 * from the device library's; both sides therefore evaluate this one fixed sequence of IEEE
 * operations.  tests/test_oracle.py checks it against libm's double-precision functions. */
#include "../gaussianimage_amd/csrc/sincos_f32.h"

#define GI2D_BLOCK 16

typedef struct {
    float pixel_offset; /* sample point of pixel (j,i) is (j+off, i+off)        */
    float center_bias;  /* xys = 0.5*W*x + 0.5*W + bias (same for y with H)     */
    float alpha_cap;    /* alpha = min(cap, opacity*exp(-sigma))                */
    float alpha_min;    /* pairs with alpha < alpha_min are skipped             */
    float bg_weight;    /* out += bg_weight * background (0: background ignored)*/
} gi2d_conv;

void oracle_default_conv(gi2d_conv *c)
{
    c->pixel_offset = 0.0f;
    c->center_bias = 0.0f;
    c->alpha_cap = 1.0f;
    c->alpha_min = 1.0f / 255.0f;
    c->bg_weight = 0.0f;
}

/* float -> int, truncating toward zero, saturating, NaN -> 0 */
static int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483520.0f) return 2147483520;
    if (v <= -2147483520.0f) return -2147483520;
    return (int)v;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* Conic (inverse covariance) and integer 3-sigma radius of a 2x2 covariance.
 * Returns 0 (early-out) for a zero or non-finite determinant. */
static int cov_to_conic_radius(float sxx, float sxy, float syy, float conic[3], float *radius)
{
    float det = sxx * syy - sxy * sxy;
    if (det == 0.0f || !(fabsf(det) <= 3.0e38f)) return 0;
    float inv_det = 1.0f / det;
    conic[0] = syy * inv_det;
    conic[1] = -sxy * inv_det;
    conic[2] = sxx * inv_det;
    float b = 0.5f * (sxx + syy);
    float disc = fmaxf(0.1f, b * b - det);
    float s = sqrtf(disc);
    float v1 = b + s;
    float v2 = b - s;
    float r = ceilf(3.0f * sqrtf(fmaxf(v1, v2)));
    if (!(r <= 3.0e38f)) return 0;
    *radius = r;
    return 1;
}

/* Tile box [x0,x1) x [y0,y1) touched by a disc of `radius` pixels at (cx,cy). */
void oracle_tile_box(float cx, float cy, float radius, int tiles_x, int tiles_y, int box[4])
{
    float tcx = cx / (float)GI2D_BLOCK, tcy = cy / (float)GI2D_BLOCK;
    float tr = radius / (float)GI2D_BLOCK;
    box[0] = imin(imax(0, f2i_sat(tcx - tr)), tiles_x);
    box[1] = imin(imax(0, f2i_sat(tcx + tr + 1.0f)), tiles_x);
    box[2] = imin(imax(0, f2i_sat(tcy - tr)), tiles_y);
    box[3] = imin(imax(0, f2i_sat(tcy + tr + 1.0f)), tiles_y);
}

static void project_tail(int g, float mx, float my, float sxx, float sxy, float syy, int H, int W,
                         int tiles_x, int tiles_y, const gi2d_conv *cv, float *xys, float *depths,
                         int *radii, float *conics, int *num_tiles_hit)
{
    xys[2 * g] = xys[2 * g + 1] = 0.0f;
    depths[g] = 0.0f;
    radii[g] = 0;
    conics[3 * g] = conics[3 * g + 1] = conics[3 * g + 2] = 0.0f;
    num_tiles_hit[g] = 0;
    float conic[3], radius;
    if (!cov_to_conic_radius(sxx, sxy, syy, conic, &radius)) return;
    float cx = 0.5f * (float)W * mx + 0.5f * (float)W + cv->center_bias;
    float cy = 0.5f * (float)H * my + 0.5f * (float)H + cv->center_bias;
    conics[3 * g] = conic[0];
    conics[3 * g + 1] = conic[1];
    conics[3 * g + 2] = conic[2];
    xys[2 * g] = cx;
    xys[2 * g + 1] = cy;
    radii[g] = f2i_sat(radius);
    int box[4];
    oracle_tile_box(cx, cy, radius, tiles_x, tiles_y, box);
    int area = (box[1] - box[0]) * (box[3] - box[2]);
    if (area <= 0) return;
    num_tiles_hit[g] = area;
}

/* project_gaussians_2d (gaussianimage_cholesky.py:77): means in (-1,1) NDC,
 * L_elements = (l11, l21, l22). */
void oracle_project_cholesky_fwd(int N, const float *means, const float *L, int H, int W,
                                 int tiles_x, int tiles_y, const gi2d_conv *cv, float *xys,
                                 float *depths, int *radii, float *conics, int *num_tiles_hit)
{
#pragma omp parallel for schedule(static)
    for (int g = 0; g < N; ++g) {
        float l11 = L[3 * g], l21 = L[3 * g + 1], l22 = L[3 * g + 2];
        float sxx = l11 * l11;
        float sxy = l11 * l21;
        float syy = l21 * l21 + l22 * l22;
        project_tail(g, means[2 * g], means[2 * g + 1], sxx, sxy, syy, H, W, tiles_x, tiles_y, cv,
                     xys, depths, radii, conics, num_tiles_hit);
    }
}

/* project_gaussians_2d_scale_rot (gaussianimage_rs.py:75): scales (s1,s2),
 * rotation theta in radians; Sigma = R diag(s)^2 R^T, R = [[c,-s],[s,c]]. */
void oracle_project_rs_fwd(int N, const float *means, const float *scales, const float *rot, int H,
                           int W, int tiles_x, int tiles_y, const gi2d_conv *cv, float *xys,
                           float *depths, int *radii, float *conics, int *num_tiles_hit)
{
#pragma omp parallel for schedule(static)
    for (int g = 0; g < N; ++g) {
        float c, s;
        gi2d_sincos_f32(rot[g], &s, &c);
        float s1 = scales[2 * g], s2 = scales[2 * g + 1];
        float q1 = s1 * s1, q2 = s2 * s2;
        float sxx = c * c * q1 + s * s * q2;
        float sxy = c * s * (q1 - q2);
        float syy = s * s * q1 + c * c * q2;
        project_tail(g, means[2 * g], means[2 * g + 1], sxx, sxy, syy, H, W, tiles_x, tiles_y, cv,
                     xys, depths, radii, conics, num_tiles_hit);
    }
}

/* Tile binning.  Gaussians with radius <= 0 are not binned; every other one is
 * entered in each tile of its box.  Lists are grouped by tile (row-major tile
 * id) and ascending in Gaussian id inside a tile.  tile_offsets has T+1
 * entries.  Returns the number of intersections; ids may be NULL to count. */
int64_t oracle_bin_tiles(int N, const float *xys, const int *radii, int tiles_x, int tiles_y,
                         int *tile_offsets, int *ids, int64_t ids_capacity)
{
    int T = tiles_x * tiles_y;
    int *cnt = (int *)calloc((size_t)T + 1, sizeof(int));
    /* Threads own disjoint bands of tile rows and each walks all Gaussians in ascending order:
     * the lists are those of the serial loop, whatever the thread count. */
#pragma omp parallel
    {
        int nt = omp_get_num_threads(), me = omp_get_thread_num();
        int r0 = (int)((int64_t)tiles_y * me / nt), r1 = (int)((int64_t)tiles_y * (me + 1) / nt);
        for (int g = 0; g < N && r0 < r1; ++g) {
            if (radii[g] <= 0) continue;
            int box[4];
            oracle_tile_box(xys[2 * g], xys[2 * g + 1], (float)radii[g], tiles_x, tiles_y, box);
            for (int ty = imax(box[2], r0); ty < imin(box[3], r1); ++ty)
                for (int tx = box[0]; tx < box[1]; ++tx) cnt[ty * tiles_x + tx]++;
        }
    }
    int64_t total = 0;
    for (int t = 0; t < T; ++t) {
        tile_offsets[t] = (int)total;
        total += cnt[t];
    }
    tile_offsets[T] = (int)total;
    if (ids && total <= ids_capacity) {
        memset(cnt, 0, sizeof(int) * (size_t)T);
#pragma omp parallel
        {
            int nt = omp_get_num_threads(), me = omp_get_thread_num();
            int r0 = (int)((int64_t)tiles_y * me / nt);
            int r1 = (int)((int64_t)tiles_y * (me + 1) / nt);
            for (int g = 0; g < N && r0 < r1; ++g) {
                if (radii[g] <= 0) continue;
                int box[4];
                oracle_tile_box(xys[2 * g], xys[2 * g + 1], (float)radii[g], tiles_x, tiles_y, box);
                for (int ty = imax(box[2], r0); ty < imin(box[3], r1); ++ty)
                    for (int tx = box[0]; tx < box[1]; ++tx) {
                        int t = ty * tiles_x + tx;
                        ids[tile_offsets[t] + cnt[t]++] = g;
                    }
            }
        }
    }
    free(cnt);
    return total;
}

/* One pixel/Gaussian pair.  Returns 1 when the pair contributes. */
static inline int pair_eval(float px, float py, float gx, float gy, float a, float b, float c,
                            float opac, const gi2d_conv *cv, float *dx_, float *dy_, float *e_,
                            float *alpha_)
{
    float dx = gx - px, dy = gy - py;
    float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
    float e = expf(-sigma);
    float alpha = fminf(cv->alpha_cap, opac * e);
    *dx_ = dx;
    *dy_ = dy;
    *e_ = e;
    *alpha_ = alpha;
    if (sigma < 0.0f || alpha < cv->alpha_min) return 0;
    return 1;
}

/* rasterize_gaussians_sum forward (gaussianimage_cholesky.py:81-85): out is
 * [H,W,3].  margin (optional, [H,W]) receives the smallest relative distance
 * of any pair of that pixel to a skip threshold; the parity tests use it to
 * set aside pixels whose value hinges on the last bits of exp(). */
void oracle_rasterize_sum_fwd(int N, const float *xys, const float *conics, const float *colors,
                              const float *opacity, const float *background, int H, int W,
                              int tiles_x, int tiles_y, const int *tile_offsets, const int *ids,
                              const gi2d_conv *cv, float *out, float *margin)
{
    (void)N;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles_x * tiles_y; ++t) {
        int tx = t % tiles_x, ty = t / tiles_x;
        int beg = tile_offsets[t], end = tile_offsets[t + 1];
        for (int i = ty * GI2D_BLOCK; i < imin(H, (ty + 1) * GI2D_BLOCK); ++i)
            for (int j = tx * GI2D_BLOCK; j < imin(W, (tx + 1) * GI2D_BLOCK); ++j) {
                float px = (float)j + cv->pixel_offset, py = (float)i + cv->pixel_offset;
                float acc[3] = {0.0f, 0.0f, 0.0f};
                float mrg = 1.0e30f;
                for (int k = beg; k < end; ++k) {
                    int g = ids[k];
                    float dx, dy, e, alpha;
                    int on = pair_eval(px, py, xys[2 * g], xys[2 * g + 1], conics[3 * g],
                                       conics[3 * g + 1], conics[3 * g + 2], opacity[g], cv, &dx,
                                       &dy, &e, &alpha);
                    if (margin) {
                        float m = fabsf(opacity[g] * e - cv->alpha_min) / cv->alpha_min;
                        if (m < mrg) mrg = m;
                    }
                    if (!on) continue;
                    acc[0] = acc[0] + colors[3 * g] * alpha;
                    acc[1] = acc[1] + colors[3 * g + 1] * alpha;
                    acc[2] = acc[2] + colors[3 * g + 2] * alpha;
                }
                size_t p = (size_t)i * W + j;
                for (int ch = 0; ch < 3; ++ch)
                    out[3 * p + ch] = acc[ch] + cv->bg_weight * background[ch];
                if (margin) margin[p] = mrg;
            }
    }
}

/* rasterize_gaussians_sum backward: gradients of sum(out * v_out) with respect
 * to xys [N,2], conics [N,3], colors [N,3], opacity [N].  Derived from the
 * forward above; partial sums are kept in double. */
void oracle_rasterize_sum_bwd(int N, const float *xys, const float *conics, const float *colors,
                              const float *opacity, int H, int W, int tiles_x, int tiles_y,
                              const int *tile_offsets, const int *ids, const gi2d_conv *cv,
                              const float *v_out, float *v_xy, float *v_conic, float *v_colors,
                              float *v_opacity)
{
    double *acc = (double *)calloc((size_t)N * 9, sizeof(double));
    for (int t = 0; t < tiles_x * tiles_y; ++t) {
        int tx = t % tiles_x, ty = t / tiles_x;
        for (int k = tile_offsets[t]; k < tile_offsets[t + 1]; ++k) {
            int g = ids[k];
            float gx = xys[2 * g], gy = xys[2 * g + 1];
            float a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
            float opac = opacity[g];
            double *A = acc + (size_t)g * 9;
            for (int i = ty * GI2D_BLOCK; i < imin(H, (ty + 1) * GI2D_BLOCK); ++i)
                for (int j = tx * GI2D_BLOCK; j < imin(W, (tx + 1) * GI2D_BLOCK); ++j) {
                    float px = (float)j + cv->pixel_offset, py = (float)i + cv->pixel_offset;
                    float dx, dy, e, alpha;
                    if (!pair_eval(px, py, gx, gy, a, b, c, opac, cv, &dx, &dy, &e, &alpha))
                        continue;
                    const float *vo = v_out + 3 * ((size_t)i * W + j);
                    double v_alpha = (double)colors[3 * g] * vo[0] +
                                     (double)colors[3 * g + 1] * vo[1] +
                                     (double)colors[3 * g + 2] * vo[2];
                    A[5] += (double)alpha * vo[0];
                    A[6] += (double)alpha * vo[1];
                    A[7] += (double)alpha * vo[2];
                    if (opac * e > cv->alpha_cap) continue; /* capped: alpha is constant */
                    double v_sigma = -(double)opac * e * v_alpha;
                    A[8] += (double)e * v_alpha;
                    A[0] += v_sigma * ((double)a * dx + (double)b * dy);
                    A[1] += v_sigma * ((double)b * dx + (double)c * dy);
                    A[2] += v_sigma * 0.5 * (double)dx * dx;
                    A[3] += v_sigma * (double)dx * dy;
                    A[4] += v_sigma * 0.5 * (double)dy * dy;
                }
        }
    }
    for (int g = 0; g < N; ++g) {
        const double *A = acc + (size_t)g * 9;
        v_xy[2 * g] = (float)A[0];
        v_xy[2 * g + 1] = (float)A[1];
        v_conic[3 * g] = (float)A[2];
        v_conic[3 * g + 1] = (float)A[3];
        v_conic[3 * g + 2] = (float)A[4];
        v_colors[3 * g] = (float)A[5];
        v_colors[3 * g + 1] = (float)A[6];
        v_colors[3 * g + 2] = (float)A[7];
        v_opacity[g] = (float)A[8];
    }
    free(acc);
}

/* The shared sine / cosine, exported for the tests. */
void oracle_sincos_f32(int n, const float *theta, float *s, float *c)
{
    for (int i = 0; i < n; ++i) gi2d_sincos_f32(theta[i], s + i, c + i);
}

/* Gradient of the covariance from the gradient of its inverse (the conic):
 * with V = [[va, vb/2],[vb/2, vc]] and C = Sigma^-1, dL/dSigma = -C V C.
 * Returned as (gxx, gxy, gyy) where gxy is the derivative with respect to the
 * single off-diagonal parameter sxy. */
static void conic_grad_to_cov_grad(const double conic[3], const float *v_conic, double g[3])
{
    double a = conic[0], b = conic[1], c = conic[2];
    double va = v_conic[0], vb = 0.5 * v_conic[1], vc = v_conic[2];
    /* M = C V */
    double m00 = a * va + b * vb, m01 = a * vb + b * vc;
    double m10 = b * va + c * vb, m11 = b * vb + c * vc;
    /* G = -M C */
    double g00 = -(m00 * a + m01 * b), g01 = -(m00 * b + m01 * c);
    double g10 = -(m10 * a + m11 * b), g11 = -(m10 * b + m11 * c);
    g[0] = g00;
    g[1] = g01 + g10;
    g[2] = g11;
}

static int bwd_active(float sxx, float sxy, float syy)
{
    float conic[3], radius;
    return cov_to_conic_radius(sxx, sxy, syy, conic, &radius);
}

/* project_gaussians_2d backward: (v_xys, v_conics) -> (v_means, v_L). */
void oracle_project_cholesky_bwd(int N, const float *means, const float *L, int H, int W,
                                 const float *v_xy, const float *v_conic, float *v_means,
                                 float *v_L)
{
    (void)means;
#pragma omp parallel for schedule(static)
    for (int g = 0; g < N; ++g) {
        float l11 = L[3 * g], l21 = L[3 * g + 1], l22 = L[3 * g + 2];
        v_means[2 * g] = v_means[2 * g + 1] = 0.0f;
        v_L[3 * g] = v_L[3 * g + 1] = v_L[3 * g + 2] = 0.0f;
        if (!bwd_active(l11 * l11, l11 * l21, l21 * l21 + l22 * l22)) continue;
        double sxx = (double)l11 * l11, sxy = (double)l11 * l21;
        double syy = (double)l21 * l21 + (double)l22 * l22;
        double det = sxx * syy - sxy * sxy;
        double conic[3] = {syy / det, -sxy / det, sxx / det};
        double gs[3];
        conic_grad_to_cov_grad(conic, v_conic + 3 * g, gs);
        v_means[2 * g] = (float)(0.5 * W * (double)v_xy[2 * g]);
        v_means[2 * g + 1] = (float)(0.5 * H * (double)v_xy[2 * g + 1]);
        v_L[3 * g] = (float)(2.0 * l11 * gs[0] + l21 * gs[1]);
        v_L[3 * g + 1] = (float)(l11 * gs[1] + 2.0 * l21 * gs[2]);
        v_L[3 * g + 2] = (float)(2.0 * l22 * gs[2]);
    }
}

/* project_gaussians_2d_scale_rot backward: -> (v_means, v_scales, v_rot). */
void oracle_project_rs_bwd(int N, const float *means, const float *scales, const float *rot, int H,
                           int W, const float *v_xy, const float *v_conic, float *v_means,
                           float *v_scales, float *v_rot)
{
    (void)means;
    for (int g = 0; g < N; ++g) {
        v_means[2 * g] = v_means[2 * g + 1] = 0.0f;
        v_scales[2 * g] = v_scales[2 * g + 1] = 0.0f;
        v_rot[g] = 0.0f;
        float cf, sf;
        gi2d_sincos_f32(rot[g], &sf, &cf);
        float s1f = scales[2 * g], s2f = scales[2 * g + 1];
        float q1f = s1f * s1f, q2f = s2f * s2f;
        if (!bwd_active(cf * cf * q1f + sf * sf * q2f, cf * sf * (q1f - q2f),
                        sf * sf * q1f + cf * cf * q2f))
            continue;
        double c = cos((double)rot[g]), s = sin((double)rot[g]);
        double s1 = s1f, s2 = s2f, q1 = s1 * s1, q2 = s2 * s2;
        double sxx = c * c * q1 + s * s * q2, sxy = c * s * (q1 - q2);
        double syy = s * s * q1 + c * c * q2;
        double det = sxx * syy - sxy * sxy;
        double conic[3] = {syy / det, -sxy / det, sxx / det};
        double gs[3];
        conic_grad_to_cov_grad(conic, v_conic + 3 * g, gs);
        v_means[2 * g] = (float)(0.5 * W * (double)v_xy[2 * g]);
        v_means[2 * g + 1] = (float)(0.5 * H * (double)v_xy[2 * g + 1]);
        double gq1 = gs[0] * c * c + gs[1] * c * s + gs[2] * s * s;
        double gq2 = gs[0] * s * s - gs[1] * c * s + gs[2] * c * c;
        v_scales[2 * g] = (float)(2.0 * s1 * gq1);
        v_scales[2 * g + 1] = (float)(2.0 * s2 * gq2);
        /* d/dtheta: sxx' = -2cs(q1-q2), sxy' = (c^2-s^2)(q1-q2), syy' = 2cs(q1-q2) */
        double dq = q1 - q2;
        v_rot[g] = (float)(dq * (-2.0 * c * s * gs[0] + (c * c - s * s) * gs[1] +
                                 2.0 * c * s * gs[2]));
    }
}
