// gi2d.hip -- MI355X (gfx950, wave64) kernels and C ABI of the accumulated-summation
// 2D Gaussian rasterizer: projection, tile binning, rasterize_sum forward/backward, Adan step.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see build.py).
// -ffp-contract=off is part of the contract: the projection's integer outputs (radii, tile
// boxes) must not depend on the compiler's choice of fused multiply-adds.  Where a fused
// multiply-add is wanted in a hot loop it is written explicitly with __builtin_fmaf.
//
// Frame pipeline (forward): TWO launches in steady state.
//   1. project_*_fwd_kernel  -- per Gaussian: conic, radius, tile box, the API outputs, and for
//      every tile of the box one returning atomic on the tile's counter followed by a store of
//      the Gaussian id into the tile's segment (fixed capacity C ids per tile).
//   2. rasterize_sum_fwd_kernel<FINALIZE> -- per tile: sorts its segment (ids arrive in atomic
//      order; ascending order makes runs reproducible and lets the backward search lists),
//      publishes the list length, returns the counter to zero, then rasterises.
// Every launch costs ~2 us plus ~1-1.5 us per dependent memory round trip on this part, and a
// 768x512 frame with 30k Gaussians is only a few us of arithmetic, so the design minimises
// launches and dependent rounds rather than bytes.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <new>

#include "../../include/gi2d.h"

#define WAVE 64
// Workgroup size of the per-Gaussian kernels that issue scattered atomics: one wave, so that
// the launch spreads over every CU.
#define PG_THREADS 64
// Per-tile counters live one per 128-byte line: device-scope atomics on one line serialise at
// the memory side, and ~48 Gaussians per tile hammering 32 neighbouring counters of one line
// made the atomic pass three times slower.
#define CNT_STRIDE 32
#define LOG2E 1.4426950408889634f
#define PART_STRIDE 9  // backward partial sums per (tile, Gaussian): v_xy 2, v_conic 3, v_rgb 3, v_op 1
#define SLOT_TAB 4     // slots per Gaussian remembered by backward pass 1 (boxes are mostly <= 2x2)

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

// ------------------------------------------------------------------------------------------
// projection and tile binning (one lane per Gaussian)
// ------------------------------------------------------------------------------------------

// Enters Gaussian g in the segment of every tile of its box (clipped to the band's tile rows
// [row0,row1)).  Four returning atomics are put in flight before the first dependent store, so
// a lane's round trips overlap instead of adding up.  Ids beyond a segment's capacity are
// dropped; the counter keeps counting, which is how the overflow is noticed.
__device__ __forceinline__ void emit_tiles(int g, const TileBox &box, int row0, int row1,
                                           int tiles_x, int *tile_count, int *seg, int C)
{
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
    int tx = box.x0, ty = y0;
    for (int k0 = 0; k0 < cnt; k0 += 4) {
        int t[4], pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t[u] = (k0 + u < cnt) ? (ty - row0) * tiles_x + tx : -1;
            if (++tx >= box.x1) {
                tx = box.x0;
                ++ty;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pos[u] = C;
            if (t[u] >= 0) pos[u] = atomicAdd(&tile_count[(size_t)t[u] * CNT_STRIDE], 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pos[u] < C) seg[(size_t)t[u] * C + pos[u]] = g;
    }
}

// At most one lane per launch: copies the device status to pinned host memory, where the next
// calls read it without synchronising.  Issued first so that the slow host write overlaps the
// rest of the kernel.
__device__ __forceinline__ void publish_status(const DevStatus *status, volatile long long *h,
                                               long long generation)
{
    h[0] = status->max_count;
    h[2] = status->overflow_gen;
    h[1] = generation;
}

__device__ __forceinline__ void project_tail(int g, float mx, float my, float sxx, float sxy,
                                             float syy, int H, int W, int tiles_x, int tiles_y,
                                             float center_bias, float *xys, float *depths,
                                             int *radii, float *conics, int *num_tiles_hit,
                                             int *tile_count, int *seg, int C, int row0, int row1)
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
    if (tile_count != nullptr && irad > 0 && area > 0)
        emit_tiles(g, box, row0, row1, tiles_x, tile_count, seg, C);
}

// RAW = the model's raw parameters come in and its activations are applied here:
// means = tanh(_xyz), L = _cholesky + (0.5, 0, 0.5) (gaussianimage_cholesky.py:62,74,42).
template <bool RAW>
__global__ __launch_bounds__(256) void project_cholesky_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ L, int N, int H, int W, int tiles_x,
    int tiles_y, float center_bias, float *__restrict__ xys, float *__restrict__ depths,
    int *__restrict__ radii, float *__restrict__ conics, int *__restrict__ num_tiles_hit,
    int *tile_count, int *seg, int C, int row0, int row1, const DevStatus *status,
    volatile long long *h_status, long long generation)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && tile_count != nullptr) publish_status(status, h_status, generation);
    if (g >= N) return;
    float l11 = L[3 * g], l21 = L[3 * g + 1], l22 = L[3 * g + 2];
    float mx = means[2 * g], my = means[2 * g + 1];
    if (RAW) {
        mx = tanhf(mx);
        my = tanhf(my);
        l11 = l11 + 0.5f;
        l22 = l22 + 0.5f;
    }
    float sxx = l11 * l11;
    float sxy = l11 * l21;
    float syy = l21 * l21 + l22 * l22;
    project_tail(g, mx, my, sxx, sxy, syy, H, W, tiles_x, tiles_y, center_bias, xys, depths, radii,
                 conics, num_tiles_hit, tile_count, seg, C, row0, row1);
}

// RAW: means = tanh(_xyz), scales = |_scaling + 0.5|, rotation = sigmoid(_rotation) * 2 * pi
// (gaussianimage_rs.py:55-64).
template <bool RAW>
__global__ __launch_bounds__(256) void project_rs_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ scales,
    const float *__restrict__ rot, int N, int H, int W, int tiles_x, int tiles_y,
    float center_bias, float *__restrict__ xys, float *__restrict__ depths,
    int *__restrict__ radii, float *__restrict__ conics, int *__restrict__ num_tiles_hit,
    int *tile_count, int *seg, int C, int row0, int row1, const DevStatus *status,
    volatile long long *h_status, long long generation)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && tile_count != nullptr) publish_status(status, h_status, generation);
    if (g >= N) return;
    float theta = rot[g];
    float s1 = scales[2 * g], s2 = scales[2 * g + 1];
    float mx = means[2 * g], my = means[2 * g + 1];
    if (RAW) {
        mx = tanhf(mx);
        my = tanhf(my);
        s1 = fabsf(s1 + 0.5f);
        s2 = fabsf(s2 + 0.5f);
        theta = (1.0f / (1.0f + expf(-theta))) * 2.0f * 3.14159274f;
    }
    float s, c;
    sincosf(theta, &s, &c);
    float q1 = s1 * s1, q2 = s2 * s2;
    float sxx = c * c * q1 + s * s * q2;
    float sxy = c * s * (q1 - q2);
    float syy = s * s * q1 + c * c * q2;
    project_tail(g, mx, my, sxx, sxy, syy, H, W, tiles_x, tiles_y, center_bias, xys, depths, radii,
                 conics, num_tiles_hit, tile_count, seg, C, row0, row1);
}

// ------------------------------------------------------------------------------------------
// Codec decode (decompress_wo_ec, gaussianimage_cholesky.py:161-166 / gaussianimage_rs.py:132-139):
//   means    = tanh(float(xyz_half))
//   cholesky = code * scale + beta, then + (0.5, 0, 0.5)       (quantize.py:86-87)
//   scaling  = |code * scale + beta + 0.5|,  rotation = code * scale + beta
//   colors   = sum over the residual-VQ layers q of codebook[q][index[g][q]]   (quantize.py:146-150)
// Each product, sum and bound is rounded on its own, in the reference's order.  An index
// outside the codebook is clamped into it (a corrupt stream gives wrong colours, not a fault).
// ------------------------------------------------------------------------------------------
struct DecodedGaussian {
    float mx, my, a0, a1, a2, b0, col[3];
};

template <bool RS>
__device__ __forceinline__ DecodedGaussian decode_gaussian(const gi2d_codec &cd, int g)
{
    DecodedGaussian d;
    const __half2 h = reinterpret_cast<const __half2 *>(cd.xyz_half)[g];
    d.mx = tanhf(__low2float(h));
    d.my = tanhf(__high2float(h));
    if (RS) {
        d.a0 = fabsf((cd.cov_a_code[2 * g] * cd.cov_a_scale[0] + cd.cov_a_beta[0]) + 0.5f);
        d.a1 = fabsf((cd.cov_a_code[2 * g + 1] * cd.cov_a_scale[1] + cd.cov_a_beta[1]) + 0.5f);
        d.a2 = 0.f;
        d.b0 = cd.cov_b_code[g] * cd.cov_b_scale[0] + cd.cov_b_beta[0];
    } else {
        d.a0 = (cd.cov_a_code[3 * g] * cd.cov_a_scale[0] + cd.cov_a_beta[0]) + 0.5f;
        d.a1 = (cd.cov_a_code[3 * g + 1] * cd.cov_a_scale[1] + cd.cov_a_beta[1]) + 0.0f;
        d.a2 = (cd.cov_a_code[3 * g + 2] * cd.cov_a_scale[2] + cd.cov_a_beta[2]) + 0.5f;
        d.b0 = 0.f;
    }
    d.col[0] = d.col[1] = d.col[2] = 0.f;
    for (int q = 0; q < cd.num_quantizers; ++q) {
        int k = cd.color_index[(size_t)g * cd.num_quantizers + q];
        k = min(max(k, 0), cd.codebook_size - 1);
        const float *e = cd.codebooks + ((size_t)q * cd.codebook_size + k) * 3;
        d.col[0] += e[0];
        d.col[1] += e[1];
        d.col[2] += e[2];
    }
    return d;
}

// The decoded operator inputs as tensors (what the reference's torch expressions produce).
template <bool RS>
__global__ __launch_bounds__(256) void decode_params_kernel(gi2d_codec cd, int N,
                                                            float *__restrict__ means,
                                                            float *__restrict__ cov_a,
                                                            float *__restrict__ cov_b,
                                                            float *__restrict__ colors)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const DecodedGaussian d = decode_gaussian<RS>(cd, g);
    means[2 * g] = d.mx;
    means[2 * g + 1] = d.my;
    if (RS) {
        cov_a[2 * g] = d.a0;
        cov_a[2 * g + 1] = d.a1;
        cov_b[g] = d.b0;
    } else {
        cov_a[3 * g] = d.a0;
        cov_a[3 * g + 1] = d.a1;
        cov_a[3 * g + 2] = d.a2;
    }
    colors[3 * g] = d.col[0];
    colors[3 * g + 1] = d.col[1];
    colors[3 * g + 2] = d.col[2];
}

// Decode fused into the projection: the frame's first launch reads the coded attributes, leaves
// the decoded colours for the rasteriser and fills the tile lists.
template <bool RS>
__global__ __launch_bounds__(256) void project_decode_fwd_kernel(
    gi2d_codec cd, float *__restrict__ colors, int N, int H, int W, int tiles_x, int tiles_y,
    float center_bias, float *__restrict__ xys, float *__restrict__ depths,
    int *__restrict__ radii, float *__restrict__ conics, int *__restrict__ num_tiles_hit,
    int *tile_count, int *seg, int C, int row0, int row1, const DevStatus *status,
    volatile long long *h_status, long long generation)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && tile_count != nullptr) publish_status(status, h_status, generation);
    if (g >= N) return;
    const DecodedGaussian d = decode_gaussian<RS>(cd, g);
    colors[3 * g] = d.col[0];
    colors[3 * g + 1] = d.col[1];
    colors[3 * g + 2] = d.col[2];
    float sxx, sxy, syy;
    if (RS) {
        float s, c;
        sincosf(d.b0, &s, &c);
        const float q1 = d.a0 * d.a0, q2 = d.a1 * d.a1;
        sxx = c * c * q1 + s * s * q2;
        sxy = c * s * (q1 - q2);
        syy = s * s * q1 + c * c * q2;
    } else {
        sxx = d.a0 * d.a0;
        sxy = d.a0 * d.a1;
        syy = d.a1 * d.a1 + d.a2 * d.a2;
    }
    project_tail(g, d.mx, d.my, sxx, sxy, syy, H, W, tiles_x, tiles_y, center_bias, xys, depths,
                 radii, conics, num_tiles_hit, tile_count, seg, C, row0, row1);
}

// Binning from caller-made xys / radii (no projection in front of it).
__global__ __launch_bounds__(256) void bin_tiles_kernel(const float *__restrict__ xys,
                                                         const int *__restrict__ radii, int N,
                                                         int tiles_x, int tiles_y, int row0,
                                                         int row1, int *tile_count, int *seg, int C,
                                                         const DevStatus *status,
                                                         volatile long long *h_status,
                                                         long long generation)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) publish_status(status, h_status, generation);
    if (g >= N) return;
    const int r = radii[g];
    const float cx = xys[2 * g], cy = xys[2 * g + 1];
    if (r <= 0) return;
    TileBox box = tile_box(cx, cy, (float)r, tiles_x, tiles_y);
    emit_tiles(g, box, row0, row1, tiles_x, tile_count, seg, C);
}

__global__ __launch_bounds__(256) void zero_ints_kernel(int *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
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

__global__ __launch_bounds__(256) void project_cholesky_bwd_kernel(
    const float *__restrict__ L, int N, int H, int W, const float *__restrict__ v_xy,
    const float *__restrict__ v_conic, float *__restrict__ v_means, float *__restrict__ v_L)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float l11 = L[3 * g], l21 = L[3 * g + 1], l22 = L[3 * g + 2];
    float sxx = l11 * l11, sxy = l11 * l21, syy = l21 * l21 + l22 * l22;
    float a, b, c, radius;
    float vm0 = 0.f, vm1 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (cov_to_conic_radius(sxx, sxy, syy, a, b, c, radius)) {
        float gxx, gxy, gyy;
        conic_grad_to_cov_grad(a, b, c, v_conic[3 * g], v_conic[3 * g + 1], v_conic[3 * g + 2],
                               gxx, gxy, gyy);
        vm0 = 0.5f * (float)W * v_xy[2 * g];
        vm1 = 0.5f * (float)H * v_xy[2 * g + 1];
        o0 = 2.0f * l11 * gxx + l21 * gxy;
        o1 = l11 * gxy + 2.0f * l21 * gyy;
        o2 = 2.0f * l22 * gyy;
    }
    v_means[2 * g] = vm0;
    v_means[2 * g + 1] = vm1;
    v_L[3 * g] = o0;
    v_L[3 * g + 1] = o1;
    v_L[3 * g + 2] = o2;
}

__global__ __launch_bounds__(256) void project_rs_bwd_kernel(
    const float *__restrict__ scales, const float *__restrict__ rot, int N, int H, int W,
    const float *__restrict__ v_xy, const float *__restrict__ v_conic,
    float *__restrict__ v_means, float *__restrict__ v_scales, float *__restrict__ v_rot)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float s, c;
    sincosf(rot[g], &s, &c);
    float s1 = scales[2 * g], s2 = scales[2 * g + 1];
    float q1 = s1 * s1, q2 = s2 * s2;
    float sxx = c * c * q1 + s * s * q2;
    float sxy = c * s * (q1 - q2);
    float syy = s * s * q1 + c * c * q2;
    float a, b, cc, radius;
    float vm0 = 0.f, vm1 = 0.f, vs0 = 0.f, vs1 = 0.f, vr = 0.f;
    if (cov_to_conic_radius(sxx, sxy, syy, a, b, cc, radius)) {
        float gxx, gxy, gyy;
        conic_grad_to_cov_grad(a, b, cc, v_conic[3 * g], v_conic[3 * g + 1], v_conic[3 * g + 2],
                               gxx, gxy, gyy);
        vm0 = 0.5f * (float)W * v_xy[2 * g];
        vm1 = 0.5f * (float)H * v_xy[2 * g + 1];
        float gq1 = gxx * c * c + gxy * c * s + gyy * s * s;
        float gq2 = gxx * s * s - gxy * c * s + gyy * c * c;
        vs0 = 2.0f * s1 * gq1;
        vs1 = 2.0f * s2 * gq2;
        float dq = q1 - q2;
        vr = dq * (-2.0f * c * s * gxx + (c * c - s * s) * gxy + 2.0f * c * s * gyy);
    }
    v_means[2 * g] = vm0;
    v_means[2 * g + 1] = vm1;
    v_scales[2 * g] = vs0;
    v_scales[2 * g + 1] = vs1;
    v_rot[g] = vr;
}

// ------------------------------------------------------------------------------------------
// per-tile sorting
// ------------------------------------------------------------------------------------------

// ascending in-place bitonic network; elements past n are a virtual +infinity
template <typename P>
__device__ __forceinline__ void bitonic_sort(P a, int n, int lane, int nthreads)
{
    int npow = 1;
    while (npow < n) npow <<= 1;
    const int half = npow >> 1;
    for (int k = 2; k <= npow; k <<= 1) {
        const int hk = k >> 1;
        for (int i = lane; i < half; i += nthreads) {
            int blk = i / hk, off = i - blk * hk;
            int lo = blk * k + off, hi = blk * k + k - 1 - off;
            if (hi < n) {
                int x = a[lo], y = a[hi];
                if (x > y) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
        __syncthreads();
        for (int j = hk >> 1; j >= 1; j >>= 1) {
            for (int i = lane; i < half; i += nthreads) {
                int lo = 2 * j * (i / j) + (i % j), hi = lo + j;
                if (hi < n) {
                    int x = a[lo], y = a[hi];
                    if (x > y) {
                        a[lo] = y;
                        a[hi] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Rank sort of up to 64*R distinct ints held R per lane: every value is broadcast once with
// v_readlane and each lane counts how many are smaller than its own.  No LDS, no barrier.
template <int R>
__device__ __forceinline__ void rank_sort_regs(int *ids, int beg, int n, int lane)
{
    int v[R], rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int idx = r * WAVE + lane;
        v[r] = idx < n ? ids[beg + idx] : INT_MAX;
        rank[r] = 0;
    }
#pragma unroll
    for (int r2 = 0; r2 < R; ++r2) {
        const int lim = __builtin_amdgcn_readfirstlane(min(WAVE, n - r2 * WAVE));
        for (int k = 0; k < lim; ++k) {
            const int sv = __builtin_amdgcn_readlane(v[r2], k);
#pragma unroll
            for (int r = 0; r < R; ++r) rank[r] += (sv < v[r]) ? 1 : 0;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * WAVE + lane < n) ids[beg + rank[r]] = v[r];
}


#define SORT_LDS_MAX 4096

// What a tile does once per binning, whoever gets there first (the standalone kernel below or
// the first rasterisation): publish the list length, return the counter to zero for the next
// frame, and note a nearly full or overflowed segment for the capacity policy.
__device__ __forceinline__ void finalize_tile_words(int t, int n_raw, int C, int *tile_count,
                                                    int *tile_len, DevStatus *status,
                                                    long long generation)
{
    tile_len[t] = n_raw;
    tile_count[(size_t)t * CNT_STRIDE] = 0;
    // Raising the published maximum is rare: only lists past half the capacity try, and only
    // those above the value already there reach the atomic (thousands of tiles hammering one
    // word would otherwise serialise at the memory side).
    if (2 * n_raw > C && n_raw > status->max_count) atomicMax(&status->max_count, n_raw);
    if (n_raw > C) status->overflow_gen = generation;
}

// One wave per tile: puts the tile's ids in ascending order and finalises the tile.  The
// launch lasts as long as its slowest tile, so lists up to 256 entries stay on the register
// path; longer ones use a bitonic network in LDS, beyond the LDS budget in place in memory.
// Used when nothing rasterises right after a binning (backward first, list read-back, the
// synchronous capacity check); otherwise the forward rasteriser does this work itself.
__global__ __launch_bounds__(WAVE) void finalize_tiles_kernel(int *tile_count, int *tile_len,
                                                              int *seg, int C, DevStatus *status,
                                                              long long generation)
{
    __shared__ int s_ids[SORT_LDS_MAX];
    const int t = blockIdx.x, lane = threadIdx.x;
    const int n_raw = tile_count[(size_t)t * CNT_STRIDE];
    const int n = min(n_raw, C);
    int *ids = seg + (size_t)t * C;
    if (n > 1) {
        if (n <= WAVE) {
            rank_sort_regs<1>(ids, 0, n, lane);
        } else if (n <= 2 * WAVE) {
            rank_sort_regs<2>(ids, 0, n, lane);
        } else if (n <= 4 * WAVE) {
            rank_sort_regs<4>(ids, 0, n, lane);
        } else if (n <= SORT_LDS_MAX) {
            for (int i = lane; i < n; i += WAVE) s_ids[i] = ids[i];
            __syncthreads();
            bitonic_sort(s_ids, n, lane, WAVE);
            for (int i = lane; i < n; i += WAVE) ids[i] = s_ids[i];
        } else {
            bitonic_sort(ids, n, lane, WAVE);
        }
    }
    if (lane == 0) finalize_tile_words(t, n_raw, C, tile_count, tile_len, status, generation);
}

// Copies every finalised list to its place in the compact id array (one wave per tile).
__global__ __launch_bounds__(WAVE) void compact_lists_kernel(const int *__restrict__ seg, int C,
                                                             const int *__restrict__ tile_len,
                                                             const int *__restrict__ offsets,
                                                             int *__restrict__ ids)
{
    const int t = blockIdx.x;
    const int n = min(tile_len[t], C), off = offsets[t];
    for (int k = threadIdx.x; k < n; k += WAVE) ids[off + k] = seg[(size_t)t * C + k];
}

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

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share one).  This maps the
// launch index to a tile index so that each XCD works on a contiguous run of tiles: neighbouring
// tiles share Gaussians (L2 hits on the gathers) and image cache lines.  Bijective for any T.
__device__ __forceinline__ int xcd_tile(int bid, int T)
{
    const int q = T >> 3, r = T & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// ------------------------------------------------------------------------------------------
// rasterize_sum forward: one 256-lane workgroup per 16x16 tile, one wave per 8x8 quadrant, one
// pixel per lane.  The tile's Gaussians are staged through LDS in batches of 256; while staging,
// each entry's support rectangle yields a 4-bit quadrant mask, a ballot turns the masks of 64
// entries into one 64-bit scalar per wave, and the wave walks only the set bits (scalar bit
// scan), reading each surviving entry's record from LDS as a broadcast.
//
// FINALIZE = true (first pass after a binning): the workgroup first puts its segment in
// ascending id order -- for a list of up to 256 ids that is a rank computed from LDS broadcasts
// while the records are staged, so it costs no extra round trip to memory -- writes the sorted
// ids back for the later passes and the backward, and finalises the tile's words.
// ------------------------------------------------------------------------------------------

// Tile counts from which the forward rasteriser switches to 2 / 4 pixels per lane (measured on
// MI355X: below ~2 resident workgroups per CU slot the 4-wave layout is faster).
#define GI2D_PPL2_MIN_TILES 8192
#define GI2D_PPL4_MIN_TILES INT_MAX  // never chosen automatically: 2 was the best large-frame shape
#define FWD_LDS_INTS 2816  // 11 KB of LDS: the staging records, or the sort buffer of a long list

// L2 = true adds the fitting loop's epilogue (gaussianimage_cholesky.py:83-85,104 with
// utils.py:24-25): clamp to [0,1], L2 loss against the target (stored [3,H,W] like the model's
// gt_image), and the gradient of the mean-squared error with respect to the unclamped image
// written to `v_img` [H,W,3]; each tile leaves its sum of squared errors in loss_part[tile].
//
// SETS > 1 renders further colour sets of the same Gaussians in the same pass (the model's
// forward rasterises three colour tensors over one projection,
// gaussianimage_cholesky.py:81-97): the geometry, the culling and the alpha of a (pixel,
// Gaussian) pair are computed once and feed 3*SETS accumulators.
struct ExtraSets {
    const float *colors[2];  // [N,3] each
    float *out[2];           // [H,W,3] each
    // optional second form of the first image: clamped to [0,1] and planar, [3][rows][W] -- the
    // model's "render" tensor (gaussianimage_cholesky.py:83-84) without the torch ops behind it
    float *out_chw;
    size_t chw_plane;        // rows * W of the band being rendered
};

//
// PPL (pixels per lane) = 1, 2 or 4 shrinks the workgroup to 4, 2 or 1 waves, each wave
// rendering PPL quadrants.  A frame of many sparse tiles is bound by the chain of dependent
// memory rounds of each workgroup, not by arithmetic; smaller workgroups put 2x / 4x as many
// tiles in flight per CU.  Every pixel still adds its Gaussians in list order, so all variants
// give the same bits.
template <bool FINALIZE, bool L2, int SETS, int PPL>
__global__ __launch_bounds__(256 / PPL) void rasterize_sum_fwd_kernel(
    int *tile_count, int *tile_len, int *seg, int C, const float *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, float bg0, float bg1, float bg2, int H, int W, int tiles_x,
    int row0, int T, gi2d_conv cv, int cull, DevStatus *status, long long generation,
    float *__restrict__ out, const float *__restrict__ gt, float *__restrict__ v_img,
    float *__restrict__ loss_part, ExtraSets extra)
{
    static_assert(SETS >= 1 && SETS <= 3 && !(L2 && SETS > 1), "unsupported variant");
    static_assert(PPL == 1 || ((PPL == 2 || PPL == 4) && SETS == 1), "unsupported variant");
    constexpr int BATCH = 256 / PPL;            // lanes of the workgroup = entries staged at once
    constexpr int NX = 3 * (SETS - 1);          // extra colour floats per entry
    constexpr int NX4 = (NX + 3) / 4;           // ... in float4 records
    __shared__ __attribute__((aligned(16))) int smem[FWD_LDS_INTS];
    __shared__ float4 s_x[NX4 ? NX4 * 256 : 1];  // extra sets: record q of entry k at [q*256+k]
    float4 *s_a = reinterpret_cast<float4 *>(smem);         // gx, gy, -0.5*a*log2e, -b*log2e
    float4 *s_b = reinterpret_cast<float4 *>(smem + 1024);  // -0.5*c*log2e, opacity, r, g
    float *s_c = reinterpret_cast<float *>(smem + 2048);    // b
    int *s_m = smem + 2304;                                 // quadrant mask (bit = 2*qy + qx)
    int *s_id = smem + 2560;                                // ids of the batch (rank sort)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_tile(blockIdx.x, T);
    const int tx = t % tiles_x, ty = row0 + t / tiles_x;
    // this wave's quadrants are wave*PPL .. wave*PPL+PPL-1; a lane is the same pixel of each
    float px[PPL], py[PPL], r[PPL], g[PPL], b[PPL];
    size_t pix[PPL];
    bool inside[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        const int q = wave * PPL + u;
        const int lx = ((q & 1) << 3) + (lane & 7), ly = ((q >> 1) << 3) + (lane >> 3);
        const int j = tx * GI2D_BLOCK + lx, i = ty * GI2D_BLOCK + ly;
        inside[u] = (i < H) && (j < W);
        px[u] = (float)j + cv.pixel_offset;
        py[u] = (float)i + cv.pixel_offset;
        pix[u] = (size_t)(i - row0 * GI2D_BLOCK) * W + j;
        r[u] = g[u] = b[u] = 0.f;
    }
    int *lst = seg + (size_t)t * C;
    // the first batch's id is loaded before the list length is known (a segment always has C
    // words): one round trip to memory instead of two
    int gid_first = tid < C ? lst[tid] : 0;
    const int n_raw = FINALIZE ? tile_count[(size_t)t * CNT_STRIDE] : tile_len[t];
    const int n_all = min(n_raw, C);
    float xacc[NX ? NX : 1];
#pragma unroll
    for (int q = 0; q < NX; ++q) xacc[q] = 0.f;
    if (FINALIZE) {
        __syncthreads();  // every wave has read the counter before lane 0 returns it to zero
        if (n_all > BATCH) {  // long list: sort it where it lies before the batches are read
            if (n_all <= FWD_LDS_INTS) {
                for (int k = tid; k < n_all; k += BATCH) smem[k] = lst[k];
                __syncthreads();
                bitonic_sort(smem, n_all, tid, BATCH);
                for (int k = tid; k < n_all; k += BATCH) lst[k] = smem[k];
            } else {
                bitonic_sort(lst, n_all, tid, BATCH);
            }
            __syncthreads();
        }
        if (tid == 0) finalize_tile_words(t, n_raw, C, tile_count, tile_len, status, generation);
    }
    const int x_hi = min(GI2D_BLOCK, W - tx * GI2D_BLOCK) - 1;
    const int y_hi = min(GI2D_BLOCK, H - ty * GI2D_BLOCK) - 1;
    const float ox = (float)(tx * GI2D_BLOCK) + cv.pixel_offset;
    const float oy = (float)(ty * GI2D_BLOCK) + cv.pixel_offset;
    for (int base = 0; base < n_all; base += BATCH) {
        const int n = min(BATCH, n_all - base);
        __syncthreads();
        // (a long list was just re-sorted in place: the early copy of its first ids is stale)
        const bool early = base == 0 && !(FINALIZE && n_all > BATCH);
        const int gid = tid < n ? (early ? gid_first : lst[base + tid]) : INT_MAX;
        int slot = tid;
        if (FINALIZE && n_all <= BATCH) {
            // rank of my id among the batch = its place in ascending order (ids are distinct)
            s_id[tid] = gid;
            __syncthreads();
            if (tid < n) {  // (waves wholly past the list skip the counting)
                int rank = 0;
                for (int k = 0; k < n; k += 4) {
                    const int4 q = *reinterpret_cast<const int4 *>(s_id + k);
                    rank += (q.x < gid ? 1 : 0) + (q.y < gid ? 1 : 0) + (q.z < gid ? 1 : 0) +
                            (q.w < gid ? 1 : 0);
                }
                lst[rank] = gid;
                slot = rank;
            }
        }
        int mask = 0;
        if (tid < n) {
            const float gx = xys[2 * gid], gy = xys[2 * gid + 1];
            const float ca = conics[3 * gid], cb = conics[3 * gid + 1], cc = conics[3 * gid + 2];
            const float op = opacity[gid];
            s_a[slot] = make_float4(gx, gy, (-0.5f * LOG2E) * ca, (-LOG2E) * cb);
            s_b[slot] = make_float4((-0.5f * LOG2E) * cc, op, colors[3 * gid], colors[3 * gid + 1]);
            s_c[slot] = colors[3 * gid + 2];
            if (NX) {
                float xc[NX4 ? 4 * NX4 : 1];
#pragma unroll
                for (int q = 0; q < 4 * NX4; ++q)
                    xc[q] = q < NX ? extra.colors[q / 3][3 * gid + q % 3] : 0.f;
#pragma unroll
                for (int q = 0; q < NX4; ++q)
                    s_x[q * 256 + slot] =
                        make_float4(xc[4 * q], xc[4 * q + 1], xc[4 * q + 2], xc[4 * q + 3]);
            }
            int x0 = 0, x1 = x_hi, y0 = 0, y1 = y_hi;
            if (cull) support_rect(gx, gy, ca, cb, cc, op, cv.alpha_min, ox, oy, x0, x1, y0, y1);
            if (x0 <= x1 && y0 <= y1) {
                const int mx = (x0 < 8 ? 1 : 0) | (x1 >= 8 ? 2 : 0);
                const int my = (y0 < 8 ? 1 : 0) | (y1 >= 8 ? 2 : 0);
                mask = ((my & 1) ? mx : 0) | ((my & 2) ? (mx << 2) : 0);
            }
        }
        s_m[slot] = mask;  // lanes past the batch keep slot = tid >= n and store 0
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int m = s_m[c0 + lane];
#pragma unroll
            for (int u = 0; u < PPL; ++u) {
                unsigned long long bits = __ballot((m >> (wave * PPL + u)) & 1);
                while (bits) {
                    const int k = c0 + (int)__builtin_ctzll(bits);
                    bits &= bits - 1;
                    const float4 A = s_a[k];
                    const float4 B = s_b[k];
                    const float cb = s_c[k];
                    const float dx = A.x - px[u], dy = A.y - py[u];
                    // p = -sigma*log2(e) = (A'dx + B'dy)dx + C'dy*dy
                    const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                    const float p = __builtin_fmaf(B.x * dy, dy, q * dx);
                    const float e = __builtin_amdgcn_exp2f(p);
                    const float alpha = fminf(cv.alpha_cap, B.y * e);
                    const float w = (p > 0.f || alpha < cv.alpha_min) ? 0.f : alpha;
                    r[u] = __builtin_fmaf(B.z, w, r[u]);
                    g[u] = __builtin_fmaf(B.w, w, g[u]);
                    b[u] = __builtin_fmaf(cb, w, b[u]);
                    if (NX) {
#pragma unroll
                        for (int q4 = 0; q4 < NX4; ++q4) {
                            const float4 X = s_x[q4 * 256 + k];
                            const float xv[4] = {X.x, X.y, X.z, X.w};
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (4 * q4 + v < NX)
                                    xacc[4 * q4 + v] =
                                        __builtin_fmaf(xv[v], w, xacc[4 * q4 + v]);
                        }
                    }
                }
            }
        }
    }
    float sq[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        float rr = r[u] + cv.bg_weight * bg0;
        float gg = g[u] + cv.bg_weight * bg1;
        float bb = b[u] + cv.bg_weight * bg2;
        if (n_raw > C) rr = gg = bb = __builtin_nanf("");  // overflowed segment: never silently wrong
        if (inside[u] && out != nullptr) {
            float *o = out + 3 * pix[u];
            o[0] = rr;
            o[1] = gg;
            o[2] = bb;
        }
        if (inside[u] && extra.out_chw != nullptr) {
            float *o = extra.out_chw + pix[u];
            // torch.clamp keeps a NaN (the overflow marker must survive); fmin/fmax would drop it
            o[0] = rr != rr ? rr : fminf(fmaxf(rr, 0.f), 1.f);
            o[extra.chw_plane] = gg != gg ? gg : fminf(fmaxf(gg, 0.f), 1.f);
            o[2 * extra.chw_plane] = bb != bb ? bb : fminf(fmaxf(bb, 0.f), 1.f);
        }
        sq[u] = 0.f;
        if (L2 && inside[u]) {
            const size_t plane = (size_t)H * W;
            const size_t gp = pix[u] + (size_t)(row0 * GI2D_BLOCK) * W;  // index in the full frame
            const float scale = 2.0f / (3.0f * (float)plane);
            const float v[3] = {rr, gg, bb};
            float *vo = v_img + 3 * pix[u];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float x = v[ch];
                const float d = fminf(fmaxf(x, 0.f), 1.f) - gt[ch * plane + gp];
                sq[u] = __builtin_fmaf(d, d, sq[u]);
                vo[ch] = (x >= 0.f && x <= 1.f) ? scale * d : 0.f;  // clamp passes its interior
            }
        }
    }
    if (NX && inside[0]) {
        const float bgs[3] = {bg0, bg1, bg2};
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            float v = xacc[q] + cv.bg_weight * bgs[q % 3];
            if (n_raw > C) v = __builtin_nanf("");
            extra.out[q / 3][3 * pix[0] + q % 3] = v;
        }
    }
    if (L2) {
        // tile sum in a fixed order whatever the variant: the lanes of a quadrant by butterfly,
        // the four quadrants through LDS
        __syncthreads();
        float *s_red = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            float v = sq[u];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
            if (lane == 0) s_red[wave * PPL + u] = v;
        }
        __syncthreads();
        if (tid == 0) loss_part[t] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// ------------------------------------------------------------------------------------------
// rasterize_sum backward.
//
// Pass 1 (one workgroup per tile): a quad of lanes per list entry.  The quad walks only the
// pixels of the entry's support rectangle (bounding box of the ellipse alpha >= alpha_min),
// round-robin over its four lanes, keeps the entry's nine partial sums in registers, adds the
// four lanes with two DPP steps and writes the sums to the entry's own slot of a [slots][9]
// slab -- no cross-lane tree, no atomics -- and notes the slot in a small per-Gaussian table.
// Pass 2 (one lane per Gaussian): adds the Gaussian's slots in tile order; slots of the first
// SLOT_TAB tiles come from the table, the rest by binary search (lists ascend in Gaussian id).
// ------------------------------------------------------------------------------------------

// sum over the LPE (4, 8 or 16) neighbouring lanes of an entry's group, result in every lane
template <int LPE>
__device__ __forceinline__ float group_sum(float v)
{
    v = quad_sum(v);
    if (LPE >= 8) v = v + dpp_mov<0x141>(v);   // row_half_mirror: lane i <-> 7-i of each 8
    if (LPE >= 16) v = v + dpp_mov<0x140>(v);  // row_mirror: lane i <-> 15-i of each 16
    return v;
}

// The batches of one tile with LPE lanes per list entry (256 / LPE entries per batch).
template <int LPE>
__device__ __forceinline__ void bwd_tile(
    const int *__restrict__ lst, const int *s_first, int n_raw, size_t slot0, int tx, int ty,
    const float4 *s_v, const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0,
    const gi2d_conv &cv, int cull, float *__restrict__ partials, int *__restrict__ slot_tab)
{
    constexpr int EPB = 256 / LPE;  // entries per batch
    const int tid = threadIdx.x;
    const int x_hi = min(GI2D_BLOCK, W - tx * GI2D_BLOCK) - 1;
    const int y_hi = min(GI2D_BLOCK, H - ty * GI2D_BLOCK) - 1;
    const float ox = (float)(tx * GI2D_BLOCK) + cv.pixel_offset;
    const float oy = (float)(ty * GI2D_BLOCK) + cv.pixel_offset;
    const int q = tid & (LPE - 1);
    for (int base = 0; base < n_raw; base += EPB) {
        const int e = base + tid / LPE;
        float acc[PART_STRIDE];
#pragma unroll
        for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
        int gid = 0, rad = 0;
        float gx = 0.f, gy = 0.f;
        if (e < n_raw) {
            gid = base == 0 ? s_first[e] : lst[e];  // (the first 64 ids are already here)
            gx = xys[2 * gid];
            gy = xys[2 * gid + 1];
            rad = radii[gid];
            const float a = conics[3 * gid], b = conics[3 * gid + 1], c = conics[3 * gid + 2];
            const float op = opacity[gid];
            const float cr = colors[3 * gid], cg = colors[3 * gid + 1], cb = colors[3 * gid + 2];
            int x0 = 0, x1 = x_hi, y0 = 0, y1 = y_hi;
            if (cull) support_rect(gx, gy, a, b, c, op, cv.alpha_min, ox, oy, x0, x1, y0, y1);
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            const int cnt = (w > 0 && h > 0) ? w * h : 0;
            // The lanes of the group take the rectangle's pixels round-robin.  LPE / w and
            // q / w for w in 1..16 from the hardware reciprocal (exact here: the quotients'
            // fractional parts are multiples of 1/w, well clear of the 1-ulp error).
            const float rw = __builtin_amdgcn_rcpf((float)max(w, 1));
            const int step_y = cnt ? (int)((float)LPE * rw + 1.0e-3f) : 0;
            const int step_x = cnt ? LPE - step_y * w : 0;
            const int qy = (int)(((float)q + 0.5f) * rw);
            int x = x0 + q - qy * w, y = y0 + qy;
            const float pa = (-0.5f * LOG2E) * a, pb = (-LOG2E) * b, pc = (-0.5f * LOG2E) * c;
            for (int k = q; k < cnt; k += LPE) {
                const float dx = gx - (ox + (float)x), dy = gy - (oy + (float)y);
                const float p = __builtin_fmaf(pc * dy, dy, __builtin_fmaf(pb, dy, pa * dx) * dx);
                const float ex = __builtin_amdgcn_exp2f(p);
                const float oe = op * ex;
                const float alpha = fminf(cv.alpha_cap, oe);
                const bool on = !(p > 0.f || alpha < cv.alpha_min);
                const float4 v = s_v[y * GI2D_BLOCK + x];
                const float am = on ? alpha : 0.f;
                acc[5] = __builtin_fmaf(am, v.x, acc[5]);
                acc[6] = __builtin_fmaf(am, v.y, acc[6]);
                acc[7] = __builtin_fmaf(am, v.z, acc[7]);
                const float v_alpha = __builtin_fmaf(cb, v.z, __builtin_fmaf(cg, v.y, cr * v.x));
                const bool live = on && !(oe > cv.alpha_cap);  // a capped alpha is a constant
                const float va = live ? v_alpha : 0.f;
                const float vs = -oe * va;
                acc[8] = __builtin_fmaf(ex, va, acc[8]);
                // moments of vs over the rectangle; the conic enters once, after the loop
                const float hx = vs * dx, hy = vs * dy;
                acc[0] += hx;
                acc[1] += hy;
                acc[2] = __builtin_fmaf(hx, dx, acc[2]);
                acc[3] = __builtin_fmaf(hx, dy, acc[3]);
                acc[4] = __builtin_fmaf(hy, dy, acc[4]);
                x += step_x;
                y += step_y;
                if (x > x1) {
                    x -= w;
                    ++y;
                }
            }
            // v_xy = sum vs * (conic . d) = conic . (sum vs * d); v_conic = (1/2, 1, 1/2) * moments
            const float m0 = acc[0], m1 = acc[1];
            acc[0] = __builtin_fmaf(b, m1, a * m0);
            acc[1] = __builtin_fmaf(c, m1, b * m0);
            acc[2] *= 0.5f;
            acc[4] *= 0.5f;
        }
#pragma unroll
        for (int k = 0; k < PART_STRIDE; ++k) acc[k] = group_sum<LPE>(acc[k]);
        if (e < n_raw && q == 0) {
            float *dst = partials + (slot0 + e) * PART_STRIDE;
#pragma unroll
            for (int k = 0; k < PART_STRIDE; ++k) dst[k] = acc[k];
            // where pass 2 finds this slot without searching: index of this tile in the
            // Gaussian's own (band-clipped) tile box, first SLOT_TAB tiles only
            if (rad > 0) {
                TileBox box = tile_box(gx, gy, (float)rad, tiles_x, tiles_y);
                const int by0 = max(box.y0, row0);
                const int kb = (ty - by0) * (box.x1 - box.x0) + (tx - box.x0);
                if (kb >= 0 && kb < SLOT_TAB) slot_tab[(size_t)gid * SLOT_TAB + kb] = e;
            }
        }
    }
}

__global__ __launch_bounds__(256) void rasterize_sum_bwd_kernel(
    const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0, int row1,
    int T, gi2d_conv cv, int cull, const float *__restrict__ v_out,
    float *__restrict__ partials, int *__restrict__ slot_tab)
{
    __shared__ float4 s_v[256];
    __shared__ int s_first[64];
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, T);
    const int *lst = seg + (size_t)t * C;
    // the first 64 ids before the length is known (a segment always has C >= 64 words)
    if (tid < 64) s_first[tid] = lst[tid];
    const int n_raw = tile_len[t];
    if (n_raw == 0 || n_raw > C) return;  // an overflowed tile is reported, not half-summed
    const size_t slot0 = (size_t)t * C;
    const int tx = t % tiles_x, ty = row0 + t / tiles_x;
    {
        const int j = tx * GI2D_BLOCK + (tid & 15), i = ty * GI2D_BLOCK + (tid >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < H && j < W) {
            const float *p = v_out + 3 * ((size_t)(i - row0 * GI2D_BLOCK) * W + j);
            v.x = p[0];
            v.y = p[1];
            v.z = p[2];
        }
        s_v[tid] = v;
    }
    __syncthreads();
    // a short list gets more lanes per entry: its rectangles are walked in fewer steps
#define GI2D_BWD_TILE(LPE)                                                                      \
    bwd_tile<LPE>(lst, s_first, n_raw, slot0, tx, ty, s_v, xys, radii, conics, colors, opacity, \
                  H, W, tiles_x, tiles_y, row0, cv, cull, partials, slot_tab)
    if (n_raw <= 16) GI2D_BWD_TILE(16);
    else if (n_raw <= 32) GI2D_BWD_TILE(8);
    else GI2D_BWD_TILE(4);
#undef GI2D_BWD_TILE
}

// Adds up Gaussian g's slots of the partial-sum slab, in the row-major order of its tile box.
__device__ __forceinline__ void gather_partials(int g, int r, float cx, float cy, int tiles_x,
                                                int tiles_y, int row0, int row1,
                                                const int *__restrict__ tile_len,
                                                const int *__restrict__ seg, int C,
                                                const float *__restrict__ partials,
                                                const int *__restrict__ slot_tab,
                                                float acc[PART_STRIDE])
{
#pragma unroll
    for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
    const int4 tab = *reinterpret_cast<const int4 *>(slot_tab + (size_t)g * SLOT_TAB);
    if (r <= 0) return;
    TileBox box = tile_box(cx, cy, (float)r, tiles_x, tiles_y);
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
    // The first SLOT_TAB tiles: list position remembered by pass 1; the loads of the four
    // slots are independent of each other.
    int tl[SLOT_TAB], ln[SLOT_TAB];
    const int pos[SLOT_TAB] = {tab.x, tab.y, tab.z, tab.w};
#pragma unroll
    for (int k = 0; k < SLOT_TAB; ++k) {
        const int ty = w > 0 ? y0 + k / w : 0, tx = w > 0 ? box.x0 + k % w : 0;
        tl[k] = k < cnt ? (ty - row0) * tiles_x + tx : 0;
        ln[k] = k < cnt ? tile_len[tl[k]] : 0;
    }
    float part[SLOT_TAB][PART_STRIDE];
#pragma unroll
    for (int k = 0; k < SLOT_TAB; ++k) {
        // an overflowed tile (ln > C) wrote no partial sums
        const bool live = k < cnt && ln[k] <= C && pos[k] >= 0 && pos[k] < ln[k];
        const float *src = partials + ((size_t)tl[k] * C + (live ? pos[k] : 0)) * PART_STRIDE;
#pragma unroll
        for (int u = 0; u < PART_STRIDE; ++u) part[k][u] = live ? src[u] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < SLOT_TAB; ++k)
#pragma unroll
        for (int u = 0; u < PART_STRIDE; ++u) acc[u] += part[k][u];
    // larger boxes: the rest by binary search (lists ascend in Gaussian id)
    for (int k = SLOT_TAB; k < cnt; ++k) {
        const int ty = y0 + k / w, tx = box.x0 + k % w;
        const int t = (ty - row0) * tiles_x + tx;
        const int n = tile_len[t];
        if (n > C) continue;
        const int *lst = seg + (size_t)t * C;
        int lo = 0, hi = n;
        while (lo < hi) {  // first position with id >= g
            int mid = (lo + hi) >> 1;
            if (lst[mid] < g)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < n && lst[lo] == g) {
            const float *src = partials + ((size_t)t * C + lo) * PART_STRIDE;
#pragma unroll
            for (int u = 0; u < PART_STRIDE; ++u) acc[u] += src[u];
        }
    }
}

__global__ __launch_bounds__(256) void gather_grads_kernel(
    const float *__restrict__ xys, const int *__restrict__ radii, int N, int tiles_x, int tiles_y,
    int row0, int row1, const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ partials, const int *__restrict__ slot_tab,
    float *__restrict__ v_xy, float *__restrict__ v_conic, float *__restrict__ v_colors,
    float *__restrict__ v_opacity)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float acc[PART_STRIDE];
    gather_partials(g, radii[g], xys[2 * g], xys[2 * g + 1], tiles_x, tiles_y, row0, row1,
                    tile_len, seg, C, partials, slot_tab, acc);
    v_xy[2 * g] = acc[0];
    v_xy[2 * g + 1] = acc[1];
    v_conic[3 * g] = acc[2];
    v_conic[3 * g + 1] = acc[3];
    v_conic[3 * g + 2] = acc[4];
    v_colors[3 * g] = acc[5];
    v_colors[3 * g + 1] = acc[6];
    v_colors[3 * g + 2] = acc[7];
    v_opacity[g] = acc[8];
}

// ------------------------------------------------------------------------------------------
// Adan step, fused: one launch over all parameter tensors, one read and one write of each of
// p, m, n, diff, neg_pre_grad (the reference's default path makes 18 multi-tensor passes).
// Operation order per element follows /root/reference/optimizer.py:319-362.
// ------------------------------------------------------------------------------------------

#define ADAN_MAX_TENSORS 8

struct AdanTensors {
    float *p[ADAN_MAX_TENSORS];
    float *g[ADAN_MAX_TENSORS];
    float *m[ADAN_MAX_TENSORS];
    float *n[ADAN_MAX_TENSORS];
    float *d[ADAN_MAX_TENSORS];
    float *prev[ADAN_MAX_TENSORS];
    long long numel[ADAN_MAX_TENSORS];
};

struct AdanScalars {
    float beta1, omb1, beta2, omb2, beta3, omb3;
    float bc3_sqrt, eps, neg_step, neg_step_diff, decay_mul, decay_div;
    int no_prox, first_step;
};

// One element of the update, optimizer.py:319-362 in order.
__device__ __forceinline__ void adan_element(const AdanScalars &h, float g, float &p, float &m,
                                             float &n, float &d, float &prev)
{
    const float np0 = h.first_step ? -g : prev;     // optimizer.py:187-189
    const float diff = np0 + g;                     // :323
    m = m * h.beta1 + h.omb1 * g;                   // :325-326
    d = d * h.beta2 + h.omb2 * diff;                // :328-330
    const float u = diff * h.beta2 + g;             // :332-333
    n = n * h.beta3 + h.omb3 * (u * u);             // :334-338
    const float denom = sqrtf(n) / h.bc3_sqrt + h.eps;  // :340-342
    if (h.no_prox) p = p * h.decay_mul;             // :348
    p = p + h.neg_step * (m / denom);               // :349 / :355
    p = p + h.neg_step_diff * (d / denom);          // :350 / :356
    if (!h.no_prox) p = p / h.decay_div;            // :360
    prev = -g;                                      // :361-362
}

__device__ __forceinline__ void adan_inplace(const AdanScalars &h, float g, float *p, float *m,
                                             float *n, float *d, float *prev, size_t i)
{
    float pi = p[i], mi = m[i], ni = n[i], di = d[i], pv = prev[i];
    adan_element(h, g, pi, mi, ni, di, pv);
    p[i] = pi;
    m[i] = mi;
    n[i] = ni;
    d[i] = di;
    prev[i] = pv;
}

// Tail of a fitting iteration, one lane per Gaussian: sum the Gaussian's backward partials,
// chain them through the projection and the model's activations to the raw parameters
// (gaussianimage_cholesky.py:62,74 / gaussianimage_rs.py:55-64), and take the Adan step on its
// 8 (Cholesky) or 8 (RS: 2+2+1+3) parameters -- no gradient tensor is materialised.
// Tensor slots: 0 = _xyz[N,2], 1 = _cholesky[N,3] or _scaling[N,2], 2 = _rotation[N,1] (RS),
// 3 = _features_dc[N,3].
// One parameter element with its optimizer state, held in registers.
struct AdanSlot {
    float p, m, n, d, prev;
};

__device__ __forceinline__ AdanSlot adan_load(const AdanTensors &T, int k, size_t i)
{
    AdanSlot s;
    s.p = T.p[k][i];
    s.m = T.m[k][i];
    s.n = T.n[k][i];
    s.d = T.d[k][i];
    s.prev = T.prev[k][i];
    return s;
}

__device__ __forceinline__ void adan_apply(const AdanScalars &h, const AdanTensors &T, int k,
                                           size_t i, AdanSlot s, float g)
{
    adan_element(h, g, s.p, s.m, s.n, s.d, s.prev);
    T.p[k][i] = s.p;
    T.m[k][i] = s.m;
    T.n[k][i] = s.n;
    T.d[k][i] = s.d;
    T.prev[k][i] = s.prev;
}

template <bool RS>
__global__ __launch_bounds__(256) void fit_tail_kernel(
    AdanTensors T, AdanScalars h, const float *__restrict__ xys, const int *__restrict__ radii,
    int N, int H, int W, int tiles_x, int tiles_y, int row0, int row1,
    const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ partials, const int *__restrict__ slot_tab)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    // Parameters and optimizer state first: these loads do not depend on the gather below, so
    // they are in flight while its three dependent rounds (table, lengths, partials) complete.
    const size_t g2 = 2 * (size_t)g, g3 = 3 * (size_t)g;
    const AdanSlot x0 = adan_load(T, 0, g2), x1 = adan_load(T, 0, g2 + 1);
    const AdanSlot c0 = adan_load(T, 1, RS ? g2 : g3), c1 = adan_load(T, 1, (RS ? g2 : g3) + 1);
    const AdanSlot c2 = adan_load(T, RS ? 2 : 1, RS ? (size_t)g : g3 + 2);
    const AdanSlot f0 = adan_load(T, 3, g3), f1 = adan_load(T, 3, g3 + 1),
                   f2 = adan_load(T, 3, g3 + 2);
    float acc[PART_STRIDE];
    gather_partials(g, radii[g], xys[2 * g], xys[2 * g + 1], tiles_x, tiles_y, row0, row1,
                    tile_len, seg, C, partials, slot_tab, acc);
    const float t0 = tanhf(x0.p), t1 = tanhf(x1.p);
    // means = tanh(_xyz); xys = 0.5*(W,H)*means + ...
    const float gx0 = 0.5f * (float)W * acc[0] * (1.0f - t0 * t0);
    const float gx1 = 0.5f * (float)H * acc[1] * (1.0f - t1 * t1);
    float a, b, c, radius;
    float vm0 = 0.f, vm1 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (!RS) {
        const float l11 = c0.p + 0.5f, l21 = c1.p, l22 = c2.p + 0.5f;
        if (cov_to_conic_radius(l11 * l11, l11 * l21, l21 * l21 + l22 * l22, a, b, c, radius)) {
            float gxx, gxy, gyy;
            conic_grad_to_cov_grad(a, b, c, acc[2], acc[3], acc[4], gxx, gxy, gyy);
            o0 = 2.0f * l11 * gxx + l21 * gxy;
            o1 = l11 * gxy + 2.0f * l21 * gyy;
            o2 = 2.0f * l22 * gyy;
            vm0 = gx0;
            vm1 = gx1;
        }
    } else {
        const float u1 = c0.p + 0.5f, u2 = c1.p + 0.5f;
        const float s1 = fabsf(u1), s2 = fabsf(u2);
        const float sg = 1.0f / (1.0f + expf(-c2.p));
        const float theta = sg * 2.0f * 3.14159274f;
        float sn, cs;
        sincosf(theta, &sn, &cs);
        const float q1 = s1 * s1, q2 = s2 * s2;
        if (cov_to_conic_radius(cs * cs * q1 + sn * sn * q2, cs * sn * (q1 - q2),
                                sn * sn * q1 + cs * cs * q2, a, b, c, radius)) {
            float gxx, gxy, gyy;
            conic_grad_to_cov_grad(a, b, c, acc[2], acc[3], acc[4], gxx, gxy, gyy);
            const float gq1 = gxx * cs * cs + gxy * cs * sn + gyy * sn * sn;
            const float gq2 = gxx * sn * sn - gxy * cs * sn + gyy * cs * cs;
            // d|u|/du = sign(u) (0 at 0, like torch.abs)
            o0 = 2.0f * s1 * gq1 * (u1 > 0.f ? 1.f : (u1 < 0.f ? -1.f : 0.f));
            o1 = 2.0f * s2 * gq2 * (u2 > 0.f ? 1.f : (u2 < 0.f ? -1.f : 0.f));
            const float dq = q1 - q2;
            const float vtheta = dq * (-2.0f * cs * sn * gxx + (cs * cs - sn * sn) * gxy +
                                       2.0f * cs * sn * gyy);
            o2 = vtheta * 2.0f * 3.14159274f * sg * (1.0f - sg);
            vm0 = gx0;
            vm1 = gx1;
        }
    }
    adan_apply(h, T, 0, g2, x0, vm0);
    adan_apply(h, T, 0, g2 + 1, x1, vm1);
    adan_apply(h, T, 1, RS ? g2 : g3, c0, o0);
    adan_apply(h, T, 1, (RS ? g2 : g3) + 1, c1, o1);
    adan_apply(h, T, RS ? 2 : 1, RS ? (size_t)g : g3 + 2, c2, o2);
    adan_apply(h, T, 3, g3, f0, acc[5]);
    adan_apply(h, T, 3, g3 + 1, f1, acc[6]);
    adan_apply(h, T, 3, g3 + 2, f2, acc[7]);
}

__global__ __launch_bounds__(256) void adan_step_kernel(AdanTensors T, AdanScalars h,
                                                        const float *__restrict__ clip_dev)
{
    const int k = blockIdx.y;
    float *__restrict__ p = T.p[k];
    float *__restrict__ gr = T.g[k];
    float *__restrict__ m = T.m[k];
    float *__restrict__ n = T.n[k];
    float *__restrict__ d = T.d[k];
    float *__restrict__ prev = T.prev[k];
    const long long numel = T.numel[k];
    const float clip = clip_dev ? clip_dev[0] : 1.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel;
         i += (long long)gridDim.x * blockDim.x) {
        float g = gr[i];
        if (clip_dev) {
            g = g * clip;
            gr[i] = g;  // the reference scales the gradient in place
        }
        adan_inplace(h, g, p, m, n, d, prev, (size_t)i);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

struct gi2d_workspace {
    int device = 0;
    int tiles_cap = 0;          // tiles the per-tile arrays are allocated for
    int seg_cap = 0;            // C: ids per tile segment
    int *d_count = nullptr;     // [tiles_cap][CNT_STRIDE] per-tile counters, zero between frames
    int *d_len = nullptr;       // [tiles_cap] list length of the finalised lists
    int *d_seg = nullptr;       // [tiles_cap][seg_cap] Gaussian ids per tile
    int64_t part_slots = 0;
    float *d_partials = nullptr;  // [tiles][seg_cap][9] backward partial sums
    int64_t slot_cap = 0;
    int *d_slot_tab = nullptr;  // [n][SLOT_TAB] list position of a Gaussian in its first tiles
    DevStatus *d_status = nullptr;
    long long *h_status = nullptr;  // pinned: [0] max count, [1] generation, [2] overflow gen.
    long long *h_status_dev = nullptr;
    int64_t generation = 0;
    int64_t overflow_ack = 0;
    int tiles_x = 0, tiles_y = 0, n_binned = 0;
    int row0 = 0, row1 = INT_MAX;    // tile rows requested by the caller (a band)
    int bin_row0 = 0, bin_row1 = 0;  // tile rows of the current lists
    bool counters_dirty = false;     // counters may be non-zero
    bool lists_raw = false;          // lists filled but not yet sorted / finalised
    bool lists_valid = false;        // lists belong to the binning that (tiles_x, ...) describe
    bool prefilled = false;          // project_*_fwd filled the lists for the shape below
    int pre_n = 0, pre_tx = 0, pre_ty = 0, pre_r0 = 0, pre_r1 = 0;
    int cull = 1;
    int raster_ppl = 0;  // 0 = choose by tile count
    int min_seg_cap = 64;
};

#define HIP_TRY(expr)                                \
    do {                                             \
        hipError_t err__ = (expr);                   \
        if (err__ != hipSuccess) return (int)err__;  \
    } while (0)

static inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GI2D_OK : (int)e;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int m) { return (a + m - 1) / m * m; }

namespace {
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (dev < 0) return;  // nothing to guard
        if (hipGetDevice(&prev) != hipSuccess) ok = false;
        if (ok && prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
}  // namespace

// (Re)allocates the per-tile arrays for T tiles of C ids each; the device is current.
// Contents are lost: callers re-bin afterwards.
static int ensure_lists(gi2d_workspace *ws, int T, int C)
{
    if (T <= ws->tiles_cap && C <= ws->seg_cap) return GI2D_OK;
    T = T > ws->tiles_cap ? T : ws->tiles_cap;
    C = C > ws->seg_cap ? C : ws->seg_cap;
    if ((int64_t)T * C >= (int64_t)INT_MAX) return GI2D_E_CAPACITY;
    HIP_TRY(hipDeviceSynchronize());
    if (ws->d_count) (void)hipFree(ws->d_count);
    if (ws->d_len) (void)hipFree(ws->d_len);
    if (ws->d_seg) (void)hipFree(ws->d_seg);
    if (ws->d_partials) (void)hipFree(ws->d_partials);
    ws->d_count = ws->d_len = ws->d_seg = nullptr;
    ws->d_partials = nullptr;
    ws->part_slots = 0;
    ws->tiles_cap = ws->seg_cap = 0;
    HIP_TRY(hipMalloc(&ws->d_count, sizeof(int) * CNT_STRIDE * (size_t)T));
    HIP_TRY(hipMalloc(&ws->d_len, sizeof(int) * (size_t)T));
    HIP_TRY(hipMalloc(&ws->d_seg, sizeof(int) * (size_t)T * C));
    HIP_TRY(hipMemset(ws->d_count, 0, sizeof(int) * CNT_STRIDE * (size_t)T));
    HIP_TRY(hipMemset(ws->d_len, 0, sizeof(int) * (size_t)T));
    ws->tiles_cap = T;
    ws->seg_cap = C;
    ws->counters_dirty = false;
    ws->lists_raw = false;
    ws->lists_valid = false;
    ws->prefilled = false;
    ws->tiles_x = ws->tiles_y = 0;  // no valid lists
    return GI2D_OK;
}

// Segment capacity wanted for a largest per-tile count `seen`: twice it, in whole waves.
static inline int wanted_cap(const gi2d_workspace *ws, long long seen)
{
    long long c = 2 * seen;
    if (c < ws->min_seg_cap) c = ws->min_seg_cap;
    if (c > (long long)INT_MAX / 4) c = INT_MAX / 4;
    return round_up((int)c, WAVE);
}

static int zero_counters(gi2d_workspace *ws, hipStream_t st)
{
    const int n = ws->tiles_cap * CNT_STRIDE;
    if (n == 0) return GI2D_OK;
    zero_ints_kernel<<<cdiv(n, 256), 256, 0, st>>>(ws->d_count, n);
    return launch_status();
}

// Sorts and finalises raw lists (when no forward rasterisation does it).
static int finalize_lists(gi2d_workspace *ws, hipStream_t st)
{
    if (!ws->lists_raw) return GI2D_OK;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (T > 0) {
        finalize_tiles_kernel<<<T, WAVE, 0, st>>>(ws->d_count, ws->d_len, ws->d_seg, ws->seg_cap,
                                                  ws->d_status, ws->generation);
        int rc = launch_status();
        if (rc != GI2D_OK) return rc;
    }
    ws->lists_raw = false;
    ws->counters_dirty = false;
    return GI2D_OK;
}

extern "C" {

int gi2d_version(void) { return GI2D_VERSION; }

const char *gi2d_error_string(int status)
{
    switch (status) {
        case GI2D_OK: return "ok";
        case GI2D_E_INVALID: return "invalid argument";
        case GI2D_E_STATE: return "workspace holds no tile lists for this call";
        case GI2D_E_CAPACITY: return "tile-list capacity exceeded";
        case GI2D_E_PREV_OVERFLOW:
            return "an earlier asynchronous frame overflowed a tile list (those tiles of its "
                   "image are NaN); the workspace was grown, call again";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown error";
    }
}

void gi2d_default_conv(gi2d_conv *c)
{
    c->pixel_offset = 0.0f;
    c->center_bias = 0.0f;
    c->alpha_cap = 1.0f;
    c->alpha_min = 1.0f / 255.0f;
    c->bg_weight = 0.0f;
}

int gi2d_workspace_create(int device, gi2d_workspace **out)
{
    if (!out) return GI2D_E_INVALID;
    *out = nullptr;
    DeviceGuard guard(device);
    if (!guard.ok) return GI2D_E_INVALID;
    gi2d_workspace *ws = new (std::nothrow) gi2d_workspace();
    if (!ws) return (int)hipErrorOutOfMemory;
    ws->device = device;
    hipError_t e = hipMalloc(&ws->d_status, sizeof(DevStatus));
    if (e == hipSuccess) e = hipMemset(ws->d_status, 0, sizeof(DevStatus));
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&ws->h_status, 4 * sizeof(long long), hipHostMallocMapped);
    if (e == hipSuccess) {
        for (int i = 0; i < 4; ++i) ws->h_status[i] = 0;
        e = hipHostGetDevicePointer((void **)&ws->h_status_dev, ws->h_status, 0);
    }
    if (e != hipSuccess) {
        gi2d_workspace_destroy(ws);
        return (int)e;
    }
    *out = ws;
    return GI2D_OK;
}

int gi2d_workspace_destroy(gi2d_workspace *ws)
{
    if (!ws) return GI2D_OK;
    DeviceGuard guard(ws->device);
    (void)hipDeviceSynchronize();
    if (ws->d_count) (void)hipFree(ws->d_count);
    if (ws->d_len) (void)hipFree(ws->d_len);
    if (ws->d_seg) (void)hipFree(ws->d_seg);
    if (ws->d_partials) (void)hipFree(ws->d_partials);
    if (ws->d_slot_tab) (void)hipFree(ws->d_slot_tab);
    if (ws->d_status) (void)hipFree(ws->d_status);
    if (ws->h_status) (void)hipHostFree(ws->h_status);
    delete ws;
    return GI2D_OK;
}

int gi2d_workspace_reserve(gi2d_workspace *ws, int64_t per_tile_capacity, int max_tiles)
{
    if (!ws || per_tile_capacity < 0 || max_tiles < 0) return GI2D_E_INVALID;
    if (per_tile_capacity > INT_MAX / 4) return GI2D_E_CAPACITY;
    DeviceGuard guard(ws->device);
    if (!guard.ok) return GI2D_E_INVALID;
    const int C = round_up(per_tile_capacity > 1 ? (int)per_tile_capacity : 1, WAVE);
    return ensure_lists(ws, max_tiles > 1 ? max_tiles : 1, C);
}

int gi2d_workspace_set_option(gi2d_workspace *ws, int option, int value)
{
    if (!ws) return GI2D_E_INVALID;
    switch (option) {
        case GI2D_OPT_CULL: ws->cull = value ? 1 : 0; return GI2D_OK;
        case GI2D_OPT_RASTER_PPL:
            if (value != 0 && value != 1 && value != 2 && value != 4) return GI2D_E_INVALID;
            ws->raster_ppl = value;
            return GI2D_OK;
        default: return GI2D_E_INVALID;
    }
}

int gi2d_workspace_set_rows(gi2d_workspace *ws, int row_begin, int row_end)
{
    if (!ws || row_begin < 0 || row_end < row_begin) return GI2D_E_INVALID;
    ws->row0 = row_begin;
    ws->row1 = row_end;
    return GI2D_OK;
}

int gi2d_workspace_info(gi2d_workspace *ws, int64_t info[8])
{
    if (!ws || !info) return GI2D_E_INVALID;
    info[0] = ws->seg_cap;
    info[1] = ws->h_status[0];
    info[2] = ws->generation;
    info[3] = ws->tiles_x;
    info[4] = ws->tiles_y;
    info[5] = ws->n_binned;
    info[6] = ws->bin_row0;
    info[7] = ws->bin_row1;
    return GI2D_OK;
}

static int check_proj_args(const void *a, const void *b, int N, int H, int W, int tiles_x,
                           int tiles_y, const gi2d_conv *conv, const void *o0, const void *o1,
                           const void *o2, const void *o3, const void *o4)
{
    if (N < 0 || H < 1 || W < 1 || !conv) return GI2D_E_INVALID;
    if (tiles_x != cdiv(W, GI2D_BLOCK) || tiles_y != cdiv(H, GI2D_BLOCK)) return GI2D_E_INVALID;
    if (N > 0 && (!a || !b || !o0 || !o1 || !o2 || !o3 || !o4)) return GI2D_E_INVALID;
    return GI2D_OK;
}

// True when the next binning of this shape must learn its list lengths synchronously.
static bool shape_is_new(const gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int r0, int r1)
{
    return ws->seg_cap == 0 || ws->generation == 0 || N != ws->n_binned ||
           tiles_x != ws->tiles_x || tiles_y != ws->tiles_y || r0 != ws->bin_row0 ||
           r1 != ws->bin_row1;
}

// Prepares ws for a projection that also fills the tile lists.  Leaves ws->prefilled false when
// the shape is new (the lists' capacity is unknown): gi2d_bin_tiles then bins synchronously.
static int begin_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, hipStream_t st)
{
    const int r0 = min(ws->row0, tiles_y), r1 = min(ws->row1, tiles_y);
    const int T = (r1 - r0) * tiles_x;
    ws->prefilled = false;
    if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1)) return GI2D_OK;
    int rc = ensure_lists(ws, max(1, T), wanted_cap(ws, ws->h_status[0]));
    if (rc != GI2D_OK) return rc;
    if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1)) return GI2D_OK;  // reallocated
    if (ws->counters_dirty) {
        rc = zero_counters(ws, st);
        if (rc != GI2D_OK) return rc;
    }
    ws->counters_dirty = true;
    ws->lists_raw = false;
    ws->lists_valid = false;  // the old lists are about to be overwritten
    ws->prefilled = true;
    ws->pre_n = N;
    ws->pre_tx = tiles_x;
    ws->pre_ty = tiles_y;
    ws->pre_r0 = r0;
    ws->pre_r1 = r1;
    return GI2D_OK;
}

// kind 0 = Cholesky (cov_a = L[N,3]), 1 = rotation-scale (cov_a = scales[N,2], cov_b = rotation).
// raw = the model's raw parameters are given and activated inside the kernel.
static int check_codec(int kind, const gi2d_codec *cd, int N)
{
    if ((kind != 0 && kind != 1) || !cd || N < 0) return GI2D_E_INVALID;
    if (cd->num_quantizers < 1 || cd->num_quantizers > GI2D_MAX_VQ_LAYERS || cd->codebook_size < 1)
        return GI2D_E_INVALID;
    if (N == 0) return GI2D_OK;
    if (!cd->xyz_half || !cd->cov_a_code || !cd->cov_a_scale || !cd->cov_a_beta ||
        !cd->color_index || !cd->codebooks)
        return GI2D_E_INVALID;
    if (kind == 1 && (!cd->cov_b_code || !cd->cov_b_scale || !cd->cov_b_beta))
        return GI2D_E_INVALID;
    return GI2D_OK;
}

// codec != NULL: the inputs are the coded attributes (means / cov_a / cov_b unused) and the
// decoded colours are written to colors_out.
static int project_impl(int kind, bool raw, const float *means, const float *cov_a,
                        const float *cov_b, int N, int H, int W, int tiles_x, int tiles_y,
                        const gi2d_conv *conv, float *xys, float *depths, int *radii,
                        float *conics, int *num_tiles_hit, gi2d_workspace *ws, void *stream,
                        const gi2d_codec *codec = nullptr, float *colors_out = nullptr)
{
    if (codec) {
        int rcc = check_codec(kind, codec, N);
        if (rcc != GI2D_OK) return rcc;
        if (N > 0 && !colors_out) return GI2D_E_INVALID;
        means = reinterpret_cast<const float *>(codec->xyz_half);
        cov_a = codec->cov_a_code;
        cov_b = codec->cov_b_code;
    }
    int rc = check_proj_args(means, cov_a, N, H, W, tiles_x, tiles_y, conv, xys, depths, radii,
                             conics, num_tiles_hit);
    if (rc != GI2D_OK) return rc;
    if (kind == 1 && N > 0 && !cov_b) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    bool fill = false;
    DeviceGuard guard(ws ? ws->device : -1);
    if (ws) {
        if (!guard.ok) return GI2D_E_INVALID;
        rc = begin_prefill(ws, N, tiles_x, tiles_y, st);
        if (rc != GI2D_OK) return rc;
        fill = ws->prefilled;
    }
    if (N == 0) return GI2D_OK;
    int *cnt = fill ? ws->d_count : nullptr, *seg = fill ? ws->d_seg : nullptr;
    const int C = fill ? ws->seg_cap : 0, r0 = fill ? ws->pre_r0 : 0, r1 = fill ? ws->pre_r1 : 0;
    DevStatus *status = fill ? ws->d_status : nullptr;
    long long *hs = fill ? ws->h_status_dev : nullptr;
    const long long gen = fill ? ws->generation : 0;
    const dim3 grid(cdiv(N, PG_THREADS)), block(PG_THREADS);
#define GI2D_PROJ_ARGS N, H, W, tiles_x, tiles_y, conv->center_bias, xys, depths, radii, conics, \
                       num_tiles_hit, cnt, seg, C, r0, r1, status, hs, gen
    if (codec) {
        if (kind == 0)
            project_decode_fwd_kernel<false><<<grid, block, 0, st>>>(*codec, colors_out, GI2D_PROJ_ARGS);
        else
            project_decode_fwd_kernel<true><<<grid, block, 0, st>>>(*codec, colors_out, GI2D_PROJ_ARGS);
    } else if (kind == 0) {
        if (raw)
            project_cholesky_fwd_kernel<true><<<grid, block, 0, st>>>(means, cov_a, GI2D_PROJ_ARGS);
        else
            project_cholesky_fwd_kernel<false><<<grid, block, 0, st>>>(means, cov_a, GI2D_PROJ_ARGS);
    } else {
        if (raw)
            project_rs_fwd_kernel<true><<<grid, block, 0, st>>>(means, cov_a, cov_b, GI2D_PROJ_ARGS);
        else
            project_rs_fwd_kernel<false><<<grid, block, 0, st>>>(means, cov_a, cov_b, GI2D_PROJ_ARGS);
    }
#undef GI2D_PROJ_ARGS
    return launch_status();
}

int gi2d_project_cholesky_fwd(const float *means2d, const float *L_elements, int N, int H, int W,
                              int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                              float *depths, int *radii, float *conics, int *num_tiles_hit,
                              gi2d_workspace *ws, void *stream)
{
    return project_impl(0, false, means2d, L_elements, nullptr, N, H, W, tiles_x, tiles_y, conv,
                        xys, depths, radii, conics, num_tiles_hit, ws, stream);
}

int gi2d_project_rs_fwd(const float *means2d, const float *scales, const float *rotation, int N,
                        int H, int W, int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                        float *depths, int *radii, float *conics, int *num_tiles_hit,
                        gi2d_workspace *ws, void *stream)
{
    return project_impl(1, false, means2d, scales, rotation, N, H, W, tiles_x, tiles_y, conv, xys,
                        depths, radii, conics, num_tiles_hit, ws, stream);
}

int gi2d_project_cholesky_bwd(const float *means2d, const float *L_elements, int N, int H, int W,
                              const float *v_xys, const float *v_conics, float *v_means2d,
                              float *v_L_elements, void *stream)
{
    (void)means2d;
    if (N < 0 || H < 1 || W < 1) return GI2D_E_INVALID;
    if (N == 0) return GI2D_OK;
    if (!L_elements || !v_xys || !v_conics || !v_means2d || !v_L_elements) return GI2D_E_INVALID;
    project_cholesky_bwd_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(
        L_elements, N, H, W, v_xys, v_conics, v_means2d, v_L_elements);
    return launch_status();
}

int gi2d_project_rs_bwd(const float *means2d, const float *scales, const float *rotation, int N,
                        int H, int W, const float *v_xys, const float *v_conics, float *v_means2d,
                        float *v_scales, float *v_rotation, void *stream)
{
    (void)means2d;
    if (N < 0 || H < 1 || W < 1) return GI2D_E_INVALID;
    if (N == 0) return GI2D_OK;
    if (!scales || !rotation || !v_xys || !v_conics || !v_means2d || !v_scales || !v_rotation)
        return GI2D_E_INVALID;
    project_rs_bwd_kernel<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(
        scales, rotation, N, H, W, v_xys, v_conics, v_means2d, v_scales, v_rotation);
    return launch_status();
}

// Fills the lists from xys / radii with the standalone kernel.
static int launch_bin(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                      int tiles_y, int r0, int r1, hipStream_t st)
{
    int rc = GI2D_OK;
    if (ws->counters_dirty) {
        rc = zero_counters(ws, st);
        if (rc != GI2D_OK) return rc;
    }
    ws->counters_dirty = true;
    bin_tiles_kernel<<<max(1, cdiv(N, PG_THREADS)), PG_THREADS, 0, st>>>(
        xys, radii, N, tiles_x, tiles_y, r0, r1, ws->d_count, ws->d_seg, ws->seg_cap,
        ws->d_status, ws->h_status_dev, ws->generation);
    return launch_status();
}

int gi2d_bin_tiles(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                   int tiles_y, int flags, void *stream)
{
    if (!ws || N < 0 || tiles_x < 1 || tiles_y < 1) return GI2D_E_INVALID;
    if ((int64_t)tiles_x * tiles_y >= (int64_t)INT_MAX / 2) return GI2D_E_INVALID;
    if (N > 0 && (!xys || !radii)) return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    if (!guard.ok) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int r0 = min(ws->row0, tiles_y), r1 = min(ws->row1, tiles_y);
    const int T = (r1 - r0) * tiles_x;
    const bool nogrow = (flags & GI2D_BIN_NOGROW) != 0;
    bool sync = (flags & GI2D_BIN_SYNC) != 0;
    bool prev_overflow = false;

    // an earlier asynchronous frame whose lists did not fit: tell the caller (once)
    if (ws->h_status[2] > ws->overflow_ack) {
        ws->overflow_ack = ws->h_status[2];
        prev_overflow = true;
    }

    bool prefilled = ws->prefilled && ws->pre_n == N && ws->pre_tx == tiles_x &&
                     ws->pre_ty == tiles_y && ws->pre_r0 == r0 && ws->pre_r1 == r1;
    ws->prefilled = false;
    if (!(flags & GI2D_BIN_COUNTED)) prefilled = false;

    if (!prefilled) {
        // Size the lists.  Steady state needs no synchronisation: the capacity keeps a 2x
        // margin over the largest per-tile count seen from completed frames.  A new problem
        // shape says nothing about its counts: learn them once, synchronously.
        if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1) && !nogrow) sync = true;
        if (nogrow) {
            if (T > ws->tiles_cap || ws->seg_cap == 0) return GI2D_E_CAPACITY;
        } else {
            int rc = ensure_lists(ws, max(1, T), wanted_cap(ws, sync ? 0 : ws->h_status[0]));
            if (rc != GI2D_OK) return rc;
        }
    }
    ws->generation += 1;
    ws->tiles_x = tiles_x;
    ws->tiles_y = tiles_y;
    ws->n_binned = N;
    ws->bin_row0 = r0;
    ws->bin_row1 = r1;
    ws->lists_valid = false;
    if (!prefilled) {
        int rc = launch_bin(ws, xys, radii, N, tiles_x, tiles_y, r0, r1, st);
        if (rc != GI2D_OK) return rc;
    }
    ws->lists_raw = true;
    ws->lists_valid = true;

    if (sync) {
        // finalise now, read the list lengths, and redo the binning if a segment overflowed
        for (int attempt = 0; attempt < 2; ++attempt) {
            int rc = finalize_lists(ws, st);
            if (rc != GI2D_OK) return rc;
            int maxlen = 0;
            if (T > 0) {
                int *h_len = (int *)malloc(sizeof(int) * (size_t)T);
                if (!h_len) return (int)hipErrorOutOfMemory;
                hipError_t e = hipMemcpyAsync(h_len, ws->d_len, sizeof(int) * (size_t)T,
                                              hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) {
                    free(h_len);
                    return (int)e;
                }
                for (int i = 0; i < T; ++i) maxlen = h_len[i] > maxlen ? h_len[i] : maxlen;
                free(h_len);
            } else {
                HIP_TRY(hipStreamSynchronize(st));
            }
            if (maxlen > ws->h_status[0]) ws->h_status[0] = maxlen;
            const bool fits = maxlen <= ws->seg_cap;
            if (!fits) ws->overflow_ack = ws->generation;  // reported (or repaired) right here
            if (nogrow) {
                if (!fits) return GI2D_E_CAPACITY;
                break;
            }
            // keep the asynchronous policy's 2x margin from the start, so that the next frames
            // of this shape never have to reallocate
            const int want = wanted_cap(ws, maxlen);
            if (want <= ws->seg_cap) break;
            if (attempt == 1) return GI2D_E_CAPACITY;
            rc = ensure_lists(ws, max(1, T), want);
            if (rc != GI2D_OK) return rc;
            ws->tiles_x = tiles_x;
            ws->tiles_y = tiles_y;
            rc = launch_bin(ws, xys, radii, N, tiles_x, tiles_y, r0, r1, st);
            if (rc != GI2D_OK) return rc;
            ws->lists_raw = true;
            ws->lists_valid = true;
        }
    }
    return prev_overflow ? GI2D_E_PREV_OVERFLOW : GI2D_OK;
}

int gi2d_copy_bins(gi2d_workspace *ws, int *tile_offsets, int *ids, int64_t ids_capacity,
                   void *stream)
{
    if (!ws || !tile_offsets) return GI2D_E_INVALID;
    if (ws->generation == 0 || ws->tiles_x == 0 || !ws->lists_valid) return GI2D_E_STATE;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = finalize_lists(ws, st);
    if (rc != GI2D_OK) return rc;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    const int C = ws->seg_cap;
    int *h_len = (int *)malloc(sizeof(int) * ((size_t)T + 1));
    int *h_off = (int *)malloc(sizeof(int) * ((size_t)T + 1));
    if (!h_len || !h_off) {
        free(h_len);
        free(h_off);
        return (int)hipErrorOutOfMemory;
    }
    hipError_t e = hipSuccess;
    if (T > 0) e = hipMemcpyAsync(h_len, ws->d_len, sizeof(int) * (size_t)T, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    int64_t total = 0;
    bool over = false;
    for (int t = 0; t < T && e == hipSuccess; ++t) {
        h_off[t] = (int)total;
        if (h_len[t] > C) over = true;
        total += h_len[t];
    }
    h_off[T] = (int)total;
    rc = GI2D_OK;
    if (e != hipSuccess) rc = (int)e;
    else if (over || total >= INT_MAX) rc = GI2D_E_CAPACITY;
    else if (ids && total > ids_capacity) rc = GI2D_E_CAPACITY;
    if (rc == GI2D_OK) {
        e = hipMemcpyAsync(tile_offsets, h_off, sizeof(int) * ((size_t)T + 1), hipMemcpyDefault, st);
        // compaction through a temporary on the device (the destinations may be host memory);
        // a test / diagnostics path, not a hot one
        int *d_off = nullptr, *d_ids = nullptr;
        if (ids && total > 0 && e == hipSuccess) {
            e = hipMalloc(&d_off, sizeof(int) * ((size_t)T + 1));
            if (e == hipSuccess) e = hipMalloc(&d_ids, sizeof(int) * (size_t)total);
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_off, h_off, sizeof(int) * ((size_t)T + 1),
                                   hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                compact_lists_kernel<<<T, WAVE, 0, st>>>(ws->d_seg, C, ws->d_len, d_off, d_ids);
                e = hipGetLastError();
            }
            if (e == hipSuccess)
                e = hipMemcpyAsync(ids, d_ids, sizeof(int) * (size_t)total, hipMemcpyDefault, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (d_off) (void)hipFree(d_off);
        if (d_ids) (void)hipFree(d_ids);
        if (e != hipSuccess) rc = (int)e;
    }
    free(h_len);
    free(h_off);
    return rc;
}

static int check_raster_args(gi2d_workspace *ws, int N, int H, int W, const gi2d_conv *conv)
{
    if (!ws || !conv || N < 0 || H < 1 || W < 1) return GI2D_E_INVALID;
    if (ws->generation == 0 || ws->tiles_x == 0 || !ws->lists_valid) return GI2D_E_STATE;
    if (ws->tiles_x != cdiv(W, GI2D_BLOCK) || ws->tiles_y != cdiv(H, GI2D_BLOCK) ||
        ws->n_binned != N)
        return GI2D_E_STATE;
    return GI2D_OK;
}

// gt / v_img / loss_part non-NULL selects the fitting epilogue (then out_img may be NULL).
// sets = number of colour sets (1..3); sets beyond the first come in `extra`.
static int raster_fwd_impl(gi2d_workspace *ws, const float *xys, const float *conics,
                           const float *colors, const float *opacity, const float *background,
                           int N, int H, int W, const gi2d_conv *conv, float *out_img,
                           const float *gt, float *v_img, float *loss_part, void *stream,
                           int sets = 1, ExtraSets extra = ExtraSets())
{
    int rc = check_raster_args(ws, N, H, W, conv);
    if (rc != GI2D_OK) return rc;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    if (T == 0) {  // empty band
        ws->lists_raw = false;
        return GI2D_OK;
    }
    const bool l2 = gt != nullptr;
    if (l2 && (!v_img || !loss_part || sets != 1)) return GI2D_E_INVALID;
    if ((!out_img && !l2 && !extra.out_chw) ||
        (N > 0 && (!xys || !conics || !colors || !opacity)))
        return GI2D_E_INVALID;
    if (extra.out_chw) {
        const int y0 = ws->bin_row0 * GI2D_BLOCK;
        extra.chw_plane = (size_t)(min(H, ws->bin_row1 * GI2D_BLOCK) - y0) * (size_t)W;
    }
    if (conv->bg_weight != 0.0f && !background) return GI2D_E_INVALID;
    float bg[3] = {0.f, 0.f, 0.f};
    if (conv->bg_weight != 0.0f) {
        // three floats, needed as kernel arguments
        HIP_TRY(hipMemcpyAsync(bg, background, sizeof(bg), hipMemcpyDefault, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
#define GI2D_RASTER_ARGS ws->d_count, ws->d_len, ws->d_seg, ws->seg_cap, xys, conics, colors, \
                         opacity, bg[0], bg[1], bg[2], H, W, ws->tiles_x, ws->bin_row0, T, *conv, \
                         ws->cull, ws->d_status, ws->generation, out_img, gt, v_img, loss_part, \
                         extra
    const bool fin = ws->lists_raw;
    // pixels per lane: many sparse tiles -> smaller workgroups, more tiles in flight per CU
    int ppl = 1;
    if (sets == 1) {
        ppl = ws->raster_ppl;
        if (ppl == 0) ppl = T >= GI2D_PPL4_MIN_TILES ? 4 : (T >= GI2D_PPL2_MIN_TILES ? 2 : 1);
    }
#define GI2D_RASTER_LAUNCH(FIN, LL, SETS, PPL) \
    rasterize_sum_fwd_kernel<FIN, LL, SETS, PPL><<<T, 256 / PPL, 0, st>>>(GI2D_RASTER_ARGS)
#define GI2D_RASTER_PICK(SETS, PPL)                         \
    do {                                                    \
        if (fin && l2) GI2D_RASTER_LAUNCH(true, true, SETS, PPL);   \
        else if (fin) GI2D_RASTER_LAUNCH(true, false, SETS, PPL);   \
        else if (l2) GI2D_RASTER_LAUNCH(false, true, SETS, PPL);    \
        else GI2D_RASTER_LAUNCH(false, false, SETS, PPL);           \
    } while (0)
    if (sets == 3) {
        if (fin) GI2D_RASTER_LAUNCH(true, false, 3, 1);
        else GI2D_RASTER_LAUNCH(false, false, 3, 1);
    } else if (sets == 2) {
        if (fin) GI2D_RASTER_LAUNCH(true, false, 2, 1);
        else GI2D_RASTER_LAUNCH(false, false, 2, 1);
    } else if (ppl == 4) {
        GI2D_RASTER_PICK(1, 4);
    } else if (ppl == 2) {
        GI2D_RASTER_PICK(1, 2);
    } else {
        GI2D_RASTER_PICK(1, 1);
    }
#undef GI2D_RASTER_PICK
#undef GI2D_RASTER_LAUNCH
#undef GI2D_RASTER_ARGS
    if (fin) {
        ws->lists_raw = false;
        ws->counters_dirty = false;
    }
    return launch_status();
}

int gi2d_rasterize_sum_fwd(gi2d_workspace *ws, const float *xys, const float *conics,
                           const float *colors, const float *opacity, const float *background,
                           int N, int H, int W, const gi2d_conv *conv, float *out_img,
                           void *stream)
{
    return raster_fwd_impl(ws, xys, conics, colors, opacity, background, N, H, W, conv, out_img,
                           nullptr, nullptr, nullptr, stream);
}

int gi2d_rasterize_sum_fwd_multi(gi2d_workspace *ws, const float *xys, const float *conics,
                                 int num_sets, const float *const *colors, const float *opacity,
                                 const float *background, int N, int H, int W,
                                 const gi2d_conv *conv, float *const *out_imgs, void *stream)
{
    if (num_sets < 1 || num_sets > GI2D_MAX_COLOR_SETS || !colors || !out_imgs)
        return GI2D_E_INVALID;
    ExtraSets extra = ExtraSets();
    for (int s = 0; s < num_sets; ++s) {
        if (!out_imgs[s] || (N > 0 && !colors[s])) return GI2D_E_INVALID;
        if (s > 0) {
            extra.colors[s - 1] = colors[s];
            extra.out[s - 1] = out_imgs[s];
        }
    }
    return raster_fwd_impl(ws, xys, conics, colors[0], opacity, background, N, H, W, conv,
                           out_imgs[0], nullptr, nullptr, nullptr, stream, num_sets, extra);
}

// Backward pass 1 (per-tile partial sums into the slab); shared by the operator backward and
// the fused fitting step.
static int raster_bwd_pass1(gi2d_workspace *ws, const float *xys, const int *radii,
                            const float *conics, const float *colors, const float *opacity, int N,
                            int H, int W, const gi2d_conv *conv, const float *v_out,
                            hipStream_t st)
{
    int rc = finalize_lists(ws, st);
    if (rc != GI2D_OK) return rc;
    const int64_t slots = (int64_t)ws->tiles_cap * ws->seg_cap;
    if (ws->part_slots < slots) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->d_partials) (void)hipFree(ws->d_partials);
        ws->d_partials = nullptr;
        ws->part_slots = 0;
        HIP_TRY(hipMalloc(&ws->d_partials, sizeof(float) * PART_STRIDE * (size_t)slots));
        ws->part_slots = slots;
    }
    if (ws->slot_cap < N) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->d_slot_tab) (void)hipFree(ws->d_slot_tab);
        ws->d_slot_tab = nullptr;
        ws->slot_cap = 0;
        HIP_TRY(hipMalloc(&ws->d_slot_tab, sizeof(int) * SLOT_TAB * (size_t)N));
        HIP_TRY(hipMemset(ws->d_slot_tab, 0, sizeof(int) * SLOT_TAB * (size_t)N));
        ws->slot_cap = N;
    }
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (T > 0) {
        rasterize_sum_bwd_kernel<<<T, 256, 0, st>>>(
            ws->d_len, ws->d_seg, ws->seg_cap, xys, radii, conics, colors, opacity, H, W,
            ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1, T, *conv, ws->cull, v_out,
            ws->d_partials, ws->d_slot_tab);
        rc = launch_status();
    }
    return rc;
}

int gi2d_rasterize_sum_bwd(gi2d_workspace *ws, const float *xys, const int *radii,
                           const float *conics, const float *colors, const float *opacity, int N,
                           int H, int W, const gi2d_conv *conv, const float *v_out, float *v_xys,
                           float *v_conics, float *v_colors, float *v_opacity, void *stream)
{
    int rc = check_raster_args(ws, N, H, W, conv);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (!xys || !radii || !conics || !colors || !opacity || (!v_out && T > 0) || !v_xys ||
        !v_conics || !v_colors || !v_opacity)
        return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    rc = raster_bwd_pass1(ws, xys, radii, conics, colors, opacity, N, H, W, conv, v_out, st);
    if (rc != GI2D_OK) return rc;
    gather_grads_kernel<<<cdiv(N, 256), 256, 0, st>>>(
        xys, radii, N, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1, ws->d_len, ws->d_seg,
        ws->seg_cap, ws->d_partials, ws->d_slot_tab, v_xys, v_conics, v_colors, v_opacity);
    return launch_status();
}

// One call per frame: projection (filling the lists), binning bookkeeping, rasterisation.
static int render_tail(gi2d_workspace *ws, int N, int H, int W, int tiles_x, int tiles_y,
                       const gi2d_conv *conv, const float *xys, const int *radii,
                       const float *conics, const float *colors, const float *opacity,
                       const float *background, float *out_img, void *ev_begin, void *ev_end,
                       void *stream, float *out_chw = nullptr)
{
    int rc = gi2d_bin_tiles(ws, xys, radii, N, tiles_x, tiles_y, GI2D_BIN_COUNTED, stream);
    if (rc != GI2D_OK && rc != GI2D_E_PREV_OVERFLOW) return rc;
    const int late = rc;
    if (ev_begin) HIP_TRY(hipEventRecord((hipEvent_t)ev_begin, (hipStream_t)stream));
    ExtraSets extra = ExtraSets();
    extra.out_chw = out_chw;
    rc = raster_fwd_impl(ws, xys, conics, colors, opacity, background, N, H, W, conv, out_img,
                         nullptr, nullptr, nullptr, stream, 1, extra);
    if (rc != GI2D_OK) return rc;
    if (ev_end) HIP_TRY(hipEventRecord((hipEvent_t)ev_end, (hipStream_t)stream));
    return late;
}

int gi2d_render_cholesky_fwd(gi2d_workspace *ws, const float *means2d, const float *L_elements,
                             const float *colors, const float *opacity, const float *background,
                             int N, int H, int W, const gi2d_conv *conv, float *xys, float *depths,
                             int *radii, float *conics, int *num_tiles_hit, float *out_img,
                             void *raster_begin_event, void *raster_end_event, void *stream)
{
    if (!ws) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    int rc = gi2d_project_cholesky_fwd(means2d, L_elements, N, H, W, tiles_x, tiles_y, conv, xys,
                                       depths, radii, conics, num_tiles_hit, ws, stream);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, raster_begin_event, raster_end_event, stream);
}

int gi2d_render_rs_fwd(gi2d_workspace *ws, const float *means2d, const float *scales,
                       const float *rotation, const float *colors, const float *opacity,
                       const float *background, int N, int H, int W, const gi2d_conv *conv,
                       float *xys, float *depths, int *radii, float *conics, int *num_tiles_hit,
                       float *out_img, void *raster_begin_event, void *raster_end_event,
                       void *stream)
{
    if (!ws) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    int rc = gi2d_project_rs_fwd(means2d, scales, rotation, N, H, W, tiles_x, tiles_y, conv, xys,
                                 depths, radii, conics, num_tiles_hit, ws, stream);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, raster_begin_event, raster_end_event, stream);
}

int gi2d_decode_params(int kind, const gi2d_codec *codec, int N, float *means2d, float *cov_a,
                       float *cov_b, float *colors, void *stream)
{
    int rc = check_codec(kind, codec, N);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    if (!means2d || !cov_a || !colors || (kind == 1 && !cov_b)) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0)
        decode_params_kernel<false><<<cdiv(N, 256), 256, 0, st>>>(*codec, N, means2d, cov_a, cov_b,
                                                                  colors);
    else
        decode_params_kernel<true><<<cdiv(N, 256), 256, 0, st>>>(*codec, N, means2d, cov_a, cov_b,
                                                                 colors);
    return launch_status();
}

int gi2d_decode_render(gi2d_workspace *ws, int kind, const gi2d_codec *codec, const float *opacity,
                       const float *background, int N, int H, int W, const gi2d_conv *conv,
                       float *xys, float *depths, int *radii, float *conics, int *num_tiles_hit,
                       float *colors, float *out_img, float *render_chw, void *stream)
{
    if (!ws || (!out_img && !render_chw)) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    int rc = project_impl(kind, false, nullptr, nullptr, nullptr, N, H, W, tiles_x, tiles_y, conv,
                          xys, depths, radii, conics, num_tiles_hit, ws, stream, codec, colors);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, nullptr, nullptr, stream, render_chw);
}

static int adan_scalars(const gi2d_adan_hyper *hyper, AdanScalars &h)
{
    if (!hyper || hyper->step < 1) return GI2D_E_INVALID;
    const double b1 = hyper->beta1, b2 = hyper->beta2, b3 = hyper->beta3, lr = hyper->lr;
    const double bc1 = 1.0 - pow(b1, (double)hyper->step);
    const double bc2 = 1.0 - pow(b2, (double)hyper->step);
    const double bc3 = 1.0 - pow(b3, (double)hyper->step);
    h.beta1 = (float)b1;
    h.omb1 = (float)(1.0 - b1);
    h.beta2 = (float)b2;
    h.omb2 = (float)(1.0 - b2);
    h.beta3 = (float)b3;
    h.omb3 = (float)(1.0 - b3);
    h.bc3_sqrt = (float)sqrt(bc3);
    h.eps = (float)hyper->eps;
    h.neg_step = (float)(-(lr / bc1));
    h.neg_step_diff = (float)(-(lr * b2 / bc2));
    h.decay_mul = (float)(1.0 - lr * hyper->weight_decay);
    h.decay_div = (float)(1.0 + lr * hyper->weight_decay);
    h.no_prox = hyper->no_prox ? 1 : 0;
    h.first_step = hyper->step == 1 ? 1 : 0;
    return GI2D_OK;
}

int gi2d_fit_forward(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                     const float *cov_b, const float *features, const float *opacity,
                     const float *gt_chw, int N, int H, int W, const gi2d_conv *conv, float *xys,
                     float *depths, int *radii, float *conics, int *num_tiles_hit, float *image,
                     float *v_image, float *loss_partials, void *stream)
{
    if (!ws || (kind != 0 && kind != 1) || !gt_chw) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    int rc = project_impl(kind, true, xyz, cov_a, cov_b, N, H, W, tiles_x, tiles_y, conv, xys,
                          depths, radii, conics, num_tiles_hit, ws, stream);
    if (rc != GI2D_OK) return rc;
    rc = gi2d_bin_tiles(ws, xys, radii, N, tiles_x, tiles_y, GI2D_BIN_COUNTED, stream);
    if (rc != GI2D_OK && rc != GI2D_E_PREV_OVERFLOW) return rc;
    const int late = rc;
    rc = raster_fwd_impl(ws, xys, conics, features, opacity, nullptr, N, H, W, conv, image, gt_chw,
                         v_image, loss_partials, stream);
    return rc != GI2D_OK ? rc : late;
}

int gi2d_fit_backward_step(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                           float *features, const float *opacity, int N, int H, int W,
                           const gi2d_conv *conv, const float *xys, const int *radii,
                           const float *conics, const float *v_image, float *const *exp_avg,
                           float *const *exp_avg_sq, float *const *exp_avg_diff,
                           float *const *neg_pre_grad, const gi2d_adan_hyper *hyper, void *stream)
{
    if (kind != 0 && kind != 1) return GI2D_E_INVALID;
    int rc = check_raster_args(ws, N, H, W, conv);
    if (rc != GI2D_OK) return rc;
    AdanScalars h;
    rc = adan_scalars(hyper, h);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (!xyz || !cov_a || (kind == 1 && !cov_b) || !features || !opacity || !xys || !radii ||
        !conics || (!v_image && T > 0) || !exp_avg || !exp_avg_sq || !exp_avg_diff ||
        !neg_pre_grad)
        return GI2D_E_INVALID;
    AdanTensors Tn;
    float *ps[4] = {xyz, cov_a, kind == 1 ? cov_b : xyz, features};
    for (int k = 0; k < ADAN_MAX_TENSORS; ++k) {
        const int src = k < 4 ? k : 0;
        if (k < 4 && (k != 2 || kind == 1) &&
            (!exp_avg[src] || !exp_avg_sq[src] || !exp_avg_diff[src] || !neg_pre_grad[src]))
            return GI2D_E_INVALID;
        Tn.p[k] = ps[src];
        Tn.g[k] = nullptr;
        Tn.m[k] = exp_avg[src];
        Tn.n[k] = exp_avg_sq[src];
        Tn.d[k] = exp_avg_diff[src];
        Tn.prev[k] = neg_pre_grad[src];
        Tn.numel[k] = 0;
    }
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    rc = raster_bwd_pass1(ws, xys, radii, conics, features, opacity, N, H, W, conv, v_image, st);
    if (rc != GI2D_OK) return rc;
    if (kind == 0)
        fit_tail_kernel<false><<<cdiv(N, 256), 256, 0, st>>>(
            Tn, h, xys, radii, N, H, W, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1,
            ws->d_len, ws->d_seg, ws->seg_cap, ws->d_partials, ws->d_slot_tab);
    else
        fit_tail_kernel<true><<<cdiv(N, 256), 256, 0, st>>>(
            Tn, h, xys, radii, N, H, W, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1,
            ws->d_len, ws->d_seg, ws->seg_cap, ws->d_partials, ws->d_slot_tab);
    return launch_status();
}

int gi2d_adan_step(int num_tensors, float *const *params, float *const *grads,
                   float *const *exp_avg, float *const *exp_avg_sq, float *const *exp_avg_diff,
                   float *const *neg_pre_grad, const int64_t *numel, const gi2d_adan_hyper *hyper,
                   const float *clip_dev, void *stream)
{
    if (num_tensors < 0) return GI2D_E_INVALID;
    AdanScalars h;
    int rcs = adan_scalars(hyper, h);
    if (rcs != GI2D_OK) return rcs;
    if (num_tensors == 0) return GI2D_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !exp_avg_diff || !neg_pre_grad || !numel)
        return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < num_tensors; base += ADAN_MAX_TENSORS) {
        AdanTensors T;
        const int cnt = num_tensors - base < ADAN_MAX_TENSORS ? num_tensors - base : ADAN_MAX_TENSORS;
        long long longest = 0;
        for (int k = 0; k < ADAN_MAX_TENSORS; ++k) {
            const int src = base + (k < cnt ? k : 0);
            if (numel[src] < 0) return GI2D_E_INVALID;
            if (numel[src] > 0 && (!params[src] || !grads[src] || !exp_avg[src] ||
                                   !exp_avg_sq[src] || !exp_avg_diff[src] || !neg_pre_grad[src]))
                return GI2D_E_INVALID;
            T.p[k] = params[src];
            T.g[k] = grads[src];
            T.m[k] = exp_avg[src];
            T.n[k] = exp_avg_sq[src];
            T.d[k] = exp_avg_diff[src];
            T.prev[k] = neg_pre_grad[src];
            T.numel[k] = k < cnt ? (long long)numel[src] : 0;
            if (T.numel[k] > longest) longest = T.numel[k];
        }
        if (longest == 0) continue;
        long long blocks = (longest + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        adan_step_kernel<<<dim3((unsigned)blocks, (unsigned)cnt), 256, 0, st>>>(T, h, clip_dev);
        int rc = launch_status();
        if (rc != GI2D_OK) return rc;
    }
    return GI2D_OK;
}

}  // extern "C"
