// gi2d.hip -- MI355X (gfx950, wave64) kernels and C ABI of the accumulated-summation
// 2D Gaussian rasterizer: projection, tile binning, rasterize_sum forward/backward.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see build.py).
// -ffp-contract=off is part of the contract: the projection's integer outputs (radii, tile
// boxes) must not depend on the compiler's choice of fused multiply-adds.  Where a fused
// multiply-add is wanted in a hot loop it is written explicitly with __builtin_fmaf.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <new>

#include "../../include/gi2d.h"

#define WAVE 64
// Workgroup size of the per-Gaussian kernels that issue scattered atomics: one wave, so that
// the launch spreads over every CU (a CU retires scattered atomics slowly; many CUs do not).
#define PG_THREADS 64
// Per-tile counters live one per 128-byte line: device-scope atomics on one line serialise at
// the memory side, and ~48 Gaussians per tile hammering 32 neighbouring counters of one line
// made the two atomic passes the slowest kernels of the frame.
#define CNT_STRIDE 32

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------

// status words written by the scan kernel and read by everything after it
struct DevStatus {
    int overflow;        // 1: the lists do not fit, later kernels skip their work
    int pad;
    long long total;     // number of (tile, Gaussian) intersections
    long long generation;
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
// projection (one lane per Gaussian)
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void project_tail(int g, float mx, float my, float sxx, float sxy,
                                             float syy, int H, int W, int tiles_x, int tiles_y,
                                             float center_bias, float *xys, float *depths,
                                             int *radii, float *conics, int *num_tiles_hit,
                                             int *tile_count, int row0, int row1)
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
        const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = box.x0; tx < box.x1; ++tx)
                atomicAdd(&tile_count[((ty - row0) * tiles_x + tx) * CNT_STRIDE], 1);
    }
}

__global__ __launch_bounds__(256) void project_cholesky_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ L, int N, int H, int W, int tiles_x,
    int tiles_y, float center_bias, float *__restrict__ xys, float *__restrict__ depths,
    int *__restrict__ radii, float *__restrict__ conics, int *__restrict__ num_tiles_hit,
    int *tile_count, int row0, int row1)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float l11 = L[3 * g], l21 = L[3 * g + 1], l22 = L[3 * g + 2];
    float sxx = l11 * l11;
    float sxy = l11 * l21;
    float syy = l21 * l21 + l22 * l22;
    project_tail(g, means[2 * g], means[2 * g + 1], sxx, sxy, syy, H, W, tiles_x, tiles_y,
                 center_bias, xys, depths, radii, conics, num_tiles_hit, tile_count, row0, row1);
}

__global__ __launch_bounds__(256) void project_rs_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ scales,
    const float *__restrict__ rot, int N, int H, int W, int tiles_x, int tiles_y,
    float center_bias, float *__restrict__ xys, float *__restrict__ depths,
    int *__restrict__ radii, float *__restrict__ conics, int *__restrict__ num_tiles_hit,
    int *tile_count, int row0, int row1)
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
    project_tail(g, means[2 * g], means[2 * g + 1], sxx, sxy, syy, H, W, tiles_x, tiles_y,
                 center_bias, xys, depths, radii, conics, num_tiles_hit, tile_count, row0, row1);
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
// tile binning: count -> scan -> fill -> per-tile sort
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void zero_ints_kernel(int *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(256) void count_tiles_kernel(const float *__restrict__ xys,
                                                           const int *__restrict__ radii, int N,
                                                           int tiles_x, int tiles_y, int row0,
                                                           int row1, int *tile_count)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    int r = radii[g];
    if (r <= 0) return;
    TileBox box = tile_box(xys[2 * g], xys[2 * g + 1], (float)r, tiles_x, tiles_y);
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = box.x0; tx < box.x1; ++tx)
            atomicAdd(&tile_count[((ty - row0) * tiles_x + tx) * CNT_STRIDE], 1);
}

// One workgroup: exclusive scan of the per-tile counts into tile_offsets[T+1]; leaves the
// counters at zero so that the fill kernel can use them as per-tile cursors.
__global__ __launch_bounds__(1024) void scan_tiles_kernel(int *tile_count, int *tile_offsets, int T,
                                                           long long ids_cap, DevStatus *status,
                                                           long long generation)
{
    __shared__ long long wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (T + 1023) / 1024;
    const int b = min(T, tid * chunk), e = min(T, b + chunk);
    // up to SCAN_REG counters per lane stay in registers: one round trip to memory, not two
    constexpr int SCAN_REG = 8;
    int c[SCAN_REG];
    long long s = 0;
    if (chunk <= SCAN_REG) {
#pragma unroll
        for (int u = 0; u < SCAN_REG; ++u)
            c[u] = (b + u < e) ? tile_count[(size_t)(b + u) * CNT_STRIDE] : 0;
#pragma unroll
        for (int u = 0; u < SCAN_REG; ++u) s += c[u];
    } else {
        for (int i = b; i < e; ++i) s += tile_count[(size_t)i * CNT_STRIDE];
    }
    long long inc = s;
    for (int d = 1; d < 64; d <<= 1) {
        long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    long long wbase = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        long long v = wave_tot[w];
        if (w < wave) wbase += v;
        total += v;
    }
    const bool over = total > ids_cap || total >= (long long)INT_MAX;
    long long run = wbase + inc - s;
    if (chunk <= SCAN_REG) {
#pragma unroll
        for (int u = 0; u < SCAN_REG; ++u)
            if (b + u < e) {
                tile_offsets[b + u] = (int)(run < (long long)INT_MAX ? run : (long long)INT_MAX);
                run += c[u];
                tile_count[(size_t)(b + u) * CNT_STRIDE] = 0;
            }
    } else {
        for (int i = b; i < e; ++i) {
            int ci = tile_count[(size_t)i * CNT_STRIDE];
            tile_offsets[i] = (int)(run < (long long)INT_MAX ? run : (long long)INT_MAX);
            run += ci;
            tile_count[(size_t)i * CNT_STRIDE] = 0;
        }
    }
    if (tid == 0) {
        // tile_offsets[T] doubles as the overflow sentinel every later kernel reads together
        // with its own list bounds: -1 = the lists do not fit, skip
        tile_offsets[T] = over ? -1 : (int)total;
        status->overflow = over ? 1 : 0;
        status->total = total;
        status->generation = generation;
    }
}

__global__ __launch_bounds__(256) void fill_tiles_kernel(const float *__restrict__ xys,
                                                          const int *__restrict__ radii, int N,
                                                          int tiles_x, int tiles_y, int row0,
                                                          int row1,
                                                          const int *__restrict__ tile_offsets,
                                                          int *tile_cursor, int *ids, int T,
                                                          const DevStatus *status,
                                                          volatile long long *h_status)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        // Host-visible copy of the count (pinned memory, read without synchronisation by the
        // next calls).  Issued first so that the slow host write overlaps the rest of the kernel.
        const long long total = status->total, gen = status->generation;
        h_status[0] = total;
        h_status[1] = gen;
        if (status->overflow) h_status[2] = gen;
    }
    if (g >= N) return;
    const int r = radii[g];
    const float cx = xys[2 * g], cy = xys[2 * g + 1];
    if (tile_offsets[T] < 0) return;  // overflow
    if (r <= 0) return;
    TileBox box = tile_box(cx, cy, (float)r, tiles_x, tiles_y);
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
    // Four returning atomics are put in flight before the first dependent store: a lane's
    // round trips overlap instead of adding up.
    int tx = box.x0, ty = y0;
    for (int k0 = 0; k0 < cnt; k0 += 4) {
        int t[4], pos[4], off[4];
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
            pos[u] = 0;
            off[u] = 0;
            if (t[u] >= 0) {
                pos[u] = atomicAdd(&tile_cursor[(size_t)t[u] * CNT_STRIDE], 1);
                off[u] = tile_offsets[t[u]];
            }
        }
        // slot 0 is always live here; dead slots repeat its store so that the four stores
        // issue back to back without control flow (and the waits it brings) between them
        const int first = off[0] + pos[0];
#pragma unroll
        for (int u = 0; u < 4; ++u) ids[t[u] >= 0 ? off[u] + pos[u] : first] = g;
    }
}

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

#define SORT_LDS_MAX 4096

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

// One wave per tile: puts the tile's Gaussian ids in ascending order (the fill order depends
// on atomic arrival) and returns the tile's cursor to zero for the next binning.  The launch
// lasts as long as its slowest tile, so lists up to 256 entries stay on the register path;
// longer ones use a bitonic network in LDS, and beyond the LDS budget in place in memory.
__global__ __launch_bounds__(WAVE) void sort_tiles_kernel(const int *__restrict__ tile_offsets,
                                                          int *ids, int *tile_cursor, int T)
{
    __shared__ int s_ids[SORT_LDS_MAX];
    const int t = blockIdx.x, lane = threadIdx.x;
    // the three list words are independent loads: one round trip
    const int beg = tile_offsets[t], fin = tile_offsets[t + 1], sentinel = tile_offsets[T];
    const int n = sentinel < 0 ? 0 : fin - beg;
    if (n > 1) {
        if (n <= WAVE) {
            rank_sort_regs<1>(ids, beg, n, lane);
        } else if (n <= 2 * WAVE) {
            rank_sort_regs<2>(ids, beg, n, lane);
        } else if (n <= 4 * WAVE) {
            rank_sort_regs<4>(ids, beg, n, lane);
        } else if (n <= SORT_LDS_MAX) {
            for (int i = lane; i < n; i += WAVE) s_ids[i] = ids[beg + i];
            __syncthreads();
            bitonic_sort(s_ids, n, lane, WAVE);
            for (int i = lane; i < n; i += WAVE) ids[beg + i] = s_ids[i];
        } else {
            bitonic_sort(ids + beg, n, lane, WAVE);
        }
    }
    if (lane == 0) tile_cursor[(size_t)t * CNT_STRIDE] = 0;
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
    const float det = a * c - b * b;
    const float ratio = op / alpha_min;
    if (!(ratio >= 1.0f)) {
        x1 = x0 - 1;  // opacity below the cut: no pixel can reach alpha_min
    } else if (a > 0.f && c > 0.f && det > 1.0e-3f * a * c && ratio < 3.0e38f) {
        const float tau = 1.001f * __logf(ratio) + 1.0e-3f;
        const float hx = 1.001f * sqrtf(2.0f * tau * c / det) + 1.0e-3f;
        const float hy = 1.001f * sqrtf(2.0f * tau * a / det) + 1.0e-3f;
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
// ------------------------------------------------------------------------------------------

#define LOG2E 1.4426950408889634f

__global__ __launch_bounds__(256) void rasterize_sum_fwd_kernel(
    const int *__restrict__ tile_offsets, const int *__restrict__ ids,
    const float *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacity, float bg0, float bg1,
    float bg2, int H, int W, int tiles_x, int row0, int T, gi2d_conv cv, int cull,
    float *__restrict__ out)
{
    __shared__ float4 s_a[256];  // gx, gy, -0.5*a*log2e, -b*log2e
    __shared__ float4 s_b[256];  // -0.5*c*log2e, opacity, colour r, colour g
    __shared__ float s_c[256];   // colour b
    __shared__ int s_m[256];     // quadrant mask of the entry (bit = 2*qy + qx)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_tile(blockIdx.x, T);
    const int tx = t % tiles_x, ty = row0 + t / tiles_x;
    const int lx = ((wave & 1) << 3) + (lane & 7), ly = ((wave >> 1) << 3) + (lane >> 3);
    const int j = tx * GI2D_BLOCK + lx, i = ty * GI2D_BLOCK + ly;
    const bool inside = (i < H) && (j < W);
    const float px = (float)j + cv.pixel_offset, py = (float)i + cv.pixel_offset;
    float r = 0.f, g = 0.f, b = 0.f;
    // list bounds and the overflow sentinel: three independent loads, one round trip
    const int beg = tile_offsets[t], end = tile_offsets[t + 1], sentinel = tile_offsets[T];
    if (sentinel < 0) {
        r = g = b = __builtin_nanf("");  // the lists did not fit: never a silently wrong image
    } else {
        const int x_hi = min(GI2D_BLOCK, W - tx * GI2D_BLOCK) - 1;
        const int y_hi = min(GI2D_BLOCK, H - ty * GI2D_BLOCK) - 1;
        const float ox = (float)(tx * GI2D_BLOCK) + cv.pixel_offset;
        const float oy = (float)(ty * GI2D_BLOCK) + cv.pixel_offset;
        for (int base = beg; base < end; base += 256) {
            const int n = min(256, end - base);
            __syncthreads();
            int mask = 0;
            if (tid < n) {
                const int gid = ids[base + tid];
                const float gx = xys[2 * gid], gy = xys[2 * gid + 1];
                const float ca = conics[3 * gid], cb = conics[3 * gid + 1], cc = conics[3 * gid + 2];
                const float op = opacity[gid];
                s_a[tid] = make_float4(gx, gy, (-0.5f * LOG2E) * ca, (-LOG2E) * cb);
                s_b[tid] = make_float4((-0.5f * LOG2E) * cc, op, colors[3 * gid],
                                       colors[3 * gid + 1]);
                s_c[tid] = colors[3 * gid + 2];
                int x0 = 0, x1 = x_hi, y0 = 0, y1 = y_hi;
                if (cull) support_rect(gx, gy, ca, cb, cc, op, cv.alpha_min, ox, oy, x0, x1, y0, y1);
                if (x0 <= x1 && y0 <= y1) {
                    const int mx = (x0 < 8 ? 1 : 0) | (x1 >= 8 ? 2 : 0);
                    const int my = (y0 < 8 ? 1 : 0) | (y1 >= 8 ? 2 : 0);
                    mask = ((my & 1) ? mx : 0) | ((my & 2) ? (mx << 2) : 0);
                }
            }
            s_m[tid] = mask;
            __syncthreads();
            for (int c0 = 0; c0 < n; c0 += 64) {
                unsigned long long bits = __ballot((s_m[c0 + lane] >> wave) & 1);
                while (bits) {
                    const int k = c0 + (int)__builtin_ctzll(bits);
                    bits &= bits - 1;
                    const float4 A = s_a[k];
                    const float4 B = s_b[k];
                    const float cb = s_c[k];
                    const float dx = A.x - px, dy = A.y - py;
                    // p = -sigma*log2(e) = (A'dx + B'dy)dx + C'dy*dy
                    const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                    const float p = __builtin_fmaf(B.x * dy, dy, q * dx);
                    const float e = __builtin_amdgcn_exp2f(p);
                    const float alpha = fminf(cv.alpha_cap, B.y * e);
                    const float w = (p > 0.f || alpha < cv.alpha_min) ? 0.f : alpha;
                    r = __builtin_fmaf(B.z, w, r);
                    g = __builtin_fmaf(B.w, w, g);
                    b = __builtin_fmaf(cb, w, b);
                }
            }
        }
        r = r + cv.bg_weight * bg0;
        g = g + cv.bg_weight * bg1;
        b = b + cv.bg_weight * bg2;
    }
    if (inside) {
        float *o = out + 3 * ((size_t)(i - row0 * GI2D_BLOCK) * W + j);
        o[0] = r;
        o[1] = g;
        o[2] = b;
    }
}

// ------------------------------------------------------------------------------------------
// rasterize_sum backward.
//
// Pass 1 (one workgroup per tile): lane = (list entry, 4-row strip of the tile).  Each lane
// walks only the pixels of its strip that lie inside the entry's support rectangle (the
// bounding box of the ellipse alpha >= alpha_min), keeps the entry's nine partial sums in
// registers, the four strips are added with two DPP steps, and the tile's sums are written to
// the entry's own slot of a [slots][9] slab -- no cross-lane tree, no atomics.
// Pass 2 (one lane per Gaussian): adds the Gaussian's slots in tile order; a slot is found by
// binary search because tile lists are ascending in Gaussian id.
// ------------------------------------------------------------------------------------------

#define PART_STRIDE 9
#define SLOT_TAB 4  // slots per Gaussian remembered by pass 1 (tile boxes are mostly <= 2x2)

__global__ __launch_bounds__(256) void rasterize_sum_bwd_kernel(
    const int *__restrict__ tile_offsets, const int *__restrict__ ids,
    const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0, int row1,
    int T, gi2d_conv cv, int cull, const float *__restrict__ v_out,
    float *__restrict__ partials, int *__restrict__ slot_tab)
{
    __shared__ float4 s_v[256];
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, T);
    const int beg = tile_offsets[t], end = tile_offsets[t + 1], sentinel = tile_offsets[T];
    if (sentinel < 0 || beg == end) return;
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
    const int x_hi = min(GI2D_BLOCK, W - tx * GI2D_BLOCK) - 1;
    const int y_hi = min(GI2D_BLOCK, H - ty * GI2D_BLOCK) - 1;
    const float ox = (float)(tx * GI2D_BLOCK) + cv.pixel_offset;
    const float oy = (float)(ty * GI2D_BLOCK) + cv.pixel_offset;
    const int q = tid & 3;
    for (int base = beg; base < end; base += 64) {
        const int e = base + (tid >> 2);
        float acc[PART_STRIDE];
#pragma unroll
        for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
        int gid = 0, rad = 0;
        float gx = 0.f, gy = 0.f;
        if (e < end) {
            gid = ids[e];
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
            // the four lanes of the quad take the rectangle's pixels round-robin
            const int step_y = cnt ? 4 / w : 0, step_x = cnt ? 4 - step_y * w : 0;
            int x = x0 + q, y = y0;
            while (cnt && x > x1) {
                x -= w;
                ++y;
            }
            const float pa = (-0.5f * LOG2E) * a, pb = (-LOG2E) * b, pc = (-0.5f * LOG2E) * c;
            for (int k = q; k < cnt; k += 4) {
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
                acc[0] = __builtin_fmaf(vs, __builtin_fmaf(b, dy, a * dx), acc[0]);
                acc[1] = __builtin_fmaf(vs, __builtin_fmaf(c, dy, b * dx), acc[1]);
                const float hx = vs * dx, hy = vs * dy;
                acc[2] = __builtin_fmaf(0.5f * hx, dx, acc[2]);
                acc[3] = __builtin_fmaf(hx, dy, acc[3]);
                acc[4] = __builtin_fmaf(0.5f * hy, dy, acc[4]);
                x += step_x;
                y += step_y;
                if (x > x1) {
                    x -= w;
                    ++y;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PART_STRIDE; ++k) acc[k] = quad_sum(acc[k]);
        if (e < end && q == 0) {
            float *dst = partials + (size_t)e * PART_STRIDE;
#pragma unroll
            for (int k = 0; k < PART_STRIDE; ++k) dst[k] = acc[k];
            // where the second pass finds this slot without searching: index of this tile in
            // the Gaussian's own (band-clipped) tile box, first SLOT_TAB tiles only
            if (rad > 0) {
                TileBox box = tile_box(gx, gy, (float)rad, tiles_x, tiles_y);
                const int by0 = max(box.y0, row0);
                const int kb = (ty - by0) * (box.x1 - box.x0) + (tx - box.x0);
                if (kb >= 0 && kb < SLOT_TAB) slot_tab[(size_t)gid * SLOT_TAB + kb] = e;
            }
        }
    }
}

__global__ __launch_bounds__(256) void gather_grads_kernel(
    const float *__restrict__ xys, const int *__restrict__ radii, int N, int tiles_x, int tiles_y,
    int row0, int row1, int T, const int *__restrict__ tile_offsets, const int *__restrict__ ids,
    const float *__restrict__ partials, const int *__restrict__ slot_tab,
    float *__restrict__ v_xy, float *__restrict__ v_conic, float *__restrict__ v_colors,
    float *__restrict__ v_opacity)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float acc[PART_STRIDE];
#pragma unroll
    for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
    const int r = radii[g];
    const float cx = xys[2 * g], cy = xys[2 * g + 1];
    const int4 tab = *reinterpret_cast<const int4 *>(slot_tab + (size_t)g * SLOT_TAB);
    const int sentinel = tile_offsets[T];
    if (r > 0 && sentinel >= 0) {
        TileBox box = tile_box(cx, cy, (float)r, tiles_x, tiles_y);
        const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
        const int w = box.x1 - box.x0;
        const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
        // the first SLOT_TAB tiles: slots remembered by pass 1, loads independent of each other
        {
            const int slots[SLOT_TAB] = {tab.x, tab.y, tab.z, tab.w};
            float part[SLOT_TAB][PART_STRIDE];
#pragma unroll
            for (int k = 0; k < SLOT_TAB; ++k) {
                const float *src = partials + (size_t)(k < cnt ? slots[k] : 0) * PART_STRIDE;
#pragma unroll
                for (int u = 0; u < PART_STRIDE; ++u) part[k][u] = (k < cnt) ? src[u] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < SLOT_TAB; ++k)
#pragma unroll
                for (int u = 0; u < PART_STRIDE; ++u) acc[u] += part[k][u];
        }
        // larger boxes: the rest by binary search (tile lists ascend in Gaussian id)
        for (int k = SLOT_TAB; k < cnt; ++k) {
            const int ty = y0 + k / w, tx = box.x0 + k % w;
            const int t = (ty - row0) * tiles_x + tx;
            int lo = tile_offsets[t], hi = tile_offsets[t + 1];
            const int fin = hi;
            while (lo < hi) {  // first slot with id >= g
                int mid = (lo + hi) >> 1;
                if (ids[mid] < g)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo < fin && ids[lo] == g) {
                const float *src = partials + (size_t)lo * PART_STRIDE;
#pragma unroll
                for (int u = 0; u < PART_STRIDE; ++u) acc[u] += src[u];
            }
        }
    }
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
        const float np0 = h.first_step ? -g : prev[i];   // optimizer.py:187-189
        const float diff = np0 + g;                       // :323
        const float mi = m[i] * h.beta1 + h.omb1 * g;     // :325-326
        const float di = d[i] * h.beta2 + h.omb2 * diff;  // :328-330
        const float u = diff * h.beta2 + g;               // :332-333
        const float ni = n[i] * h.beta3 + h.omb3 * (u * u);  // :334-338
        const float denom = sqrtf(ni) / h.bc3_sqrt + h.eps;  // :340-342
        float pi = p[i];
        if (h.no_prox) pi = pi * h.decay_mul;             // :348
        pi = pi + h.neg_step * (mi / denom);              // :349 / :355
        pi = pi + h.neg_step_diff * (di / denom);         // :350 / :356
        if (!h.no_prox) pi = pi / h.decay_div;            // :360
        m[i] = mi;
        d[i] = di;
        n[i] = ni;
        p[i] = pi;
        prev[i] = -g;                                     // :361-362
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

struct gi2d_workspace {
    int device = 0;
    int tiles_cap = 0;
    int *d_count = nullptr;     // [tiles_cap] per-tile counters / cursors, zero between calls
    int *d_offsets = nullptr;   // [tiles_cap + 1]
    int64_t ids_cap = 0;
    int *d_ids = nullptr;
    int64_t part_cap = 0;
    float *d_partials = nullptr;
    int64_t slot_cap = 0;
    int *d_slot_tab = nullptr;  // [n][SLOT_TAB] slot of a Gaussian in its first tiles (backward)
    DevStatus *d_status = nullptr;
    long long *h_status = nullptr;   // pinned: [0] total, [1] generation, [2] overflow generation
    long long *h_status_dev = nullptr;
    int64_t generation = 0;
    int64_t overflow_ack = 0;
    int tiles_x = 0, tiles_y = 0, n_binned = 0;
    int row0 = 0, row1 = INT_MAX;          // tile rows requested by the caller (a band)
    int bin_row0 = 0, bin_row1 = 0;        // tile rows of the current lists
    bool counters_dirty = false;  // counters may be non-zero (a binning was cut short)
    bool counted = false;         // project_*_fwd accumulated counts for (n, tiles) below
    int counted_n = 0, counted_tx = 0, counted_ty = 0, counted_r0 = 0, counted_r1 = 0;
    int cull = 1;
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

namespace {
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
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
            return "an earlier asynchronous frame overflowed the tile lists (its image is NaN); "
                   "the workspace was grown, call again";
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
    if (ws->d_offsets) (void)hipFree(ws->d_offsets);
    if (ws->d_ids) (void)hipFree(ws->d_ids);
    if (ws->d_partials) (void)hipFree(ws->d_partials);
    if (ws->d_slot_tab) (void)hipFree(ws->d_slot_tab);
    if (ws->d_status) (void)hipFree(ws->d_status);
    if (ws->h_status) (void)hipHostFree(ws->h_status);
    delete ws;
    return GI2D_OK;
}

// (Re)allocates the per-tile arrays; callers have made the device current.
static int ensure_tiles(gi2d_workspace *ws, int T)
{
    if (T <= ws->tiles_cap) return GI2D_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (ws->d_count) (void)hipFree(ws->d_count);
    if (ws->d_offsets) (void)hipFree(ws->d_offsets);
    ws->d_count = ws->d_offsets = nullptr;
    ws->tiles_cap = 0;
    HIP_TRY(hipMalloc(&ws->d_count, sizeof(int) * CNT_STRIDE * (size_t)T));
    HIP_TRY(hipMalloc(&ws->d_offsets, sizeof(int) * ((size_t)T + 1)));
    HIP_TRY(hipMemset(ws->d_count, 0, sizeof(int) * CNT_STRIDE * (size_t)T));
    HIP_TRY(hipMemset(ws->d_offsets, 0, sizeof(int) * ((size_t)T + 1)));
    ws->tiles_cap = T;
    ws->counters_dirty = false;
    ws->counted = false;
    ws->tiles_x = ws->tiles_y = 0;
    return GI2D_OK;
}

static int ensure_ids(gi2d_workspace *ws, int64_t need)
{
    if (need <= ws->ids_cap) return GI2D_OK;
    if (need >= (int64_t)INT_MAX) return GI2D_E_CAPACITY;
    HIP_TRY(hipDeviceSynchronize());
    if (ws->d_ids) (void)hipFree(ws->d_ids);
    ws->d_ids = nullptr;
    ws->ids_cap = 0;
    HIP_TRY(hipMalloc(&ws->d_ids, sizeof(int) * (size_t)need));
    ws->ids_cap = need;
    return GI2D_OK;
}

int gi2d_workspace_reserve(gi2d_workspace *ws, int64_t max_intersections, int max_tiles)
{
    if (!ws || max_intersections < 0 || max_tiles < 0) return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    if (!guard.ok) return GI2D_E_INVALID;
    int rc = ensure_tiles(ws, max_tiles);
    if (rc != GI2D_OK) return rc;
    return ensure_ids(ws, max_intersections);
}

int gi2d_workspace_set_option(gi2d_workspace *ws, int option, int value)
{
    if (!ws) return GI2D_E_INVALID;
    switch (option) {
        case GI2D_OPT_CULL: ws->cull = value ? 1 : 0; return GI2D_OK;
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
    info[0] = ws->ids_cap;
    info[1] = ws->h_status[1] == ws->generation ? ws->h_status[0] : -1;
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

// Prepares ws for a projection that also counts tiles.
static int begin_fused_count(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, hipStream_t st)
{
    const int r0 = min(ws->row0, tiles_y), r1 = min(ws->row1, tiles_y);
    int rc = ensure_tiles(ws, max(1, (r1 - r0) * tiles_x));
    if (rc != GI2D_OK) return rc;
    if (ws->counters_dirty || ws->counted) {
        int T = ws->tiles_cap * CNT_STRIDE;
        zero_ints_kernel<<<cdiv(T, 256), 256, 0, st>>>(ws->d_count, T);
        rc = launch_status();
        if (rc != GI2D_OK) return rc;
    }
    ws->counters_dirty = true;
    ws->counted = true;
    ws->counted_n = N;
    ws->counted_tx = tiles_x;
    ws->counted_ty = tiles_y;
    ws->counted_r0 = r0;
    ws->counted_r1 = r1;
    return GI2D_OK;
}

int gi2d_project_cholesky_fwd(const float *means2d, const float *L_elements, int N, int H, int W,
                              int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                              float *depths, int *radii, float *conics, int *num_tiles_hit,
                              gi2d_workspace *ws, void *stream)
{
    int rc = check_proj_args(means2d, L_elements, N, H, W, tiles_x, tiles_y, conv, xys, depths,
                             radii, conics, num_tiles_hit);
    if (rc != GI2D_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    int *tile_count = nullptr;
    if (ws) {
        rc = begin_fused_count(ws, N, tiles_x, tiles_y, st);
        if (rc != GI2D_OK) return rc;
        tile_count = ws->d_count;
    }
    if (N == 0) return GI2D_OK;
    project_cholesky_fwd_kernel<<<cdiv(N, PG_THREADS), PG_THREADS, 0, st>>>(
        means2d, L_elements, N, H, W, tiles_x, tiles_y, conv->center_bias, xys, depths, radii,
        conics, num_tiles_hit, tile_count, ws ? ws->counted_r0 : 0, ws ? ws->counted_r1 : 0);
    return launch_status();
}

int gi2d_project_rs_fwd(const float *means2d, const float *scales, const float *rotation, int N,
                        int H, int W, int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                        float *depths, int *radii, float *conics, int *num_tiles_hit,
                        gi2d_workspace *ws, void *stream)
{
    int rc = check_proj_args(means2d, scales, N, H, W, tiles_x, tiles_y, conv, xys, depths, radii,
                             conics, num_tiles_hit);
    if (rc != GI2D_OK) return rc;
    if (N > 0 && !rotation) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int *tile_count = nullptr;
    if (ws) {
        rc = begin_fused_count(ws, N, tiles_x, tiles_y, st);
        if (rc != GI2D_OK) return rc;
        tile_count = ws->d_count;
    }
    if (N == 0) return GI2D_OK;
    project_rs_fwd_kernel<<<cdiv(N, PG_THREADS), PG_THREADS, 0, st>>>(means2d, scales, rotation, N, H, W,
                                                       tiles_x, tiles_y, conv->center_bias, xys,
                                                       depths, radii, conics, num_tiles_hit,
                                                       tile_count, ws ? ws->counted_r0 : 0,
                                                       ws ? ws->counted_r1 : 0);
    return launch_status();
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

    // an earlier asynchronous frame that did not fit: grow for it, tell the caller
    if (ws->h_status[2] > ws->overflow_ack) {
        ws->overflow_ack = ws->h_status[2];
        prev_overflow = true;
    }

    bool counted = (flags & GI2D_BIN_COUNTED) && ws->counted && ws->counted_n == N &&
                   ws->counted_tx == tiles_x && ws->counted_ty == tiles_y &&
                   ws->counted_r0 == r0 && ws->counted_r1 == r1;
    if ((flags & GI2D_BIN_COUNTED) && !counted) return GI2D_E_STATE;
    if (!counted) {
        if (nogrow && T > ws->tiles_cap) return GI2D_E_CAPACITY;
        int rc = ensure_tiles(ws, max(1, T));
        if (rc != GI2D_OK) return rc;
        if (ws->counters_dirty || ws->counted) {
            zero_ints_kernel<<<cdiv(ws->tiles_cap * CNT_STRIDE, 256), 256, 0, st>>>(
                ws->d_count, ws->tiles_cap * CNT_STRIDE);
            rc = launch_status();
            if (rc != GI2D_OK) return rc;
        }
        ws->counters_dirty = true;
        if (N > 0) {
            count_tiles_kernel<<<cdiv(N, PG_THREADS), PG_THREADS, 0, st>>>(xys, radii, N, tiles_x, tiles_y, r0,
                                                             r1, ws->d_count);
            rc = launch_status();
            if (rc != GI2D_OK) return rc;
        }
    }
    ws->counted = false;

    // Size the lists.  Steady state needs no synchronisation: the capacity keeps a 2x margin
    // over the largest count seen from completed frames.
    if (!nogrow) {
        int64_t seen = ws->h_status[0];
        // a new problem shape says nothing about its count: learn it once, synchronously
        if (ws->ids_cap == 0 || N != ws->n_binned || tiles_x != ws->tiles_x ||
            tiles_y != ws->tiles_y || r0 != ws->bin_row0 || r1 != ws->bin_row1)
            sync = true;
        if (!sync && 2 * seen > ws->ids_cap) {
            int rc = ensure_ids(ws, 2 * seen + 1024);
            if (rc != GI2D_OK) return rc;
        }
    }

    ws->generation += 1;
    ws->tiles_x = tiles_x;
    ws->tiles_y = tiles_y;
    ws->n_binned = N;
    ws->bin_row0 = r0;
    ws->bin_row1 = r1;
    scan_tiles_kernel<<<1, 1024, 0, st>>>(ws->d_count, ws->d_offsets, T,
                                          sync && !nogrow ? (long long)INT_MAX - 1
                                                          : (long long)ws->ids_cap,
                                          ws->d_status, ws->generation);
    int rc = launch_status();
    if (rc != GI2D_OK) return rc;
    ws->counters_dirty = false;  // the scan left the counters at zero

    if (sync) {
        DevStatus hs;
        HIP_TRY(hipMemcpyAsync(&hs, ws->d_status, sizeof(hs), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        int64_t total = hs.total;
        if (total > ws->ids_cap) {
            if (nogrow) {
                ws->overflow_ack = ws->generation;  // reported here, not again on the next call
                return GI2D_E_CAPACITY;
            }
            rc = ensure_ids(ws, total + total / 2 + 1024);
            if (rc != GI2D_OK) return rc;
        }
    }
    {
        // always launched: its first lane also publishes the count to the host
        ws->counters_dirty = true;
        fill_tiles_kernel<<<max(1, cdiv(N, PG_THREADS)), PG_THREADS, 0, st>>>(
            xys, radii, N, tiles_x, tiles_y, r0, r1, ws->d_offsets, ws->d_count, ws->d_ids, T,
            ws->d_status, ws->h_status_dev);
        rc = launch_status();
        if (rc != GI2D_OK) return rc;
    }
    if (T > 0) {
        sort_tiles_kernel<<<T, WAVE, 0, st>>>(ws->d_offsets, ws->d_ids, ws->d_count, T);
        rc = launch_status();
        if (rc != GI2D_OK) return rc;
        ws->counters_dirty = false;  // the sort returned every cursor to zero
    }
    return prev_overflow ? GI2D_E_PREV_OVERFLOW : GI2D_OK;
}

int gi2d_copy_bins(gi2d_workspace *ws, int *tile_offsets, int *ids, int64_t ids_capacity,
                   void *stream)
{
    if (!ws || !tile_offsets || ws->generation == 0) return ws ? GI2D_E_STATE : GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(st));
    if (ws->h_status[2] == ws->generation) return GI2D_E_CAPACITY;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    HIP_TRY(hipMemcpyAsync(tile_offsets, ws->d_offsets, sizeof(int) * ((size_t)T + 1),
                           hipMemcpyDefault, st));
    int64_t total = ws->h_status[0];
    if (ids && total > 0) {
        if (total > ids_capacity) return GI2D_E_CAPACITY;
        HIP_TRY(hipMemcpyAsync(ids, ws->d_ids, sizeof(int) * (size_t)total, hipMemcpyDefault, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return GI2D_OK;
}

static int check_raster_args(gi2d_workspace *ws, int N, int H, int W, const gi2d_conv *conv)
{
    if (!ws || !conv || N < 0 || H < 1 || W < 1) return GI2D_E_INVALID;
    if (ws->generation == 0) return GI2D_E_STATE;
    if (ws->tiles_x != cdiv(W, GI2D_BLOCK) || ws->tiles_y != cdiv(H, GI2D_BLOCK) ||
        ws->n_binned != N)
        return GI2D_E_STATE;
    return GI2D_OK;
}

int gi2d_rasterize_sum_fwd(gi2d_workspace *ws, const float *xys, const float *conics,
                           const float *colors, const float *opacity, const float *background,
                           int N, int H, int W, const gi2d_conv *conv, float *out_img,
                           void *stream)
{
    int rc = check_raster_args(ws, N, H, W, conv);
    if (rc != GI2D_OK) return rc;
    if (ws->tiles_x * (ws->bin_row1 - ws->bin_row0) == 0) return GI2D_OK;  // empty band
    if (!out_img || (N > 0 && (!xys || !conics || !colors || !opacity))) return GI2D_E_INVALID;
    if (conv->bg_weight != 0.0f && !background) return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    float bg[3] = {0.f, 0.f, 0.f};
    if (conv->bg_weight != 0.0f) {
        // three floats, needed as kernel arguments
        HIP_TRY(hipMemcpyAsync(bg, background, sizeof(bg), hipMemcpyDefault, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (T == 0) return GI2D_OK;
    rasterize_sum_fwd_kernel<<<T, 256, 0, st>>>(ws->d_offsets, ws->d_ids, xys, conics, colors,
                                                opacity, bg[0], bg[1], bg[2], H, W, ws->tiles_x,
                                                ws->bin_row0, T, *conv, ws->cull, out_img);
    return launch_status();
}

int gi2d_rasterize_sum_bwd(gi2d_workspace *ws, const float *xys, const int *radii,
                           const float *conics, const float *colors, const float *opacity, int N,
                           int H, int W, const gi2d_conv *conv, const float *v_out, float *v_xys,
                           float *v_conics, float *v_colors, float *v_opacity, void *stream)
{
    int rc = check_raster_args(ws, N, H, W, conv);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    const bool empty_band = ws->tiles_x * (ws->bin_row1 - ws->bin_row0) == 0;
    if (!xys || !radii || !conics || !colors || !opacity || (!v_out && !empty_band) || !v_xys ||
        !v_conics || !v_colors || !v_opacity)
        return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    if (ws->part_cap < ws->ids_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->d_partials) (void)hipFree(ws->d_partials);
        ws->d_partials = nullptr;
        ws->part_cap = 0;
        HIP_TRY(hipMalloc(&ws->d_partials, sizeof(float) * PART_STRIDE * (size_t)ws->ids_cap));
        ws->part_cap = ws->ids_cap;
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
            ws->d_offsets, ws->d_ids, xys, radii, conics, colors, opacity, H, W, ws->tiles_x,
            ws->tiles_y, ws->bin_row0, ws->bin_row1, T, *conv, ws->cull, v_out, ws->d_partials,
            ws->d_slot_tab);
        rc = launch_status();
        if (rc != GI2D_OK) return rc;
    }
    gather_grads_kernel<<<cdiv(N, 256), 256, 0, st>>>(
        xys, radii, N, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1, T, ws->d_offsets,
        ws->d_ids, ws->d_partials, ws->d_slot_tab, v_xys, v_conics, v_colors, v_opacity);
    return launch_status();
}

int gi2d_adan_step(int num_tensors, float *const *params, float *const *grads,
                   float *const *exp_avg, float *const *exp_avg_sq, float *const *exp_avg_diff,
                   float *const *neg_pre_grad, const int64_t *numel, const gi2d_adan_hyper *hyper,
                   const float *clip_dev, void *stream)
{
    if (num_tensors < 0 || !hyper || hyper->step < 1) return GI2D_E_INVALID;
    if (num_tensors == 0) return GI2D_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !exp_avg_diff || !neg_pre_grad || !numel)
        return GI2D_E_INVALID;
    const double b1 = hyper->beta1, b2 = hyper->beta2, b3 = hyper->beta3, lr = hyper->lr;
    const double bc1 = 1.0 - pow(b1, (double)hyper->step);
    const double bc2 = 1.0 - pow(b2, (double)hyper->step);
    const double bc3 = 1.0 - pow(b3, (double)hyper->step);
    AdanScalars h;
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
