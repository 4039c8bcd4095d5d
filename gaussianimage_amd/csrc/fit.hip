// fit.hip -- the Adan step over all parameter tensors in one launch, and the fused fitting
// iteration (forward with loss epilogue; backward rasterisation + gather + projection backward +
// activation backward + Adan in place).
#include "adan.h"
#include "gather.h"
#include "host.h"
#include "projection.h"
#include "raster_bwd_tile.h"
#include "raster_fwd_tile.h"

// ------------------------------------------------------------------------------------------
// Adan step, fused: one launch over all parameter tensors, one read and one write of each of
// p, m, n, diff, neg_pre_grad (the reference's default path makes 18 multi-tensor passes).
// Operation order per element follows /root/reference/optimizer.py:319-362.
// ------------------------------------------------------------------------------------------

// What the chained variant needs to project the updated parameters and fill the next
// iteration's tile lists (into the workspace's second list set).
struct ChainArgs {
    float center_bias;
    ProjOut out;       // the projection outputs to overwrite (xys / radii are also read first)
    TileLists lists;   // the next frame's lists
    ListCull cull;
    RecordSink rec;    // the next frame's per-Gaussian records
    const DevStatus *status;
    volatile long long *h_status;
    long long generation;
};

// Per-Gaussian gradients of the rasteriser handed in from outside (GRADS = true): the sums a
// tile-row sharded fit has reduced over its ranks, in place of this workspace's partial sums.
struct GradIn {
    const float *v_xy, *v_conic, *v_colors;  // [N,2], [N,3], [N,3]
};

// CHAIN = true: having updated Gaussian g's parameters, the lane goes straight on to the next
// iteration's projection of g (model activations included) -- outputs overwritten in place,
// tile lists of the next frame filled -- so the next iteration needs no projection launch.
//
// gridDim.y = the images of a batch that is fitted in lockstep: image f has Gaussians
// [f*N, (f+1)*N) of every parameter, state and per-Gaussian tensor and its own run of tile lists;
// ids in the lists are image-local.
template <bool RS, bool CHAIN, bool GRADS>
__global__ __launch_bounds__(256) void fit_tail_kernel(
    AdanTensors T, AdanScalars h, float *xys, int *radii, int N, int H, int W, int tiles_x,
    int tiles_y, int row0, int row1, const int *__restrict__ tile_len,
    const int *__restrict__ seg, int C, const float *__restrict__ partials,
    const float *__restrict__ gslots, size_t gslot_n, int stamp, ChainArgs ch, GradIn gin)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (CHAIN && g == 0 && f == 0) publish_status(ch.status, ch.h_status, ch.generation);
    if (g >= N) return;
    if (f != 0) {
        const size_t n = (size_t)f * N, t = (size_t)f * tiles_x * (row1 - row0);
        const size_t dims[4] = {2, RS ? 2 : 3, 1, 3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t o = n * dims[k];
            T.p[k] += o;
            T.m[k] += o;
            T.n[k] += o;
            T.d[k] += o;
            T.prev[k] += o;
        }
        xys += 2 * n;
        radii += n;
        tile_len += t;
        seg += t * C;
        partials += t * C * PART_STRIDE;
        gslots += n * GSLOT_STRIDE;
        if (GRADS) {
            gin.v_xy += 2 * n;
            gin.v_conic += 3 * n;
            gin.v_colors += 3 * n;
        }
        if (CHAIN) {
            ch.out.xys += 2 * n;
            ch.out.depths += n;
            ch.out.radii += n;
            ch.out.conics += 3 * n;
            ch.out.num_tiles_hit += n;
            ch.lists.tile_count += t * CNT_STRIDE;
            ch.lists.seg += t * ch.lists.C;
            if (ch.cull.opacity != nullptr) ch.cull.opacity += n;
            if (ch.rec.rec != nullptr) {
                ch.rec.rec += n * REC_FLOATS;
                ch.rec.opacity += n;
            }
        }
    }
    // Parameters, optimizer state and the Gaussian's own partial-sum slots depend on nothing but
    // g: one round trip for all of them.
    const size_t g2 = 2 * (size_t)g, g3 = 3 * (size_t)g;
    const AdanSlot x0 = adan_load(T, 0, g2), x1 = adan_load(T, 0, g2 + 1);
    const AdanSlot c0 = adan_load(T, 1, RS ? g2 : g3), c1 = adan_load(T, 1, (RS ? g2 : g3) + 1);
    const AdanSlot c2 = adan_load(T, RS ? 2 : 1, RS ? (size_t)g : g3 + 2);
    const AdanSlot f0 = adan_load(T, 3, g3), f1 = adan_load(T, 3, g3 + 1),
                   f2 = adan_load(T, 3, g3 + 2);
    float acc[PART_STRIDE];
    if (GRADS) {
        acc[0] = gin.v_xy[g2];
        acc[1] = gin.v_xy[g2 + 1];
        acc[2] = gin.v_conic[g3];
        acc[3] = gin.v_conic[g3 + 1];
        acc[4] = gin.v_conic[g3 + 2];
        acc[5] = gin.v_colors[g3];
        acc[6] = gin.v_colors[g3 + 1];
        acc[7] = gin.v_colors[g3 + 2];
        acc[8] = 0.f;
    } else {
        gather_partials(g, radii[g], xys[2 * g], xys[2 * g + 1], tiles_x, tiles_y, row0, row1,
                        tile_len, seg, C, partials, gslots, gslot_n, stamp, acc);
    }
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
        gi2d_sincos_f32(theta, &sn, &cs);
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
    const float nx0 = adan_apply(h, T, 0, g2, x0, vm0);
    const float nx1 = adan_apply(h, T, 0, g2 + 1, x1, vm1);
    const float nc0 = adan_apply(h, T, 1, RS ? g2 : g3, c0, o0);
    const float nc1 = adan_apply(h, T, 1, (RS ? g2 : g3) + 1, c1, o1);
    const float nc2 = adan_apply(h, T, RS ? 2 : 1, RS ? (size_t)g : g3 + 2, c2, o2);
    const float nf0 = adan_apply(h, T, 3, g3, f0, acc[5]);
    const float nf1 = adan_apply(h, T, 3, g3 + 1, f1, acc[6]);
    const float nf2 = adan_apply(h, T, 3, g3 + 2, f2, acc[7]);
    if (CHAIN) {
        // the next iteration's projection of this Gaussian: the same arithmetic as
        // project_*_fwd_kernel<RAW = true> on the parameters just written
        const float mx = tanhf(nx0), my = tanhf(nx1);
        float sxx, sxy, syy;
        if (!RS) {
            const float l11 = nc0 + 0.5f, l21 = nc1, l22 = nc2 + 0.5f;
            sxx = l11 * l11;
            sxy = l11 * l21;
            syy = l21 * l21 + l22 * l22;
        } else {
            const float s1 = fabsf(nc0 + 0.5f), s2 = fabsf(nc1 + 0.5f);
            const float theta = (1.0f / (1.0f + expf(-nc2))) * 2.0f * 3.14159274f;
            float s, c;
            gi2d_sincos_f32(theta, &s, &c);
            const float q1 = s1 * s1, q2 = s2 * s2;
            sxx = c * c * q1 + s * s * q2;
            sxy = c * s * (q1 - q2);
            syy = s * s * q1 + c * c * q2;
        }
        project_tail(g, mx, my, sxx, sxy, syy, H, W, tiles_x, tiles_y, ch.center_bias, ch.out,
                     ch.lists, ch.cull, ch.rec, nf0, nf1, nf2);
    }
}

// ------------------------------------------------------------------------------------------
// Forward, loss and backward rasterisation of one tile in one workgroup.  For the L2 loss the
// gradient with respect to a pixel depends on that pixel alone, so a tile that has rendered its
// 256 pixels can go straight on to its backward pass: the loss gradient stays in LDS (it never
// becomes an image in memory), the tile's sorted list is still at hand, and the backward of one
// tile overlaps the forward of another instead of waiting for the whole frame behind a launch
// boundary.  The arithmetic is that of rasterize_sum_fwd_kernel<.., L2> followed by
// rasterize_sum_bwd_kernel: the same bits.
// ------------------------------------------------------------------------------------------

// LDS the two phases use one after the other: the forward's staging records, then the backward's
// work-piece buffers
#define FIT_POOL_BWD_INTS \
    (3 * 4 * BWD_BATCH + (BWD_BATCH + 4) + BWD_MAX_PIECES * PART_STRIDE + BWD_MAX_PIECES / 4)
// ... of the one-wave form (32 entries per batch, 48 pieces)
#define FIT64_POOL_BWD_INTS \
    (3 * 4 * 32 + (32 + 4) + BWD64_MAX_PIECES * PART_STRIDE + BWD64_MAX_PIECES / 4)

// The backward of one tile inside the fused kernel, workgroup of 256 lanes: direct paths for
// short lists, equal work pieces beyond.
__device__ __forceinline__ void fit_bwd_256(
    int *pool, const float *s_v, int *s_w, const int *lst, int gid_first, int n_raw,
    size_t slot0, int tx, int ty, const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0,
    const gi2d_conv &cv, int cull, float *__restrict__ partials, float *__restrict__ gslots,
    size_t gslot_n, int stamp)
{
    const int tid = threadIdx.x;
    if (n_raw <= BWD_PIECES_MIN_LIST) {
        int *s_first = pool;  // (the forward's records are dead by now)
        if (tid < 64) s_first[tid] = gid_first;
        __syncthreads();
#define GI2D_BWD_TILE(LPE)                                                                       \
    bwd_tile<LPE>(lst, s_first, n_raw, slot0, tx, ty, s_v, xys, radii, conics, colors, opacity,  \
                  H, W, tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n, stamp)
        if (n_raw <= 16) GI2D_BWD_TILE(16);
        else GI2D_BWD_TILE(8);
#undef GI2D_BWD_TILE
    } else {
        float4 *s_r0 = reinterpret_cast<float4 *>(pool);
        float4 *s_r1 = s_r0 + BWD_BATCH, *s_r2 = s_r1 + BWD_BATCH;
        int *s_start = reinterpret_cast<int *>(s_r2 + BWD_BATCH);
        float *s_piece = reinterpret_cast<float *>(s_start + BWD_BATCH + 4);
        unsigned char *s_owner =
            reinterpret_cast<unsigned char *>(s_piece + BWD_MAX_PIECES * PART_STRIDE);
        bwd_tile_pieces(lst, gid_first, n_raw, slot0, tx, ty, s_v, s_r0, s_r1, s_r2, s_start, s_w,
                        s_piece, s_owner, xys, radii, conics, colors, opacity, H, W, tiles_x, tiles_y, row0,
                        cv, cull, partials, gslots, gslot_n, stamp);
    }
}

// ... and of a single wave (sparse frames: many tiles with a handful of Gaussians each): the
// fewer entries, the more lanes each gets.
__device__ __forceinline__ void fit_bwd_64(
    int *pool, const float *s_v, const int *lst, int gid_first, int n_raw, size_t slot0, int tx,
    int ty, const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0,
    const gi2d_conv &cv, int cull, float *__restrict__ partials, float *__restrict__ gslots,
    size_t gslot_n, int stamp)
{
    int *s_first = pool;
    s_first[threadIdx.x] = gid_first;
    __syncthreads();
#define GI2D_BWD_TILE(LPE)                                                                       \
    bwd_tile<LPE, 64>(lst, s_first, n_raw, slot0, tx, ty, s_v, xys, radii, conics, colors,       \
                      opacity, H, W, tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n, \
                      stamp)
    if (n_raw <= 4) {
        GI2D_BWD_TILE(16);
    } else if (n_raw <= 8) {
        GI2D_BWD_TILE(8);
    } else if (n_raw <= BWD64_PIECES_MIN_LIST) {
        GI2D_BWD_TILE(4);
    } else {
        // longer lists: equal work pieces, as in the 256-lane form (with four lanes per entry
        // every batch of 16 entries lasts as long as its largest entry)
        __syncthreads();  // (s_first is about to be overwritten)
        float4 *s_r0 = reinterpret_cast<float4 *>(pool);
        float4 *s_r1 = s_r0 + 32, *s_r2 = s_r1 + 32;
        int *s_start = reinterpret_cast<int *>(s_r2 + 32);
        float *s_piece = reinterpret_cast<float *>(s_start + 32 + 4);
        unsigned char *s_owner =
            reinterpret_cast<unsigned char *>(s_piece + BWD64_MAX_PIECES * PART_STRIDE);
        bwd_tile_pieces<64>(lst, gid_first, n_raw, slot0, tx, ty, s_v, s_r0, s_r1, s_r2, s_start,
                            nullptr, s_piece, s_owner, xys, radii, conics, colors, opacity, H, W,
                            tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n, stamp);
    }
#undef GI2D_BWD_TILE
}

// PPL = 1: 256 lanes per tile; PPL = 4: one wave per tile (the forward renders four pixels per
// lane, the backward gives each entry 4 to 16 lanes and has no work-piece path).
template <bool FINALIZE, int PPL, bool PACKED>
// (7 waves per SIMD for the 256-lane form, as in rasterize_sum_bwd_kernel; the one-wave form is
// limited by its LDS)
__global__ __launch_bounds__(256 / PPL)
__attribute__((amdgpu_waves_per_eu(PPL == 1 ? 7 : 6, 8))) void
fit_raster_kernel(int *tile_count, int *tile_len, int *seg, int C, const float *__restrict__ xys,
                  const int *__restrict__ radii, const float *__restrict__ conics,
                  const float *__restrict__ colors, const float *__restrict__ opacity,
                  const float *__restrict__ rec, const float *__restrict__ background, int H,
                  int W, int tiles_x, int tiles_y, int row0, int T, gi2d_conv cv, int cull,
                  DevStatus *status,
                  long long generation, float *__restrict__ out, const float *__restrict__ gt,
                  float *__restrict__ loss_part, float *__restrict__ partials,
                  float *__restrict__ gslots, size_t gslot_n, int stamp, int N)
{
    static_assert(PPL == 1 || PPL == 4, "unsupported variant");
    if (blockIdx.y != 0) {  // image f of a batch (see fit_tail_kernel)
        const size_t n = (size_t)blockIdx.y * N, t = (size_t)blockIdx.y * T;
        tile_count += t * CNT_STRIDE;
        tile_len += t;
        seg += t * C;
        xys += 2 * n;
        radii += n;
        conics += 3 * n;
        colors += 3 * n;
        opacity += n;
        rec += n * REC_FLOATS;
        if (out != nullptr) out += (size_t)blockIdx.y * H * W * 3;
        gt += (size_t)blockIdx.y * H * W * 3;
        loss_part += t;
        partials += t * C * PART_STRIDE;
        gslots += n * GSLOT_STRIDE;
    }
    constexpr int NT = 256 / PPL;
    constexpr int FWD_INTS = FWD_LDS_INTS_PER_ENTRY * NT;
    // LDS the two phases use one after the other (forward staging records, then the backward's
    // buffers), with the tile's sorted list (lists of up to NT ids) behind the forward's
    // records, in a part the backward writes only after every lane has fetched its id
    constexpr int POOL_INTS =
        PPL == 1 ? (FIT_POOL_BWD_INTS > FWD_INTS + NT ? FIT_POOL_BWD_INTS : FWD_INTS + NT)
                 : (FIT64_POOL_BWD_INTS > FWD_INTS + NT ? FIT64_POOL_BWD_INTS : FWD_INTS + NT);
    __shared__ __attribute__((aligned(16))) int pool[POOL_INTS];
    __shared__ float s_v[VTILE_FLOATS];   // the tile's loss gradient
    __shared__ int s_w[4];
    int *s_sorted = pool + FWD_INTS;
    const int n_raw = fwd_tile<FINALIZE, true, 1, PPL, true, PACKED>(
        pool, nullptr, s_v, s_sorted, tile_count, tile_len, seg, C, xys, conics, colors, opacity,
        rec, background, H, W, tiles_x, row0, T, cv, cull, status, generation, out, gt, nullptr,
        loss_part, ExtraSets());
    // (every wave is past its last read of the forward's records; the list, the loss gradient
    // and the sorted ids are visible to the whole workgroup)
    __syncthreads();
    if (n_raw == 0 || n_raw > C) return;  // an overflowed tile is reported, not half-summed
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, T);
    const int tx = t % tiles_x, ty = row0 + t / tiles_x;
    const int *lst = seg + (size_t)t * C;
    const size_t slot0 = (size_t)t * C;
    int gid_first;
    if (FINALIZE && n_raw <= NT)
        gid_first = s_sorted[tid];
    else
        gid_first = tid < min(C, min(NT, BWD_BATCH)) ? lst[tid] : 0;
    if (PPL == 1)
        fit_bwd_256(pool, s_v, s_w, lst, gid_first, n_raw, slot0, tx, ty, xys, radii, conics,
                    colors, opacity, H, W, tiles_x, tiles_y, row0, cv, cull, partials, gslots,
                    gslot_n, stamp);
    else
        fit_bwd_64(pool, s_v, lst, gid_first, n_raw, slot0, tx, ty, xys, radii, conics, colors,
                   opacity, H, W, tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n,
                   stamp);
}

// Pass 2 on its own (gi2d_fit_raster): every Gaussian's partial sums of THIS workspace's tiles.
__global__ __launch_bounds__(256) void gather_grads_kernel_fit(
    const float *__restrict__ xys, const int *__restrict__ radii, int N, int tiles_x, int tiles_y,
    int row0, int row1, const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ partials, const float *__restrict__ gslots, size_t gslot_n,
    int stamp, float *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_colors, float *__restrict__ v_opacity)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float acc[PART_STRIDE];
    gather_partials(g, radii[g], xys[2 * g], xys[2 * g + 1], tiles_x, tiles_y, row0, row1,
                    tile_len, seg, C, partials, gslots, gslot_n, stamp, acc);
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


// The projection of the raw parameters (skipped when a chained step has already done it into
// these outputs) and the binning bookkeeping of a fitting iteration.
static int fit_project_bin(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                           const float *cov_b, const float *features, const float *opacity, int N,
                           int H, int W,
                           const gi2d_conv *conv, float *xys, float *depths, int *radii,
                           float *conics, int *num_tiles_hit, void *stream, int frames = 1)
{
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    const bool chained = ws->chain_pending && ws->prefilled && ws->chain_kind == kind &&
                         ws->chain_xys == (const void *)xys && ws->pre_n == N &&
                         ws->pre_tx == tiles_x && ws->pre_ty == tiles_y &&
                         ws->pre_frames == frames;
    if (!chained) {
        ProjExtras ex;
        ex.opacity = opacity;
        ex.colors = features;
        ex.frames = frames;
        int rc = project_impl(kind, true, xyz, cov_a, cov_b, N, H, W, tiles_x, tiles_y, conv, xys,
                              depths, radii, conics, num_tiles_hit, ws, stream, ex);
        if (rc != GI2D_OK) return rc;
    }
    return bin_tiles_impl(ws, xys, radii, N, tiles_x, tiles_y, GI2D_BIN_COUNTED, stream, frames);
}

extern "C" {

int gi2d_fit_forward(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                     const float *cov_b, const float *features, const float *opacity,
                     const float *background, const float *gt_chw, int N, int H, int W,
                     const gi2d_conv *conv, float *xys,
                     float *depths, int *radii, float *conics, int *num_tiles_hit, float *image,
                     float *v_image, float *loss_partials, void *stream)
{
    if (!ws || (kind != 0 && kind != 1) || !gt_chw) return GI2D_E_INVALID;
    int rc = fit_project_bin(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys, depths,
                             radii, conics, num_tiles_hit, stream);
    if (rc != GI2D_OK && rc != GI2D_E_PREV_OVERFLOW) return rc;
    const int late = rc;
    rc = raster_fwd_impl(ws, xys, conics, features, opacity, background, N, H, W, conv, image,
                         gt_chw, v_image, loss_partials, stream);
    return rc != GI2D_OK ? rc : late;
}

// chain != NULL: the projection outputs to overwrite for the next iteration (see
// gi2d_fit_backward_step_chain); the const inputs xys / radii / conics then alias them.
struct ChainOutputs {
    float *xys, *depths, *conics;
    int *radii, *num_tiles_hit;
};

static int fit_backward_impl(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                             float *features, const float *opacity, int N, int H, int W,
                             const gi2d_conv *conv, const float *xys, const int *radii,
                             const float *conics, const float *v_image, float *const *exp_avg,
                             float *const *exp_avg_sq, float *const *exp_avg_diff,
                             float *const *neg_pre_grad, const gi2d_adan_hyper *hyper,
                             void *stream, const ChainOutputs *chain, bool pass1_done = false,
                             const GradIn *grads = nullptr, int frames = 1)
{
    if (kind != 0 && kind != 1) return GI2D_E_INVALID;
    // (several images only through the fused rasterisation kernel, which has done pass 1)
    if (frames < 1 || (frames > 1 && (!pass1_done || grads))) return GI2D_E_INVALID;
    int rc = check_raster_args(ws, N, H, W, conv, frames);
    if (rc != GI2D_OK) return rc;
    AdanScalars h;
    rc = adan_scalars(hyper, h);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (!xyz || !cov_a || (kind == 1 && !cov_b) || !features || !opacity || !xys || !radii ||
        !conics || (!v_image && T > 0 && !pass1_done) || !exp_avg || !exp_avg_sq || !exp_avg_diff ||
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
    if (!pass1_done) {
        rc = raster_bwd_pass1(ws, xys, radii, conics, features, opacity, N, H, W, conv, v_image,
                              st);
        if (rc != GI2D_OK) return rc;
    }
    // (a batch of four images and more projects with its own launch instead: the chained lanes
    // fill the lists with one counter atomic per (Gaussian, tile) pair, the batch projection
    // aggregates them in LDS, and the part's rate of returning atomics is what a chained tail
    // of 24 images would wait for)
    // (nor does a frame whose projection fills the lists through cells of tiles: the same
    // atomics, 25 us of them at 3840x2160 / 200 000)
    bool chained = chain != nullptr && frames < 4 &&
                   !(frames == 1 &&
                     cells_form(ws, N, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1)) &&
                   begin_chain_prefill(ws, N, ws->tiles_x, ws->tiles_y, frames);
    if (chained && !ws->no_records && ensure_records(ws, (int64_t)frames * N) != GI2D_OK) {
        (void)hipGetLastError();
        chained = false;
    }
    ChainArgs ch = ChainArgs();
    if (chained) {
        ch.center_bias = conv->center_bias;
        ch.out = ProjOut{chain->xys, chain->depths, chain->radii, chain->conics,
                         chain->num_tiles_hit};
        ch.lists = TileLists{ws->alt_count, ws->alt_seg, ws->seg_cap, ws->bin_row0, ws->bin_row1};
        ch.cull.opacity = ws->cull ? opacity : nullptr;
        ch.cull.alpha_min = conv->alpha_min;
        ch.cull.pixel_offset = conv->pixel_offset;
        if (!ws->no_records) ch.rec = RecordSink{ws->d_rec, opacity, conv->alpha_cap};
        ch.status = ws->d_status;
        ch.h_status = ws->h_status_dev;
        ch.generation = ws->generation;
    }
    const GradIn gin = grads ? *grads : GradIn{nullptr, nullptr, nullptr};
    // (the kernel reads xys / radii of the current frame and, chained, overwrites them)
    float *xys_io = chained ? chain->xys : const_cast<float *>(xys);
    int *radii_io = chained ? chain->radii : const_cast<int *>(radii);
#define GI2D_TAIL_ARGS Tn, h, xys_io, radii_io, N, H, W, ws->tiles_x, ws->tiles_y, ws->bin_row0, \
                       ws->bin_row1, ws->d_len, ws->d_seg, ws->seg_cap, ws->d_partials,          \
                       ws->d_gslots, (size_t)ws->slot_cap, ws->bwd_stamp, ch, gin
    // one wave per workgroup when the lanes also issue the list-filling atomics, so that the
    // launch spreads over every CU (as in the projection kernels)
    const int threads = chained ? PG_THREADS : 256;
    const dim3 grid(cdiv(N, threads), frames), block(threads);
    {
    Probe probe(ws, GI2D_PROBE_FIT_TAIL, st);
#define GI2D_TAIL(RS, GR)                                                                   \
    do {                                                                                    \
        if (chained) fit_tail_kernel<RS, true, GR><<<grid, block, 0, st>>>(GI2D_TAIL_ARGS);  \
        else fit_tail_kernel<RS, false, GR><<<grid, block, 0, st>>>(GI2D_TAIL_ARGS);         \
    } while (0)
    if (kind == 0 && grads) GI2D_TAIL(false, true);
    else if (kind == 0) GI2D_TAIL(false, false);
    else if (grads) GI2D_TAIL(true, true);
    else GI2D_TAIL(true, false);
#undef GI2D_TAIL
    }
#undef GI2D_TAIL_ARGS
    rc = launch_status();
    if (rc == GI2D_OK && chained) {
        commit_chain_prefill(ws, N, ws->tiles_x, ws->tiles_y, kind, chain->xys, frames);
        // the records the tail wrote are the next frame's, for these tensors
        ws->rec_valid = ch.rec.rec != nullptr;
        ws->rec_xys = chain->xys;
        ws->rec_colors = features;
        ws->rec_opacity = opacity;
    }
    return rc;
}

int gi2d_fit_backward_step(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                           float *features, const float *opacity, int N, int H, int W,
                           const gi2d_conv *conv, const float *xys, const int *radii,
                           const float *conics, const float *v_image, float *const *exp_avg,
                           float *const *exp_avg_sq, float *const *exp_avg_diff,
                           float *const *neg_pre_grad, const gi2d_adan_hyper *hyper, void *stream)
{
    return fit_backward_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys,
                             radii, conics, v_image, exp_avg, exp_avg_sq, exp_avg_diff,
                             neg_pre_grad, hyper, stream, nullptr);
}

int gi2d_fit_backward_step_chain(gi2d_workspace *ws, int kind, float *xyz, float *cov_a,
                                 float *cov_b, float *features, const float *opacity, int N, int H,
                                 int W, const gi2d_conv *conv, float *xys, float *depths,
                                 int *radii, float *conics, int *num_tiles_hit,
                                 const float *v_image, float *const *exp_avg,
                                 float *const *exp_avg_sq, float *const *exp_avg_diff,
                                 float *const *neg_pre_grad, const gi2d_adan_hyper *hyper,
                                 void *stream)
{
    if (!xys || !depths || !radii || !conics || !num_tiles_hit) return GI2D_E_INVALID;
    const ChainOutputs chain = {xys, depths, conics, radii, num_tiles_hit};
    return fit_backward_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys,
                             radii, conics, v_image, exp_avg, exp_avg_sq, exp_avg_diff,
                             neg_pre_grad, hyper, stream, &chain);
}

// Projection (unless a chained tail prepared it), binning bookkeeping and the fused rasterisation
// kernel (forward, loss, backward pass 1) of one fitting iteration.
static int fit_raster_impl(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                           const float *cov_b, const float *features, const float *opacity,
                           const float *background, const float *gt_chw, int N, int H, int W,
                           const gi2d_conv *conv, float *xys, float *depths, int *radii,
                           float *conics, int *num_tiles_hit, float *image, float *loss_partials,
                           void *stream, int *late_out, bool *ran, int frames = 1)
{
    *ran = false;
    if (frames < 1) return GI2D_E_INVALID;
    if (!ws || (kind != 0 && kind != 1) || !gt_chw || !loss_partials) return GI2D_E_INVALID;
    if (N > 0 && (!xys || !depths || !radii || !conics || !num_tiles_hit)) return GI2D_E_INVALID;
    if (N > 0 && (!xyz || !cov_a || (kind == 1 && !cov_b) || !features || !opacity))
        return GI2D_E_INVALID;
    if (conv && conv->bg_weight != 0.0f && !background) return GI2D_E_INVALID;
    int rc = fit_project_bin(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys,
                             depths, radii, conics, num_tiles_hit, stream, frames);
    if (rc != GI2D_OK && rc != GI2D_E_PREV_OVERFLOW) return rc;
    *late_out = rc;
    rc = check_raster_args(ws, N, H, W, conv, frames);
    if (rc != GI2D_OK) return rc;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (T == 0) {  // empty band
        ws->lists_raw = false;
        return GI2D_OK;
    }
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)frames * N >= (int64_t)INT_MAX / 16) return GI2D_E_INVALID;
    rc = ensure_bwd_buffers(ws, frames * N);
    if (rc != GI2D_OK) return rc;
    rc = next_bwd_stamp(ws, st);
    if (rc != GI2D_OK) return rc;
    const bool packed = ws->rec_valid && ws->rec_xys == (const void *)xys &&
                        ws->rec_colors == (const void *)features &&
                        ws->rec_opacity == (const void *)opacity;
#define GI2D_FIT_RASTER_ARGS ws->d_count, ws->d_len, ws->d_seg, ws->seg_cap, xys, radii, conics, \
                             features, opacity, ws->d_rec, background, H, W, ws->tiles_x,        \
                             ws->tiles_y,                                                        \
                             ws->bin_row0, T, *conv, ws->cull, ws->d_status, ws->generation,     \
                             image, gt_chw, loss_partials, ws->d_partials, ws->d_gslots,         \
                             (size_t)ws->slot_cap, ws->bwd_stamp, N
    // (a launch of many tiles: one wave per tile, as in the forward rasteriser -- unless the
    // lists are long, which the one-wave backward walks without its work-piece path)
    const bool one_wave =
        frames > 1 && ws->raster_ppl == 0
            ? (long long)T * frames >= GI2D_PPL4_MIN_TILES && ws->h_status[0] <= 128
            : sparse_frame(ws, T);
    const dim3 grid(T, frames);
#define GI2D_FIT_RASTER(FIN, PK)                                                            \
    do {                                                                                    \
        if (one_wave) fit_raster_kernel<FIN, 4, PK><<<grid, 64, 0, st>>>(GI2D_FIT_RASTER_ARGS); \
        else fit_raster_kernel<FIN, 1, PK><<<grid, 256, 0, st>>>(GI2D_FIT_RASTER_ARGS);         \
    } while (0)
    {
    Probe probe(ws, GI2D_PROBE_FIT_RASTER, st);
    if (ws->lists_raw) {
        if (packed) GI2D_FIT_RASTER(true, true);
        else GI2D_FIT_RASTER(true, false);
        ws->lists_raw = false;
        ws->counters_dirty = false;
    } else {
        if (packed) GI2D_FIT_RASTER(false, true);
        else GI2D_FIT_RASTER(false, false);
    }
    }
#undef GI2D_FIT_RASTER
#undef GI2D_FIT_RASTER_ARGS
    *ran = true;
    return launch_status();
}

int gi2d_fit_step(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                  float *features, const float *opacity, const float *background,
                  const float *gt_chw, int N, int H, int W, const gi2d_conv *conv, float *xys,
                  float *depths, int *radii, float *conics, int *num_tiles_hit, float *image,
                  float *loss_partials, float *const *exp_avg, float *const *exp_avg_sq,
                  float *const *exp_avg_diff, float *const *neg_pre_grad,
                  const gi2d_adan_hyper *hyper, void *stream)
{
    if (!hyper || hyper->step < 1) return GI2D_E_INVALID;
    int late = GI2D_OK;
    bool ran = false;
    int rc = fit_raster_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, background, gt_chw, N,
                             H, W, conv, xys, depths, radii, conics, num_tiles_hit, image,
                             loss_partials, stream, &late, &ran);
    if (rc != GI2D_OK) return rc;
    if (!ran) return late;  // empty band
    const ChainOutputs chain = {xys, depths, conics, radii, num_tiles_hit};
    rc = fit_backward_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys,
                           radii, conics, nullptr, exp_avg, exp_avg_sq, exp_avg_diff, neg_pre_grad,
                           hyper, stream, &chain, true);
    return rc != GI2D_OK ? rc : late;
}

int gi2d_fit_step_batch(gi2d_workspace *ws, int kind, int frames, float *xyz, float *cov_a,
                        float *cov_b, float *features, const float *opacity,
                        const float *background, const float *gt_chw, int N, int H, int W,
                        const gi2d_conv *conv, float *xys, float *depths, int *radii,
                        float *conics, int *num_tiles_hit, float *images, float *loss_partials,
                        float *const *exp_avg, float *const *exp_avg_sq,
                        float *const *exp_avg_diff, float *const *neg_pre_grad,
                        const gi2d_adan_hyper *hyper, void *stream)
{
    if (!hyper || hyper->step < 1 || frames < 1) return GI2D_E_INVALID;
    int late = GI2D_OK;
    bool ran = false;
    int rc = fit_raster_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, background, gt_chw, N,
                             H, W, conv, xys, depths, radii, conics, num_tiles_hit, images,
                             loss_partials, stream, &late, &ran, frames);
    if (rc != GI2D_OK) return rc;
    if (!ran) return late;
    const ChainOutputs chain = {xys, depths, conics, radii, num_tiles_hit};
    rc = fit_backward_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys,
                           radii, conics, nullptr, exp_avg, exp_avg_sq, exp_avg_diff, neg_pre_grad,
                           hyper, stream, &chain, true, nullptr, frames);
    return rc != GI2D_OK ? rc : late;
}

int gi2d_fit_raster(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                    const float *cov_b, const float *features, const float *opacity,
                    const float *background, const float *gt_chw, int N, int H, int W,
                    const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                    int *num_tiles_hit, float *image, float *loss_partials, float *v_xys,
                    float *v_conics, float *v_colors, float *v_opacity, void *stream)
{
    if (N > 0 && (!v_xys || !v_conics || !v_colors || !v_opacity)) return GI2D_E_INVALID;
    int late = GI2D_OK;
    bool ran = false;
    int rc = fit_raster_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, background, gt_chw, N,
                             H, W, conv, xys, depths, radii, conics, num_tiles_hit, image,
                             loss_partials, stream, &late, &ran);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return late;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    if (!ran) {  // an empty band contributes zeros
        rc = ensure_bwd_buffers(ws, N);
        if (rc == GI2D_OK) rc = next_bwd_stamp(ws, st);
        if (rc != GI2D_OK) return rc;
    }
    {
        Probe probe(ws, GI2D_PROBE_GATHER, st);
        gather_grads_kernel_fit<<<cdiv(N, 256), 256, 0, st>>>(
            xys, radii, N, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1, ws->d_len,
            ws->d_seg, ws->seg_cap, ws->d_partials, ws->d_gslots, (size_t)ws->slot_cap,
            ws->bwd_stamp, v_xys, v_conics, v_colors, v_opacity);
    }
    rc = launch_status();
    return rc != GI2D_OK ? rc : late;
}

int gi2d_fit_tail(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                  float *features, const float *opacity, int N, int H, int W,
                  const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                  int *num_tiles_hit, const float *v_xys, const float *v_conics,
                  const float *v_colors, float *const *exp_avg, float *const *exp_avg_sq,
                  float *const *exp_avg_diff, float *const *neg_pre_grad,
                  const gi2d_adan_hyper *hyper, void *stream)
{
    if (!xys || !depths || !radii || !conics || !num_tiles_hit) return GI2D_E_INVALID;
    if (N > 0 && (!v_xys || !v_conics || !v_colors)) return GI2D_E_INVALID;
    const ChainOutputs chain = {xys, depths, conics, radii, num_tiles_hit};
    const GradIn gin = {v_xys, v_conics, v_colors};
    return fit_backward_impl(ws, kind, xyz, cov_a, cov_b, features, opacity, N, H, W, conv, xys,
                             radii, conics, nullptr, exp_avg, exp_avg_sq, exp_avg_diff,
                             neg_pre_grad, hyper, stream, &chain, true, &gin);
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
