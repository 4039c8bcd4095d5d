// raster_bwd_tile.h -- the backward rasteriser's work for one tile (shared by the backward kernel
// and the fused fitting kernel).
#pragma once
#include "gather.h"
#include "host.h"
#include "projection.h"
#include "support.h"
#include "tiles.h"

// The tile's loss gradient in LDS: three planes (red, green, blue) of 256 floats,
// v[ch * VTILE_PLANE + y * 16 + x] -- 3 KB instead of 4 as float4s, which is what the one-wave
// forms' occupancy is short of.
#define VTILE_PLANE 256
#define VTILE_FLOATS (3 * VTILE_PLANE)

#define BWD_BATCH 128                  // entries per batch of the work-piece path
// Lists longer than this take the work-piece path.  Measured on MI355X: a 50k-Gaussian fit of a
// ~1020x768 image spends 37 us per backward in its fitted regime with the threshold at 64, 30 us
// at 32 or 16; frames at initialisation (small supports) are unaffected either way.
#ifndef BWD_PIECES_MIN_LIST
#define BWD_PIECES_MIN_LIST 32
#endif
// ... of the one-wave forms (32 entries per batch, 16 quads, 64 pieces): with four lanes per
// entry a batch of 16 entries lasts as long as its largest entry
#ifndef BWD64_PIECES_MIN_LIST
#define BWD64_PIECES_MIN_LIST 16
#endif
#define BWD_QUADS 64                   // quads of lanes per workgroup
#define BWD_MAX_ROUNDS 4               // rounds of BWD_QUADS pieces a batch is cut into, at most
#define BWD_MAX_PIECES (BWD_QUADS * BWD_MAX_ROUNDS)
#define BWD64_MAX_ROUNDS 3             // ... of the one-wave form (16 quads)
#define BWD64_MAX_PIECES (16 * BWD64_MAX_ROUNDS)

// inclusive prefix sum over the NT (256 or 64) lanes of the workgroup (s_w: 4 ints of LDS, unused
// by a single wave); total in `all`
template <int NT = 256>
__device__ __forceinline__ int block_inclusive_scan(int v, int *s_w, int &all)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_inclusive_scan(v);
    if (NT == 64) {
        all = __builtin_amdgcn_readlane(v, 63);
        return v;
    }
    __syncthreads();  // (s_w may still be read from an earlier scan)
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    const int w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    all = w0 + w1 + w2 + w3;
    return v + (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
}

// sum over the LPE (4, 8 or 16) neighbouring lanes of an entry's group, result in every lane
template <int LPE>
__device__ __forceinline__ float group_sum(float v)
{
    v = quad_sum(v);
    if (LPE >= 8) v = v + dpp_mov<0x141>(v);   // row_half_mirror: lane i <-> 7-i of each 8
    if (LPE >= 16) v = v + dpp_mov<0x140>(v);  // row_mirror: lane i <-> 15-i of each 16
    return v;
}

// ---- the pixel loop of an entry ---------------------------------------------------------------
// Vector instructions do not all issue at the same rate on this part (measured,
// tools/micro/op_rates.hip): v_fma / v_mul / v_add / v_sub_f32, v_mov, v_add_u32 and v_and_b32
// take 2 cycles per wave64 instruction; compares, v_cndmask, v_min / v_max, conversions, shifts
// and anything with an SGPR operand take 4; v_exp_f32 takes 8.  The loop below therefore walks the
// rectangle in FLOATING POINT (the pixel coordinates are small integers, exact in fp32) and
// replaces every select by a multiplication with a 0/1 factor made by one v_fma with the clamp
// modifier, so that a step of a plain entry is 2-cycle instructions but for the exponential and
// the loop's own compare.

// Pixels k0, k0 + STRIDE, ... < k_end of the rectangle of width w whose first row starts at x0
// and whose last column is x1 (tile coordinates), pixel k = (x0 + k % w, y0 + k / w); (x, y) is
// pixel k0.  Adds the lane's share of the nine partial sums: [0..1] sum vs*d, [2..4] the second
// moments of vs, [5..7] sum alpha*v_out, [8] sum exp * v_alpha.  PLAIN: the entry's exponent is
// never positive and its alpha never reaches the cap (projection.h), so alpha = opacity * exp.
template <int STRIDE, bool PLAIN>
__device__ __forceinline__ void bwd_pixels(float (&acc)[PART_STRIDE], int k0, int k_end, int x,
                                           int y, int x1, int w, float gx, float gy, float ox,
                                           float oy, float a, float b, float c, float op, float cr,
                                           float cg, float cb, const float *s_v,
                                           const gi2d_conv &cv, const StepConsts &kc)
{
    // STRIDE / w and STRIDE % w for w in 1..16 from the hardware reciprocal (exact here: the
    // quotient's fractional part is a multiple of 1/w, well clear of the 1-ulp error)
    const float wf = (float)w;
    const float sy = floorf((float)STRIDE * __builtin_amdgcn_rcpf(wf) + 1.0e-3f);
    const float sx = (float)STRIDE - sy * wf;
    const float x1f = (float)x1, nwf = -wf;
    float fx = (float)x, fy = (float)y;
    const float pa = (-0.5f * LOG2E) * a, pb = (-LOG2E) * b, pc = (-0.5f * LOG2E) * c;
    const char *v_base = reinterpret_cast<const char *>(s_v);
    for (int k = k0; k < k_end; k += STRIDE) {
        const float dx = gx - (ox + fx), dy = gy - (oy + fy);
        const float p = __builtin_fmaf(pc * dy, dy, __builtin_fmaf(pb, dy, pa * dx) * dx);
        const float ex = __builtin_amdgcn_exp2f(p);
        const float oe = op * ex;
        // byte offset of pixel (x, y) in a plane (4 bytes each, 16 per row): 2^23 + 64*y + 4*x
        // is exact, and the low bits of its mantissa are the offset
        const float af = __builtin_fmaf(fy, 64.0f, __builtin_fmaf(fx, 4.0f, 8388608.0f));
        const float *vp = reinterpret_cast<const float *>(v_base + (__float_as_int(af) & 0xffff));
        const float3 v = make_float3(vp[0], vp[VTILE_PLANE], vp[2 * VTILE_PLANE]);
        const float v_alpha = __builtin_fmaf(cb, v.z, __builtin_fmaf(cg, v.y, cr * v.x));
        float am, va;
        if (PLAIN) {
            const float on = fma_clamp01(oe, kc.step_scale, kc.step_bias);
            am = oe * on;
            va = v_alpha * on;
        } else {
            const float alpha = fminf(cv.alpha_cap, oe);
            const bool on = !(p > 0.f || alpha < cv.alpha_min);
            am = on ? alpha : 0.f;
            const bool live = on && !(oe > cv.alpha_cap);  // a capped alpha is a constant
            va = live ? v_alpha : 0.f;
        }
        acc[5] = __builtin_fmaf(am, v.x, acc[5]);
        acc[6] = __builtin_fmaf(am, v.y, acc[6]);
        acc[7] = __builtin_fmaf(am, v.z, acc[7]);
        const float vs = -oe * va;
        acc[8] = __builtin_fmaf(ex, va, acc[8]);
        // moments of vs over the rectangle; the conic enters once, after the loop
        const float hx = vs * dx, hy = vs * dy;
        acc[0] += hx;
        acc[1] += hy;
        acc[2] = __builtin_fmaf(hx, dx, acc[2]);
        acc[3] = __builtin_fmaf(hx, dy, acc[3]);
        acc[4] = __builtin_fmaf(hy, dy, acc[4]);
        // next pixel: STRIDE further along the rows (at most one wrap: sx < w)
        fx += sx;
        fy += sy;
        const float wrap = sub_clamp01(fx, x1f);  // 1 past the last column (by >= 1), else 0
        fx = __builtin_fmaf(nwf, wrap, fx);
        fy += wrap;
    }
}

// The batches of one tile with LPE lanes per list entry, in a workgroup of NT lanes (NT / LPE
// entries per batch).
template <int LPE, int NT = 256>
__device__ __forceinline__ void bwd_tile(
    const int *lst, const int *s_first, int n_raw, size_t slot0, int tx, int ty,
    const float *s_v, const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0,
    const gi2d_conv &cv, int cull, float *__restrict__ partials, float *__restrict__ gslots,
    size_t gslot_n, int stamp)
{
    constexpr int EPB = NT / LPE;  // entries per batch
    const int tid = threadIdx.x;
    const int x_hi = min(GI2D_BLOCK, W - tx * GI2D_BLOCK) - 1;
    const int y_hi = min(GI2D_BLOCK, H - ty * GI2D_BLOCK) - 1;
    const float ox = (float)(tx * GI2D_BLOCK) + cv.pixel_offset;
    const float oy = (float)(ty * GI2D_BLOCK) + cv.pixel_offset;
    const int q = tid & (LPE - 1);
    const StepConsts kc = step_consts(cv);
    for (int base = 0; base < n_raw; base += EPB) {
        const int e = base + tid / LPE;
        float acc[PART_STRIDE];
#pragma unroll
        for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
        int gid = 0, rad = 0;
        float gx = 0.f, gy = 0.f;
        bool touched = false;  // the entry's support reaches this tile
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
            touched = cnt > 0;
            // The lanes of the group take the rectangle's pixels round-robin.  q / w for w in
            // 1..16 from the hardware reciprocal (exact here: the quotient's fractional part is
            // a multiple of 1/w, well clear of the 1-ulp error).
            const float rw = __builtin_amdgcn_rcpf((float)max(w, 1));
            const int qy = (int)(((float)q + 0.5f) * rw);
            const int x = x0 + q - qy * w, y = y0 + qy;
            if (plain_entry(a, b, c, op, cv.alpha_cap))
                bwd_pixels<LPE, true>(acc, q, cnt, x, y, x1, w, gx, gy, ox, oy, a, b, c, op, cr, cg,
                                      cb, s_v, cv, kc);
            else
                bwd_pixels<LPE, false>(acc, q, cnt, x, y, x1, w, gx, gy, ox, oy, a, b, c, op, cr,
                                       cg, cb, s_v, cv, kc);
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
            // index of this tile in the Gaussian's own (band-clipped) tile box: the first GSLOTS
            // tiles write to the Gaussian's slots (with the stamp), the others to the tile's slab
            TileBox box = tile_box(gx, gy, (float)rad, tiles_x, tiles_y);
            const int by0 = max(box.y0, row0);
            const int kb = (ty - by0) * (box.x1 - box.x0) + (tx - box.x0);
            if (rad > 0 && kb >= 0 && kb < GSLOTS) {
                // (an entry that cannot reach the tile leaves its slot's old stamp: pass 2
                // then counts it as the zero it is)
                if (!touched) continue;
                float2 *dst = reinterpret_cast<float2 *>(
                    gslots + ((size_t)kb * gslot_n + gid) * GSLOT_STRIDE);
                dst[0] = make_float2(acc[0], acc[1]);
                dst[1] = make_float2(acc[2], acc[3]);
                dst[2] = make_float2(acc[4], acc[5]);
                dst[3] = make_float2(acc[6], acc[7]);
                dst[4] = make_float2(acc[8], __int_as_float(stamp));
            } else {
                float *dst = partials + (slot0 + e) * PART_STRIDE;
#pragma unroll
                for (int k = 0; k < PART_STRIDE; ++k) dst[k] = acc[k];
            }
        }
    }
}

// The work-piece path (long lists): the batches of one tile, in a workgroup of NT lanes (256 or
// 64): NT / 2 entries per batch, NT / 4 quads, at most 256 or 48 pieces (MAX_PIECES).  s_r0 / s_r1 /
// s_r2 hold NT / 2 float4 each, s_start NT / 2 + 1 ints, s_piece MAX_PIECES x PART_STRIDE floats,
// s_owner MAX_PIECES bytes.
template <int NT = 256>
__device__ __forceinline__ void bwd_tile_pieces(
    const int *lst, int gid_first, int n_raw, size_t slot0, int tx, int ty,
    const float *s_v, float4 *s_r0, float4 *s_r1, float4 *s_r2, int *s_start, int *s_w,
    float *s_piece, unsigned char *s_owner, const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0,
    const gi2d_conv &cv, int cull, float *__restrict__ partials, float *__restrict__ gslots,
    size_t gslot_n, int stamp)
{
    // (one wave: at most three rounds -- 48 pieces for 32 entries -- which keeps the form's LDS
    // within what six workgroups per SIMD may take)
    constexpr int MAX_ROUNDS = NT == 64 ? BWD64_MAX_ROUNDS : BWD_MAX_ROUNDS;
    constexpr int EB = NT / 2, QUADS = NT / 4, MAX_PIECES = QUADS * MAX_ROUNDS;
    static_assert(EB == BWD_BATCH || NT == 64, "buffers are sized by BWD_BATCH for 256 lanes");
    const int tid = threadIdx.x, q = tid & 3, quad = tid >> 2;
    const int x_hi = min(GI2D_BLOCK, W - tx * GI2D_BLOCK) - 1;
    const int y_hi = min(GI2D_BLOCK, H - ty * GI2D_BLOCK) - 1;
    const float ox = (float)(tx * GI2D_BLOCK) + cv.pixel_offset;
    const float oy = (float)(ty * GI2D_BLOCK) + cv.pixel_offset;
    const StepConsts kc = step_consts(cv);
    for (int base = 0; base < n_raw; base += EB) {
        const int n = min(EB, n_raw - base);
        // ---- A: stage entry `tid` ----------------------------------------------------------
        int cnt = 0;
        if (tid < n) {
            const int gid = base == 0 ? gid_first : lst[base + tid];
            const float gx = xys[2 * gid], gy = xys[2 * gid + 1];
            const float a = conics[3 * gid], b = conics[3 * gid + 1], c = conics[3 * gid + 2];
            const float op = opacity[gid];
            const int rad = radii[gid];
            int x0 = 0, x1 = x_hi, y0 = 0, y1 = y_hi;
            if (cull) support_rect(gx, gy, a, b, c, op, cv.alpha_min, ox, oy, x0, x1, y0, y1);
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            cnt = (w > 0 && h > 0) ? w * h : 0;
            const int rect = cnt ? (x0 | (x1 << 8) | (y0 << 16) | (y1 << 24)) : 0;
            s_r0[tid] = make_float4(gx, gy, a, b);
            s_r1[tid] = make_float4(c, op, colors[3 * gid], colors[3 * gid + 1]);
            // (bit 31 of the staged radius: a general entry, see bwd_pixels)
            const int general = plain_entry(a, b, c, op, cv.alpha_cap) ? 0 : (int)0x80000000;
            s_r2[tid] = make_float4(colors[3 * gid + 2], __int_as_float(gid),
                                    __int_as_float(rad | general), __int_as_float(rect));
        }
        // ---- B: pieces ---------------------------------------------------------------------
        // The batch's pixels are cut into about QUADS pieces per EB / 4 entries (at most
        // BWD_MAX_ROUNDS x QUADS): the quads then make that many rounds over pieces of about
        // equal size.  With fewer, larger pieces a round lasts as long as its largest entry while
        // most quads hold small ones (measured on fitted-size supports: 3 rounds of ~55 pixel
        // steps for 128 entries, where 4 rounds of ~17 do).
        int total = 0;
        (void)block_inclusive_scan<NT>(cnt, s_w, total);
        const int rounds = min(MAX_ROUNDS, (n + EB / 4 - 1) / (EB / 4));
        // expected pieces = total / chunk + n / 2 (half a piece of rounding per entry)
        int chunk = max(8, (int)((float)total * __builtin_amdgcn_rcpf(
                                     (float)(QUADS * rounds - (n >> 1)))) + 1);
        // ceil(cnt / chunk) <= 32 (cnt <= 256, chunk >= 8) from the hardware reciprocal: exact,
        // the quotient's fractional part is a multiple of 1/chunk
        float rchunk = __builtin_amdgcn_rcpf((float)chunk);
        int my_pieces = cnt ? (int)(((float)(cnt + chunk - 1) + 0.5f) * rchunk) : 0;
        int n_pieces = 0;
        int incl = block_inclusive_scan<NT>(my_pieces, s_w, n_pieces);
        if (n_pieces > MAX_PIECES) {
            // (rare: many entries just over a multiple of the chunk)  sum ceil(cnt / chunk) <=
            // total / chunk + n, so this chunk fits whatever the counts are
            chunk = max(chunk, (int)((float)total * __builtin_amdgcn_rcpf(
                                         (float)(MAX_PIECES - n))) + 2);
            rchunk = __builtin_amdgcn_rcpf((float)chunk);
            my_pieces = cnt ? (int)(((float)(cnt + chunk - 1) + 0.5f) * rchunk) : 0;
            incl = block_inclusive_scan<NT>(my_pieces, s_w, n_pieces);
        }
        if (tid < n) {
            const int p0 = incl - my_pieces;
            s_start[tid] = p0;
            for (int j = 0; j < my_pieces; ++j) s_owner[p0 + j] = (unsigned char)tid;
        }
        if (tid == 0) s_start[n] = n_pieces;
        __syncthreads();
        // ---- C: a quad per piece -----------------------------------------------------------
        for (int p = quad; p < n_pieces; p += QUADS) {
            const int e = s_owner[p];
            const float4 r0 = s_r0[e], r1 = s_r1[e], r2 = s_r2[e];
            const float gx = r0.x, gy = r0.y, a = r0.z, b = r0.w, c = r1.x, op = r1.y;
            const float cr = r1.z, cg = r1.w, cb = r2.x;
            const int rect = __float_as_int(r2.w);
            const int x0 = rect & 255, x1 = (rect >> 8) & 255, y0 = (rect >> 16) & 255;
            const int y1 = (rect >> 24) & 255;
            const int w = x1 - x0 + 1, ecnt = w * (y1 - y0 + 1);
            const int k_beg = (p - s_start[e]) * chunk, k_end = min(ecnt, k_beg + chunk);
            // pixel k of the rectangle is (x0 + k % w, y0 + k / w); the quad's lanes take the
            // piece's pixels round-robin.  Small quotients from the hardware reciprocal (exact:
            // fractional parts are multiples of 1/w).
            const float rw = __builtin_amdgcn_rcpf((float)w);
            const int k0 = k_beg + q;
            const int ky = (int)(((float)k0 + 0.5f) * rw);
            const int x = x0 + k0 - ky * w, y = y0 + ky;
            float acc[PART_STRIDE];
#pragma unroll
            for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
            if (__float_as_int(r2.z) >= 0)
                bwd_pixels<4, true>(acc, k0, k_end, x, y, x1, w, gx, gy, ox, oy, a, b, c, op, cr, cg,
                                    cb, s_v, cv, kc);
            else
                bwd_pixels<4, false>(acc, k0, k_end, x, y, x1, w, gx, gy, ox, oy, a, b, c, op, cr,
                                     cg, cb, s_v, cv, kc);
#pragma unroll
            for (int k = 0; k < PART_STRIDE; ++k) acc[k] = quad_sum(acc[k]);
            if (q == 0) {
#pragma unroll
                for (int k = 0; k < PART_STRIDE; ++k) s_piece[p * PART_STRIDE + k] = acc[k];
            }
        }
        __syncthreads();
        // ---- D: entry `tid` adds its pieces and stores -------------------------------------
        if (tid < n) {
            const float4 r0 = s_r0[tid], r1 = s_r1[tid], r2 = s_r2[tid];
            const int gid = __float_as_int(r2.y), rad = __float_as_int(r2.z) & 0x7fffffff;
            // index of this tile in the Gaussian's own (band-clipped) tile box: the first GSLOTS
            // tiles write to the Gaussian's slots (with the stamp), the others to the tile's slab
            TileBox box = tile_box(r0.x, r0.y, (float)rad, tiles_x, tiles_y);
            const int by0 = max(box.y0, row0);
            const int kb = (ty - by0) * (box.x1 - box.x0) + (tx - box.x0);
            const bool own = rad > 0 && kb >= 0 && kb < GSLOTS;
            // (an entry that cannot reach the tile leaves its own slot's old stamp: pass 2 then
            // counts it as the zero it is; a slab slot is always written)
            if (my_pieces > 0 || !own) {
                float sum[PART_STRIDE];
#pragma unroll
                for (int k = 0; k < PART_STRIDE; ++k) sum[k] = 0.f;
                const int p0 = s_start[tid];
                for (int p = p0; p < p0 + my_pieces; ++p)
#pragma unroll
                    for (int k = 0; k < PART_STRIDE; ++k) sum[k] += s_piece[p * PART_STRIDE + k];
                // v_xy = sum vs * (conic . d) = conic . (sum vs * d); v_conic = (1/2, 1, 1/2) * moments
                const float a = r0.z, b = r0.w, c = r1.x;
                const float m0 = sum[0], m1 = sum[1];
                sum[0] = __builtin_fmaf(b, m1, a * m0);
                sum[1] = __builtin_fmaf(c, m1, b * m0);
                sum[2] *= 0.5f;
                sum[4] *= 0.5f;
                if (own) {
                    float2 *dst = reinterpret_cast<float2 *>(
                        gslots + ((size_t)kb * gslot_n + gid) * GSLOT_STRIDE);
                    dst[0] = make_float2(sum[0], sum[1]);
                    dst[1] = make_float2(sum[2], sum[3]);
                    dst[2] = make_float2(sum[4], sum[5]);
                    dst[3] = make_float2(sum[6], sum[7]);
                    dst[4] = make_float2(sum[8], __int_as_float(stamp));
                } else {
                    float *dst = partials + (slot0 + base + tid) * PART_STRIDE;
#pragma unroll
                    for (int k = 0; k < PART_STRIDE; ++k) dst[k] = sum[k];
                }
            }
        }
        __syncthreads();  // the staging buffers are reused by the next batch
    }
}

