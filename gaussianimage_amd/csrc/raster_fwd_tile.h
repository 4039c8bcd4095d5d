// raster_fwd_tile.h -- the forward rasteriser's work for one tile (shared by the forward kernels
// and the fused fitting kernel).
#pragma once
#include "host.h"
#include "projection.h"
#include <type_traits>

// LDS words per staged entry: three float4 records (the third holds blue and the quadrant mask)
// and the id.  The records are three arrays of the same 16-byte stride, so that the pair loop
// reaches all of an entry's words from one address register.  A workgroup of BATCH lanes holds
// BATCH entries (13 KB for 256 lanes, 3.25 KB for 64: the small shapes need the small footprint
// to reach their occupancy); the same words are the sort buffer of a list longer than one batch.
#define FWD_LDS_INTS_PER_ENTRY 13

// L2 = true adds the fitting loop's epilogue (gaussianimage_cholesky.py:83-85,104 with
// utils.py:24-25): clamp to [0,1], L2 loss against the target (stored [3,H,W] like the model's
// gt_image), and the gradient of the mean-squared error with respect to the unclamped image
// written to `v_img` [H,W,3]; each tile leaves its sum of squared errors in loss_part[tile].
//
// SETS > 1 renders further colour sets of the same Gaussians in the same pass (the model's
// forward rasterises three colour tensors over one projection,
// gaussianimage_cholesky.py:81-97): the geometry, the culling and the alpha of a (pixel,
// Gaussian) pair are computed once and feed 3*SETS accumulators.

//
// PPL (pixels per lane) = 1, 2 or 4 shrinks the workgroup to 4, 2 or 1 waves, each wave
// rendering PPL quadrants.  A frame of many sparse tiles is bound by the chain of dependent
// memory rounds of each workgroup, not by arithmetic; smaller workgroups put 2x / 4x as many
// tiles in flight per CU.  Every pixel still adds its Gaussians in list order, so all variants
// give the same bits.
//
// KEEP = true (the fused fitting kernel; with L2): the loss gradient of the tile's pixels stays
// in LDS (keep_v[ch * 256 + ly * 16 + lx]) instead of going to v_img, and a list of up to one batch
// (256 / PPL) of ids that this pass sorts (FINALIZE) is also left, sorted, in keep_ids.
// `smem` holds FWD_LDS_INTS_PER_ENTRY * (256 / PPL) ints (16-byte aligned), `s_x` the extra
// colour sets' records.  Returns the tile's raw list length.
//
// PACKED = true: the Gaussians' attributes come from the per-Gaussian records a projection wrote
// (`rec`, projection.h) -- one 48-byte line per list entry; false: from the separately passed
// tensors, turned into the same record while staging.  Same bits either way.
template <bool FINALIZE, bool L2, int SETS, int PPL, bool KEEP, bool PACKED = false>
__device__ __forceinline__ int fwd_tile(
    int *smem, float4 *s_x, float *keep_v, int *keep_ids, int *tile_count, int *tile_len, int *seg, int C, const float *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, const float *__restrict__ rec,
    const float *__restrict__ background, int H, int W,
    int tiles_x,
    int row0, int T, gi2d_conv cv, int cull, DevStatus *status, long long generation,
    float *__restrict__ out, const float *__restrict__ gt, float *__restrict__ v_img,
    float *__restrict__ loss_part, ExtraSets extra)
{
    static_assert(SETS >= 1 && SETS <= 3 && !(L2 && SETS > 1), "unsupported variant");
    static_assert(PPL == 1 || ((PPL == 2 || PPL == 4) && SETS == 1), "unsupported variant");
    static_assert(!KEEP || L2, "unsupported variant");
    static_assert(!PACKED || SETS == 1, "unsupported variant");
    constexpr int BATCH = 256 / PPL;            // lanes of the workgroup = entries staged at once
    constexpr int NX = 3 * (SETS - 1);          // extra colour floats per entry
    constexpr int NX4 = (NX + 3) / 4;           // ... in float4 records
    constexpr int FWD_LDS_INTS = FWD_LDS_INTS_PER_ENTRY * BATCH;
    // s_x: record q of entry k of the extra sets at [q*256+k]
    float4 *s_a = reinterpret_cast<float4 *>(smem);         // gx, gy, -0.5*a*log2e, -b*log2e
    // -0.5*c*log2e, log2(opacity) (plain entries) or opacity (general entries), r, g
    float4 *s_b = reinterpret_cast<float4 *>(smem + 4 * BATCH);
    // b, quadrant mask (bit = 2*qy + qx; bit 4 = general entry), -, -
    float4 *s_c = reinterpret_cast<float4 *>(smem + 8 * BATCH);
    int *s_id = smem + 12 * BATCH;                          // ids of the batch (rank sort)
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
    // The loss epilogue's target values are requested now and used after the pair loops: their
    // round trip runs under the tile's own work instead of behind it.
    float gtv[PPL][3];
    if (L2) {
        const size_t plane = (size_t)H * W;
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const size_t gp = pix[u] + (size_t)(row0 * GI2D_BLOCK) * W;  // index in the full frame
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) gtv[u][ch] = inside[u] ? gt[ch * plane + gp] : 0.f;
        }
    }
    int *lst = seg + (size_t)t * C;
    // the first batch's id is loaded before the list length is known (a segment always has C
    // words): one round trip to memory instead of two
    int gid_first = tid < C ? lst[tid] : 0;
    const int n_raw = FINALIZE ? tile_count[CNT_AT(t)] : tile_len[t];
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
                if (KEEP) keep_ids[rank] = gid;
                slot = rank;
            }
        }
        int mask = 0;
        float blue = 0.f;
        if (tid < n) {
            Record R;
            if (PACKED) {
                const float4 *src = reinterpret_cast<const float4 *>(rec) + 3 * (size_t)gid;
                R.r0 = src[0];
                R.r1 = src[1];
                R.r2 = src[2];
            } else {
                R = make_record(xys[2 * gid], xys[2 * gid + 1], conics[3 * gid], conics[3 * gid + 1],
                                conics[3 * gid + 2], opacity[gid], cv.alpha_cap, cv.alpha_min,
                                colors[3 * gid], colors[3 * gid + 1], colors[3 * gid + 2]);
            }
            s_a[slot] = R.r0;
            s_b[slot] = R.r1;
            blue = R.r2.x;
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
            const int flags = __float_as_int(R.r2.y);
            int x0 = 0, x1 = x_hi, y0 = 0, y1 = y_hi;
            if (cull)
                support_rect_from((flags >> 1) & 3, R.r2.z, R.r2.w, R.r0.x, R.r0.y, ox, oy, x0, x1,
                                  y0, y1);
            if (x0 <= x1 && y0 <= y1) {
                const int mx = (x0 < 8 ? 1 : 0) | (x1 >= 8 ? 2 : 0);
                const int my = (y0 < 8 ? 1 : 0) | (y1 >= 8 ? 2 : 0);
                mask = ((my & 1) ? mx : 0) | ((my & 2) ? (mx << 2) : 0);
            }
            mask |= (flags & 1) << 4;  // a general entry
        }
        // lanes past the batch keep slot = tid >= n and store an empty mask
        s_c[slot] = make_float4(blue, __int_as_float(mask), 0.f, 0.f);
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int m = __float_as_int(s_c[c0 + lane].y);
            const unsigned long long general_bits = __ballot((m >> 4) & 1);
            // MIXED = false_type: every entry of the chunk is plain (the usual case): the loop
            // body has no branch at all; otherwise a scalar branch per entry picks the form
            // (the entry index is uniform over the wave).  Entries are taken in list order
            // either way.
            // the weight of entry (A, B) on pixel u of this lane (a plain entry)
            auto plain_weight = [&](const float4 &A, const float4 &B, int u) {
                const float dx = A.x - px[u], dy = A.y - py[u];
                const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                // p = log2(opacity) - sigma*log2(e) <= log2(opacity): alpha = exp2(p)
                // <= opacity <= cap, and sigma < 0 cannot happen
                const float p = __builtin_fmaf(B.x * dy, dy, __builtin_fmaf(q, dx, B.y));
                const float alpha = __builtin_amdgcn_exp2f(p);
                return alpha < cv.alpha_min ? 0.f : alpha;
            };
            auto chunk = [&](auto mixed) {
#pragma unroll
                for (int u = 0; u < PPL; ++u) {
                    unsigned long long bits = __ballot((m >> (wave * PPL + u)) & 1);
                    if (!decltype(mixed)::value && NX == 0) {
                        // Two entries per trip: an evaluation is one chain of dependent
                        // instructions behind an LDS read, and the waves of a SIMD that are in
                        // their loops at the same time are too few to hide it; the two chains
                        // of a trip are independent up to the sums, which still take the
                        // entries in list order (the same bits).  (Measured on one box, 24
                        // frames of 768x512 / 30 000: one entry per trip 66.8 us, two 64.3,
                        // three 65.9, four -- 68 registers -- 67.0.)
                        while (bits & (bits - 1)) {
                            const int k1 = c0 + (int)__builtin_ctzll(bits);
                            bits &= bits - 1;
                            const int k2 = c0 + (int)__builtin_ctzll(bits);
                            bits &= bits - 1;
                            const float4 A1 = s_a[k1], B1 = s_b[k1];
                            const float c1 = s_c[k1].x;
                            const float4 A2 = s_a[k2], B2 = s_b[k2];
                            const float c2 = s_c[k2].x;
                            const float w1 = plain_weight(A1, B1, u);
                            const float w2 = plain_weight(A2, B2, u);
                            r[u] = __builtin_fmaf(B1.z, w1, r[u]);
                            g[u] = __builtin_fmaf(B1.w, w1, g[u]);
                            b[u] = __builtin_fmaf(c1, w1, b[u]);
                            r[u] = __builtin_fmaf(B2.z, w2, r[u]);
                            g[u] = __builtin_fmaf(B2.w, w2, g[u]);
                            b[u] = __builtin_fmaf(c2, w2, b[u]);
                        }
                    }
                    while (bits) {
                        const int kk = (int)__builtin_ctzll(bits);
                        const int k = c0 + kk;
                        bits &= bits - 1;
                        const float4 A = s_a[k];
                        const float4 B = s_b[k];
                        const float cb = s_c[k].x;
                        float w;
                        if (!decltype(mixed)::value || !((general_bits >> kk) & 1ull)) {
                            w = plain_weight(A, B, u);
                        } else {
                            const float dx = A.x - px[u], dy = A.y - py[u];
                            const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                            const float p = __builtin_fmaf(B.x * dy, dy, q * dx);
                            const float e = __builtin_amdgcn_exp2f(p);
                            const float alpha = fminf(cv.alpha_cap, B.y * e);
                            w = (p > 0.f || alpha < cv.alpha_min) ? 0.f : alpha;
                        }
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
            };
            if (general_bits == 0ull) chunk(std::false_type());
            else chunk(std::true_type());
        }
    }
    // The background is read on the device (three uniform loads) and added only when it counts;
    // an overflowed segment turns the tile's pixels into NaN (never silently wrong).  Both are
    // uniform over the workgroup and rare: real branches, nothing on the usual path.
    float bg[3] = {0.f, 0.f, 0.f};
    if (cv.bg_weight != 0.f) {
        bg[0] = cv.bg_weight * background[0];
        bg[1] = cv.bg_weight * background[1];
        bg[2] = cv.bg_weight * background[2];
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            r[u] += bg[0];
            g[u] += bg[1];
            b[u] += bg[2];
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) xacc[q] += bg[q % 3];
    }
    if (n_raw > C) {
        const float nan = __builtin_nanf("");
#pragma unroll
        for (int u = 0; u < PPL; ++u) r[u] = g[u] = b[u] = nan;
#pragma unroll
        for (int q = 0; q < NX; ++q) xacc[q] = nan;
    }
    float sq[PPL];
    float keep_vo[PPL][3];  // (pixels outside the image keep a zero gradient)
#pragma unroll
    for (int u = 0; u < PPL; ++u) keep_vo[u][0] = keep_vo[u][1] = keep_vo[u][2] = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        const float rr = r[u], gg = g[u], bb = b[u];
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
            const float scale = 2.0f / (3.0f * (float)((size_t)H * W));
            const float v[3] = {rr, gg, bb};
            float vo[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float x = v[ch];
                const float d = fminf(fmaxf(x, 0.f), 1.f) - gtv[u][ch];
                sq[u] = __builtin_fmaf(d, d, sq[u]);
                vo[ch] = (x >= 0.f && x <= 1.f) ? scale * d : 0.f;  // clamp passes its interior
            }
            if (!KEEP) {
                float *dst = v_img + 3 * pix[u];
                dst[0] = vo[0];
                dst[1] = vo[1];
                dst[2] = vo[2];
            } else {
                keep_vo[u][0] = vo[0];
                keep_vo[u][1] = vo[1];
                keep_vo[u][2] = vo[2];
            }
        }
    }
    if (NX && inside[0]) {
#pragma unroll
        for (int q = 0; q < NX; ++q) extra.out[q / 3][3 * pix[0] + q % 3] = xacc[q];
    }
    if (KEEP) {
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const int q = wave * PPL + u;
            const int lx = ((q & 1) << 3) + (lane & 7), ly = ((q >> 1) << 3) + (lane >> 3);
            // (three planes of 256 floats: raster_bwd_tile.h)
            keep_v[ly * GI2D_BLOCK + lx] = keep_vo[u][0];
            keep_v[256 + ly * GI2D_BLOCK + lx] = keep_vo[u][1];
            keep_v[512 + ly * GI2D_BLOCK + lx] = keep_vo[u][2];
        }
    }
    if (L2) {
        // tile sum in a fixed order whatever the variant: the lanes of a quadrant by butterfly,
        // the four quadrants through LDS
        __syncthreads();
        float *s_red = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const float v = wave_sum_lane0(sq[u]);
            if (lane == 0) s_red[wave * PPL + u] = v;
        }
        __syncthreads();
        if (tid == 0) loss_part[t] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
    return n_raw;
}
