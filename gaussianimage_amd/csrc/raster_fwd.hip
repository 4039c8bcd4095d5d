// raster_fwd.hip -- rasterize_gaussians_sum forward and the one-call frame entry points.
#include "host.h"
#include "support.h"
#include "tiles.h"

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

// Tile counts from which the forward rasteriser switches to 2 / 4 pixels per lane.  Measured on
// MI355X (kernel time, first pass): 3840x2160 / 200k (32 400 tiles, ~15 Gaussians per tile)
// 59 / 44 / 38 us with 1 / 2 / 4; ~1020x768 / 50k (10 752 tiles) 14.6 / 14.6 / 20.3 us;
// 768x512 / 30k (1536 tiles, ~48 per tile) 10 / 11.5 / 18 us.
#define GI2D_PPL2_MIN_TILES 12288
#define GI2D_PPL4_MIN_TILES 20480
// LDS words per staged entry: two float4 records, blue, quadrant mask, id.  A workgroup of
// BATCH lanes holds BATCH entries (11 KB for 256 lanes, 2.75 KB for 64: the small shapes need
// the small footprint to reach their occupancy); the same words are the sort buffer of a list
// longer than one batch.
#define FWD_LDS_INTS_PER_ENTRY 11

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
template <bool FINALIZE, bool L2, int SETS, int PPL>
__global__ __launch_bounds__(256 / PPL) void rasterize_sum_fwd_kernel(
    int *tile_count, int *tile_len, int *seg, int C, const float *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, const float *__restrict__ background, int H, int W,
    int tiles_x,
    int row0, int T, gi2d_conv cv, int cull, DevStatus *status, long long generation,
    float *__restrict__ out, const float *__restrict__ gt, float *__restrict__ v_img,
    float *__restrict__ loss_part, ExtraSets extra)
{
    static_assert(SETS >= 1 && SETS <= 3 && !(L2 && SETS > 1), "unsupported variant");
    static_assert(PPL == 1 || ((PPL == 2 || PPL == 4) && SETS == 1), "unsupported variant");
    constexpr int BATCH = 256 / PPL;            // lanes of the workgroup = entries staged at once
    constexpr int NX = 3 * (SETS - 1);          // extra colour floats per entry
    constexpr int NX4 = (NX + 3) / 4;           // ... in float4 records
    constexpr int FWD_LDS_INTS = FWD_LDS_INTS_PER_ENTRY * BATCH;
    __shared__ __attribute__((aligned(16))) int smem[FWD_LDS_INTS];
    __shared__ float4 s_x[NX4 ? NX4 * 256 : 1];  // extra sets: record q of entry k at [q*256+k]
    float4 *s_a = reinterpret_cast<float4 *>(smem);         // gx, gy, -0.5*a*log2e, -b*log2e
    float4 *s_b = reinterpret_cast<float4 *>(smem + 4 * BATCH);  // -0.5*c*log2e, opacity, r, g
    float *s_c = reinterpret_cast<float *>(smem + 8 * BATCH);    // b
    int *s_m = smem + 9 * BATCH;                            // quadrant mask (bit = 2*qy + qx)
    int *s_id = smem + 10 * BATCH;                          // ids of the batch (rank sort)
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
    // the background is read on the device (three uniform loads) and only when it counts
    float bg[3] = {0.f, 0.f, 0.f};
    if (cv.bg_weight != 0.f) {
        bg[0] = background[0];
        bg[1] = background[1];
        bg[2] = background[2];
    }
    float sq[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        float rr = r[u] + cv.bg_weight * bg[0];
        float gg = g[u] + cv.bg_weight * bg[1];
        float bb = b[u] + cv.bg_weight * bg[2];
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
        const float *bgs = bg;
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
// host side
// ------------------------------------------------------------------------------------------

int check_raster_args(gi2d_workspace *ws, int N, int H, int W, const gi2d_conv *conv)
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
int raster_fwd_impl(gi2d_workspace *ws, const float *xys, const float *conics,
                           const float *colors, const float *opacity, const float *background,
                           int N, int H, int W, const gi2d_conv *conv, float *out_img,
                           const float *gt, float *v_img, float *loss_part, void *stream,
                           int sets, ExtraSets extra)
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
#define GI2D_RASTER_ARGS ws->d_count, ws->d_len, ws->d_seg, ws->seg_cap, xys, conics, colors, \
                         opacity, background, H, W, ws->tiles_x, ws->bin_row0, T, *conv, \
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


extern "C" {

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

int gi2d_render_cholesky_fwd(gi2d_workspace *ws, const float *means2d, const float *L_elements,
                             const float *colors, const float *opacity, const float *background,
                             int N, int H, int W, const gi2d_conv *conv, float *xys, float *depths,
                             int *radii, float *conics, int *num_tiles_hit, float *out_img,
                             void *raster_begin_event, void *raster_end_event, void *stream)
{
    if (!ws) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    int rc = project_impl(0, false, means2d, L_elements, nullptr, N, H, W, tiles_x, tiles_y, conv,
                          xys, depths, radii, conics, num_tiles_hit, ws, stream, nullptr, nullptr,
                          opacity);
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
    int rc = project_impl(1, false, means2d, scales, rotation, N, H, W, tiles_x, tiles_y, conv, xys,
                          depths, radii, conics, num_tiles_hit, ws, stream, nullptr, nullptr,
                          opacity);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, raster_begin_event, raster_end_event, stream);
}

int gi2d_decode_render(gi2d_workspace *ws, int kind, const gi2d_codec *codec, const float *opacity,
                       const float *background, int N, int H, int W, const gi2d_conv *conv,
                       float *xys, float *depths, int *radii, float *conics, int *num_tiles_hit,
                       float *colors, float *out_img, float *render_chw, void *stream)
{
    if (!ws || (!out_img && !render_chw)) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    int rc = project_impl(kind, false, nullptr, nullptr, nullptr, N, H, W, tiles_x, tiles_y, conv,
                          xys, depths, radii, conics, num_tiles_hit, ws, stream, codec, colors,
                          opacity);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, nullptr, nullptr, stream, render_chw);
}

}  // extern "C"
