// raster_fwd.hip -- rasterize_gaussians_sum forward and the one-call frame entry points.
#include "raster_fwd_tile.h"

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

// gridDim.y = frames of a batch: frame f has its own run of T tile lists, its own N Gaussians
// in every per-Gaussian tensor and its own H x W image (ids in the lists are frame-local).
template <bool FINALIZE, bool L2, int SETS, int PPL, bool PACKED>
__global__ __launch_bounds__(256 / PPL) void rasterize_sum_fwd_kernel(
    int *tile_count, int *tile_len, int *seg, int C, const float *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, const float *__restrict__ rec,
    const float *__restrict__ background, int H, int W, int tiles_x, int row0, int T, gi2d_conv cv,
    int cull, DevStatus *status, long long generation, float *__restrict__ out,
    const float *__restrict__ gt, float *__restrict__ v_img, float *__restrict__ loss_part,
    ExtraSets extra, int N)
{
    constexpr int NX4 = (3 * (SETS - 1) + 3) / 4;
    __shared__ __attribute__((aligned(16))) int smem[FWD_LDS_INTS_PER_ENTRY * (256 / PPL)];
    __shared__ float4 s_x[NX4 ? NX4 * 256 : 1];
    const int f = blockIdx.y;
    if (f != 0) {
        const size_t n = (size_t)f * N, t = (size_t)f * T;
        tile_count += t * CNT_STRIDE;
        tile_len += t;
        seg += t * C;
        if (PACKED) {
            rec += n * REC_FLOATS;
        } else {
            xys += 2 * n;
            conics += 3 * n;
            colors += 3 * n;
            opacity += n;
        }
        out += (size_t)f * H * W * 3;
    }
    (void)fwd_tile<FINALIZE, L2, SETS, PPL, false, PACKED>(
        smem, s_x, nullptr, nullptr, tile_count, tile_len, seg, C, xys, conics, colors, opacity,
        rec, background, H, W, tiles_x, row0, T, cv, cull, status, generation, out, gt, v_img,
        loss_part, extra);
}
// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

int check_raster_args(gi2d_workspace *ws, int N, int H, int W, const gi2d_conv *conv, int frames)
{
    if (!ws || !conv || N < 0 || H < 1 || W < 1) return GI2D_E_INVALID;
    if (ws->generation == 0 || ws->tiles_x == 0 || !ws->lists_valid) return GI2D_E_STATE;
    if (ws->tiles_x != cdiv(W, GI2D_BLOCK) || ws->tiles_y != cdiv(H, GI2D_BLOCK) ||
        ws->n_binned != N || ws->frames != frames)
        return GI2D_E_STATE;
    return GI2D_OK;
}

// gt / v_img / loss_part non-NULL selects the fitting epilogue (then out_img may be NULL).
// sets = number of colour sets (1..3); sets beyond the first come in `extra`.
int raster_fwd_impl(gi2d_workspace *ws, const float *xys, const float *conics,
                           const float *colors, const float *opacity, const float *background,
                           int N, int H, int W, const gi2d_conv *conv, float *out_img,
                           const float *gt, float *v_img, float *loss_part, void *stream,
                           int sets, ExtraSets extra, int frames)
{
    int rc = check_raster_args(ws, N, H, W, conv, frames);
    if (rc != GI2D_OK) return rc;
    const int T = ws_tiles_per_frame(ws);
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    if (T == 0) {  // empty band
        ws->lists_raw = false;
        return GI2D_OK;
    }
    const bool l2 = gt != nullptr;
    if (l2 && (!v_img || !loss_part || sets != 1)) return GI2D_E_INVALID;
    if (frames > 1 && (l2 || sets != 1 || extra.out_chw || !out_img)) return GI2D_E_INVALID;
    if ((!out_img && !l2 && !extra.out_chw) ||
        (N > 0 && (!xys || !conics || !colors || !opacity)))
        return GI2D_E_INVALID;
    if (extra.out_chw) {
        const int y0 = ws->bin_row0 * GI2D_BLOCK;
        extra.chw_plane = (size_t)(min(H, ws->bin_row1 * GI2D_BLOCK) - y0) * (size_t)W;
    }
    if (conv->bg_weight != 0.0f && !background) return GI2D_E_INVALID;
    // the per-Gaussian records of the projection that filled these lists, when it wrote them
    // for exactly these tensors
    const bool packed = sets == 1 && ws->rec_valid && ws->rec_xys == (const void *)xys &&
                        ws->rec_colors == (const void *)colors &&
                        ws->rec_opacity == (const void *)opacity;
#define GI2D_RASTER_ARGS ws->d_count, ws->d_len, ws->d_seg, ws->seg_cap, xys, conics, colors, \
                         opacity, ws->d_rec, background, H, W, ws->tiles_x, ws->bin_row0, T, *conv, \
                         ws->cull, ws->d_status, ws->generation, out_img, gt, v_img, loss_part, \
                         extra, N
    const bool fin = ws->lists_raw;
    // pixels per lane: many sparse tiles -> smaller workgroups, more tiles in flight per CU
    int ppl = 1;
    if (sets == 1) {
        ppl = ws->raster_ppl;
        // (by the tiles of the whole LAUNCH: with enough workgroups to fill the chip several
        // times over, one wave per tile pays the per-tile work -- staging, the sort, the
        // barriers -- once instead of four times, whatever the tiles hold.  Measured at 768x512 /
        // 30 000 Gaussians, 1 / 2 / 4 pixels per lane: 8 frames 37 / 32 / 33 us, 24 frames
        // 95 / 76 / 69 us; one frame 11 / 13 / 16 us.  Long lists -- fitted Gaussians, three times
        // the footprint, up to 300 per tile -- keep two waves: 24 frames 237 / 227 / 250 us)
        if (ppl == 0) {
            // (small batches: two frames of 1536 tiles 84 / 89 / 80 k frames/s, four 106 / 110 / 108)
            const long long launch_tiles = (long long)T * frames;
            const long long two_from = frames > 1 ? 3072 : GI2D_PPL2_MIN_TILES;
            ppl = launch_tiles >= GI2D_PPL4_MIN_TILES ? 4 : (launch_tiles >= two_from ? 2 : 1);
            if (ppl == 4 && frames > 1 && ws->h_status[0] > 128) ppl = 2;
        }
    }
    const dim3 grid(T, frames);
#define GI2D_RASTER_LAUNCH(FIN, LL, SETS, PPL, PK) \
    rasterize_sum_fwd_kernel<FIN, LL, SETS, PPL, PK><<<grid, 256 / PPL, 0, st>>>(GI2D_RASTER_ARGS)
#define GI2D_RASTER_PICK2(SETS, PPL, PK)                        \
    do {                                                        \
        if (fin && l2) GI2D_RASTER_LAUNCH(true, true, SETS, PPL, PK);   \
        else if (fin) GI2D_RASTER_LAUNCH(true, false, SETS, PPL, PK);   \
        else if (l2) GI2D_RASTER_LAUNCH(false, true, SETS, PPL, PK);    \
        else GI2D_RASTER_LAUNCH(false, false, SETS, PPL, PK);           \
    } while (0)
#define GI2D_RASTER_PICK(SETS, PPL)                  \
    do {                                             \
        if (packed) GI2D_RASTER_PICK2(SETS, PPL, true);  \
        else GI2D_RASTER_PICK2(SETS, PPL, false);        \
    } while (0)
    {
    Probe probe(ws, GI2D_PROBE_RASTER_FWD, st);
    if (sets == 3) {
        if (fin) GI2D_RASTER_LAUNCH(true, false, 3, 1, false);
        else GI2D_RASTER_LAUNCH(false, false, 3, 1, false);
    } else if (sets == 2) {
        if (fin) GI2D_RASTER_LAUNCH(true, false, 2, 1, false);
        else GI2D_RASTER_LAUNCH(false, false, 2, 1, false);
    } else if (ppl == 4) {
        GI2D_RASTER_PICK(1, 4);
    } else if (ppl == 2) {
        GI2D_RASTER_PICK(1, 2);
    } else {
        GI2D_RASTER_PICK(1, 1);
    }
    }
#undef GI2D_RASTER_PICK
#undef GI2D_RASTER_PICK2
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
                       void *stream, float *out_chw = nullptr, int frames = 1)
{
    int rc = bin_tiles_impl(ws, xys, radii, N, tiles_x, tiles_y, GI2D_BIN_COUNTED, stream, frames);
    if (rc != GI2D_OK && rc != GI2D_E_PREV_OVERFLOW) return rc;
    const int late = rc;
    if (ev_begin) HIP_TRY(hipEventRecord((hipEvent_t)ev_begin, (hipStream_t)stream));
    ExtraSets extra = ExtraSets();
    extra.out_chw = out_chw;
    rc = raster_fwd_impl(ws, xys, conics, colors, opacity, background, N, H, W, conv, out_img,
                         nullptr, nullptr, nullptr, stream, 1, extra, frames);
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
    ProjExtras ex;
    ex.opacity = opacity;
    ex.colors = colors;
    int rc = project_impl(0, false, means2d, L_elements, nullptr, N, H, W, tiles_x, tiles_y, conv,
                          xys, depths, radii, conics, num_tiles_hit, ws, stream, ex);
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
    ProjExtras ex;
    ex.opacity = opacity;
    ex.colors = colors;
    int rc = project_impl(1, false, means2d, scales, rotation, N, H, W, tiles_x, tiles_y, conv, xys,
                          depths, radii, conics, num_tiles_hit, ws, stream, ex);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, raster_begin_event, raster_end_event, stream);
}

int gi2d_render_frame(const gi2d_frame_args *a)
{
    if (!a) return GI2D_E_INVALID;
    if (a->kind == 0)
        return gi2d_render_cholesky_fwd(a->ws, a->means2d, a->cov_a, a->colors, a->opacity,
                                        a->background, a->N, a->H, a->W, &a->conv, a->xys,
                                        a->depths, a->radii, a->conics, a->num_tiles_hit,
                                        a->out_img, a->raster_begin_event, a->raster_end_event,
                                        a->stream);
    if (a->kind == 1)
        return gi2d_render_rs_fwd(a->ws, a->means2d, a->cov_a, a->cov_b, a->colors, a->opacity,
                                  a->background, a->N, a->H, a->W, &a->conv, a->xys, a->depths,
                                  a->radii, a->conics, a->num_tiles_hit, a->out_img,
                                  a->raster_begin_event, a->raster_end_event, a->stream);
    return GI2D_E_INVALID;
}

int gi2d_render_batch(gi2d_workspace *ws, int kind, int frames, const float *means2d,
                      const float *cov_a, const float *cov_b, const float *colors,
                      const float *opacity, const float *background, int N, int H, int W,
                      const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                      int *num_tiles_hit, float *out_imgs, void *raster_begin_event,
                      void *raster_end_event, void *stream)
{
    if (!ws || (kind != 0 && kind != 1) || frames < 1) return GI2D_E_INVALID;
    if (H < 1 || W < 1) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    if ((int64_t)frames * N >= (int64_t)INT_MAX / 16) return GI2D_E_INVALID;
    ProjExtras ex;
    ex.opacity = opacity;
    ex.colors = colors;
    ex.frames = frames;
    int rc = project_impl(kind, false, means2d, cov_a, cov_b, N, H, W, tiles_x, tiles_y, conv, xys,
                          depths, radii, conics, num_tiles_hit, ws, stream, ex);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_imgs, raster_begin_event, raster_end_event, stream, nullptr,
                       frames);
}

int gi2d_decode_render(gi2d_workspace *ws, int kind, const gi2d_codec *codec, const float *opacity,
                       const float *background, int N, int H, int W, const gi2d_conv *conv,
                       float *xys, float *depths, int *radii, float *conics, int *num_tiles_hit,
                       float *colors, float *out_img, float *render_chw, void *stream)
{
    if (!ws || (!out_img && !render_chw)) return GI2D_E_INVALID;
    const int tiles_x = cdiv(W, GI2D_BLOCK), tiles_y = cdiv(H, GI2D_BLOCK);
    ProjExtras ex;
    ex.codec = codec;
    ex.colors_out = colors;
    ex.opacity = opacity;
    int rc = project_impl(kind, false, nullptr, nullptr, nullptr, N, H, W, tiles_x, tiles_y, conv,
                          xys, depths, radii, conics, num_tiles_hit, ws, stream, ex);
    if (rc != GI2D_OK) return rc;
    return render_tail(ws, N, H, W, tiles_x, tiles_y, conv, xys, radii, conics, colors, opacity,
                       background, out_img, nullptr, nullptr, stream, render_chw);
}

}  // extern "C"
