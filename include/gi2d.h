/*
 * gi2d.h -- C ABI of the MI355X (gfx950) accumulated-summation 2D Gaussian
 * rasterizer: projection, tile binning, rasterize_sum forward and backward.
 *
 * The reference reaches this path through three Python operators of an
 * un-vendored `gsplat` extension (/root/reference/.gitmodules:1-3); the native
 * FFI under them is not part of the reference tree, so each entry point below
 * cites the Python call site whose work it performs.
 *
 * Conventions: every pointer is a DEVICE pointer on the workspace's device
 * unless named h_*; tensors are dense row-major fp32 / int32; `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  All functions
 * return 0 on success, a negative GI2D_E_* code for a rejected argument or
 * state, or a positive hipError_t.  Nothing throws.  A workspace serves one
 * host thread and one stream at a time.
 */
#ifndef GI2D_H
#define GI2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GI2D_VERSION 100 /* 0.1.0 */
#define GI2D_BLOCK 16    /* tile edge in pixels; the reference hard-codes 16 (train.py:54) */

#define GI2D_OK 0
#define GI2D_E_INVALID (-1)       /* bad pointer / size / unsupported block size           */
#define GI2D_E_STATE (-2)         /* e.g. rasterize before any bin_tiles on this workspace  */
#define GI2D_E_CAPACITY (-3)      /* a tile's list exceeds the reserved per-tile capacity   */
#define GI2D_E_PREV_OVERFLOW (-4) /* an EARLIER asynchronous frame overflowed a tile list   */
                                  /* (those tiles of its image are NaN); the workspace has  */
                                  /* been grown, call again                                 */

/* flags of gi2d_bin_tiles */
#define GI2D_BIN_SYNC 1    /* wait for the list lengths and size the lists to fit            */
#define GI2D_BIN_COUNTED 2 /* the lists were already filled by gi2d_project_*_fwd(ws)        */
#define GI2D_BIN_NOGROW 4  /* never reallocate (fixed-memory callers)                        */

/* Conventions the reference cannot settle (its operator source is absent);
 * see gi2d_default_conv for the defaults. */
typedef struct gi2d_conv {
    float pixel_offset; /* pixel (j,i) is sampled at (j+off, i+off)            */
    float center_bias;  /* xys = 0.5*W*x + 0.5*W + bias                         */
    float alpha_cap;    /* alpha = min(cap, opacity*exp(-sigma))                */
    float alpha_min;    /* pairs with alpha < alpha_min contribute nothing      */
    float bg_weight;    /* out += bg_weight*background (0 = background ignored) */
} gi2d_conv;

typedef struct gi2d_workspace gi2d_workspace;

int gi2d_version(void);
/* Identifies the sources this library was built from (a hash the build script computes over
 * them): a binding can refuse a library that is older than the sources next to it. */
const char *gi2d_build_id(void);
const char *gi2d_error_string(int status);
void gi2d_default_conv(gi2d_conv *conv);

/* Workspace: owns the per-tile counters, the Gaussian-id lists (one fixed-
 * capacity segment per tile) and the backward partial-sum slab, so that the
 * three raster passes of GaussianImage_Cholesky.forward()
 * (gaussianimage_cholesky.py:81,89,95) share one binning and no call needs a
 * host synchronisation in steady state.  gi2d_workspace_reserve pre-allocates
 * room for max_tiles tiles of per_tile_capacity ids each. */
int gi2d_workspace_create(int device, gi2d_workspace **out);
int gi2d_workspace_destroy(gi2d_workspace *ws);
int gi2d_workspace_reserve(gi2d_workspace *ws, int64_t per_tile_capacity, int max_tiles);
/* Options: GI2D_OPT_CULL (default 1) lets the rasterisers skip pixels outside
 * a Gaussian's support rectangle, and lets the entry points that know the
 * opacities when they project (gi2d_render_*_fwd, gi2d_fit_*,
 * gi2d_decode_render) list a Gaussian only in the tiles its support can touch;
 * 0 lists every tile of the radius box and evaluates every pixel of every
 * listed tile (same results, used by the tests to prove that). */
#define GI2D_OPT_CULL 1
/* GI2D_OPT_RASTER_PPL: pixels per lane of the forward rasteriser: 1, 2 or 4
 * (workgroups of 4, 2 or 1 waves per tile), 0 (default) = chosen from the tile
 * count.  A tuning knob: every setting gives the same FORWARD bits (images, losses).  The
 * backward and fused fitting kernels follow the same setting, and their one-wave form adds a
 * Gaussian's per-pixel terms in another order than the four-wave form: gradients agree to
 * rounding, not to the bit, across settings. */
#define GI2D_OPT_RASTER_PPL 2
/* GI2D_OPT_DROP_PENDING (any value): forget the projection a chained fitting
 * step prepared for the next iteration (see gi2d_fit_backward_step_chain); the
 * next gi2d_fit_forward projects again.  For callers that change the
 * parameters themselves between two iterations. */
#define GI2D_OPT_DROP_PENDING 3
/* GI2D_OPT_RECORDS (default 1): the entry points that know colours and opacities when they
 * project (gi2d_render_*, gi2d_fit_*, gi2d_decode_render) also leave one packed 48-byte record
 * per Gaussian for the rasteriser, which then gathers one line per list entry instead of four
 * tensors.  0 = the rasteriser always reads the tensors (same results; used by the tests to
 * prove that, and by callers that cannot spare 48 bytes per Gaussian). */
#define GI2D_OPT_RECORDS 5
/* GI2D_OPT_PROJECT_AGG: how a projection fills the tile lists.  1 = workgroups of 1024 lanes
 * count their Gaussians' tiles in LDS and take one counter atomic per tile they touched; 2 = one
 * counter atomic per (Gaussian, tile) pair; 3 = by bands of tile rows: the projection leaves a
 * packed tile box per Gaussian and a second launch, one workgroup per (band, frame), owns the
 * band's tiles and needs no counter atomics (frames of at most 65 536 Gaussians; two launches,
 * measured slower than form 1 at 8 x 30 000 Gaussians); 4 = through cells of tiles: the
 * projection sorts (id, tile box) into per-cell buckets, a second launch, one workgroup per
 * (cell, frame), owns the cell's tiles -- no atomic on any tile counter, for frames of many
 * tiles whose Gaussians are in no spatial order (buckets: frames x cells x N x 8 bytes of
 * workspace memory, at most 1 GiB); 0 (default) = for batches of four frames and more form 4
 * when a frame has at least 2048 tiles or lists have been longer than 128, else form 1; for
 * single frames (or bands of tile rows) form 4 from 8192 tiles and 131 072 Gaussians, else
 * form 2.  A tuning knob: the lists are the same. */
#define GI2D_OPT_PROJECT_AGG 6
/* GI2D_OPT_BWD_SERIAL (testing aid): positions the counter of backward launches, from which
 * the stamps of the backward's partial sums derive (1 .. INT_MAX, then the slots are cleared
 * and the stamps start over), so that a test can cross the wrap. */
#define GI2D_OPT_BWD_SERIAL 4
/* GI2D_OPT_FAIL_SIDE_ALLOCS (testing aid): while non-zero, the allocations of the list-filling
 * forms' side buffers (cell buckets, packed tile boxes) report "out of memory" without calling
 * hipMalloc; the projection must then fall back to a form that needs none and give the same
 * lists. */
#define GI2D_OPT_FAIL_SIDE_ALLOCS 7
int gi2d_workspace_set_option(gi2d_workspace *ws, int option, int value);
/* Instrumentation: from now on the library records begin_event right before and end_event right
 * after every launch of one kind of kernel on this workspace (hipEvent_t, created by the caller
 * with timing enabled), on the stream of the launch; kernel 0 or NULL events switch it off.
 * A benchmark reads the launch's duration from the pair after synchronising.  Costs two event
 * records per probed launch and nothing otherwise. */
#define GI2D_PROBE_PROJECT 1     /* project_*_fwd_kernel (with list filling)           */
#define GI2D_PROBE_RASTER_FWD 2  /* rasterize_sum_fwd_kernel                            */
#define GI2D_PROBE_RASTER_BWD 3  /* rasterize_sum_bwd_kernel / _wave_kernel (pass 1)    */
#define GI2D_PROBE_GATHER 4      /* gather_grads_kernel (pass 2)                        */
#define GI2D_PROBE_FIT_RASTER 5  /* fit_raster_kernel (forward + loss + backward pass 1) */
#define GI2D_PROBE_FIT_TAIL 6    /* fit_tail_kernel (gather + chain + Adan + projection) */
int gi2d_workspace_set_probe(gi2d_workspace *ws, int kernel, void *begin_event, void *end_event);
/* Tile-row band: binning and rasterisation cover tile rows [row_begin,row_end)
 * of the frame only (row_end is clipped to the frame); out_img / v_out then
 * point at the band's first pixel row.  Default: the whole frame.  This is
 * the partition of the tile-row sharding across GPUs; it has no counterpart
 * in the single-GPU reference. */
int gi2d_workspace_set_rows(gi2d_workspace *ws, int row_begin, int row_end);
/* info[0]=per-tile list capacity, [1]=largest per-tile count published by
 * completed frames, [2]=binning generation, [3]=tiles_x, [4]=tiles_y, [5]=Gaussians binned
 * (per frame), [6],[7]=tile rows [begin,end) of the current lists, [8]=frames of the current
 * lists (1 unless gi2d_render_batch made them), [9]=1 if the rasteriser's per-Gaussian records
 * of the current lists are valid */
int gi2d_workspace_info(gi2d_workspace *ws, int64_t info[10]);

/* project_gaussians_2d(means2d, L_elements, H, W, tile_bounds)
 * (gaussianimage_cholesky.py:77).  Outputs xys[N,2], depths[N], radii[N],
 * conics[N,3], num_tiles_hit[N].  With ws != NULL the tile lists of the
 * binning are filled in the same launch (then pass GI2D_BIN_COUNTED). */
int gi2d_project_cholesky_fwd(const float *means2d, const float *L_elements, int N, int H, int W,
                              int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                              float *depths, int *radii, float *conics, int *num_tiles_hit,
                              gi2d_workspace *ws, void *stream);

/* project_gaussians_2d_scale_rot(means2d, scales, rotation, H, W, tile_bounds)
 * (gaussianimage_rs.py:75). */
int gi2d_project_rs_fwd(const float *means2d, const float *scales, const float *rotation, int N,
                        int H, int W, int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                        float *depths, int *radii, float *conics, int *num_tiles_hit,
                        gi2d_workspace *ws, void *stream);

/* Backward of the two projections, reached through loss.backward()
 * (gaussianimage_cholesky.py:116, gaussianimage_rs.py:87). */
int gi2d_project_cholesky_bwd(const float *means2d, const float *L_elements, int N, int H, int W,
                              const float *v_xys, const float *v_conics, float *v_means2d,
                              float *v_L_elements, void *stream);
int gi2d_project_rs_bwd(const float *means2d, const float *scales, const float *rotation, int N,
                        int H, int W, const float *v_xys, const float *v_conics, float *v_means2d,
                        float *v_scales, float *v_rotation, void *stream);

/* Tile binning: the hidden first half of rasterize_gaussians_sum
 * (gaussianimage_cholesky.py:81 passes xys, radii, num_tiles_hit for it).
 * Groups (tile, Gaussian) intersections by tile, ascending Gaussian id inside
 * a tile. */
int gi2d_bin_tiles(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                   int tiles_y, int flags, void *stream);

/* Copies the current lists out in compact form (device or host destinations):
 * tile_offsets gets tiles+1 ints (exclusive scan of the list lengths), ids
 * (may be NULL) tile_offsets[tiles] ints.  Synchronises the stream. */
int gi2d_copy_bins(gi2d_workspace *ws, int *tile_offsets, int *ids, int64_t ids_capacity,
                   void *stream);

/* rasterize_gaussians_sum(xys, depths, radii, conics, num_tiles_hit, colors,
 * opacity, H, W, BLOCK_H, BLOCK_W, background=..., return_alpha=False)
 * forward (gaussianimage_cholesky.py:81-82): out_img[H,W,3], using the lists
 * of the last gi2d_bin_tiles on ws.  colors[N,3], opacity[N]; background[3] is
 * device memory like the rest and is read only when conv->bg_weight != 0 (it
 * may be NULL otherwise). */
int gi2d_rasterize_sum_fwd(gi2d_workspace *ws, const float *xys, const float *conics,
                           const float *colors, const float *opacity, const float *background,
                           int N, int H, int W, const gi2d_conv *conv, float *out_img,
                           void *stream);

/* The same forward for num_sets (1..GI2D_MAX_COLOR_SETS) colour tensors over one
 * set of Gaussians, in one pass: colors[s] is [N,3], out_imgs[s] is [H,W,3]
 * (host arrays of device pointers).  Replaces the three rasterize_gaussians_sum
 * calls of GaussianImage_Cholesky.forward (gaussianimage_cholesky.py:81-97:
 * features, gauss_colors, 1-gauss_colors) with a single launch; each image is
 * bit-identical to the one gi2d_rasterize_sum_fwd gives for that colour set. */
#define GI2D_MAX_COLOR_SETS 3
int gi2d_rasterize_sum_fwd_multi(gi2d_workspace *ws, const float *xys, const float *conics,
                                 int num_sets, const float *const *colors, const float *opacity,
                                 const float *background, int N, int H, int W,
                                 const gi2d_conv *conv, float *const *out_imgs, void *stream);

/* Its backward (gaussianimage_cholesky.py:116): v_out[H,W,3] ->
 * v_xys[N,2], v_conics[N,3], v_colors[N,3], v_opacity[N].  Deterministic: no
 * floating-point atomics. */
int gi2d_rasterize_sum_bwd(gi2d_workspace *ws, const float *xys, const int *radii,
                           const float *conics, const float *colors, const float *opacity, int N,
                           int H, int W, const gi2d_conv *conv, const float *v_out, float *v_xys,
                           float *v_conics, float *v_colors, float *v_opacity, void *stream);

/* One call per frame: gi2d_project_*_fwd (filling the lists) + gi2d_bin_tiles +
 * gi2d_rasterize_sum_fwd on the workspace's current tile rows; the forward of
 * gaussianimage_cholesky.py:77-82 / gaussianimage_rs.py:75-77 without the
 * Python between the operators.  raster_*_event: optional hipEvent_t recorded
 * on `stream` right before / after the rasterize launch (instrumentation). */
int gi2d_render_cholesky_fwd(gi2d_workspace *ws, const float *means2d, const float *L_elements,
                             const float *colors, const float *opacity, const float *background,
                             int N, int H, int W, const gi2d_conv *conv, float *xys, float *depths,
                             int *radii, float *conics, int *num_tiles_hit, float *out_img,
                             void *raster_begin_event, void *raster_end_event, void *stream);
int gi2d_render_rs_fwd(gi2d_workspace *ws, const float *means2d, const float *scales,
                       const float *rotation, const float *colors, const float *opacity,
                       const float *background, int N, int H, int W, const gi2d_conv *conv,
                       float *xys, float *depths, int *radii, float *conics, int *num_tiles_hit,
                       float *out_img, void *raster_begin_event, void *raster_end_event,
                       void *stream);

/* The same call with its arguments in one struct, for render loops over
 * persistent buffers: a host that prepares the struct once pays one pointer per
 * frame instead of twenty arguments (through a foreign-function interface that
 * is most of a frame's host time).  kind 0 = gi2d_render_cholesky_fwd
 * (cov_a = L_elements, cov_b unused), kind 1 = gi2d_render_rs_fwd (cov_a =
 * scales, cov_b = rotation).  No counterpart in the reference. */
typedef struct gi2d_frame_args {
    gi2d_workspace *ws;
    int kind;
    const float *means2d, *cov_a, *cov_b, *colors, *opacity, *background;
    int N, H, W;
    gi2d_conv conv;
    float *xys, *depths;
    int *radii;
    float *conics;
    int *num_tiles_hit;
    float *out_img;
    void *raster_begin_event, *raster_end_event; /* optional hipEvent_t */
    void *stream;
} gi2d_frame_args;
int gi2d_render_frame(const gi2d_frame_args *args);

/* A BATCH of `frames` independent frames of one shape in one call and two launches: frame f
 * takes Gaussians [f*N, (f+1)*N) of every input tensor ([frames][N][...], contiguous) and
 * writes its projection outputs to the same rows and its image to out_imgs[f] ([frames][H][W][3]).
 * The images of a dataset rendered one after the other (the loop over images of
 * /root/reference/train.py:366-372 and the decode loop of test_quantize.py:71-74 visit one
 * image at a time) leave most of the chip idle at 768x512; a batch fills it.  Each frame is
 * bit-identical to the one gi2d_render_frame renders from the same inputs.  Whole frames only
 * (no tile-row band).  No counterpart in the reference. */
int gi2d_render_batch(gi2d_workspace *ws, int kind, int frames, const float *means2d,
                      const float *cov_a, const float *cov_b, const float *colors,
                      const float *opacity, const float *background, int N, int H, int W,
                      const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                      int *num_tiles_hit, float *out_imgs, void *raster_begin_event,
                      void *raster_end_event, void *stream);

/* gi2d_render_frame for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  args->ws must be NULL: the tile lists live in a file-scope device array of the
 * library (about 12.4 MiB, one copy per GPU the call is used on, held until the process ends)
 * on the CURRENT device, and the call itself -- after its first use on a device, which looks up
 * the array's address -- calls no hipMalloc / hipFree and nothing else that allocates.
 * Limits: at most GI2D_NOALLOC_MAX_TILES tiles (a 3840x2160 frame has 32 400) and
 * tiles * capacity <= GI2D_NOALLOC_MAX_LIST_INTS ids; beyond them GI2D_E_CAPACITY.
 * per_tile_capacity: ids reserved per tile (rounded up to a multiple of 64); 0 = as many as the
 * array gives this frame, at most 4096.  There is no growth: a tile whose list is longer has
 * its pixels set to NaN (never silently wrong), as in the asynchronous workspace path.
 * Results are bit-identical to gi2d_render_frame's.  Every call on a GPU shares that GPU's
 * array: calls whose GPU work overlaps on one GPU (two streams, two threads) race; calls on one
 * stream are ordered and each clears what it needs cleared.  The host side is serialised by a
 * lock inside the library (two host threads may call it; their launches must still not overlap
 * on one device).  The FIRST call on a device looks up the array's address
 * (hipGetSymbolAddress) and must be made outside graph capture; a captured call bakes that
 * call's list generation and capacity into the kernel arguments, which is what a replay needs. */
#define GI2D_NOALLOC_MAX_TILES 34816
#define GI2D_NOALLOC_MAX_LIST_INTS 2097152
int gi2d_render_frame_noalloc(const gi2d_frame_args *args, int per_tile_capacity);

/* Adan optimizer step for the parameter tensors of the fitted image, the step
 * that follows the path in every training iteration
 * (gaussianimage_cholesky.py:120 -> optimizer.py:125-235, update rule
 * optimizer.py:296-362).  One launch over all tensors.  Host arrays of device
 * pointers; `step` is the 1-based step count AFTER the increment of
 * optimizer.py:166-169; step 1 takes neg_pre_grad = -grad (optimizer.py:187).
 * clip_dev: optional device scalar multiplied into the gradients in place
 * (the global-norm clip of optimizer.py:133-150), NULL = 1. */
typedef struct gi2d_adan_hyper {
    double lr, beta1, beta2, beta3, eps, weight_decay;
    int no_prox;
    int step;
} gi2d_adan_hyper;
int gi2d_adan_step(int num_tensors, float *const *params, float *const *grads,
                   float *const *exp_avg, float *const *exp_avg_sq, float *const *exp_avg_diff,
                   float *const *neg_pre_grad, const int64_t *numel, const gi2d_adan_hyper *hyper,
                   const float *clip_dev, void *stream);

/* Fused fitting iteration on the model's RAW parameters -- the caller's loop of
 * gaussianimage_cholesky.py:101-124 / gaussianimage_rs.py:82-95 (forward, L2
 * loss, backward, Adan step) in FOUR launches (three with the chained step
 * below) and no gradient tensors.
 * kind 0: _xyz[N,2], cov_a = _cholesky[N,3], cov_b unused, features[N,3];
 * kind 1: cov_a = _scaling[N,2], cov_b = _rotation[N,1].  The activations
 * (tanh, +cholesky_bound / |.+0.5|, sigmoid*2pi) are applied inside.
 *
 * gi2d_fit_forward: projection + binning + rasterisation with the loss
 * epilogue: v_image[H,W,3] = d mse(clamp(img,0,1), gt) / d img, with gt stored
 * [3,H,W] like the model's gt_image; loss_partials[tiles] = per-tile sums of
 * squared errors (mse = sum / (3*H*W)); image[H,W,3] optional (NULL to skip);
 * background[3] as in gi2d_rasterize_sum_fwd (may be NULL while bg_weight is 0).
 * gi2d_fit_backward_step: backward rasterisation, then per Gaussian the sum of
 * its partials, the chain through projection and activations, and the Adan
 * update of its parameters in place.  State arrays hold 4 device pointers
 * each, in the order (_xyz, cov_a, cov_b, features); slot 2 is ignored for
 * kind 0. */
int gi2d_fit_forward(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                     const float *cov_b, const float *features, const float *opacity,
                     const float *background, const float *gt_chw, int N, int H, int W,
                     const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                     int *num_tiles_hit, float *image, float *v_image, float *loss_partials,
                     void *stream);
int gi2d_fit_backward_step(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                           float *features, const float *opacity, int N, int H, int W,
                           const gi2d_conv *conv, const float *xys, const int *radii,
                           const float *conics, const float *v_image, float *const *exp_avg,
                           float *const *exp_avg_sq, float *const *exp_avg_diff,
                           float *const *neg_pre_grad, const gi2d_adan_hyper *hyper, void *stream);

/* gi2d_fit_backward_step that also starts the NEXT iteration: the lane that
 * has just updated a Gaussian's parameters projects them at once (activations
 * included), overwrites xys / depths / radii / conics / num_tiles_hit with the
 * next frame's values and fills the next frame's tile lists into a second list
 * set of the workspace.  The next gi2d_fit_forward on the same workspace, kind
 * and xys buffer then starts at the rasteriser: three launches per iteration.
 * When the lists cannot be double-buffered at that moment (a capacity growth is
 * due, a new shape) it behaves exactly like gi2d_fit_backward_step.  No
 * counterpart in the reference (its loop re-projects in Python each
 * iteration, gaussianimage_cholesky.py:101-103). */
int gi2d_fit_backward_step_chain(gi2d_workspace *ws, int kind, float *xyz, float *cov_a,
                                 float *cov_b, float *features, const float *opacity, int N, int H,
                                 int W, const gi2d_conv *conv, float *xys, float *depths,
                                 int *radii, float *conics, int *num_tiles_hit,
                                 const float *v_image, float *const *exp_avg,
                                 float *const *exp_avg_sq, float *const *exp_avg_diff,
                                 float *const *neg_pre_grad, const gi2d_adan_hyper *hyper,
                                 void *stream);

/* One whole fitting iteration in one call and TWO launches: gi2d_fit_forward
 * and gi2d_fit_backward_step_chain with the two rasterisation passes in one
 * kernel.  With the L2 loss the gradient at a pixel depends on that pixel
 * alone, so a tile that has rendered goes straight on to its backward pass:
 * the loss gradient stays on chip (there is no v_image), the tile's sorted
 * list is used twice.  Same arithmetic, same results as the two calls.
 * (Frames large enough to fill their lists through cells of tiles, see
 * GI2D_OPT_PROJECT_AGG, project with that launch pair instead of inside the
 * tail: four launches.)
 * image[H,W,3] optional (NULL to skip); loss_partials[tiles] as above.  Returns
 * as gi2d_fit_forward does (GI2D_E_PREV_OVERFLOW = this iteration is complete,
 * an earlier one lost some tiles).  No counterpart in the reference (its loop
 * is Python: gaussianimage_cholesky.py:101-124). */
int gi2d_fit_step(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                  float *features, const float *opacity, const float *background,
                  const float *gt_chw, int N, int H, int W, const gi2d_conv *conv, float *xys,
                  float *depths, int *radii, float *conics, int *num_tiles_hit, float *image,
                  float *loss_partials, float *const *exp_avg, float *const *exp_avg_sq,
                  float *const *exp_avg_diff, float *const *neg_pre_grad,
                  const gi2d_adan_hyper *hyper, void *stream);

/* gi2d_fit_step for `frames` images of one size fitted in lockstep -- the same iteration number,
 * learning rate and N Gaussians each -- in one call (two launches for fewer than four images;
 * three from four on, where the next lists are filled by the batch projection instead of the
 * tail's lanes): every parameter, optimizer-state
 * and per-Gaussian tensor is [frames][N][...] (the four state arrays hold one pointer per
 * parameter tensor, as in gi2d_fit_step), gt_chw is [frames][3][H][W], images (optional)
 * [frames][H][W][3], loss_partials [frames][tiles].  Image f's parameters after the call are
 * bit-identical to gi2d_fit_step's on slice f when both run the same workgroup shape
 * (GI2D_OPT_RASTER_PPL; left to itself a batch of many tiles takes one wave per tile where a
 * single 768x512 image takes four, and the backward then adds a Gaussian's per-pixel terms in
 * another order: last-bit differences in the gradients).  The reference fits the images of a dataset one
 * after the other (/root/reference/train.py:366-372); one 768x512 fit leaves most of the chip
 * idle, a batch fills it.  Whole frames only.  No counterpart in the reference. */
int gi2d_fit_step_batch(gi2d_workspace *ws, int kind, int frames, float *xyz, float *cov_a,
                        float *cov_b, float *features, const float *opacity,
                        const float *background, const float *gt_chw, int N, int H, int W,
                        const gi2d_conv *conv, float *xys, float *depths, int *radii,
                        float *conics, int *num_tiles_hit, float *images, float *loss_partials,
                        float *const *exp_avg, float *const *exp_avg_sq,
                        float *const *exp_avg_diff, float *const *neg_pre_grad,
                        const gi2d_adan_hyper *hyper, void *stream);

/* gi2d_fit_step split where a tile-row sharded fit exchanges gradients (every rank holds all
 * parameters and the optimizer state, and rasterises its own band of tile rows, see
 * gi2d_workspace_set_rows):
 *   gi2d_fit_raster -- projection (unless the previous tail prepared it), binning, the fused
 *       rasterisation kernel on the workspace's tile rows, and the sum of every Gaussian's
 *       partial sums over THESE rows: v_xys[N,2], v_conics[N,3], v_colors[N,3], v_opacity[N].
 *       gt_chw is the whole [3,H,W] target (only the band's rows are read); image, if given,
 *       is the band; loss_partials has one entry per tile of the band.
 *   (the caller adds the four tensors over the ranks: one all-reduce of 9 N floats)
 *   gi2d_fit_tail -- per Gaussian the chain through projection and activations, the Adan
 *       update in place and the projection of the updated parameters for the next iteration,
 *       from the reduced v_xys / v_conics / v_colors.
 * With one rank (the whole frame) the two calls give the bits of gi2d_fit_step.  No counterpart
 * in the reference, which is single-GPU (gaussianimage_cholesky.py:116-123 is the step split). */
int gi2d_fit_raster(gi2d_workspace *ws, int kind, const float *xyz, const float *cov_a,
                    const float *cov_b, const float *features, const float *opacity,
                    const float *background, const float *gt_chw, int N, int H, int W,
                    const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                    int *num_tiles_hit, float *image, float *loss_partials, float *v_xys,
                    float *v_conics, float *v_colors, float *v_opacity, void *stream);
int gi2d_fit_tail(gi2d_workspace *ws, int kind, float *xyz, float *cov_a, float *cov_b,
                  float *features, const float *opacity, int N, int H, int W,
                  const gi2d_conv *conv, float *xys, float *depths, int *radii, float *conics,
                  int *num_tiles_hit, const float *v_xys, const float *v_conics,
                  const float *v_colors, float *const *exp_avg, float *const *exp_avg_sq,
                  float *const *exp_avg_diff, float *const *neg_pre_grad,
                  const gi2d_adan_hyper *hyper, void *stream);

/* ---- codec decode path ------------------------------------------------------
 * The coded attributes of one image, as compress_wo_ec() leaves them
 * (gaussianimage_cholesky.py:154-159, gaussianimage_rs.py:125-130) together
 * with the quantizers' parameters; all device pointers.
 *   xyz_half     [N,2] IEEE half: encoding_dict["xyz"]
 *   cov_a_code   kind 0: [N,3] quant_cholesky_elements; kind 1: [N,2]
 *                quant_scaling -- integer codes held as float32, as the
 *                reference holds them (quantize.py:82-84)
 *   cov_a_scale, cov_a_beta   [3] / [2]: UniformQuantizer.scale / .beta
 *   cov_b_code   kind 1: [N,1] quant_rotation with cov_b_scale/_beta [1]; kind 0: unused
 *   color_index  [N,Q] int32: feature_dc_index (Q residual-VQ layers)
 *   codebooks    [Q,K,3]: layer._codebook.embed[0] of each layer (quantize.py:146-150)
 */
#define GI2D_MAX_VQ_LAYERS 8
typedef struct gi2d_codec {
    const uint16_t *xyz_half;
    const float *cov_a_code, *cov_a_scale, *cov_a_beta;
    const float *cov_b_code, *cov_b_scale, *cov_b_beta;
    const int32_t *color_index;
    const float *codebooks;
    int num_quantizers; /* Q, 1..GI2D_MAX_VQ_LAYERS */
    int codebook_size;  /* K */
} gi2d_codec;

/* The decoded operator inputs of decompress_wo_ec (gaussianimage_cholesky.py:
 * 163-166, gaussianimage_rs.py:135-139): means2d = tanh(float(xyz)),
 * kind 0: cov_a = cholesky_elements + cholesky_bound [N,3];
 * kind 1: cov_a = |scaling + bound| [N,2], cov_b = rotation [N,1];
 * colors[N,3] = sum of the layers' code vectors. */
int gi2d_decode_params(int kind, const gi2d_codec *codec, int N, float *means2d, float *cov_a,
                       float *cov_b, float *colors, void *stream);

/* decompress_wo_ec as a whole (gaussianimage_cholesky.py:161-172,
 * gaussianimage_rs.py:132-145; the loop test_quantize.py:71-74 times) in two
 * launches: decode + projection + list filling, then rasterisation.  Outputs:
 * the projection's tensors, the decoded colors[N,3], and the image as
 * out_img[H,W,3] (unclamped, may be NULL) and/or render_chw[3,H,W] =
 * clamp(out_img,0,1) permuted, the model's "render" (may be NULL). */
int gi2d_decode_render(gi2d_workspace *ws, int kind, const gi2d_codec *codec, const float *opacity,
                       const float *background, int N, int H, int W, const gi2d_conv *conv,
                       float *xys, float *depths, int *radii, float *conics, int *num_tiles_hit,
                       float *colors, float *out_img, float *render_chw, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GI2D_H */
