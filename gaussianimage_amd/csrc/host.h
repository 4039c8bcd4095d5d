// host.h -- host-side state of a workspace and the internal entry points the translation
// units of libgi2d share (none of them is exported).
#pragma once
#include "common.h"

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
    float *d_gslots = nullptr;  // [n][GSLOTS][GSLOT_STRIDE] per-Gaussian partial sums + stamp
    DevStatus *d_status = nullptr;
    long long *h_status = nullptr;  // pinned: [0] max count, [1] generation, [2] overflow gen.
    long long *h_status_dev = nullptr;
    int64_t generation = 0;
    int64_t overflow_ack = 0;
    int64_t bwd_serial = 0;     // backward pass-1 launches so far
    int bwd_stamp = 0;          // stamp of the latest one (1 .. INT_MAX), 0 = none yet
    int tiles_x = 0, tiles_y = 0, n_binned = 0;
    int frames = 1;                  // frames of the current lists (a batch: frames x tiles lists)
    int row0 = 0, row1 = INT_MAX;    // tile rows requested by the caller (a band)
    int bin_row0 = 0, bin_row1 = 0;  // tile rows of the current lists
    bool counters_dirty = false;     // counters may be non-zero
    bool lists_raw = false;          // lists filled but not yet sorted / finalised
    bool lists_valid = false;        // lists belong to the binning that (tiles_x, ...) describe
    bool prefilled = false;          // project_*_fwd filled the lists for the shape below
    int pre_n = 0, pre_tx = 0, pre_ty = 0, pre_r0 = 0, pre_r1 = 0, pre_frames = 1;
    // The rasteriser's per-Gaussian records (projection.h), written by a projection that knew
    // the colours and opacities; valid for exactly these tensors of the current binning.
    float *d_rec = nullptr;
    int64_t rec_cap = 0;             // Gaussians the buffer holds
    bool rec_valid = false;
    const void *rec_xys = nullptr, *rec_colors = nullptr, *rec_opacity = nullptr;
    int cull = 1;
    int raster_ppl = 0;  // 0 = choose by tile count
    int min_seg_cap = 64;
    // instrumentation (gi2d_workspace_set_probe): events recorded around one kind of launch
    int probe_kernel = 0;
    hipEvent_t probe_begin = nullptr, probe_end = nullptr;
    // list filling of a projection: 0 auto, 1 aggregated in LDS per workgroup, 2 direct (one
    // counter atomic per pair), 3 by bands of tile rows (no counter atomics)
    int project_agg = 0;
    bool fail_side_allocs = false;  // testing aid: bucket / box allocations report out of memory
    int *d_box = nullptr;            // packed tile boxes of the band-wise list filling
    int64_t box_cap = 0;
    // list filling through cells (large frames): per-cell counters and buckets
    int *d_cell_count = nullptr;     // [frames x cells][CNT_STRIDE], zero between calls
    int64_t cell_count_cap = 0;      // cells the counters are allocated for
    void *d_cell_bucket = nullptr;   // [frames x cells][N] (id, packed box)
    int64_t cell_bucket_cap = 0;     // entries the buckets hold in all
    bool fixed_memory = false;       // the workspace lives in memory that cannot grow
    int no_records = 0;  // 1: projections never write the rasteriser's records (no-alloc path)
    // Second list set (same tiles_cap x seg_cap): the fitting tail fills it with the NEXT
    // iteration's lists while it still reads the current ones; the sets then swap roles.
    int *alt_count = nullptr, *alt_len = nullptr, *alt_seg = nullptr;
    bool chain_pending = false;      // the lists were prefilled by a chained fitting step ...
    const void *chain_xys = nullptr;  // ... whose projection outputs live here
    int chain_kind = -1;
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

// Records the workspace's probe events right before / after a launch of the probed kind
// (scope = the launch statement).
struct Probe {
    hipEvent_t end = nullptr;
    hipStream_t st = nullptr;
    Probe(const gi2d_workspace *ws, int kernel, hipStream_t stream)
    {
        if (ws && ws->probe_kernel == kernel && ws->probe_begin && ws->probe_end) {
            (void)hipEventRecord(ws->probe_begin, stream);
            end = ws->probe_end;
            st = stream;
        }
    }
    ~Probe()
    {
        if (end) (void)hipEventRecord(end, st);
    }
};

// Further colour sets and the optional planar output of the forward rasteriser.
struct ExtraSets {
    const float *colors[2];  // [N,3] each
    float *out[2];           // [H,W,3] each
    // optional second form of the first image: clamped to [0,1] and planar, [3][rows][W] -- the
    // model's "render" tensor (gaussianimage_cholesky.py:83-84) without the torch ops behind it
    float *out_chw;
    size_t chw_plane;        // rows * W of the band being rendered
};

// Tile counts from which the forward rasteriser switches to 2 / 4 pixels per lane.  Measured on
// MI355X (kernel time, first pass): 3840x2160 / 200k (32 400 tiles, ~15 Gaussians per tile)
// 59 / 44 / 38 us with 1 / 2 / 4; ~1020x768 / 50k (10 752 tiles) 14.6 / 14.6 / 20.3 us;
// 768x512 / 30k (1536 tiles, ~48 per tile) 10 / 11.5 / 18 us.
#define GI2D_PPL2_MIN_TILES 12288
// List filling through cells of tiles (project.hip): frames from this size on take it by
// themselves -- its two launches are ~20 us of mostly fixed latencies at 200 000 Gaussians,
// one returning atomic per (Gaussian, tile) pair costs 4 us + pairs / 2e10 per second, and a
// Gaussian of the initial footprint has ~2.5 pairs (Gaussians: those a band of tile rows can
// expect to list); the buckets (cells x N entries of 8 bytes) may take this much memory.
#define GI2D_CELLS_MAX 1024
#define GI2D_CELLS_MIN_TILES 8192
#define GI2D_CELLS_MIN_TILES_BATCH 2048
#define GI2D_CELLS_MIN_GAUSSIANS 131072
#define GI2D_CELLS_MAX_BYTES ((int64_t)1 << 30)
#define GI2D_PPL4_MIN_TILES 20480

// A sparse frame -- many tiles, none of whose lists has ever held more than a wave's worth of
// ids -- gets one wave per tile in the backward and in the fused fitting kernel (measured at
// 3840x2160 / 200k: the 256-lane forms are bound by the VALU instructions their four waves spend
// on a handful of entries).  The raster_ppl option forces the choice (4 = one wave).
static inline bool sparse_frame(const gi2d_workspace *ws, int T)
{
    if (ws->raster_ppl != 0) return ws->raster_ppl == 4;
    return T >= GI2D_PPL4_MIN_TILES && ws->h_status[0] <= 64;
}

// Whether a single frame's projection fills the lists (tile rows r0 .. r1) through cells.
static inline bool cells_form(const gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int r0,
                              int r1, int frames = 1)
{
    if (ws->fixed_memory || tiles_x >= 256 || tiles_y >= 256 || r1 <= r0) return false;
    if (ws->project_agg == 4) return true;
    if (ws->project_agg != 0) return false;
    // batches (which otherwise aggregate per workgroup in LDS): frames of many tiles -- every
    // workgroup of the aggregated form pays one counter atomic per tile -- or of long lists --
    // its pairs no longer fit the stage.  Measured per 24 frames: 1020x768 / 50 000 (3072 tiles)
    // 59.5 us against 81.8 us, 768x512 / 30 000 with three times the footprint (lists of up to
    // 300) 39.4 against 51.6 us, but 768x512 / 30 000 as initialised 32.2 against 30.9 us and
    // 768x512 / 5 000 27.8 against 14.9 us.
    if (frames != 1)
        return frames >= 4 && (tiles_x * (r1 - r0) >= GI2D_CELLS_MIN_TILES_BATCH ||
                               ws->h_status[0] > 128);
    return tiles_x * (r1 - r0) >= GI2D_CELLS_MIN_TILES &&
           (int64_t)N * (r1 - r0) / tiles_y >= GI2D_CELLS_MIN_GAUSSIANS;
}

// tile lists of the current binning: all frames / one frame
static inline int ws_tiles_per_frame(const gi2d_workspace *ws)
{
    return ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
}
static inline int ws_tiles(const gi2d_workspace *ws) { return ws->frames * ws_tiles_per_frame(ws); }

// lists.hip
int ensure_lists(gi2d_workspace *ws, int T, int C);
int ensure_records(gi2d_workspace *ws, int64_t count);
int ensure_boxes(gi2d_workspace *ws, int64_t count);
int ensure_cells(gi2d_workspace *ws, int64_t cells, int N);
int bin_tiles_impl(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                   int tiles_y, int flags, void *stream, int frames);
int wanted_cap(const gi2d_workspace *ws, long long seen);
int zero_counters(gi2d_workspace *ws, hipStream_t st);
int finalize_lists(gi2d_workspace *ws, hipStream_t st);
int begin_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int frames, hipStream_t st);
bool begin_chain_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int frames = 1);
void commit_chain_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int kind,
                          const void *xys, int frames = 1);

// project.hip
// Optional parts of a projection call.  codec: the inputs are coded attributes (means / cov_a /
// cov_b unused) and the decoded colours go to colors_out.  opacity: the opacities the rasteriser
// will use -- lets the lists hold only tiles a Gaussian's support can touch; with colors (or a
// codec) as well, the projection also writes the rasteriser's per-Gaussian records.  frames: a
// batch of that many frames of N Gaussians each ([frames][N][...] tensors).
struct ProjExtras {
    const gi2d_codec *codec = nullptr;
    float *colors_out = nullptr;
    const float *opacity = nullptr;
    const float *colors = nullptr;
    int frames = 1;
};
int project_impl(int kind, bool raw, const float *means, const float *cov_a, const float *cov_b,
                 int N, int H, int W, int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                 float *depths, int *radii, float *conics, int *num_tiles_hit, gi2d_workspace *ws,
                 void *stream, const ProjExtras &ex = ProjExtras());

// raster_fwd.hip
int check_raster_args(gi2d_workspace *ws, int N, int H, int W, const gi2d_conv *conv,
                      int frames = 1);
int raster_fwd_impl(gi2d_workspace *ws, const float *xys, const float *conics, const float *colors,
                    const float *opacity, const float *background, int N, int H, int W,
                    const gi2d_conv *conv, float *out_img, const float *gt, float *v_img,
                    float *loss_part, void *stream, int sets = 1, ExtraSets extra = ExtraSets(),
                    int frames = 1);

// raster_bwd.hip
int ensure_bwd_buffers(gi2d_workspace *ws, int N);
int next_bwd_stamp(gi2d_workspace *ws, hipStream_t st);
int raster_bwd_pass1(gi2d_workspace *ws, const float *xys, const int *radii, const float *conics,
                     const float *colors, const float *opacity, int N, int H, int W,
                     const gi2d_conv *conv, const float *v_out, hipStream_t st);
