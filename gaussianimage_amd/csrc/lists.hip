// lists.hip -- the workspace: per-tile fixed-capacity segments, their counters, the capacity
// policy, binning from caller-made tensors, list read-back; plus version / error strings.
#include "host.h"
#include "tiles.h"
#define REC_FLOATS_HOST 12  // = REC_FLOATS of projection.h

// Binning from caller-made xys / radii (no projection in front of it).
// (gridDim.y frames of N Gaussians, each with its own run of tiles_per_frame lists)
__global__ __launch_bounds__(256) void bin_tiles_kernel(const float *__restrict__ xys,
                                                         const int *__restrict__ radii, int N,
                                                         int tiles_x, int tiles_y, int row0,
                                                         int row1, int *tile_count, int *seg, int C,
                                                         const DevStatus *status,
                                                         volatile long long *h_status,
                                                         long long generation, int tiles_per_frame)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (g == 0 && f == 0) publish_status(status, h_status, generation);
    if (g >= N) return;
    xys += (size_t)f * 2 * N;
    radii += (size_t)f * N;
    tile_count += (size_t)f * tiles_per_frame * CNT_STRIDE;
    seg += (size_t)f * tiles_per_frame * C;
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

// One wave per tile: puts the tile's ids in ascending order and finalises the tile.  The
// launch lasts as long as its slowest tile, so lists up to 256 entries stay on the register
// path; longer ones use a bitonic network in LDS, beyond the LDS budget in place in memory.
// Used when nothing rasterises right after a binning (backward first, list read-back, the
// synchronous capacity check); otherwise the forward rasteriser does this work itself.
__global__ __launch_bounds__(WAVE) void finalize_tiles_kernel(int *tile_count, int *tile_len,
                                                              int *seg, int C, DevStatus *status,
                                                              long long generation,
                                                              int tiles_per_frame)
{
    __shared__ int s_ids[SORT_LDS_MAX];
    const int lane = threadIdx.x;
    // (counters pair up inside a frame: the frame's words first, then the tile of the frame)
    const int f = blockIdx.x / tiles_per_frame, t = blockIdx.x - f * tiles_per_frame;
    tile_count += (size_t)f * tiles_per_frame * CNT_STRIDE;
    tile_len += (size_t)f * tiles_per_frame;
    seg += (size_t)f * tiles_per_frame * C;
    const int n_raw = tile_count[CNT_AT(t)];
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
// host side
// ------------------------------------------------------------------------------------------

static void free_alt_lists(gi2d_workspace *ws)
{
    if (ws->alt_count) (void)hipFree(ws->alt_count);
    if (ws->alt_len) (void)hipFree(ws->alt_len);
    if (ws->alt_seg) (void)hipFree(ws->alt_seg);
    ws->alt_count = ws->alt_len = ws->alt_seg = nullptr;
    ws->chain_pending = false;
}

// (Re)allocates the per-tile arrays for T tiles of C ids each; the device is current.
// Contents are lost: callers re-bin afterwards.
int ensure_lists(gi2d_workspace *ws, int T, int C)
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
    free_alt_lists(ws);
    ws->d_count = ws->d_len = ws->d_seg = nullptr;
    ws->d_partials = nullptr;
    ws->part_slots = 0;
    ws->tiles_cap = ws->seg_cap = 0;
    HIP_TRY(hipMalloc(&ws->d_count, sizeof(int) * CNT_STRIDE * (size_t)T));
    HIP_TRY(hipMalloc(&ws->d_len, sizeof(int) * (size_t)T));
    HIP_TRY(hipMalloc(&ws->d_seg, sizeof(int) * (size_t)T * C));
    HIP_TRY(hipMemset(ws->d_count, 0, sizeof(int) * CNT_STRIDE * (size_t)T));
    HIP_TRY(hipMemset(ws->d_len, 0, sizeof(int) * (size_t)T));
    // the clears run on the null stream, which orders nothing against a caller's non-blocking
    // stream: finish them before anything can use the arrays
    HIP_TRY(hipDeviceSynchronize());
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
int wanted_cap(const gi2d_workspace *ws, long long seen)
{
    long long c = 2 * seen;
    if (c < ws->min_seg_cap) c = ws->min_seg_cap;
    if (c > (long long)INT_MAX / 4) c = INT_MAX / 4;
    return round_up((int)c, WAVE);
}

int zero_counters(gi2d_workspace *ws, hipStream_t st)
{
    const int n = ws->tiles_cap * CNT_STRIDE;
    if (n == 0) return GI2D_OK;
    zero_ints_kernel<<<cdiv(n, 256), 256, 0, st>>>(ws->d_count, n);
    return launch_status();
}

// Sorts and finalises raw lists (when no forward rasterisation does it).
int finalize_lists(gi2d_workspace *ws, hipStream_t st)
{
    if (!ws->lists_raw) return GI2D_OK;
    const int T = ws_tiles(ws);
    if (T > 0) {
        finalize_tiles_kernel<<<T, WAVE, 0, st>>>(ws->d_count, ws->d_len, ws->d_seg, ws->seg_cap,
                                                  ws->d_status, ws->generation,
                                                  ws_tiles_per_frame(ws));
        int rc = launch_status();
        if (rc != GI2D_OK) return rc;
    }
    ws->lists_raw = false;
    ws->counters_dirty = false;
    return GI2D_OK;
}

// True when the next binning of this shape must learn its list lengths synchronously.
static bool shape_is_new(const gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int r0, int r1,
                         int frames = 1)
{
    return ws->seg_cap == 0 || ws->generation == 0 || N != ws->n_binned ||
           tiles_x != ws->tiles_x || tiles_y != ws->tiles_y || r0 != ws->bin_row0 ||
           r1 != ws->bin_row1 || frames != ws->frames;
}

// The per-Gaussian record buffer for `count` Gaussians; the device is current.
int ensure_records(gi2d_workspace *ws, int64_t count)
{
    if (ws->rec_cap >= count) return GI2D_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (ws->d_rec) (void)hipFree(ws->d_rec);
    ws->d_rec = nullptr;
    ws->rec_cap = 0;
    ws->rec_valid = false;
    HIP_TRY(hipMalloc(&ws->d_rec, sizeof(float) * REC_FLOATS_HOST * (size_t)count));
    ws->rec_cap = count;
    return GI2D_OK;
}

// Counters and buckets of `cells` cells (of all frames of a batch) of N entries each; the
// device is current.
int ensure_cells(gi2d_workspace *ws, int64_t cells, int N)
{
    if (cells < 1) return GI2D_E_INVALID;
    if (ws->cell_count_cap < cells) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->d_cell_count) (void)hipFree(ws->d_cell_count);
        ws->d_cell_count = nullptr;
        ws->cell_count_cap = 0;
        const size_t bytes = sizeof(int) * (size_t)cells * CNT_STRIDE;
        HIP_TRY(hipMalloc(&ws->d_cell_count, bytes));
        HIP_TRY(hipMemset(ws->d_cell_count, 0, bytes));
        HIP_TRY(hipDeviceSynchronize());  // (null-stream clear, see ensure_lists)
        ws->cell_count_cap = cells;
    }
    const int64_t want = cells * N;
    if (ws->cell_bucket_cap >= want && !ws->fail_side_allocs) return GI2D_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (ws->d_cell_bucket) (void)hipFree(ws->d_cell_bucket);
    ws->d_cell_bucket = nullptr;
    ws->cell_bucket_cap = 0;
    if (ws->fail_side_allocs) return (int)hipErrorOutOfMemory;  // (testing aid: as if no memory)
    HIP_TRY(hipMalloc(&ws->d_cell_bucket, 2 * sizeof(int) * (size_t)want));
    ws->cell_bucket_cap = want;
    return GI2D_OK;
}

// One packed tile box per Gaussian for `count` Gaussians; the device is current.
int ensure_boxes(gi2d_workspace *ws, int64_t count)
{
    if (ws->box_cap >= count && !ws->fail_side_allocs) return GI2D_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (ws->d_box) (void)hipFree(ws->d_box);
    ws->d_box = nullptr;
    ws->box_cap = 0;
    if (ws->fail_side_allocs) return (int)hipErrorOutOfMemory;  // (testing aid: as if no memory)
    HIP_TRY(hipMalloc(&ws->d_box, sizeof(int) * (size_t)count));
    ws->box_cap = count;
    return GI2D_OK;
}

// Prepares ws for a projection that also fills the tile lists.  Leaves ws->prefilled false when
// the shape is new (the lists' capacity is unknown): gi2d_bin_tiles then bins synchronously.
int begin_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int frames, hipStream_t st)
{
    const int r0 = min(ws->row0, tiles_y), r1 = min(ws->row1, tiles_y);
    if (frames < 1 || (int64_t)(r1 - r0) * tiles_x * frames >= (int64_t)INT_MAX / 2)
        return GI2D_E_INVALID;
    if (frames > 1 && (r0 != 0 || r1 != tiles_y)) return GI2D_E_INVALID;  // a batch of whole frames
    const int T = (r1 - r0) * tiles_x * frames;
    ws->prefilled = false;
    ws->rec_valid = false;
    ws->chain_pending = false;  // a new projection replaces whatever a chained step prepared
    if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1, frames)) return GI2D_OK;
    int rc = ensure_lists(ws, max(1, T), wanted_cap(ws, ws->h_status[0]));
    if (rc != GI2D_OK) return rc;
    if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1, frames)) return GI2D_OK;  // reallocated
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
    ws->pre_frames = frames;
    return GI2D_OK;
}

// The fitting tail wants to fill the NEXT iteration's lists while the current ones are still
// being read.  Possible when the current lists are finalised and valid, the shape is unchanged
// and no capacity growth is due (a growth reallocates, which would lose the current lists: the
// next projection does it instead).  Makes sure the second list set exists; the device is current.
bool begin_chain_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int frames)
{
    const int r0 = min(ws->row0, tiles_y), r1 = min(ws->row1, tiles_y);
    if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1, frames)) return false;
    if (ws->counters_dirty || ws->lists_raw || !ws->lists_valid || ws->seg_cap == 0) return false;
    if (wanted_cap(ws, ws->h_status[0]) > ws->seg_cap) return false;
    if (!ws->alt_count) {
        const size_t T = (size_t)ws->tiles_cap, C = (size_t)ws->seg_cap;
        bool ok = hipMalloc(&ws->alt_count, sizeof(int) * CNT_STRIDE * T) == hipSuccess &&
                  hipMalloc(&ws->alt_len, sizeof(int) * T) == hipSuccess &&
                  hipMalloc(&ws->alt_seg, sizeof(int) * T * C) == hipSuccess &&
                  hipMemset(ws->alt_count, 0, sizeof(int) * CNT_STRIDE * T) == hipSuccess &&
                  hipMemset(ws->alt_len, 0, sizeof(int) * T) == hipSuccess;
        ok = ok && hipDeviceSynchronize() == hipSuccess;  // the clears are done before any use
        if (!ok) {
            (void)hipGetLastError();
            free_alt_lists(ws);
            return false;
        }
    }
    return true;
}

// The chained fill has been launched into the second set: it becomes the current one, in the
// state a projection with list filling leaves behind (gi2d_bin_tiles(COUNTED) picks it up).
void commit_chain_prefill(gi2d_workspace *ws, int N, int tiles_x, int tiles_y, int kind,
                          const void *xys, int frames)
{
    int *t;
    t = ws->d_count, ws->d_count = ws->alt_count, ws->alt_count = t;
    t = ws->d_len, ws->d_len = ws->alt_len, ws->alt_len = t;
    t = ws->d_seg, ws->d_seg = ws->alt_seg, ws->alt_seg = t;
    ws->counters_dirty = true;
    ws->lists_raw = false;
    ws->lists_valid = false;
    ws->prefilled = true;
    ws->pre_n = N;
    ws->pre_tx = tiles_x;
    ws->pre_ty = tiles_y;
    ws->pre_r0 = min(ws->row0, tiles_y);
    ws->pre_r1 = min(ws->row1, tiles_y);
    ws->pre_frames = frames;
    ws->chain_pending = true;
    ws->chain_kind = kind;
    ws->chain_xys = xys;
}

extern "C" {

int gi2d_version(void) { return GI2D_VERSION; }

#ifndef GI2D_BUILD_ID
#define GI2D_BUILD_ID "unknown"
#endif
const char *gi2d_build_id(void) { return GI2D_BUILD_ID; }

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
    if (e == hipSuccess) e = hipDeviceSynchronize();  // (null-stream clear, see ensure_lists)
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
    free_alt_lists(ws);
    if (ws->d_gslots) (void)hipFree(ws->d_gslots);
    if (ws->d_rec) (void)hipFree(ws->d_rec);
    if (ws->d_box) (void)hipFree(ws->d_box);
    if (ws->d_cell_count) (void)hipFree(ws->d_cell_count);
    if (ws->d_cell_bucket) (void)hipFree(ws->d_cell_bucket);
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
        case GI2D_OPT_DROP_PENDING:
            ws->chain_pending = false;
            return GI2D_OK;
        case GI2D_OPT_PROJECT_AGG:
            if (value < 0 || value > 4) return GI2D_E_INVALID;
            ws->project_agg = value;
            return GI2D_OK;
        case GI2D_OPT_RECORDS:
            ws->no_records = value ? 0 : 1;
            ws->rec_valid = false;
            return GI2D_OK;
        case GI2D_OPT_BWD_SERIAL:
            if (value < 0) return GI2D_E_INVALID;
            ws->bwd_serial = value;
            return GI2D_OK;
        case GI2D_OPT_FAIL_SIDE_ALLOCS:
            ws->fail_side_allocs = value != 0;
            return GI2D_OK;
        default: return GI2D_E_INVALID;
    }
}

int gi2d_workspace_set_probe(gi2d_workspace *ws, int kernel, void *begin_event, void *end_event)
{
    if (!ws || kernel < 0 || kernel > GI2D_PROBE_FIT_TAIL) return GI2D_E_INVALID;
    const bool on = kernel != 0 && begin_event && end_event;
    ws->probe_kernel = on ? kernel : 0;
    ws->probe_begin = on ? (hipEvent_t)begin_event : nullptr;
    ws->probe_end = on ? (hipEvent_t)end_event : nullptr;
    return GI2D_OK;
}

int gi2d_workspace_set_rows(gi2d_workspace *ws, int row_begin, int row_end)
{
    if (!ws || row_begin < 0 || row_end < row_begin) return GI2D_E_INVALID;
    ws->row0 = row_begin;
    ws->row1 = row_end;
    return GI2D_OK;
}

int gi2d_workspace_info(gi2d_workspace *ws, int64_t info[10])
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
    info[8] = ws->frames;
    info[9] = ws->rec_valid ? 1 : 0;
    return GI2D_OK;
}

// Fills the lists from xys / radii with the standalone kernel.
static int launch_bin(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                      int tiles_y, int r0, int r1, int frames, hipStream_t st)
{
    int rc = GI2D_OK;
    if (ws->counters_dirty) {
        rc = zero_counters(ws, st);
        if (rc != GI2D_OK) return rc;
    }
    ws->counters_dirty = true;
    bin_tiles_kernel<<<dim3(max(1, cdiv(N, PG_THREADS)), frames), PG_THREADS, 0, st>>>(
        xys, radii, N, tiles_x, tiles_y, r0, r1, ws->d_count, ws->d_seg, ws->seg_cap,
        ws->d_status, ws->h_status_dev, ws->generation, (r1 - r0) * tiles_x);
    return launch_status();
}

int gi2d_bin_tiles(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                   int tiles_y, int flags, void *stream)
{
    return bin_tiles_impl(ws, xys, radii, N, tiles_x, tiles_y, flags, stream, 1);
}

}  // extern "C"

// gi2d_bin_tiles for `frames` frames of N Gaussians each (xys / radii: [frames][N][...]).
int bin_tiles_impl(gi2d_workspace *ws, const float *xys, const int *radii, int N, int tiles_x,
                   int tiles_y, int flags, void *stream, int frames)
{
    if (!ws || N < 0 || tiles_x < 1 || tiles_y < 1 || frames < 1) return GI2D_E_INVALID;
    if ((int64_t)tiles_x * tiles_y * frames >= (int64_t)INT_MAX / 2) return GI2D_E_INVALID;
    if (N > 0 && (!xys || !radii)) return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    if (!guard.ok) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int r0 = min(ws->row0, tiles_y), r1 = min(ws->row1, tiles_y);
    if (frames > 1 && (r0 != 0 || r1 != tiles_y)) return GI2D_E_INVALID;
    const int T = (r1 - r0) * tiles_x * frames;
    const bool nogrow = (flags & GI2D_BIN_NOGROW) != 0;
    bool sync = (flags & GI2D_BIN_SYNC) != 0;
    bool prev_overflow = false;

    // an earlier asynchronous frame whose lists did not fit: tell the caller (once)
    if (ws->h_status[2] > ws->overflow_ack) {
        ws->overflow_ack = ws->h_status[2];
        prev_overflow = true;
    }

    bool prefilled = ws->prefilled && ws->pre_n == N && ws->pre_tx == tiles_x &&
                     ws->pre_ty == tiles_y && ws->pre_r0 == r0 && ws->pre_r1 == r1 &&
                     ws->pre_frames == frames;
    ws->prefilled = false;
    ws->chain_pending = false;
    if (!(flags & GI2D_BIN_COUNTED)) prefilled = false;
    // records stay valid only for lists the same projection filled
    if (!prefilled || ws->rec_xys != (const void *)xys) ws->rec_valid = false;

    if (!prefilled) {
        // Size the lists.  Steady state needs no synchronisation: the capacity keeps a 2x
        // margin over the largest per-tile count seen from completed frames.  A new problem
        // shape says nothing about its counts: learn them once, synchronously.
        if (shape_is_new(ws, N, tiles_x, tiles_y, r0, r1, frames) && !nogrow) sync = true;
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
    ws->frames = frames;
    ws->bin_row0 = r0;
    ws->bin_row1 = r1;
    ws->lists_valid = false;
    if (!prefilled) {
        int rc = launch_bin(ws, xys, radii, N, tiles_x, tiles_y, r0, r1, frames, st);
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
            rc = launch_bin(ws, xys, radii, N, tiles_x, tiles_y, r0, r1, frames, st);
            if (rc != GI2D_OK) return rc;
            ws->lists_raw = true;
            ws->lists_valid = true;
        }
    }
    return prev_overflow ? GI2D_E_PREV_OVERFLOW : GI2D_OK;
}

extern "C" {

int gi2d_copy_bins(gi2d_workspace *ws, int *tile_offsets, int *ids, int64_t ids_capacity,
                   void *stream)
{
    if (!ws || !tile_offsets) return GI2D_E_INVALID;
    if (ws->generation == 0 || ws->tiles_x == 0 || !ws->lists_valid) return GI2D_E_STATE;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = finalize_lists(ws, st);
    if (rc != GI2D_OK) return rc;
    const int T = ws_tiles(ws);
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
        // (always: the copies above read h_off and the temporaries until they complete)
        const hipError_t e_sync = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e_sync;
        if (d_off) (void)hipFree(d_off);
        if (d_ids) (void)hipFree(d_ids);
        if (e != hipSuccess) rc = (int)e;
    }
    free(h_len);
    free(h_off);
    return rc;
}

}  // extern "C"
