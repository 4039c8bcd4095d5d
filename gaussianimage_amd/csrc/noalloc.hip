// noalloc.hip -- the frame call for callers that cannot allocate device memory (code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc): the tile lists live in
// a file-scope device array instead of a workspace's hipMalloc'd buffers.
#include <mutex>

#include "host.h"

// Layout of the array (bytes): per-tile counters (128 bytes per tile, CNT_AT), list lengths, status
// words, then the id segments of all tiles.
#define NA_COUNT_BYTES ((size_t)GI2D_NOALLOC_MAX_TILES * CNT_STRIDE * sizeof(int))
#define NA_LEN_BYTES ((size_t)GI2D_NOALLOC_MAX_TILES * sizeof(int))
#define NA_STATUS_BYTES 256
#define NA_SEG_BYTES ((size_t)GI2D_NOALLOC_MAX_LIST_INTS * sizeof(int))
#define NA_TOTAL_BYTES (NA_COUNT_BYTES + NA_LEN_BYTES + NA_STATUS_BYTES + NA_SEG_BYTES)
static_assert(NA_COUNT_BYTES % 256 == 0 && NA_LEN_BYTES % 256 == 0, "sections stay 256-aligned");

// One copy per GPU the library runs on, about 12.4 MiB, held until the process ends.  In host
// code this name is only a handle for hipGetSymbolAddress: the array's memory is reached through
// the per-device pointer taken at set-up, never through the name.
static __device__ __align__(256) unsigned char scratch[NA_TOTAL_BYTES];

#define NA_MAX_DEVICES 64

namespace {
struct NoAllocDevice {
    bool ready = false;
    unsigned char *base = nullptr;  // this device's copy of `scratch`
    long long h_status[4] = {0, 0, 0, 0};  // host words the capacity policy reads: stay zero
    gi2d_workspace ws;              // host-side bookkeeping only; owns no memory
};
NoAllocDevice g_dev[NA_MAX_DEVICES];
std::mutex g_setup_lock;
std::mutex g_call_lock;  // serialises the host side of gi2d_render_frame_noalloc
}  // namespace

// Once per device, with that device current: the address of its copy of the array.
static int noalloc_setup(int dev, NoAllocDevice **out)
{
    if (dev < 0 || dev >= NA_MAX_DEVICES) return GI2D_E_INVALID;
    NoAllocDevice *d = &g_dev[dev];
    std::lock_guard<std::mutex> hold(g_setup_lock);
    if (!d->ready) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr) return GI2D_E_STATE;
        d->base = static_cast<unsigned char *>(ptr);
        gi2d_workspace &w = d->ws;
        w.device = dev;
        w.d_count = reinterpret_cast<int *>(d->base);
        w.d_len = reinterpret_cast<int *>(d->base + NA_COUNT_BYTES);
        unsigned char *status = d->base + NA_COUNT_BYTES + NA_LEN_BYTES;
        w.d_status = reinterpret_cast<DevStatus *>(status);
        // the kernels publish their status words to "host" memory: here a corner of the array
        // that no one reads (the list capacity is the caller's, there is no growth policy)
        w.h_status_dev = reinterpret_cast<long long *>(status + 64);
        w.h_status = d->h_status;
        w.d_seg = reinterpret_cast<int *>(status + NA_STATUS_BYTES);
        d->ready = true;
    }
    *out = d;
    return GI2D_OK;
}

extern "C" {

int gi2d_render_frame_noalloc(const gi2d_frame_args *a, int per_tile_capacity)
{
    if (!a || a->ws != nullptr || per_tile_capacity < 0) return GI2D_E_INVALID;
    if ((a->kind != 0 && a->kind != 1) || a->N < 0 || a->H < 1 || a->W < 1) return GI2D_E_INVALID;
    const int tiles_x = cdiv(a->W, GI2D_BLOCK), tiles_y = cdiv(a->H, GI2D_BLOCK);
    const long long T = (long long)tiles_x * tiles_y;
    if (T > GI2D_NOALLOC_MAX_TILES) return GI2D_E_CAPACITY;
    // capacity per tile: whole waves; 0 = as much as the array gives this frame (at most 4096)
    long long C = per_tile_capacity > 0
                      ? round_up(per_tile_capacity, WAVE)
                      : (long long)GI2D_NOALLOC_MAX_LIST_INTS / T / WAVE * WAVE;
    if (per_tile_capacity == 0 && C > 4096) C = 4096;
    if (C < WAVE || T * C > GI2D_NOALLOC_MAX_LIST_INTS) return GI2D_E_CAPACITY;
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    NoAllocDevice *d = nullptr;
    int rc = noalloc_setup(dev, &d);
    if (rc != GI2D_OK) return rc;
    // One call at a time per process edits the bookkeeping and enqueues the launches (host work
    // only, a few microseconds): two host threads cannot interleave their edits of one device's
    // struct.  The lock says nothing about the GPU: launches of two calls that OVERLAP on one
    // device still share the array, and that remains the caller's to avoid.
    std::lock_guard<std::mutex> hold(g_call_lock);
    // The bookkeeping of a workspace whose lists already have this frame's shape and capacity:
    // nothing is learned synchronously, nothing grows, and the counters -- which may hold what
    // an interrupted earlier call left -- are cleared by a fill kernel before the projection.
    gi2d_workspace &w = d->ws;
    w.tiles_cap = (int)T;
    w.seg_cap = (int)C;
    w.min_seg_cap = WAVE;
    w.no_records = 1;  // (per-Gaussian records would need a buffer that grows with N)
    w.fixed_memory = true;
    if (w.generation == 0) w.generation = 1;
    w.n_binned = a->N;
    w.tiles_x = tiles_x;
    w.tiles_y = tiles_y;
    w.row0 = 0;
    w.row1 = INT_MAX;
    w.bin_row0 = 0;
    w.bin_row1 = tiles_y;
    w.counters_dirty = true;
    w.prefilled = false;
    w.chain_pending = false;
    gi2d_frame_args args = *a;
    args.ws = &w;
    rc = gi2d_render_frame(&args);
    // (a list longer than the capacity marks its tile's pixels NaN, as in the asynchronous
    // workspace path; there is no later call that could report it)
    return rc == GI2D_E_PREV_OVERFLOW ? GI2D_OK : rc;
}

}  // extern "C"
