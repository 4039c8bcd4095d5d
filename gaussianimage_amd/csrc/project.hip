// libgi2d -- MI355X (gfx950, wave64) kernels and C ABI of the accumulated-summation 2D Gaussian
// rasterizer.  This file: projection (Cholesky / rotation-scale / coded attributes), forward and
// backward.  lists.hip: workspace and binning; raster_fwd.hip / raster_bwd.hip: rasterize_sum;
// fit.hip: Adan and the fused fitting step.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off, one object per .hip file, linked
// into one shared library (see build.py).
// -ffp-contract=off is part of the contract: the projection's integer outputs (radii, tile
// boxes) must not depend on the compiler's choice of fused multiply-adds.  Where a fused
// multiply-add is wanted in a hot loop it is written explicitly with __builtin_fmaf.
//
// Frame pipeline (forward): TWO launches in steady state.
//   1. project_*_fwd_kernel  -- per Gaussian: conic, radius, tile box, the API outputs, and for
//      every tile of the box one returning atomic on the tile's counter followed by a store of
//      the Gaussian id into the tile's segment (fixed capacity C ids per tile).
//   2. rasterize_sum_fwd_kernel<FINALIZE> -- per tile: sorts its segment (ids arrive in atomic
//      order; ascending order makes runs reproducible and lets the backward search lists),
//      publishes the list length, returns the counter to zero, then rasterises.
// Every launch costs ~2 us plus ~1-1.5 us per dependent memory round trip on this part, and a
// 768x512 frame with 30k Gaussians is only a few us of arithmetic, so the design minimises
// launches and dependent rounds rather than bytes.
#include "host.h"
#include "projection.h"

// ------------------------------------------------------------------------------------------
// projection and tile binning (one lane per Gaussian)
// ------------------------------------------------------------------------------------------

// What every projection kernel takes besides its inputs.  A launch covers `gridDim.y` frames of
// N Gaussians each (a batch of independent frames of one shape): every per-Gaussian tensor is
// [frames][N][...], every frame has its own run of `tiles_per_frame` tile lists, and ids in the
// lists are frame-local.
struct ProjArgs {
    int N, H, W, tiles_x, tiles_y;
    float center_bias;
    ProjOut out;
    TileLists lists;
    ListCull cull;
    RecordSink rec;
    const float *rec_colors;  // [N,3] colours for the records (the decode kernel has its own)
    const DevStatus *status;
    volatile long long *h_status;
    long long generation;
    int tiles_per_frame;
};

// Moves every per-Gaussian and per-tile pointer to frame f.
__device__ __forceinline__ void to_frame(ProjArgs &a, int f)
{
    const size_t n = (size_t)f * a.N;
    a.out.xys += 2 * n;
    a.out.depths += n;
    a.out.radii += n;
    a.out.conics += 3 * n;
    a.out.num_tiles_hit += n;
    if (a.lists.tile_count != nullptr) {
        const size_t t = (size_t)f * a.tiles_per_frame;
        a.lists.tile_count += t * CNT_STRIDE;
        a.lists.seg += t * a.lists.C;
    }
    if (a.cull.opacity != nullptr) a.cull.opacity += n;
    if (a.rec.rec != nullptr) {
        a.rec.rec += n * REC_FLOATS;
        a.rec.opacity += n;
        if (a.rec_colors != nullptr) a.rec_colors += 3 * n;
    }
}

// RAW = the model's raw parameters come in and its activations are applied here:
// means = tanh(_xyz), L = _cholesky + (0.5, 0, 0.5) (gaussianimage_cholesky.py:62,74,42).
template <bool RAW>
__global__ __launch_bounds__(256) void project_cholesky_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ L, ProjArgs a)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (g == 0 && f == 0 && a.lists.tile_count != nullptr)
        publish_status(a.status, a.h_status, a.generation);
    if (g >= a.N) return;
    to_frame(a, f);
    means += (size_t)f * 2 * a.N;
    L += (size_t)f * 3 * a.N;
    float l11 = L[3 * g], l21 = L[3 * g + 1], l22 = L[3 * g + 2];
    float mx = means[2 * g], my = means[2 * g + 1];
    if (RAW) {
        mx = tanhf(mx);
        my = tanhf(my);
        l11 = l11 + 0.5f;
        l22 = l22 + 0.5f;
    }
    float sxx = l11 * l11;
    float sxy = l11 * l21;
    float syy = l21 * l21 + l22 * l22;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (a.rec.rec != nullptr) {
        cr = a.rec_colors[3 * g];
        cg = a.rec_colors[3 * g + 1];
        cb = a.rec_colors[3 * g + 2];
    }
    project_tail(g, mx, my, sxx, sxy, syy, a.H, a.W, a.tiles_x, a.tiles_y, a.center_bias, a.out,
                 a.lists, a.cull, a.rec, cr, cg, cb);
}

// RAW: means = tanh(_xyz), scales = |_scaling + 0.5|, rotation = sigmoid(_rotation) * 2 * pi
// (gaussianimage_rs.py:55-64).
template <bool RAW>
__global__ __launch_bounds__(256) void project_rs_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ scales,
    const float *__restrict__ rot, ProjArgs a)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (g == 0 && f == 0 && a.lists.tile_count != nullptr)
        publish_status(a.status, a.h_status, a.generation);
    if (g >= a.N) return;
    to_frame(a, f);
    means += (size_t)f * 2 * a.N;
    scales += (size_t)f * 2 * a.N;
    rot += (size_t)f * a.N;
    float theta = rot[g];
    float s1 = scales[2 * g], s2 = scales[2 * g + 1];
    float mx = means[2 * g], my = means[2 * g + 1];
    if (RAW) {
        mx = tanhf(mx);
        my = tanhf(my);
        s1 = fabsf(s1 + 0.5f);
        s2 = fabsf(s2 + 0.5f);
        theta = (1.0f / (1.0f + expf(-theta))) * 2.0f * 3.14159274f;
    }
    float s, c;
    gi2d_sincos_f32(theta, &s, &c);
    float q1 = s1 * s1, q2 = s2 * s2;
    float sxx = c * c * q1 + s * s * q2;
    float sxy = c * s * (q1 - q2);
    float syy = s * s * q1 + c * c * q2;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (a.rec.rec != nullptr) {
        cr = a.rec_colors[3 * g];
        cg = a.rec_colors[3 * g + 1];
        cb = a.rec_colors[3 * g + 2];
    }
    project_tail(g, mx, my, sxx, sxy, syy, a.H, a.W, a.tiles_x, a.tiles_y, a.center_bias, a.out,
                 a.lists, a.cull, a.rec, cr, cg, cb);
}

// ------------------------------------------------------------------------------------------
// The projection of a batch with the list filling aggregated in LDS.  Returning atomics on the
// tiles' counters are what the list filling of the kernels above costs (one per (Gaussian,
// tile) pair, and the part sustains only about 2e10 of them per second).  Here a workgroup of
// 1024 lanes takes a run of `chunk` Gaussians of one frame (up to AGG_ROUNDS per lane) and
//   1. projects them, counting their tiles in a histogram in LDS (LDS atomics cost next to
//      nothing);
//   2. reserves, with ONE returning atomic per tile it touched, that many places in the tile's
//      segment;
//   3. while those atomics are in flight, walks its Gaussians' boxes again and gathers the
//      pairs per tile in an LDS stage (a prefix sum of the histogram gives each tile's run, a
//      returning LDS atomic the place inside it);
//   4. copies the stage out, consecutive lanes to consecutive places of a tile's run (runs of
//      ~8 ids instead of 64 scattered 4-byte stores per wave).
// (Measured as a launch of its own at 24 x 30 000 Gaussians, steps 2-4 done one after the other
// with scattered stores took 21.7 us: 4.5 us loads and histogram, 6.8 us reservations, 4 us
// placement, 6.5 us stores.)
// A frame of N Gaussians costs (workgroups per frame) x (tiles touched) counter atomics instead
// of one per pair: 8 workgroups per 30 000-Gaussian frame -> 12 000 instead of 57 000.  Lists are
// the same sets as the direct kernels' (the order inside a list is restored by the sort that
// follows either way).  For frames of at most AGG_MAX_TILES tiles, each side < 256 tiles.
// ------------------------------------------------------------------------------------------
#define AGG_THREADS 1024
#define AGG_ROUNDS 8
#define AGG_MAX_TILES 4096
#define AGG_STAGE 16384  // (tile, Gaussian) pairs a workgroup can gather in LDS
static_assert(AGG_MAX_TILES == 4 * AGG_THREADS && AGG_MAX_TILES <= (1 << 12) &&
                  AGG_ROUNDS * AGG_THREADS <= (1 << 13),
              "the stage words and the four tiles per lane assume these sizes");

// RAW: the model's raw parameters, activated here (as in project_*_fwd_kernel<RAW>).
template <bool RS, bool RAW>
__global__ __launch_bounds__(AGG_THREADS) void project_agg_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ cov_a,
    const float *__restrict__ cov_b, ProjArgs a, int chunk)
{
    __shared__ __attribute__((aligned(16))) int hist[AGG_MAX_TILES];
    __shared__ int stage[AGG_STAGE];
    __shared__ int wave_sum[AGG_THREADS / 64];
    const int tid = threadIdx.x, f = blockIdx.y;
    if (tid == 0 && blockIdx.x == 0 && f == 0) publish_status(a.status, a.h_status, a.generation);
    to_frame(a, f);
    means += (size_t)f * 2 * a.N;
    cov_a += (size_t)f * (RS ? 2 : 3) * a.N;
    if (RS) cov_b += (size_t)f * a.N;
    *reinterpret_cast<int4 *>(hist + (AGG_MAX_TILES / AGG_THREADS) * tid) = make_int4(0, 0, 0, 0);
    __syncthreads();
    const int g0 = blockIdx.x * chunk, g1 = min(a.N, g0 + chunk);
    int boxes[AGG_ROUNDS];  // x0 | x1 << 8 | y0 << 16 | y1 << 24 of each of my Gaussians
#pragma unroll
    for (int r = 0; r < AGG_ROUNDS; ++r) {
        boxes[r] = 0;
        const int g = g0 + r * AGG_THREADS + tid;
        if (g < g1) {
            float mx = means[2 * g], my = means[2 * g + 1];
            if (RAW) {
                mx = tanhf(mx);
                my = tanhf(my);
            }
            float sxx, sxy, syy;
            if (RS) {
                float theta = cov_b[g];
                float s1 = cov_a[2 * g], s2 = cov_a[2 * g + 1];
                if (RAW) {
                    s1 = fabsf(s1 + 0.5f);
                    s2 = fabsf(s2 + 0.5f);
                    theta = (1.0f / (1.0f + expf(-theta))) * 2.0f * 3.14159274f;
                }
                float s, c;
                gi2d_sincos_f32(theta, &s, &c);
                const float q1 = s1 * s1, q2 = s2 * s2;
                sxx = c * c * q1 + s * s * q2;
                sxy = c * s * (q1 - q2);
                syy = s * s * q1 + c * c * q2;
            } else {
                float l11 = cov_a[3 * g], l21 = cov_a[3 * g + 1], l22 = cov_a[3 * g + 2];
                if (RAW) {
                    l11 = l11 + 0.5f;
                    l22 = l22 + 0.5f;
                }
                sxx = l11 * l11;
                sxy = l11 * l21;
                syy = l21 * l21 + l22 * l22;
            }
            float cr = 0.f, cg = 0.f, cb = 0.f;
            if (a.rec.rec != nullptr) {
                cr = a.rec_colors[3 * g];
                cg = a.rec_colors[3 * g + 1];
                cb = a.rec_colors[3 * g + 2];
            }
            TileBox box;
            const bool listed = project_body(g, mx, my, sxx, sxy, syy, a.H, a.W, a.tiles_x,
                                             a.tiles_y, a.center_bias, a.out, true, a.cull, a.rec,
                                             cr, cg, cb, box);
            if (listed && box.x1 > box.x0 && box.y1 > box.y0) {
                boxes[r] = box.x0 | (box.x1 << 8) | (box.y0 << 16) | (box.y1 << 24);
                for (int ty = box.y0; ty < box.y1; ++ty)
                    for (int tx = box.x0; tx < box.x1; ++tx)
                        atomicAdd(&hist[ty * a.tiles_x + tx], 1);
            }
        }
    }
    __syncthreads();
    // Each lane looks after PER_LANE consecutive tiles.  The reservations are put in flight
    // first and used last: a returning atomic's round trip is the longest wait of this kernel
    // (and every workgroup of the launch reaches this point at about the same time), so the
    // ids are meanwhile gathered per tile in LDS, where only the order inside the workgroup's
    // own run of each tile has to be settled.
    constexpr int PER_LANE = AGG_MAX_TILES / AGG_THREADS;
    const int t4 = PER_LANE * tid;
    int cnt[PER_LANE], base[PER_LANE], off[PER_LANE];
    {
        const int4 c = *reinterpret_cast<const int4 *>(hist + t4);
        cnt[0] = c.x;
        cnt[1] = c.y;
        cnt[2] = c.z;
        cnt[3] = c.w;
    }
    // (the counters of tiles 2j and 2j+1 are neighbouring words: one 64-bit atomic reserves for
    // both; a pair this workgroup never touched keeps zeros nobody reads)
    static_assert(PER_LANE % 2 == 0, "tiles are reserved in pairs");
#pragma unroll
    for (int k = 0; k < PER_LANE; k += 2) {
        base[k] = base[k + 1] = 0;
        if ((cnt[k] | cnt[k + 1]) != 0) {
            unsigned long long *pair =
                reinterpret_cast<unsigned long long *>(a.lists.tile_count + CNT_AT(t4 + k));
            const unsigned long long old = atomicAdd(
                pair, (unsigned long long)(unsigned)cnt[k] |
                          ((unsigned long long)(unsigned)cnt[k + 1] << 32));
            base[k] = (int)(unsigned)old;
            base[k + 1] = (int)(unsigned)(old >> 32);
        }
    }
    // exclusive prefix of the counts over the tiles: where each tile's run starts in the stage
    const int lane = tid & 63, wave = tid >> 6;
    const int mine = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
    const int incl = wave_inclusive_scan(mine);
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();  // (also: every lane has taken its counts out of the histogram)
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < AGG_THREADS / 64; ++w) {
        const int v = wave_sum[w];
        before += w < wave ? v : 0;
        total += v;
    }
    off[0] = before + incl - mine;
#pragma unroll
    for (int k = 1; k < PER_LANE; ++k) off[k] = off[k - 1] + cnt[k - 1];
    const int C = a.lists.C;
    if (total <= AGG_STAGE) {
        // the histogram words become the tiles' cursors into the stage
        *reinterpret_cast<int4 *>(hist + t4) = make_int4(off[0], off[1], off[2], off[3]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < AGG_ROUNDS; ++r) {
            const int b = boxes[r];
            if (b == 0) continue;
            const int x0 = b & 255, x1 = (b >> 8) & 255, y0 = (b >> 16) & 255, y1 = (b >> 24) & 255;
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const int t = ty * a.tiles_x + tx;
                    // tile (12 bits) and the Gaussian's place in this workgroup's run (13 bits)
                    stage[atomicAdd(&hist[t], 1)] = (t << 13) | (r * AGG_THREADS + tid);
                }
        }
        __syncthreads();
        // stage index -> place in the tile's segment: + (first reserved place - run start)
        *reinterpret_cast<int4 *>(hist + t4) =
            make_int4(base[0] - off[0], base[1] - off[1], base[2] - off[2], base[3] - off[3]);
        __syncthreads();
        // consecutive lanes write consecutive places of a tile's run
        for (int i = tid; i < total; i += AGG_THREADS) {
            const int w = stage[i];
            const int t = w >> 13;
            const int pos = hist[t] + i;
            // (ids beyond a segment's capacity are dropped; the counter has counted them)
            if (pos < C) a.lists.seg[(size_t)t * C + pos] = g0 + (w & 8191);
        }
        return;
    }
    // more pairs than the stage holds (large footprints): every pair's place comes from a
    // returning LDS atomic on its tile's first reserved place, and the ids go out scattered
    *reinterpret_cast<int4 *>(hist + t4) = make_int4(base[0], base[1], base[2], base[3]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < AGG_ROUNDS; ++r) {
        const int b = boxes[r];
        if (b == 0) continue;
        const int g = g0 + r * AGG_THREADS + tid;
        const int x0 = b & 255, x1 = (b >> 8) & 255, y0 = (b >> 16) & 255, y1 = (b >> 24) & 255;
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                const int t = ty * a.tiles_x + tx;
                const int pos = atomicAdd(&hist[t], 1);
                if (pos < C) a.lists.seg[(size_t)t * C + pos] = g;
            }
    }
}

// ------------------------------------------------------------------------------------------
// The projection of a batch with the list filling by BANDS of tile rows: no counter atomics at
// all.  Two launches:
//   1. project_boxes_fwd_kernel -- one lane per Gaussian, the whole chip: outputs, record, and
//      the Gaussian's (culled) tile box packed into one word;
//   2. fill_bands_kernel -- one workgroup of 1024 lanes per (band of tile rows, frame) OWNS the
//      band's tiles: it reads every box of its frame (4 bytes per Gaussian, from L2 after the
//      first band), keeps the rows of its band, and hands out list places with returning LDS
//      atomics on counters that live in LDS; at the end it stores the counters.  No other
//      workgroup touches those tiles, so the global counters see one plain store per tile.
// A workgroup scans N boxes whatever the band holds, so this form is for frames of moderate N
// split into enough bands to fill the chip: batches.  Lists are the same sets as the other
// forms' (the rasteriser's sort restores the order).  Each side < 256 tiles.
// ------------------------------------------------------------------------------------------
#define BAND_MAX_ROUNDS 64  // N <= BAND_MAX_ROUNDS * AGG_THREADS per frame

template <bool RS>
__global__ __launch_bounds__(PG_THREADS) void project_boxes_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ cov_a,
    const float *__restrict__ cov_b, ProjArgs a, int *__restrict__ boxes)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (g == 0 && f == 0) publish_status(a.status, a.h_status, a.generation);
    if (g >= a.N) return;
    to_frame(a, f);
    means += (size_t)f * 2 * a.N;
    cov_a += (size_t)f * (RS ? 2 : 3) * a.N;
    if (RS) cov_b += (size_t)f * a.N;
    const float mx = means[2 * g], my = means[2 * g + 1];
    float sxx, sxy, syy;
    if (RS) {
        float s, c;
        gi2d_sincos_f32(cov_b[g], &s, &c);
        const float s1 = cov_a[2 * g], s2 = cov_a[2 * g + 1];
        const float q1 = s1 * s1, q2 = s2 * s2;
        sxx = c * c * q1 + s * s * q2;
        sxy = c * s * (q1 - q2);
        syy = s * s * q1 + c * c * q2;
    } else {
        const float l11 = cov_a[3 * g], l21 = cov_a[3 * g + 1], l22 = cov_a[3 * g + 2];
        sxx = l11 * l11;
        sxy = l11 * l21;
        syy = l21 * l21 + l22 * l22;
    }
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (a.rec.rec != nullptr) {
        cr = a.rec_colors[3 * g];
        cg = a.rec_colors[3 * g + 1];
        cb = a.rec_colors[3 * g + 2];
    }
    TileBox box;
    const bool listed = project_body(g, mx, my, sxx, sxy, syy, a.H, a.W, a.tiles_x, a.tiles_y,
                                     a.center_bias, a.out, true, a.cull, a.rec, cr, cg, cb, box);
    int packed = 0;  // (an empty box: x0 = x1)
    if (listed && box.x1 > box.x0 && box.y1 > box.y0)
        packed = box.x0 | (box.x1 << 8) | (box.y0 << 16) | (box.y1 << 24);
    boxes[(size_t)f * a.N + g] = packed;
}

__global__ __launch_bounds__(AGG_THREADS) void fill_bands_kernel(
    const int *__restrict__ boxes, int N, int tiles_x, int tiles_y, int rows_per_band,
    int tiles_per_frame, int *__restrict__ tile_count, int *__restrict__ seg, int C)
{
    __shared__ int cnt[AGG_MAX_TILES];
    const int tid = threadIdx.x, f = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_band, r1 = min(tiles_y, r0 + rows_per_band);
    const int band_tiles = (r1 - r0) * tiles_x;
    boxes += (size_t)f * N;
    const size_t t0 = (size_t)f * tiles_per_frame + (size_t)r0 * tiles_x;  // the band's first tile
    for (int t = tid; t < band_tiles; t += AGG_THREADS) cnt[t] = 0;
    __syncthreads();
    // (a lane's loads are issued eight at a time: the scan is a chain of memory round trips
    // otherwise, one per 1024 Gaussians)
    constexpr int PF = 8;
    for (int gb = tid; gb < N; gb += PF * AGG_THREADS) {
        int bx[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int g = gb + k * AGG_THREADS;
            bx[k] = g < N ? boxes[g] : 0;
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int b = bx[k], g = gb + k * AGG_THREADS;
            const int y0 = max((b >> 16) & 255, r0), y1 = min((b >> 24) & 255, r1);
            if (y0 >= y1) continue;  // (an empty box has y0 = y1 = 0 too)
            const int x0 = b & 255, x1 = (b >> 8) & 255;
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const int t = (ty - r0) * tiles_x + tx;
                    const int pos = atomicAdd(&cnt[t], 1);
                    // (ids beyond a segment's capacity are dropped; the counter has counted them)
                    if (pos < C) seg[(t0 + t) * C + pos] = g;
                }
        }
    }
    __syncthreads();
    // (the counters were zero: the rasteriser returns them to zero after reading them)
    for (int t = tid; t < band_tiles; t += AGG_THREADS)
        if (cnt[t] > 0)
            tile_count[(size_t)f * tiles_per_frame * CNT_STRIDE + CNT_AT(r0 * tiles_x + t)] = cnt[t];
}

// ------------------------------------------------------------------------------------------
// Large frames: list filling through CELLS of 2^k x 2^k tiles, two launches, no atomic on any
// tile counter.  A frame of tens of thousands of tiles and Gaussians in no spatial order defeats
// the aggregation above (a workgroup's Gaussians touch almost as many tiles as they have pairs),
// and one returning atomic per pair is what the part sustains only 2e10 of per second (25 us
// for the 497 000 pairs of a 3840x2160 frame of 200 000 Gaussians).
//   1. project_cells_fwd_kernel -- one lane per Gaussian, workgroups of 1024: outputs, record,
//      culled tile box; a histogram of the CELLS the boxes touch in LDS, one returning atomic per
//      (workgroup, cell) on the cell's counter, and -- gathered per cell in LDS while those are
//      in flight -- (id, packed box) into the cell's bucket in runs.  A bucket holds N entries:
//      a Gaussian enters a cell once, so no bucket can overflow.
//   2. fill_cells_kernel -- one workgroup per cell OWNS the cell's tiles: it reads its bucket
//      once (a few entries per lane, kept in registers), counts the tiles of the boxes (clipped
//      to the cell) in LDS, stores the counters plainly, gathers the ids per tile in LDS and
//      copies them out in runs.
// Both passes are chains of a few dependent phases, so small cells (8 x 8 tiles: a bucket is
// about one entry per lane) serve pass 2 best; pass 1 is not bound by its reservations
// (measured at 3840x2160 / 200 000 Gaussians: 13.6 us with 510 cells, 12.7 us with 135, while
// pass 2 took 6.3 us against 10.7 us with four workgroups per 16 x 16 cell; 27.6 us for the
// one-atomic-per-pair projection).
// Lists are the same sets as the other forms' (the rasteriser's sort restores the order).  Each
// side < 256 tiles.
// ------------------------------------------------------------------------------------------
#define CELL_THREADS 1024
#define CELL_MAX GI2D_CELLS_MAX  // cells of a frame, at most (one lane, one histogram word each)
static_assert(CELL_MAX == CELL_THREADS, "one lane per cell");
#define CELL_MAX_LOG2 5      // tiles per cell side, at most 32 (a cell's tiles: one lane each)
#define CELL_STAGE1 4096     // (cell, Gaussian) pairs a workgroup of pass 1 gathers in LDS
#define CELL_STAGE2 8192     // (tile, Gaussian) pairs a workgroup of pass 2 gathers in LDS
#define CELL_KEEP 2          // bucket entries per lane that pass 2 keeps in registers

struct CellGrid {
    int log2_side;      // a cell is 2^log2_side tiles wide and high
    int cells_x, cells;
    int row0, row1;     // the tile rows the lists cover
    int *count;         // [frames][cells][CNT_STRIDE], zero between calls
    int2 *bucket;       // [frames][cells][cap]: (Gaussian id, x0 | x1 << 8 | y0 << 16 | y1 << 24)
    int cap;
};

// exclusive prefix of v over the 1024 lanes of the workgroup; the sum of all in `total`
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int *wave_sum, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_inclusive_scan(v);
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int o = wave_sum[w];
        before += w < wave ? o : 0;
        total += o;
    }
    return before + incl - v;
}

template <bool RS, bool RAW>
__global__ __launch_bounds__(CELL_THREADS) void project_cells_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ cov_a,
    const float *__restrict__ cov_b, ProjArgs a, CellGrid cg)
{
    __shared__ int chist[CELL_MAX];
    __shared__ int2 stage[CELL_STAGE1];
    __shared__ int wave_sum[CELL_THREADS / 64];
    const int tid = threadIdx.x, f = blockIdx.y;
    const int g = blockIdx.x * CELL_THREADS + tid;
    if (g == 0 && f == 0) publish_status(a.status, a.h_status, a.generation);
    to_frame(a, f);
    means += (size_t)f * 2 * a.N;
    cov_a += (size_t)f * (RS ? 2 : 3) * a.N;
    if (RS) cov_b += (size_t)f * a.N;
    cg.count += (size_t)f * cg.cells * CNT_STRIDE;
    cg.bucket += (size_t)f * cg.cells * cg.cap;
    chist[tid] = 0;
    __syncthreads();
    int packed = 0, cx0 = 0, cx1 = 0, cy0 = 0, cy1 = 0;  // the box, and the cells it touches
    if (g < a.N) {
        float mx = means[2 * g], my = means[2 * g + 1];
        if (RAW) {
            mx = tanhf(mx);
            my = tanhf(my);
        }
        float sxx, sxy, syy;
        if (RS) {
            float theta = cov_b[g];
            float s1 = cov_a[2 * g], s2 = cov_a[2 * g + 1];
            if (RAW) {
                s1 = fabsf(s1 + 0.5f);
                s2 = fabsf(s2 + 0.5f);
                theta = (1.0f / (1.0f + expf(-theta))) * 2.0f * 3.14159274f;
            }
            float sn, cs;
            gi2d_sincos_f32(theta, &sn, &cs);
            const float q1 = s1 * s1, q2 = s2 * s2;
            sxx = cs * cs * q1 + sn * sn * q2;
            sxy = cs * sn * (q1 - q2);
            syy = sn * sn * q1 + cs * cs * q2;
        } else {
            float l11 = cov_a[3 * g], l21 = cov_a[3 * g + 1], l22 = cov_a[3 * g + 2];
            if (RAW) {
                l11 = l11 + 0.5f;
                l22 = l22 + 0.5f;
            }
            sxx = l11 * l11;
            sxy = l11 * l21;
            syy = l21 * l21 + l22 * l22;
        }
        float cr = 0.f, cgr = 0.f, cb = 0.f;
        if (a.rec.rec != nullptr) {
            cr = a.rec_colors[3 * g];
            cgr = a.rec_colors[3 * g + 1];
            cb = a.rec_colors[3 * g + 2];
        }
        TileBox box;
        const bool listed = project_body(g, mx, my, sxx, sxy, syy, a.H, a.W, a.tiles_x, a.tiles_y,
                                         a.center_bias, a.out, true, a.cull, a.rec, cr, cgr, cb,
                                         box, cg.row0, cg.row1);
        const int y0 = max(box.y0, cg.row0), y1 = min(box.y1, cg.row1);
        if (listed && box.x1 > box.x0 && y1 > y0) {
            packed = box.x0 | (box.x1 << 8) | (y0 << 16) | (y1 << 24);
            cx0 = box.x0 >> cg.log2_side;
            cx1 = ((box.x1 - 1) >> cg.log2_side) + 1;
            cy0 = (y0 - cg.row0) >> cg.log2_side;
            cy1 = ((y1 - 1 - cg.row0) >> cg.log2_side) + 1;
            for (int cy = cy0; cy < cy1; ++cy)
                for (int cx = cx0; cx < cx1; ++cx) atomicAdd(&chist[cy * cg.cells_x + cx], 1);
        }
    }
    __syncthreads();
    // lane = cell: reserve (in flight until the copy-out), and find each cell's run in the stage
    const int cnt = chist[tid];  // (zero past the frame's cells)
    // (the counters of cells 2j and 2j+1 are neighbouring words, CNT_AT: the even lane reserves
    // for both with one 64-bit atomic and hands the odd lane its half)
    int base = 0;
    {
        const int cnt_other = __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, false);  // lane ^ 1
        unsigned long long old = 0ull;
        if ((tid & 1) == 0 && (cnt | cnt_other) != 0)
            old = atomicAdd(reinterpret_cast<unsigned long long *>(cg.count + CNT_AT(tid)),
                            (unsigned long long)(unsigned)cnt |
                                ((unsigned long long)(unsigned)cnt_other << 32));
        const int lo = (int)(unsigned)old, hi = (int)(unsigned)(old >> 32);
        const int hi_from_even = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
        base = (tid & 1) ? hi_from_even : lo;
    }
    int total;
    const int off = block_exclusive_scan_1024(cnt, wave_sum, total);
    if (total <= CELL_STAGE1) {
        chist[tid] = off;  // the cell's cursor into the stage
        __syncthreads();
        for (int cy = cy0; cy < cy1; ++cy)
            for (int cx = cx0; cx < cx1; ++cx) {
                const int c = cy * cg.cells_x + cx;
                stage[atomicAdd(&chist[c], 1)] = make_int2((c << 10) | tid, packed);
            }
        __syncthreads();
        chist[tid] = base - off;  // stage index -> place in the cell's bucket
        __syncthreads();
        for (int i = tid; i < total; i += CELL_THREADS) {
            const int2 e = stage[i];
            const int c = e.x >> 10;
            const int pos = chist[c] + i;
            if (pos < cg.cap)
                cg.bucket[(size_t)c * cg.cap + pos] =
                    make_int2(blockIdx.x * CELL_THREADS + (e.x & 1023), e.y);
        }
        return;
    }
    // boxes over many cells (very large footprints): places from returning LDS atomics on the
    // first reserved place, entries stored one by one
    chist[tid] = base;
    __syncthreads();
    for (int cy = cy0; cy < cy1; ++cy)
        for (int cx = cx0; cx < cx1; ++cx) {
            const int c = cy * cg.cells_x + cx;
            const int pos = atomicAdd(&chist[c], 1);
            if (pos < cg.cap) cg.bucket[(size_t)c * cg.cap + pos] = make_int2(g, packed);
        }
}

__global__ __launch_bounds__(CELL_THREADS) void fill_cells_kernel(
    CellGrid cg, int tiles_x, int *__restrict__ tile_count, int *__restrict__ seg, int C)
{
    __shared__ int hist[CELL_THREADS];     // per tile of the cell: count, then cursor
    __shared__ int start[CELL_THREADS];    // where the tile's run starts in the stage
    __shared__ int tile_of[CELL_THREADS];  // the tile's index in the lists
    __shared__ int stage[CELL_STAGE2];
    __shared__ unsigned short owner[CELL_STAGE2];
    __shared__ int wave_sum[CELL_THREADS / 64];
    const int tid = threadIdx.x, c = blockIdx.x, f = blockIdx.y;
    {
        const size_t T = (size_t)tiles_x * (cg.row1 - cg.row0);  // tile lists per frame
        tile_count += f * T * CNT_STRIDE;
        seg += f * T * C;
        cg.count += (size_t)f * cg.cells * CNT_STRIDE;
        cg.bucket += (size_t)f * cg.cells * cg.cap;
    }
    const int n = min(cg.count[CNT_AT(c)], cg.cap);
    hist[tid] = 0;
    __syncthreads();  // (every lane has read the counter)
    if (tid == 0) cg.count[CNT_AT(c)] = 0;  // for the next frame
    const int side = 1 << cg.log2_side;
    const int tx0 = (c % cg.cells_x) << cg.log2_side, tx1 = min(tiles_x, tx0 + side);
    const int ty0 = cg.row0 + ((c / cg.cells_x) << cg.log2_side), ty1 = min(cg.row1, ty0 + side);
    const int w = tx1 - tx0, nt = w * (ty1 - ty0);
    const int2 *__restrict__ src = cg.bucket + (size_t)c * cg.cap;
    // the first CELL_KEEP x 1024 entries stay in registers for the second walk; a longer
    // bucket (Gaussians crowded into one cell) is read again
    int2 kept[CELL_KEEP];
#pragma unroll
    for (int r = 0; r < CELL_KEEP; ++r) {
        const int i = r * CELL_THREADS + tid;
        kept[r] = i < n ? src[i] : make_int2(0, 0);  // (an empty box: x0 = x1)
    }
    auto walk = [&](auto &&visit) {
        auto entry = [&](const int2 e) {
            const int b = e.y;
            const int x0 = max(b & 255, tx0), x1 = min((b >> 8) & 255, tx1);
            const int y0 = max((b >> 16) & 255, ty0), y1 = min((b >> 24) & 255, ty1);
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) visit((ty - ty0) * w + tx - tx0, e.x);
        };
#pragma unroll
        for (int r = 0; r < CELL_KEEP; ++r) entry(kept[r]);
        for (int i = CELL_KEEP * CELL_THREADS + tid; i < n; i += CELL_THREADS) entry(src[i]);
    };
    walk([&](int lt, int) { atomicAdd(&hist[lt], 1); });
    __syncthreads();
    const int cnt = hist[tid];  // (zero past the cell's tiles)
    int t = 0;
    if (tid < nt) {
        const int ly = tid / w;
        t = (ty0 - cg.row0 + ly) * tiles_x + tx0 + (tid - ly * w);
        // (the counters were zero: the rasteriser returns them to zero after reading them)
        if (cnt > 0) tile_count[CNT_AT(t)] = cnt;
    }
    tile_of[tid] = t;
    int total;
    const int off = block_exclusive_scan_1024(cnt, wave_sum, total);
    start[tid] = off;
    const bool staged = total <= CELL_STAGE2;
    hist[tid] = staged ? off : 0;
    __syncthreads();
    walk([&](int lt, int id) {
        const int p = atomicAdd(&hist[lt], 1);
        if (staged) {
            stage[p] = id;
            owner[p] = (unsigned short)lt;
        } else if (p < C) {
            // (ids beyond a segment's capacity are dropped; the counter has counted them)
            seg[(size_t)tile_of[lt] * C + p] = id;
        }
    });
    if (!staged) return;
    __syncthreads();
    // consecutive lanes write consecutive places of a tile's run
    for (int i = tid; i < total; i += CELL_THREADS) {
        const int lt = owner[i];
        const int pos = i - start[lt];
        if (pos < C) seg[(size_t)tile_of[lt] * C + pos] = stage[i];
    }
}

// ------------------------------------------------------------------------------------------
// Codec decode (decompress_wo_ec, gaussianimage_cholesky.py:161-166 / gaussianimage_rs.py:132-139):
//   means    = tanh(float(xyz_half))
//   cholesky = code * scale + beta, then + (0.5, 0, 0.5)       (quantize.py:86-87)
//   scaling  = |code * scale + beta + 0.5|,  rotation = code * scale + beta
//   colors   = sum over the residual-VQ layers q of codebook[q][index[g][q]]   (quantize.py:146-150)
// Each product, sum and bound is rounded on its own, in the reference's order.  An index
// outside the codebook is clamped into it (a corrupt stream gives wrong colours, not a fault).
// ------------------------------------------------------------------------------------------
struct DecodedGaussian {
    float mx, my, a0, a1, a2, b0, col[3];
};

template <bool RS>
__device__ __forceinline__ DecodedGaussian decode_gaussian(const gi2d_codec &cd, int g)
{
    DecodedGaussian d;
    const __half2 h = reinterpret_cast<const __half2 *>(cd.xyz_half)[g];
    d.mx = tanhf(__low2float(h));
    d.my = tanhf(__high2float(h));
    if (RS) {
        d.a0 = fabsf((cd.cov_a_code[2 * g] * cd.cov_a_scale[0] + cd.cov_a_beta[0]) + 0.5f);
        d.a1 = fabsf((cd.cov_a_code[2 * g + 1] * cd.cov_a_scale[1] + cd.cov_a_beta[1]) + 0.5f);
        d.a2 = 0.f;
        d.b0 = cd.cov_b_code[g] * cd.cov_b_scale[0] + cd.cov_b_beta[0];
    } else {
        d.a0 = (cd.cov_a_code[3 * g] * cd.cov_a_scale[0] + cd.cov_a_beta[0]) + 0.5f;
        d.a1 = (cd.cov_a_code[3 * g + 1] * cd.cov_a_scale[1] + cd.cov_a_beta[1]) + 0.0f;
        d.a2 = (cd.cov_a_code[3 * g + 2] * cd.cov_a_scale[2] + cd.cov_a_beta[2]) + 0.5f;
        d.b0 = 0.f;
    }
    d.col[0] = d.col[1] = d.col[2] = 0.f;
    for (int q = 0; q < cd.num_quantizers; ++q) {
        int k = cd.color_index[(size_t)g * cd.num_quantizers + q];
        k = min(max(k, 0), cd.codebook_size - 1);
        const float *e = cd.codebooks + ((size_t)q * cd.codebook_size + k) * 3;
        d.col[0] += e[0];
        d.col[1] += e[1];
        d.col[2] += e[2];
    }
    return d;
}

// The decoded operator inputs as tensors (what the reference's torch expressions produce).
template <bool RS>
__global__ __launch_bounds__(256) void decode_params_kernel(gi2d_codec cd, int N,
                                                            float *__restrict__ means,
                                                            float *__restrict__ cov_a,
                                                            float *__restrict__ cov_b,
                                                            float *__restrict__ colors)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const DecodedGaussian d = decode_gaussian<RS>(cd, g);
    means[2 * g] = d.mx;
    means[2 * g + 1] = d.my;
    if (RS) {
        cov_a[2 * g] = d.a0;
        cov_a[2 * g + 1] = d.a1;
        cov_b[g] = d.b0;
    } else {
        cov_a[3 * g] = d.a0;
        cov_a[3 * g + 1] = d.a1;
        cov_a[3 * g + 2] = d.a2;
    }
    colors[3 * g] = d.col[0];
    colors[3 * g + 1] = d.col[1];
    colors[3 * g + 2] = d.col[2];
}

// Decode fused into the projection: the frame's first launch reads the coded attributes, leaves
// the decoded colours for the rasteriser and fills the tile lists.  (One frame per launch.)
template <bool RS>
__global__ __launch_bounds__(256) void project_decode_fwd_kernel(gi2d_codec cd,
                                                                 float *__restrict__ colors,
                                                                 ProjArgs a)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && a.lists.tile_count != nullptr)
        publish_status(a.status, a.h_status, a.generation);
    if (g >= a.N) return;
    const DecodedGaussian d = decode_gaussian<RS>(cd, g);
    colors[3 * g] = d.col[0];
    colors[3 * g + 1] = d.col[1];
    colors[3 * g + 2] = d.col[2];
    float sxx, sxy, syy;
    if (RS) {
        float s, c;
        gi2d_sincos_f32(d.b0, &s, &c);
        const float q1 = d.a0 * d.a0, q2 = d.a1 * d.a1;
        sxx = c * c * q1 + s * s * q2;
        sxy = c * s * (q1 - q2);
        syy = s * s * q1 + c * c * q2;
    } else {
        sxx = d.a0 * d.a0;
        sxy = d.a0 * d.a1;
        syy = d.a1 * d.a1 + d.a2 * d.a2;
    }
    project_tail(g, d.mx, d.my, sxx, sxy, syy, a.H, a.W, a.tiles_x, a.tiles_y, a.center_bias,
                 a.out, a.lists, a.cull, a.rec, d.col[0], d.col[1], d.col[2]);
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
    gi2d_sincos_f32(rot[g], &s, &c);
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
// host side
// ------------------------------------------------------------------------------------------

static int check_proj_args(const void *a, const void *b, int N, int H, int W, int tiles_x,
                           int tiles_y, const gi2d_conv *conv, const void *o0, const void *o1,
                           const void *o2, const void *o3, const void *o4)
{
    if (N < 0 || H < 1 || W < 1 || !conv) return GI2D_E_INVALID;
    if (tiles_x != cdiv(W, GI2D_BLOCK) || tiles_y != cdiv(H, GI2D_BLOCK)) return GI2D_E_INVALID;
    if (N > 0 && (!a || !b || !o0 || !o1 || !o2 || !o3 || !o4)) return GI2D_E_INVALID;
    return GI2D_OK;
}

// kind 0 = Cholesky (cov_a = L[N,3]), 1 = rotation-scale (cov_a = scales[N,2], cov_b = rotation).
// raw = the model's raw parameters are given and activated inside the kernel.
static int check_codec(int kind, const gi2d_codec *cd, int N)
{
    if ((kind != 0 && kind != 1) || !cd || N < 0) return GI2D_E_INVALID;
    if (cd->num_quantizers < 1 || cd->num_quantizers > GI2D_MAX_VQ_LAYERS || cd->codebook_size < 1)
        return GI2D_E_INVALID;
    if (N == 0) return GI2D_OK;
    if (!cd->xyz_half || !cd->cov_a_code || !cd->cov_a_scale || !cd->cov_a_beta ||
        !cd->color_index || !cd->codebooks)
        return GI2D_E_INVALID;
    if (kind == 1 && (!cd->cov_b_code || !cd->cov_b_scale || !cd->cov_b_beta))
        return GI2D_E_INVALID;
    return GI2D_OK;
}

int project_impl(int kind, bool raw, const float *means, const float *cov_a, const float *cov_b,
                 int N, int H, int W, int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                 float *depths, int *radii, float *conics, int *num_tiles_hit, gi2d_workspace *ws,
                 void *stream, const ProjExtras &ex)
{
    const gi2d_codec *codec = ex.codec;
    const int frames = ex.frames;
    if (frames < 1 || (codec && frames != 1)) return GI2D_E_INVALID;
    if (codec) {
        int rcc = check_codec(kind, codec, N);
        if (rcc != GI2D_OK) return rcc;
        if (N > 0 && !ex.colors_out) return GI2D_E_INVALID;
        means = reinterpret_cast<const float *>(codec->xyz_half);
        cov_a = codec->cov_a_code;
        cov_b = codec->cov_b_code;
    }
    int rc = check_proj_args(means, cov_a, N, H, W, tiles_x, tiles_y, conv, xys, depths, radii,
                             conics, num_tiles_hit);
    if (rc != GI2D_OK) return rc;
    if (kind == 1 && N > 0 && !cov_b) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    bool fill = false;
    DeviceGuard guard(ws ? ws->device : -1);
    if (ws) {
        if (!guard.ok) return GI2D_E_INVALID;
        rc = begin_prefill(ws, N, tiles_x, tiles_y, frames, st);
        if (rc != GI2D_OK) return rc;
        fill = ws->prefilled;
    } else if (frames != 1) {
        return GI2D_E_INVALID;
    }
    if (N == 0) return GI2D_OK;
    ProjArgs a = ProjArgs();
    a.N = N;
    a.H = H;
    a.W = W;
    a.tiles_x = tiles_x;
    a.tiles_y = tiles_y;
    a.center_bias = conv->center_bias;
    a.out = ProjOut{xys, depths, radii, conics, num_tiles_hit};
    // list culling needs the opacities the rasteriser will use, and can be switched off with the
    // rasteriser's own culling (GI2D_OPT_CULL)
    a.cull = ListCull{nullptr, conv->alpha_min, conv->pixel_offset};
    if (fill) {
        a.lists = TileLists{ws->d_count, ws->d_seg, ws->seg_cap, ws->pre_r0, ws->pre_r1};
        a.status = ws->d_status;
        a.h_status = ws->h_status_dev;
        a.generation = ws->generation;
        a.tiles_per_frame = (ws->pre_r1 - ws->pre_r0) * tiles_x;
        if (ws->cull) a.cull.opacity = ex.opacity;
        // the rasteriser's records: only with lists the rasteriser will actually use, and only
        // when colours and opacities are known now
        const float *rec_colors = codec ? ex.colors_out : ex.colors;
        if (ex.opacity && rec_colors && !ws->no_records) {
            rc = ensure_records(ws, (int64_t)frames * N);
            if (rc != GI2D_OK) return rc;
            a.rec = RecordSink{ws->d_rec, ex.opacity, conv->alpha_cap};
            a.rec_colors = codec ? nullptr : ex.colors;
        }
    }
    const dim3 grid(cdiv(N, PG_THREADS), frames), block(PG_THREADS);
    // The LDS-aggregated list filling pays when its few large workgroups still fill the chip:
    // batches.  One frame keeps the direct kernels (GI2D_OPT_PROJECT_AGG forces either form).
    const bool whole = fill && !codec && tiles_x < 256 && tiles_y < 256 && ws->pre_r0 == 0 &&
                       ws->pre_r1 == tiles_y;
    bool agg = whole && a.tiles_per_frame <= AGG_MAX_TILES;
    // bands: enough of them to fill the chip, each within the LDS counters; every workgroup
    // scans all N boxes of its frame, which bounds N
    int bands = whole ? max(1, min(tiles_y, 256 / frames)) : 0;
    int rows_per_band = whole ? cdiv(tiles_y, bands) : 0;
    bool by_bands = whole && !raw && rows_per_band * tiles_x <= AGG_MAX_TILES &&
                    N <= BAND_MAX_ROUNDS * AGG_THREADS;
    // (on request only: at 8 x 30 000 Gaussians its two launches take 11 + 13 us against 19.8 us
    // for the aggregated form -- the scan of all boxes by every band is a chain of memory rounds)
    if (by_bands) by_bands = ws->project_agg == 3;
    if (by_bands && ensure_boxes(ws, (int64_t)frames * N) != GI2D_OK) {
        (void)hipGetLastError();
        by_bands = false;
    }
    // cells: single large frames (buckets of N entries per cell: the cell side grows until the
    // cells fit their limits; a workspace that must not allocate never takes this form)
    CellGrid cells = CellGrid();
    bool by_cells = fill && !codec &&
                    cells_form(ws, N, tiles_x, tiles_y, ws->pre_r0, ws->pre_r1, frames);
    if (by_cells) {
        const int rows = ws->pre_r1 - ws->pre_r0;
        // cells of 8 x 8 tiles (4 x 4 for small frames, so that there are several), larger
        // while there would be too many of them or their buckets would not fit
        int k = (int64_t)frames * cdiv(tiles_x, 8) * cdiv(rows, 8) >= 128 ? 3 : 2;
        for (; k <= CELL_MAX_LOG2; ++k) {
            cells.cells_x = cdiv(tiles_x, 1 << k);
            cells.cells = cells.cells_x * cdiv(rows, 1 << k);
            if (cells.cells <= CELL_MAX &&
                (int64_t)frames * cells.cells * N * (int64_t)sizeof(int2) <= GI2D_CELLS_MAX_BYTES)
                break;
        }
        by_cells = k <= CELL_MAX_LOG2;
        if (by_cells && ensure_cells(ws, (int64_t)frames * cells.cells, N) != GI2D_OK) {
            (void)hipGetLastError();  // (no memory for the buckets: another form, not a failure)
            by_cells = false;
        }
        cells.log2_side = k;
        cells.row0 = ws->pre_r0;
        cells.row1 = ws->pre_r1;
        cells.count = ws->d_cell_count;
        cells.bucket = reinterpret_cast<int2 *>(ws->d_cell_bucket);
        cells.cap = N;
    }
    if (by_cells) by_bands = false;
    if (agg) agg = !by_bands && !by_cells &&
                   (ws->project_agg == 1 || (ws->project_agg == 0 && frames >= 4));
    {
    Probe probe(fill ? ws : nullptr, GI2D_PROBE_PROJECT, st);
    if (by_cells) {
        const dim3 cgrid(cdiv(N, CELL_THREADS), frames);
#define GI2D_CELLS(RS, RAW) \
    project_cells_fwd_kernel<RS, RAW><<<cgrid, CELL_THREADS, 0, st>>>(means, cov_a, cov_b, a, cells)
        if (kind == 0 && raw) GI2D_CELLS(false, true);
        else if (kind == 0) GI2D_CELLS(false, false);
        else if (raw) GI2D_CELLS(true, true);
        else GI2D_CELLS(true, false);
#undef GI2D_CELLS
        fill_cells_kernel<<<dim3(cells.cells, frames), CELL_THREADS, 0, st>>>(
            cells, tiles_x, a.lists.tile_count, a.lists.seg, a.lists.C);
    } else if (by_bands) {
        if (kind == 0)
            project_boxes_fwd_kernel<false><<<grid, block, 0, st>>>(means, cov_a, cov_b, a, ws->d_box);
        else
            project_boxes_fwd_kernel<true><<<grid, block, 0, st>>>(means, cov_a, cov_b, a, ws->d_box);
        bands = cdiv(tiles_y, rows_per_band);
        fill_bands_kernel<<<dim3(bands, frames), AGG_THREADS, 0, st>>>(
            ws->d_box, N, tiles_x, tiles_y, rows_per_band, a.tiles_per_frame, a.lists.tile_count,
            a.lists.seg, a.lists.C);
    } else if (agg) {
        // workgroups per frame: enough to spread over the chip, few enough to share counters
        // (measured at 8 x 30 000 Gaussians on 768x512: 8 / 16 / 30 workgroups per frame give
        // 21.3 / 19.9 / 20.5 us against 35 us for the direct kernel; at 24 x 30 000 with the
        // counters reserved in pairs, 128 / 144 / 168 / 192 / 216 workgroups in all give
        // 28.6 / 26.9 / 26.2 / 27.5 / 32.5 us)
        int per_frame = max(1, min(cdiv(N, AGG_THREADS), 168 / frames));
        per_frame = max(per_frame, cdiv(N, AGG_THREADS * AGG_ROUNDS));
        const int chunk = cdiv(N, per_frame);
        const dim3 agrid(cdiv(N, chunk), frames);
#define GI2D_AGG(RS, RAW) \
    project_agg_fwd_kernel<RS, RAW><<<agrid, AGG_THREADS, 0, st>>>(means, cov_a, cov_b, a, chunk)
        if (kind == 0 && raw) GI2D_AGG(false, true);
        else if (kind == 0) GI2D_AGG(false, false);
        else if (raw) GI2D_AGG(true, true);
        else GI2D_AGG(true, false);
#undef GI2D_AGG
    } else if (codec) {
        if (kind == 0)
            project_decode_fwd_kernel<false><<<grid, block, 0, st>>>(*codec, ex.colors_out, a);
        else
            project_decode_fwd_kernel<true><<<grid, block, 0, st>>>(*codec, ex.colors_out, a);
    } else if (kind == 0) {
        if (raw)
            project_cholesky_fwd_kernel<true><<<grid, block, 0, st>>>(means, cov_a, a);
        else
            project_cholesky_fwd_kernel<false><<<grid, block, 0, st>>>(means, cov_a, a);
    } else {
        if (raw)
            project_rs_fwd_kernel<true><<<grid, block, 0, st>>>(means, cov_a, cov_b, a);
        else
            project_rs_fwd_kernel<false><<<grid, block, 0, st>>>(means, cov_a, cov_b, a);
    }
    }
    rc = launch_status();
    if (rc == GI2D_OK && a.rec.rec != nullptr) {
        ws->rec_valid = true;
        ws->rec_xys = xys;
        ws->rec_colors = codec ? ex.colors_out : ex.colors;
        ws->rec_opacity = ex.opacity;
    }
    return rc;
}


extern "C" {

int gi2d_project_cholesky_fwd(const float *means2d, const float *L_elements, int N, int H, int W,
                              int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                              float *depths, int *radii, float *conics, int *num_tiles_hit,
                              gi2d_workspace *ws, void *stream)
{
    return project_impl(0, false, means2d, L_elements, nullptr, N, H, W, tiles_x, tiles_y, conv,
                        xys, depths, radii, conics, num_tiles_hit, ws, stream);
}

int gi2d_project_rs_fwd(const float *means2d, const float *scales, const float *rotation, int N,
                        int H, int W, int tiles_x, int tiles_y, const gi2d_conv *conv, float *xys,
                        float *depths, int *radii, float *conics, int *num_tiles_hit,
                        gi2d_workspace *ws, void *stream)
{
    return project_impl(1, false, means2d, scales, rotation, N, H, W, tiles_x, tiles_y, conv, xys,
                        depths, radii, conics, num_tiles_hit, ws, stream);
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

int gi2d_decode_params(int kind, const gi2d_codec *codec, int N, float *means2d, float *cov_a,
                       float *cov_b, float *colors, void *stream)
{
    int rc = check_codec(kind, codec, N);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    if (!means2d || !cov_a || !colors || (kind == 1 && !cov_b)) return GI2D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0)
        decode_params_kernel<false><<<cdiv(N, 256), 256, 0, st>>>(*codec, N, means2d, cov_a, cov_b,
                                                                  colors);
    else
        decode_params_kernel<true><<<cdiv(N, 256), 256, 0, st>>>(*codec, N, means2d, cov_a, cov_b,
                                                                 colors);
    return launch_status();
}

}  // extern "C"
