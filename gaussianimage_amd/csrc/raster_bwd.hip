// raster_bwd.hip -- rasterize_gaussians_sum backward: per-tile partial sums, then their gather.
#include "raster_bwd_tile.h"

// ------------------------------------------------------------------------------------------
// rasterize_sum backward.
//
// Pass 1 (one 256-lane workgroup per tile).  Short lists: a group of 4, 8 or 16 lanes per entry
// walks the entry's support rectangle.  Long lists, per batch of up to 128 entries:
//   A. one lane per entry stages the entry -- attributes and the support rectangle (bounding
//      box of the ellipse alpha >= alpha_min) -- in LDS;
//   B. the entries' pixel counts are cut into PIECES of at most `chunk` pixels, chunk chosen so
//      that the batch makes about one piece per quad of lanes: the work of a big Gaussian is
//      spread over several quads, a small one costs its quad only its own few pixels;
//   C. a quad of lanes per piece walks the piece's pixels round-robin, nine partial sums in
//      registers, adds its four lanes with two DPP steps and leaves the sums in LDS;
//   D. one lane per entry adds the entry's pieces in order and writes the result to the
//      Gaussian's own slot (or, beyond its first GSLOTS tiles, to the tile's slab).
// No floating-point atomics: every sum has a fixed order, gradients are reproducible.
// Pass 2 (one lane per Gaussian): adds the Gaussian's partial sums in tile order; those of the
// first GSLOTS tiles of its box lie in the Gaussian's own slots, the rest are found by binary
// search (lists ascend in Gaussian id).
// ------------------------------------------------------------------------------------------

// (at least 7 waves per SIMD, i.e. at most 72 registers: the two paths with their plain and
// general pixel loops take 71; held to 64 for an eighth workgroup per CU they spill, which costs
// more than the workgroup gains -- measured at 768x512 / 30 000 and 1020x768 / 50 000)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void
rasterize_sum_bwd_kernel(
    const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0, int row1,
    int T, gi2d_conv cv, int cull, const float *__restrict__ v_out,
    float *__restrict__ partials, float *__restrict__ gslots, size_t gslot_n, int stamp)
{
    __shared__ float s_v[VTILE_FLOATS];               // the tile's v_out
    __shared__ float4 s_r0[BWD_BATCH];                // gx, gy, a, b
    __shared__ float4 s_r1[BWD_BATCH];                // c, opacity, red, green
    __shared__ float4 s_r2[BWD_BATCH];                // blue, id, radius, rectangle (4 x 8 bits)
    __shared__ int s_start[BWD_BATCH + 1];            // first piece of every entry
    __shared__ int s_w[4];
    __shared__ float s_piece[BWD_MAX_PIECES * PART_STRIDE];
    // entry (of the batch) a piece belongs to; the short-list path keeps the list's first 64
    // ids in the same words
    __shared__ __attribute__((aligned(16))) unsigned char s_owner[BWD_MAX_PIECES];
    static_assert(BWD_MAX_PIECES >= 64 * sizeof(int), "s_first does not fit");
    int *s_first = reinterpret_cast<int *>(s_owner);
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, T);
    const int *lst = seg + (size_t)t * C;
    // the first ids before the length is known (a segment always has C >= 64 words)
    const int gid_first = tid < min(C, BWD_BATCH) ? lst[tid] : 0;
    const int n_raw = tile_len[t];
    if (n_raw == 0 || n_raw > C) return;  // an overflowed tile is reported, not half-summed
    const size_t slot0 = (size_t)t * C;
    const int tx = t % tiles_x, ty = row0 + t / tiles_x;
    {
        const int j = tx * GI2D_BLOCK + (tid & 15), i = ty * GI2D_BLOCK + (tid >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < H && j < W) {
            const float *p = v_out + 3 * ((size_t)(i - row0 * GI2D_BLOCK) * W + j);
            v.x = p[0];
            v.y = p[1];
            v.z = p[2];
        }
        s_v[tid] = v.x;
        s_v[VTILE_PLANE + tid] = v.y;
        s_v[2 * VTILE_PLANE + tid] = v.z;
        if (tid < 64) s_first[tid] = gid_first;
    }
    __syncthreads();
    // Short lists: a fixed group of lanes per entry -- more lanes for fewer entries -- with next
    // to no set-up.  Long lists (the fitted regime: many large Gaussians per tile) spend their
    // time in the pixel loops, where the uneven supports of a wave's entries would leave most
    // lanes idle: there the work is cut into equal pieces first.
#define GI2D_BWD_TILE(LPE)                                                                      \
    bwd_tile<LPE>(lst, s_first, n_raw, slot0, tx, ty, s_v, xys, radii, conics, colors, opacity, \
                  H, W, tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n, stamp)
    if (n_raw <= 16) GI2D_BWD_TILE(16);
    else if (n_raw <= BWD_PIECES_MIN_LIST) GI2D_BWD_TILE(8);
    else
        bwd_tile_pieces(lst, gid_first, n_raw, slot0, tx, ty, s_v, s_r0, s_r1, s_r2, s_start, s_w,
                        s_piece, s_owner, xys, radii, conics, colors, opacity, H, W, tiles_x, tiles_y, row0,
                        cv, cull, partials, gslots, gslot_n, stamp);
#undef GI2D_BWD_TILE
}

// The same for sparse frames (many tiles with a handful of Gaussians each): ONE wave per tile,
// each entry gets 4 to 16 lanes; lists beyond 16 entries are cut into work pieces.  Per tile it
// spends about half the instructions of the 256-lane form, whose four waves all pay the
// per-entry set-up.
__global__ __launch_bounds__(64) void rasterize_sum_bwd_wave_kernel(
    const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ xys, const int *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacity, int H, int W, int tiles_x, int tiles_y, int row0, int row1,
    int T, gi2d_conv cv, int cull, const float *__restrict__ v_out,
    float *__restrict__ partials, float *__restrict__ gslots, size_t gslot_n, int stamp)
{
    __shared__ float s_v[VTILE_FLOATS];
    __shared__ int s_first[64];
    __shared__ float4 s_r0[32], s_r1[32], s_r2[32];  // the work-piece path's buffers, as above
    __shared__ int s_start[32 + 1];
    __shared__ float s_piece[BWD64_MAX_PIECES * PART_STRIDE];
    __shared__ unsigned char s_owner[BWD64_MAX_PIECES];
    const int tid = threadIdx.x;
    const int t = xcd_tile(blockIdx.x, T);
    const int *lst = seg + (size_t)t * C;
    const int gid_first = tid < C ? lst[tid] : 0;  // before the length is known
    const int n_raw = tile_len[t];
    if (n_raw == 0 || n_raw > C) return;  // an overflowed tile is reported, not half-summed
    const size_t slot0 = (size_t)t * C;
    const int tx = t % tiles_x, ty = row0 + t / tiles_x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int px = u * 64 + tid;
        const int j = tx * GI2D_BLOCK + (px & 15), i = ty * GI2D_BLOCK + (px >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < H && j < W) {
            const float *p = v_out + 3 * ((size_t)(i - row0 * GI2D_BLOCK) * W + j);
            v.x = p[0];
            v.y = p[1];
            v.z = p[2];
        }
        s_v[px] = v.x;
        s_v[VTILE_PLANE + px] = v.y;
        s_v[2 * VTILE_PLANE + px] = v.z;
    }
    s_first[tid] = gid_first;
    __syncthreads();
#define GI2D_BWD_TILE(LPE)                                                                       \
    bwd_tile<LPE, 64>(lst, s_first, n_raw, slot0, tx, ty, s_v, xys, radii, conics, colors,       \
                      opacity, H, W, tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n, \
                      stamp)
    if (n_raw <= 4) GI2D_BWD_TILE(16);
    else if (n_raw <= 8) GI2D_BWD_TILE(8);
    else if (n_raw <= BWD64_PIECES_MIN_LIST) GI2D_BWD_TILE(4);
    else
        bwd_tile_pieces<64>(lst, gid_first, n_raw, slot0, tx, ty, s_v, s_r0, s_r1, s_r2, s_start,
                            nullptr, s_piece, s_owner, xys, radii, conics, colors, opacity, H, W,
                            tiles_x, tiles_y, row0, cv, cull, partials, gslots, gslot_n, stamp);
#undef GI2D_BWD_TILE
}

__global__ __launch_bounds__(256) void gather_grads_kernel(
    const float *__restrict__ xys, const int *__restrict__ radii, int N, int tiles_x, int tiles_y,
    int row0, int row1, const int *__restrict__ tile_len, const int *__restrict__ seg, int C,
    const float *__restrict__ partials, const float *__restrict__ gslots, size_t gslot_n,
    int stamp,
    float *__restrict__ v_xy, float *__restrict__ v_conic, float *__restrict__ v_colors,
    float *__restrict__ v_opacity)
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

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

// The partial-sum buffers of the backward: the tiles' slab and the Gaussians' own slots.
int ensure_bwd_buffers(gi2d_workspace *ws, int N)
{
    const int64_t slots = (int64_t)ws->tiles_cap * ws->seg_cap;
    if (ws->part_slots < slots) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->d_partials) (void)hipFree(ws->d_partials);
        ws->d_partials = nullptr;
        ws->part_slots = 0;
        HIP_TRY(hipMalloc(&ws->d_partials, sizeof(float) * PART_STRIDE * (size_t)slots));
        ws->part_slots = slots;
    }
    if (ws->slot_cap < N) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->d_gslots) (void)hipFree(ws->d_gslots);
        ws->d_gslots = nullptr;
        ws->slot_cap = 0;
        const size_t bytes = sizeof(float) * GSLOTS * GSLOT_STRIDE * (size_t)N;
        HIP_TRY(hipMalloc(&ws->d_gslots, bytes));
        HIP_TRY(hipMemset(ws->d_gslots, 0, bytes));  // stamp 0 = never written (stamps are >= 1)
        HIP_TRY(hipDeviceSynchronize());  // (null-stream clear: done before the caller's stream uses it)
        ws->slot_cap = N;
    }
    return GI2D_OK;
}

__global__ __launch_bounds__(256) void zero_floats_kernel(float *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0.f;
}

// Every backward pass 1 gets a stamp of its own, which its partial sums carry in the Gaussians'
// slots and which pass 2 asks for: an entry whose support misses a tile writes nothing there,
// and whatever an EARLIER pass left in that slot -- of another frame, or of the same binning
// with other opacities or conics -- must not count.  Stamps run 1 .. INT_MAX; before they start
// over, the slots are cleared (stamp 0 = never written).  Call after ensure_bwd_buffers.
int next_bwd_stamp(gi2d_workspace *ws, hipStream_t st)
{
    ws->bwd_serial += 1;
    ws->bwd_stamp = (int)((ws->bwd_serial - 1) % (int64_t)INT_MAX) + 1;
    if (ws->bwd_stamp == 1 && ws->bwd_serial > 1 && ws->slot_cap > 0) {
        const size_t n = (size_t)GSLOTS * GSLOT_STRIDE * (size_t)ws->slot_cap;
        zero_floats_kernel<<<2048, 256, 0, st>>>(ws->d_gslots, n);
        return launch_status();
    }
    return GI2D_OK;
}

// Backward pass 1 (per-tile partial sums into the slab); shared by the operator backward and
// the fused fitting step.
int raster_bwd_pass1(gi2d_workspace *ws, const float *xys, const int *radii,
                            const float *conics, const float *colors, const float *opacity, int N,
                            int H, int W, const gi2d_conv *conv, const float *v_out,
                            hipStream_t st)
{
    int rc = finalize_lists(ws, st);
    if (rc != GI2D_OK) return rc;
    rc = ensure_bwd_buffers(ws, N);
    if (rc != GI2D_OK) return rc;
    rc = next_bwd_stamp(ws, st);
    if (rc != GI2D_OK) return rc;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (T > 0) {
#define GI2D_BWD_ARGS ws->d_len, ws->d_seg, ws->seg_cap, xys, radii, conics, colors, opacity, H, W, \
                      ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1, T, *conv, ws->cull,     \
                      v_out, ws->d_partials, ws->d_gslots, (size_t)ws->slot_cap,                   \
                      ws->bwd_stamp
        Probe probe(ws, GI2D_PROBE_RASTER_BWD, st);
        if (sparse_frame(ws, T))
            rasterize_sum_bwd_wave_kernel<<<T, 64, 0, st>>>(GI2D_BWD_ARGS);
        else
            rasterize_sum_bwd_kernel<<<T, 256, 0, st>>>(GI2D_BWD_ARGS);
#undef GI2D_BWD_ARGS
        rc = launch_status();
    }
    return rc;
}


extern "C" {

int gi2d_rasterize_sum_bwd(gi2d_workspace *ws, const float *xys, const int *radii,
                           const float *conics, const float *colors, const float *opacity, int N,
                           int H, int W, const gi2d_conv *conv, const float *v_out, float *v_xys,
                           float *v_conics, float *v_colors, float *v_opacity, void *stream)
{
    int rc = check_raster_args(ws, N, H, W, conv);
    if (rc != GI2D_OK) return rc;
    if (N == 0) return GI2D_OK;
    const int T = ws->tiles_x * (ws->bin_row1 - ws->bin_row0);
    if (!xys || !radii || !conics || !colors || !opacity || (!v_out && T > 0) || !v_xys ||
        !v_conics || !v_colors || !v_opacity)
        return GI2D_E_INVALID;
    DeviceGuard guard(ws->device);
    hipStream_t st = (hipStream_t)stream;
    rc = raster_bwd_pass1(ws, xys, radii, conics, colors, opacity, N, H, W, conv, v_out, st);
    if (rc != GI2D_OK) return rc;
    Probe probe(ws, GI2D_PROBE_GATHER, st);
    gather_grads_kernel<<<cdiv(N, 256), 256, 0, st>>>(
        xys, radii, N, ws->tiles_x, ws->tiles_y, ws->bin_row0, ws->bin_row1, ws->d_len, ws->d_seg,
        ws->seg_cap, ws->d_partials, ws->d_gslots, (size_t)ws->slot_cap, ws->bwd_stamp, v_xys,
        v_conics, v_colors, v_opacity);
    return launch_status();
}

}  // extern "C"
