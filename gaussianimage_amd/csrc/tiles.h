// tiles.h -- device side of the per-tile lists: entering a Gaussian in the segments of its
// tile box, the per-tile sorts, and the words that finalise a tile.
#pragma once
#include "common.h"

// Enters Gaussian g in the segment of every tile of its box (clipped to the band's tile rows
// [row0,row1)).  Four returning atomics are put in flight before the first dependent store, so
// a lane's round trips overlap instead of adding up.  Ids beyond a segment's capacity are
// dropped; the counter keeps counting, which is how the overflow is noticed.
__device__ __forceinline__ void emit_tiles(int g, const TileBox &box, int row0, int row1,
                                           int tiles_x, int *tile_count, int *seg, int C)
{
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
    int tx = box.x0, ty = y0;
    for (int k0 = 0; k0 < cnt; k0 += 4) {
        int t[4], pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t[u] = (k0 + u < cnt) ? (ty - row0) * tiles_x + tx : -1;
            if (++tx >= box.x1) {
                tx = box.x0;
                ++ty;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pos[u] = C;
            if (t[u] >= 0) pos[u] = atomicAdd(&tile_count[(size_t)t[u] * CNT_STRIDE], 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pos[u] < C) seg[(size_t)t[u] * C + pos[u]] = g;
    }
}

// At most one lane per launch: copies the device status to pinned host memory, where the next
// calls read it without synchronising.  Issued first so that the slow host write overlaps the
// rest of the kernel.
__device__ __forceinline__ void publish_status(const DevStatus *status, volatile long long *h,
                                               long long generation)
{
    h[0] = status->max_count;
    h[2] = status->overflow_gen;
    h[1] = generation;
}

// ------------------------------------------------------------------------------------------
// per-tile sorting
// ------------------------------------------------------------------------------------------

// ascending in-place bitonic network; elements past n are a virtual +infinity
template <typename P>
__device__ __forceinline__ void bitonic_sort(P a, int n, int lane, int nthreads)
{
    int npow = 1;
    while (npow < n) npow <<= 1;
    const int half = npow >> 1;
    for (int k = 2; k <= npow; k <<= 1) {
        const int hk = k >> 1;
        for (int i = lane; i < half; i += nthreads) {
            int blk = i / hk, off = i - blk * hk;
            int lo = blk * k + off, hi = blk * k + k - 1 - off;
            if (hi < n) {
                int x = a[lo], y = a[hi];
                if (x > y) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
        __syncthreads();
        for (int j = hk >> 1; j >= 1; j >>= 1) {
            for (int i = lane; i < half; i += nthreads) {
                int lo = 2 * j * (i / j) + (i % j), hi = lo + j;
                if (hi < n) {
                    int x = a[lo], y = a[hi];
                    if (x > y) {
                        a[lo] = y;
                        a[hi] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Rank sort of up to 64*R distinct ints held R per lane: every value is broadcast once with
// v_readlane and each lane counts how many are smaller than its own.  No LDS, no barrier.
template <int R>
__device__ __forceinline__ void rank_sort_regs(int *ids, int beg, int n, int lane)
{
    int v[R], rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int idx = r * WAVE + lane;
        v[r] = idx < n ? ids[beg + idx] : INT_MAX;
        rank[r] = 0;
    }
#pragma unroll
    for (int r2 = 0; r2 < R; ++r2) {
        const int lim = __builtin_amdgcn_readfirstlane(min(WAVE, n - r2 * WAVE));
        for (int k = 0; k < lim; ++k) {
            const int sv = __builtin_amdgcn_readlane(v[r2], k);
#pragma unroll
            for (int r = 0; r < R; ++r) rank[r] += (sv < v[r]) ? 1 : 0;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * WAVE + lane < n) ids[beg + rank[r]] = v[r];
}


#define SORT_LDS_MAX 4096

// What a tile does once per binning, whoever gets there first (the standalone kernel below or
// the first rasterisation): publish the list length, return the counter to zero for the next
// frame, and note a nearly full or overflowed segment for the capacity policy.
__device__ __forceinline__ void finalize_tile_words(int t, int n_raw, int C, int *tile_count,
                                                    int *tile_len, DevStatus *status,
                                                    long long generation)
{
    tile_len[t] = n_raw;
    tile_count[(size_t)t * CNT_STRIDE] = 0;
    // Raising the published maximum is rare: only lists past half the capacity try, and only
    // those above the value already there reach the atomic (thousands of tiles hammering one
    // word would otherwise serialise at the memory side).
    if (2 * n_raw > C && n_raw > status->max_count) atomicMax(&status->max_count, n_raw);
    if (n_raw > C) status->overflow_gen = generation;
}
