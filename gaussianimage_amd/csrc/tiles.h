// tiles.h -- device side of the per-tile lists: entering a Gaussian in the segments of its
// tile box, the per-tile sorts, and the words that finalise a tile.
#pragma once
#include "common.h"

// Lanes of the wave that want the same tile share one atomic.  Groups the active lanes by
// their value of t (t < 0 = nothing to do): `leader` is the group's first lane, `rank` the
// lane's place in the group, `gsize` the group's size.  Returning atomics are the scarce
// resource of the list filling (about 21 G/s on the whole chip, measured, whatever the scope);
// Gaussians that are neighbours in memory AND in the image -- a fitter or an encoder that keeps
// them in a spatial order -- then cost one atomic per (wave, tile) instead of one per
// (Gaussian, tile).  Lanes in no group found keep leader = themselves (a group of one): the
// search gives up after three groups of one, which is what unordered Gaussians look like.
__device__ __forceinline__ void group_lanes_by_tile(int t, int lane, int &leader, int &rank,
                                                    int &gsize)
{
    leader = lane;
    rank = 0;
    gsize = 1;
    unsigned long long todo = __ballot(t >= 0);
    int singles = 0;
    while (todo) {
        const int first = (int)__builtin_ctzll(todo);
        const int tl = __builtin_amdgcn_readlane(t, first);
        const unsigned long long same = __ballot(t == tl);
        const int cnt = (int)__popcll(same);
        if (t == tl) {
            leader = first;
            rank = (int)__builtin_amdgcn_mbcnt_hi(
                (unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
            gsize = cnt;
        }
        todo &= ~same;
        if (cnt == 1 && ++singles == 3) break;
    }
}

// Enters Gaussian g in the segment of every tile of its box (clipped to the band's tile rows
// [row0,row1)).  Four rounds of returning atomics are put in flight before the first dependent
// store, so a lane's round trips overlap instead of adding up.  Ids beyond a segment's capacity
// are dropped; the counter keeps counting, which is how the overflow is noticed.
// Two forms, chosen per wave: when at least four of the wave's Gaussians start in the same tile
// (they lie in a spatial order), the lanes that enter the same tile in a round take their
// places from one atomic; otherwise every lane walks its own box (grouping unordered Gaussians
// would only cost instructions).  Workgroups are whole waves (lane = threadIdx.x % 64).
// lanes of a wave that must start in the same tile for the wave to share its list atomics
#ifndef EMIT_GROUP_MIN
#define EMIT_GROUP_MIN 4
#endif
#ifndef EMIT_FLIGHT
#define EMIT_FLIGHT 4   // returning atomics a lane has in flight per round
#endif
__device__ __forceinline__ void emit_tiles(int g, const TileBox &box, int row0, int row1,
                                           int tiles_x, int *tile_count, int *seg, int C)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
    int tx = box.x0, ty = y0;
    const int t_first = cnt > 0 ? (y0 - row0) * tiles_x + box.x0 : -1;
    const unsigned long long have = __ballot(t_first >= 0);
    if (have == 0ull) return;
    const int t_lead = __builtin_amdgcn_readlane(t_first, (int)__builtin_ctzll(have));
    if (__popcll(__ballot(t_first == t_lead)) < EMIT_GROUP_MIN) {
        for (int k0 = 0; k0 < cnt; k0 += EMIT_FLIGHT) {
            int t[EMIT_FLIGHT], pos[EMIT_FLIGHT];
#pragma unroll
            for (int u = 0; u < EMIT_FLIGHT; ++u) {
                t[u] = (k0 + u < cnt) ? (ty - row0) * tiles_x + tx : -1;
                if (++tx >= box.x1) {
                    tx = box.x0;
                    ++ty;
                }
            }
#pragma unroll
            for (int u = 0; u < EMIT_FLIGHT; ++u) {
                pos[u] = C;
                if (t[u] >= 0) pos[u] = atomicAdd(&tile_count[CNT_AT(t[u])], 1);
            }
#pragma unroll
            for (int u = 0; u < EMIT_FLIGHT; ++u)
                if (pos[u] < C) seg[(size_t)t[u] * C + pos[u]] = g;
        }
        return;
    }
    for (int k0 = 0; __ballot(k0 < cnt) != 0ull; k0 += EMIT_FLIGHT) {
        int t[EMIT_FLIGHT], pos[EMIT_FLIGHT], leader[EMIT_FLIGHT], rank[EMIT_FLIGHT], gsize[EMIT_FLIGHT];
#pragma unroll
        for (int u = 0; u < EMIT_FLIGHT; ++u) {
            t[u] = (k0 + u < cnt) ? (ty - row0) * tiles_x + tx : -1;
            if (++tx >= box.x1) {
                tx = box.x0;
                ++ty;
            }
            group_lanes_by_tile(t[u], lane, leader[u], rank[u], gsize[u]);
        }
#pragma unroll
        for (int u = 0; u < EMIT_FLIGHT; ++u) {
            pos[u] = 0;
            if (t[u] >= 0 && leader[u] == lane)
                pos[u] = atomicAdd(&tile_count[CNT_AT(t[u])], gsize[u]);
        }
#pragma unroll
        for (int u = 0; u < EMIT_FLIGHT; ++u) {
            const int p = __shfl(pos[u], leader[u]) + rank[u];
            if (t[u] >= 0 && p < C) seg[(size_t)t[u] * C + p] = g;
        }
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
    tile_count[CNT_AT(t)] = 0;
    // Raising the published maximum is rare: only lists past half the capacity try, and only
    // those above the value already there reach the atomic (thousands of tiles hammering one
    // word would otherwise serialise at the memory side).
    if (2 * n_raw > C && n_raw > status->max_count) atomicMax(&status->max_count, n_raw);
    if (n_raw > C) status->overflow_gen = generation;
}
