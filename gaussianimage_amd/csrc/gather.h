// gather.h -- backward pass 2 per Gaussian: the sum of its partial sums over the tiles of its box.
#pragma once
#include "common.h"

// Adds up Gaussian g's partial sums in the row-major order of its tile box.  The first GSLOTS
// tiles come from the Gaussian's own slots -- GSLOTS_EAGER of them loaded before anything else
// is known about it (one round trip), the rest once the box is known; a slot counts only if its
// stamp is this backward pass's (an overflowed or unreachable tile writes none).  Larger boxes: the
// remaining tiles by binary search in the tile's sorted list.
__device__ __forceinline__ void gather_partials(int g, int r, float cx, float cy, int tiles_x,
                                                int tiles_y, int row0, int row1,
                                                const int *__restrict__ tile_len,
                                                const int *__restrict__ seg, int C,
                                                const float *__restrict__ partials,
                                                const float *__restrict__ gslots, size_t gslot_n,
                                                int stamp, float acc[PART_STRIDE])
{
    // slot k of Gaussian g lies in plane k: [GSLOTS][gslot_n][GSLOT_STRIDE] (the planes of the
    // tile indices that occur are the only memory touched)
    float part[GSLOTS_EAGER][GSLOT_STRIDE];
    const float *mine = gslots + (size_t)g * GSLOT_STRIDE;
    const size_t plane = gslot_n * GSLOT_STRIDE;
#pragma unroll
    for (int k = 0; k < GSLOTS_EAGER; ++k) {
        const float2 *src = reinterpret_cast<const float2 *>(mine + k * plane);
#pragma unroll
        for (int u = 0; u < GSLOT_STRIDE / 2; ++u) {
            const float2 v = src[u];
            part[k][2 * u] = v.x;
            part[k][2 * u + 1] = v.y;
        }
    }
#pragma unroll
    for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
    if (r <= 0) return;
    TileBox box = tile_box(cx, cy, (float)r, tiles_x, tiles_y);
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
#pragma unroll
    for (int k = 0; k < GSLOTS_EAGER; ++k) {
        const bool live = k < cnt && __float_as_int(part[k][PART_STRIDE]) == stamp;
#pragma unroll
        for (int u = 0; u < PART_STRIDE; ++u) acc[u] += live ? part[k][u] : 0.f;
    }
    // the Gaussian's further slots, four at a time: the loads of a group are independent of
    // each other (one more round trip per group)
    const int lim = min(cnt, GSLOTS);
    for (int k0 = GSLOTS_EAGER; k0 < lim; k0 += 4) {
        float2 v[4][GSLOT_STRIDE / 2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(k0 + j, GSLOTS - 1);
            const float2 *p = reinterpret_cast<const float2 *>(mine + k * plane);
#pragma unroll
            for (int u = 0; u < GSLOT_STRIDE / 2; ++u) v[j][u] = p[u];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = k0 + j < lim && __float_as_int(v[j][4].y) == stamp;
            // (a dead slot may hold anything, NaN included: select, do not multiply)
            acc[0] += live ? v[j][0].x : 0.f;
            acc[1] += live ? v[j][0].y : 0.f;
            acc[2] += live ? v[j][1].x : 0.f;
            acc[3] += live ? v[j][1].y : 0.f;
            acc[4] += live ? v[j][2].x : 0.f;
            acc[5] += live ? v[j][2].y : 0.f;
            acc[6] += live ? v[j][3].x : 0.f;
            acc[7] += live ? v[j][3].y : 0.f;
            acc[8] += live ? v[j][4].x : 0.f;
        }
    }
    for (int k = GSLOTS; k < cnt; ++k) {
        const int ty = y0 + k / w, tx = box.x0 + k % w;
        const int t = (ty - row0) * tiles_x + tx;
        const int n = tile_len[t];
        if (n > C) continue;  // an overflowed tile wrote no partial sums
        const int *lst = seg + (size_t)t * C;
        int lo = 0, hi = n;
        while (lo < hi) {  // first position with id >= g
            int mid = (lo + hi) >> 1;
            if (lst[mid] < g)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < n && lst[lo] == g) {
            const float *p = partials + ((size_t)t * C + lo) * PART_STRIDE;
#pragma unroll
            for (int u = 0; u < PART_STRIDE; ++u) acc[u] += p[u];
        }
    }
}
