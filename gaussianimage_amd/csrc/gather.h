// gather.h -- backward pass 2 per Gaussian: the sum of its slots of the partial-sum slab.
#pragma once
#include "common.h"

// Adds up Gaussian g's slots of the partial-sum slab, in the row-major order of its tile box.
__device__ __forceinline__ void gather_partials(int g, int r, float cx, float cy, int tiles_x,
                                                int tiles_y, int row0, int row1,
                                                const int *__restrict__ tile_len,
                                                const int *__restrict__ seg, int C,
                                                const float *__restrict__ partials,
                                                const int *__restrict__ slot_tab,
                                                float acc[PART_STRIDE])
{
#pragma unroll
    for (int k = 0; k < PART_STRIDE; ++k) acc[k] = 0.f;
    const int4 tab = *reinterpret_cast<const int4 *>(slot_tab + (size_t)g * SLOT_TAB);
    if (r <= 0) return;
    TileBox box = tile_box(cx, cy, (float)r, tiles_x, tiles_y);
    const int y0 = max(box.y0, row0), y1 = min(box.y1, row1);
    const int w = box.x1 - box.x0;
    const int cnt = (w > 0 && y1 > y0) ? w * (y1 - y0) : 0;
    // The first SLOT_TAB tiles: list position remembered by pass 1; the loads of the four
    // slots are independent of each other.
    int tl[SLOT_TAB], ln[SLOT_TAB];
    const int pos[SLOT_TAB] = {tab.x, tab.y, tab.z, tab.w};
#pragma unroll
    for (int k = 0; k < SLOT_TAB; ++k) {
        const int ty = w > 0 ? y0 + k / w : 0, tx = w > 0 ? box.x0 + k % w : 0;
        tl[k] = k < cnt ? (ty - row0) * tiles_x + tx : 0;
        ln[k] = k < cnt ? tile_len[tl[k]] : 0;
    }
    float part[SLOT_TAB][PART_STRIDE];
#pragma unroll
    for (int k = 0; k < SLOT_TAB; ++k) {
        // an overflowed tile (ln > C) wrote no partial sums
        const bool live = k < cnt && ln[k] <= C && pos[k] >= 0 && pos[k] < ln[k];
        const float *src = partials + ((size_t)tl[k] * C + (live ? pos[k] : 0)) * PART_STRIDE;
#pragma unroll
        for (int u = 0; u < PART_STRIDE; ++u) part[k][u] = live ? src[u] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < SLOT_TAB; ++k)
#pragma unroll
        for (int u = 0; u < PART_STRIDE; ++u) acc[u] += part[k][u];
    // larger boxes: the rest by binary search (lists ascend in Gaussian id)
    for (int k = SLOT_TAB; k < cnt; ++k) {
        const int ty = y0 + k / w, tx = box.x0 + k % w;
        const int t = (ty - row0) * tiles_x + tx;
        const int n = tile_len[t];
        if (n > C) continue;
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
            const float *src = partials + ((size_t)t * C + lo) * PART_STRIDE;
#pragma unroll
            for (int u = 0; u < PART_STRIDE; ++u) acc[u] += src[u];
        }
    }
}
