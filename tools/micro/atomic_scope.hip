// Microbenchmark: returning atomic adds on per-tile counters (one per 128-byte line), device
// scope on one counter set versus workgroup scope on a counter set of the XCD the wave runs on.
// Prints time per launch and checks that every returned position is unique per counter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STRIDE 32
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ int xcc_id()
{
    // HW_REG_XCC_ID (20), bits [3:0]
    return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
}

template <int MODE>  // 0 device scope, 1 workgroup scope on the wave's XCD set, 2 device scope on the XCD set
__global__ void fill(const int *__restrict__ tile, int n, int T, int *counters, int *pos, int *xcc_of)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    int t = tile[g];
    int p;
    if (MODE == 0) {
        p = __hip_atomic_fetch_add(counters + (size_t)t * STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xcc_of[g] = 0;
    } else {
        const int x = xcc_id();
        int *c = counters + ((size_t)x * T + t) * STRIDE;
        if (MODE == 1) p = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else p = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xcc_of[g] = x;
    }
    pos[g] = p;
}

template <int MODE>
int run(const char *name, const int *d_tile, const std::vector<int> &h_tile, int n, int T)
{
    int *d_cnt, *d_pos, *d_x;
    size_t cbytes = sizeof(int) * STRIDE * (size_t)T * 16;
    CHECK(hipMalloc(&d_cnt, cbytes));
    CHECK(hipMalloc(&d_pos, sizeof(int) * n));
    CHECK(hipMalloc(&d_x, sizeof(int) * n));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipMemset(d_cnt, 0, cbytes));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        fill<MODE><<<(n + 63) / 64, 64>>>(d_tile, n, T, d_cnt, d_pos, d_x);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        sum += ms;
    }
    std::vector<int> pos(n), xs(n), cnt((size_t)STRIDE * T * 16);
    CHECK(hipMemcpy(pos.data(), d_pos, sizeof(int) * n, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(xs.data(), d_x, sizeof(int) * n, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(cnt.data(), d_cnt, cbytes, hipMemcpyDeviceToHost));
    // uniqueness of (set, tile, position) and counter totals
    std::vector<long long> keys(n);
    int xmax = 0;
    for (int i = 0; i < n; ++i) {
        keys[i] = ((long long)xs[i] * T + h_tile[i]) * 100000LL + pos[i];
        xmax = std::max(xmax, xs[i]);
    }
    std::sort(keys.begin(), keys.end());
    bool unique = std::adjacent_find(keys.begin(), keys.end()) == keys.end();
    long long total = 0;
    for (size_t i = 0; i < (size_t)T * 16; ++i) total += cnt[i * STRIDE];
    printf("%-44s best %.2f us  avg %.2f us  unique=%d total=%lld (n=%d) max xcc=%d\n", name, best * 1e3f,
           sum / reps * 1e3f, (int)unique, total, n, xmax);
    (void)hipFree(d_cnt);
    (void)hipFree(d_pos);
    (void)hipFree(d_x);
    return 0;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 497139, T = argc > 2 ? atoi(argv[2]) : 32400;
    std::vector<int> h_tile(n);
    srand(1);
    for (int i = 0; i < n; ++i) h_tile[i] = (int)(((long long)rand() * 65536 + rand()) % T);
    int *d_tile;
    CHECK(hipMalloc(&d_tile, sizeof(int) * n));
    CHECK(hipMemcpy(d_tile, h_tile.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    printf("n=%d atomics over T=%d counters\n", n, T);
    if (run<0>("device scope, one set", d_tile, h_tile, n, T)) return 1;
    if (run<2>("device scope, per-XCD sets", d_tile, h_tile, n, T)) return 1;
    if (run<1>("workgroup scope, per-XCD sets", d_tile, h_tile, n, T)) return 1;
    return 0;
}
