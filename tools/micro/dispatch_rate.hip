// Microbenchmark: what does a launch of many small workgroups cost when they do (almost) nothing?
// Grid of G workgroups of 64 or 256 lanes with 3.3 KB of LDS each; each lane loads one word and
// stores one.  Prints microseconds per launch (HIP events around 20 launches).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/dispatch_rate tools/micro/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_INTS>
__global__ void tiny(const int *__restrict__ in, int *__restrict__ out, int n)
{
    __shared__ int s[LDS_INTS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    s[threadIdx.x] = in[i % n];
    __syncthreads();
    out[i] = s[(threadIdx.x + 1) % blockDim.x];
}

// the same work by persistent workgroups: `groups` workgroups walk all `tiles` tiles
template <int LDS_INTS>
__global__ void persistent(const int *__restrict__ in, int *__restrict__ out, int n, int tiles)
{
    __shared__ int s[LDS_INTS];
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int i = t * blockDim.x + threadIdx.x;
        s[threadIdx.x] = in[i % n];
        __syncthreads();
        out[i] = s[(threadIdx.x + 1) % blockDim.x];
        __syncthreads();
    }
}

int main()
{
    const int tiles = 36864;
    int *in, *out;
    CHECK(hipMalloc(&in, 4 << 20));
    CHECK(hipMalloc(&out, sizeof(int) * tiles * 256));
    CHECK(hipMemset(in, 0, 4 << 20));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"launch\": \"%s\", \"us\": %.2f}\n", name, ms * 1e3 / 20);
        return 0;
    };
    if (timed("36864 workgroups x 64 lanes, 3.3 KB LDS", [&] { tiny<832><<<tiles, 64>>>(in, out, 1 << 20); })) return 1;
    if (timed("36864 workgroups x 64 lanes, 64 B LDS", [&] { tiny<64><<<tiles, 64>>>(in, out, 1 << 20); })) return 1;
    if (timed("9216 workgroups x 256 lanes, 13 KB LDS", [&] { tiny<3328><<<tiles / 4, 256>>>(in, out, 1 << 20); })) return 1;
    if (timed("1536 workgroups x 64 lanes (one frame)", [&] { tiny<832><<<1536, 64>>>(in, out, 1 << 20); })) return 1;
    if (timed("8192 persistent workgroups x 64 lanes over 36864 tiles", [&] { persistent<832><<<8192, 64>>>(in, out, 1 << 20, tiles); })) return 1;
    if (timed("2048 persistent workgroups x 64 lanes over 36864 tiles", [&] { persistent<832><<<2048, 64>>>(in, out, 1 << 20, tiles); })) return 1;
    if (timed("empty-ish 1 workgroup", [&] { tiny<64><<<1, 64>>>(in, out, 1 << 20); })) return 1;
    return 0;
}
