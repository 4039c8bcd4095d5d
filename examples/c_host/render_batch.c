/* A C host of libgi2d.so with no Python and no torch in the process: renders a batch of frames
 * through gi2d_render_batch (the call bench.py times) and the first frame again through the
 * entry point that allocates nothing, from parameters made by a small generator that the test
 * (tests/test_c_host.py) repeats in numpy.  What a C / C++ / cgo / JNI host of the reference's
 * operators would do at this boundary: device buffers of its own, raw pointers, int status.
 *
 *   render_batch FRAMES N H W OUT.bin     writes FRAMES*H*W*3 floats, prints a summary line
 *
 * Build (gaussianimage_amd/build.py: build_examples):
 *   hipcc -O2 -I include examples/c_host/render_batch.c -o examples/c_host/render_batch \
 *         -L gaussianimage_amd/csrc -lgi2d -Wl,-rpath,'$ORIGIN/../../gaussianimage_amd/csrc'
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gi2d.h"

#define CHECK_HIP(x)                                                                  \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)
#define CHECK_GI2D(x)                                                                 \
    do {                                                                              \
        int s_ = (x);                                                                 \
        if (s_ != GI2D_OK && s_ != GI2D_E_PREV_OVERFLOW) {                            \
            fprintf(stderr, "%s: %s (status %d)\n", #x, gi2d_error_string(s_), s_);   \
            return 3;                                                                 \
        }                                                                             \
    } while (0)

/* 32-bit linear congruential generator -> floats in [0, 1) with 24 random bits */
static uint32_t lcg_state;
static float lcg_unit(void)
{
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
}

int main(int argc, char **argv)
{
    if (argc != 6) {
        fprintf(stderr, "usage: %s FRAMES N H W OUT.bin\n", argv[0]);
        return 1;
    }
    const int F = atoi(argv[1]), N = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]);
    if (F < 1 || N < 1 || H < 1 || W < 1) return 1;
    const size_t G = (size_t)F * N, P = (size_t)F * H * W * 3;

    /* operator inputs as the reference's Cholesky model activates them: means in (-1, 1),
     * L = (l11, l21, l22) with l11, l22 in [0.5, 1.5), l21 in [0, 1), colours in [0, 1),
     * opacity 1 (gaussianimage_cholesky.py:30-42,62,74) */
    float *means = (float *)malloc(sizeof(float) * 2 * G), *L = (float *)malloc(sizeof(float) * 3 * G);
    float *colors = (float *)malloc(sizeof(float) * 3 * G), *opacity = (float *)malloc(sizeof(float) * G);
    float *image = (float *)malloc(sizeof(float) * P), *first = (float *)malloc(sizeof(float) * P / F);
    if (!means || !L || !colors || !opacity || !image || !first) return 2;
    lcg_state = 12345u;
    for (size_t g = 0; g < G; ++g) {
        means[2 * g] = 2.0f * lcg_unit() - 1.0f;
        means[2 * g + 1] = 2.0f * lcg_unit() - 1.0f;
        L[3 * g] = lcg_unit() + 0.5f;
        L[3 * g + 1] = lcg_unit();
        L[3 * g + 2] = lcg_unit() + 0.5f;
        colors[3 * g] = lcg_unit();
        colors[3 * g + 1] = lcg_unit();
        colors[3 * g + 2] = lcg_unit();
        opacity[g] = 1.0f;
    }

    float *d_means, *d_L, *d_colors, *d_opacity, *d_bg, *d_xys, *d_depths, *d_conics, *d_img, *d_one;
    int *d_radii, *d_nth;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMalloc((void **)&d_means, sizeof(float) * 2 * G));
    CHECK_HIP(hipMalloc((void **)&d_L, sizeof(float) * 3 * G));
    CHECK_HIP(hipMalloc((void **)&d_colors, sizeof(float) * 3 * G));
    CHECK_HIP(hipMalloc((void **)&d_opacity, sizeof(float) * G));
    CHECK_HIP(hipMalloc((void **)&d_bg, sizeof(float) * 3));
    CHECK_HIP(hipMalloc((void **)&d_xys, sizeof(float) * 2 * G));
    CHECK_HIP(hipMalloc((void **)&d_depths, sizeof(float) * G));
    CHECK_HIP(hipMalloc((void **)&d_conics, sizeof(float) * 3 * G));
    CHECK_HIP(hipMalloc((void **)&d_radii, sizeof(int) * G));
    CHECK_HIP(hipMalloc((void **)&d_nth, sizeof(int) * G));
    CHECK_HIP(hipMalloc((void **)&d_img, sizeof(float) * P));
    CHECK_HIP(hipMalloc((void **)&d_one, sizeof(float) * P / F));
    const float ones[3] = {1.0f, 1.0f, 1.0f};
    CHECK_HIP(hipMemcpy(d_means, means, sizeof(float) * 2 * G, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_L, L, sizeof(float) * 3 * G, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_colors, colors, sizeof(float) * 3 * G, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_opacity, opacity, sizeof(float) * G, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_bg, ones, sizeof(ones), hipMemcpyHostToDevice));

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    gi2d_workspace *ws = NULL;
    gi2d_conv conv;
    gi2d_default_conv(&conv);
    CHECK_GI2D(gi2d_workspace_create(0, &ws));
    /* the first call learns the shape synchronously, the following ones are asynchronous */
    for (int rep = 0; rep < 3; ++rep)
        CHECK_GI2D(gi2d_render_batch(ws, 0, F, d_means, d_L, NULL, d_colors, d_opacity, d_bg, N, H, W,
                                     &conv, d_xys, d_depths, d_radii, d_conics, d_nth, d_img, NULL,
                                     NULL, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(image, d_img, sizeof(float) * P, hipMemcpyDeviceToHost));

    /* frame 0 again through the entry point that allocates nothing (if it fits its limits) */
    int noalloc_matches = -1;
    const long long tiles = (long long)((W + GI2D_BLOCK - 1) / GI2D_BLOCK) * ((H + GI2D_BLOCK - 1) / GI2D_BLOCK);
    if (tiles <= GI2D_NOALLOC_MAX_TILES) {
        gi2d_frame_args a;
        memset(&a, 0, sizeof(a));
        a.ws = NULL;
        a.kind = 0;
        a.means2d = d_means;
        a.cov_a = d_L;
        a.colors = d_colors;
        a.opacity = d_opacity;
        a.background = d_bg;
        a.N = N;
        a.H = H;
        a.W = W;
        a.conv = conv;
        a.xys = d_xys;
        a.depths = d_depths;
        a.radii = d_radii;
        a.conics = d_conics;
        a.num_tiles_hit = d_nth;
        a.out_img = d_one;
        a.stream = stream;
        CHECK_GI2D(gi2d_render_frame_noalloc(&a, 0));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(first, d_one, sizeof(float) * P / F, hipMemcpyDeviceToHost));
        noalloc_matches = memcmp(first, image, sizeof(float) * P / F) == 0;
    }

    double sum = 0.0;
    for (size_t i = 0; i < P; ++i) sum += image[i];
    FILE *out = fopen(argv[5], "wb");
    if (!out || fwrite(image, sizeof(float), P, out) != P) return 4;
    fclose(out);
    printf("gi2d %d build %s frames %d N %d %dx%d mean %.9f noalloc_matches %d\n", gi2d_version(),
           gi2d_build_id(), F, N, W, H, sum / (double)P, noalloc_matches);
    CHECK_GI2D(gi2d_workspace_destroy(ws));
    return 0;
}
