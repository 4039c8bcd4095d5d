// Microbenchmark: how many wave64 vector instructions per second does the part issue?
//
// Streams, each at 1, 2, 4 and 8 waves per SIMD with every CU busy:
//   indep    -- v_fma_f32 on 8 independent accumulators (no dependency stalls)
//   dep      -- v_fma_f32 on ONE accumulator (every instruction waits for the one before)
//   exp14    -- 13 v_fma_f32 (8 accumulators) + 1 v_exp_f32 per group of 14
//   pairloop -- the forward rasteriser's pair loop as it stands in raster_fwd_tile.h (plain
//               entries): per evaluation 2 ds_read_b128 + 1 ds_read_b32 of a wave-uniform LDS
//               address, a scalar bit-scan step, and 14 vector instructions in one dependent
//               chain (dx, dy, q, p, exp2, compare, select, 3 FMAs into the colour sums)
//   pairloop4 -- the same with four pixels per lane (one wave renders the four quadrants of a
//               tile: four independent chains per entry)
// Prints wave-instructions per second (counting the vector instructions only) and the same
// per SIMD-cycle at the clock measured inside the kernel (s_memtime / s_memrealtime).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Xclang -target-feature -Xclang
//         -packed-fp32-ops -o tools/micro/valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define FMA8_INDEP \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n" \
    "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n" \
    "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"

struct Stamps {
    unsigned long long cyc, real;
};

__device__ __forceinline__ void stamp_begin(unsigned long long &c, unsigned long long &r)
{
    c = __builtin_amdgcn_s_memtime();
    r = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ void stamp_end(Stamps *st, unsigned long long c, unsigned long long r)
{
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        Stamps s = {c1 - c, r1 - r};
        st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
    }
}

// 32 vector instructions per iteration
__global__ void k_indep(float *out, int iters, float m, float c, Stamps *st)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int i = 0; i < iters; ++i) {
        asm volatile(FMA8_INDEP FMA8_INDEP FMA8_INDEP FMA8_INDEP
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(m), "v"(c));
    }
    stamp_end(st, t0, r0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

#define FMA8_DEP \
    "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" \
    "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" \
    "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"

__global__ void k_dep(float *out, int iters, float m, float c, Stamps *st)
{
    float a0 = threadIdx.x;
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int i = 0; i < iters; ++i)
        asm volatile(FMA8_DEP FMA8_DEP FMA8_DEP FMA8_DEP : "+v"(a0) : "v"(m), "v"(c));
    stamp_end(st, t0, r0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}

// 28 vector instructions per iteration: two groups of 13 FMAs + 1 exp
#define GROUP14 \
    FMA8_INDEP "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n" \
    "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32 %7, %6\n"

__global__ void k_exp14(float *out, int iters, float m, float c, Stamps *st)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int i = 0; i < iters; ++i) {
        asm volatile(GROUP14 GROUP14
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(m), "v"(c));
    }
    stamp_end(st, t0, r0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// The pair loop: ENTRIES staged entries in LDS, every wave evaluates all of them `rounds` times
// for PPL pixels per lane.  14 vector instructions per (entry, pixel set).
#define ENTRIES 64
template <int PPL>
__global__ void k_pairloop(float *out, int rounds, const float4 *__restrict__ rec, float alpha_min,
                           unsigned long long bits_in, Stamps *st)
{
    __shared__ float4 s_a[ENTRIES], s_b[ENTRIES], s_c[ENTRIES];
    if (threadIdx.x < ENTRIES) {
        s_a[threadIdx.x] = rec[3 * threadIdx.x];
        s_b[threadIdx.x] = rec[3 * threadIdx.x + 1];
        s_c[threadIdx.x] = rec[3 * threadIdx.x + 2];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float px[PPL], py[PPL], r[PPL], g[PPL], b[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        px[u] = (float)((lane & 7) + 8 * (u & 1));
        py[u] = (float)((lane >> 3) + 8 * (u >> 1));
        r[u] = g[u] = b[u] = 0.f;
    }
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            // (a wave-uniform mask the compiler cannot see through: a ballot, as in the kernel)
            unsigned long long bits = __ballot(((bits_in >> lane) & 1ull) != 0ull);
            while (bits) {
                const int k = (int)__builtin_ctzll(bits);
                bits &= bits - 1;
                const float4 A = s_a[k];
                const float4 B = s_b[k];
                const float cb = s_c[k].x;
                const float dx = A.x - px[u], dy = A.y - py[u];
                const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                const float p = __builtin_fmaf(B.x * dy, dy, __builtin_fmaf(q, dx, B.y));
                const float alpha = __builtin_amdgcn_exp2f(p);
                const float w = alpha < alpha_min ? 0.f : alpha;
                r[u] = __builtin_fmaf(B.z, w, r[u]);
                g[u] = __builtin_fmaf(B.w, w, g[u]);
                b[u] = __builtin_fmaf(cb, w, b[u]);
            }
        }
    }
    stamp_end(st, t0, r0);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) s += r[u] + g[u] + b[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The pair loop as a COUNTED loop over a compacted list (no bit scan): UNROLL evaluations per
// trip, records read at immediate offsets from one address register.
template <int PPL, int UNROLL>
__global__ void k_pairloop_counted(float *out, int rounds, const float4 *__restrict__ rec,
                                   float alpha_min, int n, Stamps *st)
{
    __shared__ float4 s_a[ENTRIES], s_b[ENTRIES], s_c[ENTRIES];
    if (threadIdx.x < ENTRIES) {
        s_a[threadIdx.x] = rec[3 * threadIdx.x];
        s_b[threadIdx.x] = rec[3 * threadIdx.x + 1];
        s_c[threadIdx.x] = rec[3 * threadIdx.x + 2];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float px[PPL], py[PPL], r[PPL], g[PPL], b[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        px[u] = (float)((lane & 7) + 8 * (u & 1));
        py[u] = (float)((lane >> 3) + 8 * (u >> 1));
        r[u] = g[u] = b[u] = 0.f;
    }
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            for (int j = 0; j < n; j += UNROLL) {
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) {
                    const float4 A = s_a[j + i];
                    const float4 B = s_b[j + i];
                    const float cb = s_c[j + i].x;
                    const float dx = A.x - px[u], dy = A.y - py[u];
                    const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                    const float p = __builtin_fmaf(B.x * dy, dy, __builtin_fmaf(q, dx, B.y));
                    const float alpha = __builtin_amdgcn_exp2f(p);
                    const float w = alpha < alpha_min ? 0.f : alpha;
                    r[u] = __builtin_fmaf(B.z, w, r[u]);
                    g[u] = __builtin_fmaf(B.w, w, g[u]);
                    b[u] = __builtin_fmaf(cb, w, b[u]);
                }
            }
        }
    }
    stamp_end(st, t0, r0);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) s += r[u] + g[u] + b[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The same evaluations with the record held in registers: no LDS reads in the loop (what the
// vector and scalar instructions alone allow).
template <int PPL>
__global__ void k_pairloop_regs(float *out, int rounds, const float4 *__restrict__ rec,
                                float alpha_min, unsigned long long bits_in, Stamps *st)
{
    const int lane = threadIdx.x & 63;
    const float4 A = rec[0], B = rec[1];
    const float cb = rec[2].x;
    float px[PPL], py[PPL], r[PPL], g[PPL], b[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        px[u] = (float)((lane & 7) + 8 * (u & 1));
        py[u] = (float)((lane >> 3) + 8 * (u >> 1));
        r[u] = g[u] = b[u] = 0.f;
    }
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            unsigned long long bits = __ballot(((bits_in >> lane) & 1ull) != 0ull);
            while (bits) {
                const int k = (int)__builtin_ctzll(bits);
                bits &= bits - 1;
                const float dx = A.x - px[u] + (float)k, dy = A.y - py[u];
                const float q = __builtin_fmaf(A.w, dy, A.z * dx);
                const float p = __builtin_fmaf(B.x * dy, dy, __builtin_fmaf(q, dx, B.y));
                const float alpha = __builtin_amdgcn_exp2f(p);
                const float w = alpha < alpha_min ? 0.f : alpha;
                r[u] = __builtin_fmaf(B.z, w, r[u]);
                g[u] = __builtin_fmaf(B.w, w, g[u]);
                b[u] = __builtin_fmaf(cb, w, b[u]);
            }
        }
    }
    stamp_end(st, t0, r0);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) s += r[u] + g[u] + b[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The record of each entry fetched by SCALAR loads from global memory (48-byte records, the
// next entry's requested before the current one is evaluated): no LDS and no address move in
// the loop; the record's fields are scalar operands of the vector instructions.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v *const_f4_ptr;
template <int PPL>
__global__ void k_pairloop_sload(float *out, int rounds, const float4 *__restrict__ rec,
                                 const int *__restrict__ ids, float alpha_min,
                                 unsigned long long bits_in, Stamps *st)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const_f4_ptr crec = (const_f4_ptr)(unsigned long long)(rec);
    const int my_id = ids[(wave * ENTRIES + lane) & 0xfffff];   // lane k holds entry k's Gaussian
    float px[PPL], py[PPL], r[PPL], g[PPL], b[PPL];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
        px[u] = (float)((lane & 7) + 8 * (u & 1));
        py[u] = (float)((lane >> 3) + 8 * (u >> 1));
        r[u] = g[u] = b[u] = 0.f;
    }
    unsigned long long t0, r0;
    stamp_begin(t0, r0);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            unsigned long long bits = __ballot(((bits_in >> lane) & 1ull) != 0ull);
            int k = (int)__builtin_ctzll(bits);
            int gid = __builtin_amdgcn_readlane(my_id, k);
            f4v A = crec[3 * gid], B = crec[3 * gid + 1], C = crec[3 * gid + 2];
            while (bits) {
                bits &= bits - 1;
                const f4v A0 = A, B0 = B;
                const float cb = C.x;
                if (bits) {
                    k = (int)__builtin_ctzll(bits);
                    gid = __builtin_amdgcn_readlane(my_id, k);
                    A = crec[3 * gid];
                    B = crec[3 * gid + 1];
                    C = crec[3 * gid + 2];
                }
                const float dx = A0.x - px[u], dy = A0.y - py[u];
                const float q = __builtin_fmaf(A0.w, dy, A0.z * dx);
                const float p = __builtin_fmaf(B0.x * dy, dy, __builtin_fmaf(q, dx, B0.y));
                const float alpha = __builtin_amdgcn_exp2f(p);
                const float w = alpha < alpha_min ? 0.f : alpha;
                r[u] = __builtin_fmaf(B0.z, w, r[u]);
                g[u] = __builtin_fmaf(B0.w, w, g[u]);
                b[u] = __builtin_fmaf(cb, w, b[u]);
            }
        }
    }
    stamp_end(st, t0, r0);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) s += r[u] + g[u] + b[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
int time_it(const char *name, int waves_per_simd, double instr_per_wave, F launch, Stamps *d_st,
            int cus)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = cus * waves_per_simd;   // 256 lanes each: one wave per SIMD per block
    launch(blocks);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        launch(blocks);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const int waves = blocks * 4;
    std::vector<Stamps> h(waves);
    CHECK(hipMemcpy(h.data(), d_st, sizeof(Stamps) * waves, hipMemcpyDeviceToHost));
    std::vector<double> ghz(waves), cyc(waves);
    for (int i = 0; i < waves; ++i) {
        ghz[i] = (double)h[i].cyc / ((double)h[i].real * 10.0);   // realtime ticks at 100 MHz
        cyc[i] = (double)h[i].cyc;
    }
    std::nth_element(ghz.begin(), ghz.begin() + waves / 2, ghz.end());
    std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
    const double clock = ghz[waves / 2], wave_cycles = cyc[waves / 2];
    const double rate = instr_per_wave * waves / (best * 1e-3);
    // cycles one SIMD spends per wave instruction while its waves_per_simd waves are all running
    const double cyc_per_instr = wave_cycles / (instr_per_wave * waves_per_simd);
    printf("{\"stream\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4e, "
           "\"in_kernel_clock_ghz\": %.3f, \"simd_cycles_per_wave_instr\": %.3f}\n",
           name, waves_per_simd, best, rate, clock, cyc_per_instr);
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, cus,
           prop.clockRate / 1000);
    float *d_out;
    Stamps *d_st;
    float4 *d_rec;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * cus * 8));
    CHECK(hipMalloc(&d_st, sizeof(Stamps) * 4 * cus * 8));
    std::vector<float4> rec(3 * ENTRIES);
    for (int k = 0; k < ENTRIES; ++k) {
        rec[3 * k] = make_float4(4.f + (k % 9), 3.f + (k % 7), -0.05f, -0.01f);
        rec[3 * k + 1] = make_float4(-0.04f, 0.f, 0.3f, 0.4f);
        rec[3 * k + 2] = make_float4(0.5f, 0.f, 0.f, 0.f);
    }
    CHECK(hipMalloc(&d_rec, sizeof(float4) * rec.size()));
    CHECK(hipMemcpy(d_rec, rec.data(), sizeof(float4) * rec.size(), hipMemcpyHostToDevice));
    const int iters = 20000, rounds = 600;
    const int ws[4] = {1, 2, 4, 8};
    for (int w : ws)
        if (time_it("indep", w, 32.0 * iters,
                    [&](int b) { k_indep<<<b, 256>>>(d_out, iters, 0.999f, 0.001f, d_st); }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("dep", w, 32.0 * iters,
                    [&](int b) { k_dep<<<b, 256>>>(d_out, iters, 0.999f, 0.001f, d_st); }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("exp14", w, 28.0 * iters,
                    [&](int b) { k_exp14<<<b, 256>>>(d_out, iters, 0.999f, 0.001f, d_st); }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop", w, 14.0 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop<1><<<b, 256>>>(d_out, rounds, d_rec, 1.f / 255.f, ~0ull, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop4", w, 14.0 * 4 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop<4><<<b, 256>>>(d_out, rounds, d_rec, 1.f / 255.f, ~0ull, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop_regs", w, 14.0 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop_regs<1><<<b, 256>>>(d_out, rounds, d_rec, 1.f / 255.f, ~0ull, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop_counted_u2", w, 13.0 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop_counted<1, 2><<<b, 256>>>(d_out, rounds, d_rec, 1.f / 255.f, ENTRIES, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop_counted_u4", w, 13.0 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop_counted<1, 4><<<b, 256>>>(d_out, rounds, d_rec, 1.f / 255.f, ENTRIES, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop4_counted_u4", w, 13.0 * 4 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop_counted<4, 4><<<b, 256>>>(d_out, rounds, d_rec, 1.f / 255.f, ENTRIES, d_st);
                    }, d_st, cus))
            return 1;
    // short lists, as a 768x512 / 30 000 frame has them: 16 evaluations per quadrant loop
    for (int w : ws)
        if (time_it("pairloop4_bitscan_n16", w, 14.0 * 4 * 16 * rounds * 4,
                    [&](int b) {
                        k_pairloop<4><<<b, 256>>>(d_out, rounds * 4, d_rec, 1.f / 255.f, 0xffffull, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop4_counted_u2_n16", w, 13.0 * 4 * 16 * rounds * 4,
                    [&](int b) {
                        k_pairloop_counted<4, 2><<<b, 256>>>(d_out, rounds * 4, d_rec, 1.f / 255.f, 16, d_st);
                    }, d_st, cus))
            return 1;
    // records of 1M Gaussians (48 MB: beyond the scalar caches, inside L2 + Infinity Cache), ids
    // at random as in a tile's list
    const int NG = 1 << 20;
    std::vector<float4> big(3 * (size_t)NG);
    for (int k = 0; k < NG; ++k) {
        big[3 * (size_t)k] = make_float4(4.f + (k % 9), 3.f + (k % 7), -0.05f, -0.01f);
        big[3 * (size_t)k + 1] = make_float4(-0.04f, 0.f, 0.3f, 0.4f);
        big[3 * (size_t)k + 2] = make_float4(0.5f, 0.f, 0.f, 0.f);
    }
    float4 *d_big;
    int *d_ids;
    CHECK(hipMalloc(&d_big, sizeof(float4) * big.size()));
    CHECK(hipMemcpy(d_big, big.data(), sizeof(float4) * big.size(), hipMemcpyHostToDevice));
    std::vector<int> ids(1 << 20);
    srand(1);
    for (auto &v : ids) v = (int)(((unsigned)rand() * 2654435761u) >> 12) & (NG - 1);
    CHECK(hipMalloc(&d_ids, sizeof(int) * ids.size()));
    CHECK(hipMemcpy(d_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice));
    for (int w : ws)
        if (time_it("pairloop_sload", w, 14.0 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop_sload<1><<<b, 256>>>(d_out, rounds, d_big, d_ids, 1.f / 255.f,
                                                        ~0ull, d_st);
                    }, d_st, cus))
            return 1;
    for (int w : ws)
        if (time_it("pairloop4_sload", w, 14.0 * 4 * ENTRIES * rounds,
                    [&](int b) {
                        k_pairloop_sload<4><<<b, 256>>>(d_out, rounds, d_big, d_ids, 1.f / 255.f,
                                                        ~0ull, d_st);
                    }, d_st, cus))
            return 1;
    return 0;
}
