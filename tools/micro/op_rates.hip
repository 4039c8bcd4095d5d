// Microbenchmark: issue rate of the vector instruction classes the rasteriser's loops are made
// of, one class at a time: 32 instructions per trip on 8 independent registers, 8 waves per
// SIMD, every CU busy.  Prints wave-instructions per second and SIMD cycles per instruction at
// the nominal 2.4 GHz (256 CUs x 4 SIMDs).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/op_rates tools/micro/op_rates.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, OP, PRE)                                                                   \
    __global__ void NAME(float *out, int iters, float m, float c)                               \
    {                                                                                           \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
              a6 = a0 + 6, a7 = a0 + 7;                                                         \
        for (int i = 0; i < iters; ++i) {                                                       \
            asm volatile(PRE REP8(OP) REP8(OP) REP8(OP) REP8(OP)                                \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), \
                           "+v"(a7)                                                             \
                         : "v"(m), "v"(c)                                                       \
                         : "vcc", "s20", "s21");                                                            \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
    }

#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_SUB(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
#define OP_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define OP_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define OP_CMP_S(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_CVT(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_ADD_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_MAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_MULLIT(i) "v_mul_f32 %" #i ", 0.5, %" #i "\n"
#define OP_FMA_SGPR(i) "v_fma_f32 %" #i ", %" #i ", s20, %9\n"
#define OP_MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDMASK_S(i) "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[20:21]\n"
#define OP_FMA_CLAMP(i) "v_fma_f32 %" #i ", %" #i ", %8, %9 clamp\n"
#define OP_MUL_SGPR(i) "v_mul_f32 %" #i ", s20, %" #i "\n"
#define OP_CMPX(i) "v_cmpx_lt_f32 vcc, %9, %" #i "\n"
#define OP_MUL_LEGACY(i) "v_mul_legacy_f32 %" #i ", %" #i ", %8\n"
#define OP_FMAMK(i) "v_fmamk_f32 %" #i ", %" #i ", 0x41800000, %9\n"
#define OP_FMAC_LIT(i) "v_fmac_f32 %" #i ", 0x43800000, %8\n"
#define OP_AND_LIT(i) "v_and_b32 %" #i ", 0xffff, %" #i "\n"
#define OP_MUL_NEG(i) "v_mul_f32_e64 %" #i ", %" #i ", -%8\n"
#define OP_SUB_CLAMP(i) "v_sub_f32_e64 %" #i ", %" #i ", %8 clamp\n"
#define OP_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL_HI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define OP_MUL_U24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 4, %8\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD_CO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define OP_ADDC_CO(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 1, 2\n"
#define OP_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define OP_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define OP_MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define OP_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define OP_SUB_U32(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define OP_MAX_ZERO(i) "v_max_f32 %" #i ", 0, %" #i "\n"

KERNEL(k_fma, OP_FMA, "")
KERNEL(k_fmac, OP_MAC, "")
KERNEL(k_fma_sgpr, OP_FMA_SGPR, "s_mov_b32 s20, 0x3f7fbe77\n")
KERNEL(k_mul, OP_MUL, "")
KERNEL(k_mul_lit, OP_MULLIT, "")
KERNEL(k_add, OP_ADD, "")
KERNEL(k_sub, OP_SUB, "")
KERNEL(k_min, OP_MIN, "")
KERNEL(k_max, OP_MAX, "")
KERNEL(k_med3, OP_MED3, "")
KERNEL(k_cndmask, OP_CNDMASK, "v_cmp_lt_f32 vcc, %8, %9\n")
KERNEL(k_cmp, OP_CMP, "")
KERNEL(k_cmp_sgpr, OP_CMP_S, "")
KERNEL(k_mov, OP_MOV, "")
KERNEL(k_add_u32, OP_ADDU, "")
KERNEL(k_and, OP_AND, "")
KERNEL(k_lshl, OP_LSHL, "")
KERNEL(k_cvt, OP_CVT, "")
KERNEL(k_exp, OP_EXP, "")
KERNEL(k_rcp, OP_RCP, "")
KERNEL(k_mov_dpp, OP_DPP, "")
KERNEL(k_add_dpp, OP_ADD_DPP, "")
KERNEL(k_mad_u24, OP_MAD_U24, "")
KERNEL(k_fmamk, OP_FMAMK, "")
KERNEL(k_fmac_lit, OP_FMAC_LIT, "")
KERNEL(k_and_lit, OP_AND_LIT, "")
KERNEL(k_mul_neg, OP_MUL_NEG, "")
KERNEL(k_sub_clamp, OP_SUB_CLAMP, "")
KERNEL(k_mul_lo, OP_MUL_LO, "")
KERNEL(k_mul_hi, OP_MUL_HI, "")
KERNEL(k_mul_u24, OP_MUL_U24, "")
KERNEL(k_lshl_add, OP_LSHL_ADD, "")
KERNEL(k_add3, OP_ADD3, "")
KERNEL(k_add_co, OP_ADD_CO, "")
KERNEL(k_addc_co, OP_ADDC_CO, "")
KERNEL(k_bfe, OP_BFE, "")
KERNEL(k_floor, OP_FLOOR, "")
KERNEL(k_sqrt, OP_SQRT, "")
KERNEL(k_log, OP_LOG, "")
KERNEL(k_max_i32, OP_MAX_I32, "")
KERNEL(k_or, OP_OR, "")
KERNEL(k_sub_u32, OP_SUB_U32, "")

// 64-bit forms: four independent register pairs, 16 instructions per trip
#define KERNEL64(NAME, BODY)                                                                    \
    __global__ void NAME(float *out, int iters, float m, float c)                               \
    {                                                                                           \
        unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;             \
        unsigned mi = __float_as_uint(m), ci = __float_as_uint(c);                              \
        for (int i = 0; i < iters; ++i) {                                                       \
            asm volatile(BODY BODY BODY BODY                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                               \
                         : "v"(mi), "v"(ci)                                                     \
                         : "vcc", "s20", "s21");                                                \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                \
    }
#define B_MAD_U64 "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n" \
                  "v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n"
#define B_LSHL_ADD_U64 "v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %2\n" \
                       "v_lshl_add_u64 %2, %2, 2, %3\n v_lshl_add_u64 %3, %3, 2, %0\n"
#define B_LSHL_B64 "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n" \
                   "v_lshlrev_b64 %3, 1, %3\n"
KERNEL64(k_mad_u64, B_MAD_U64)
KERNEL64(k_lshl_add_u64, B_LSHL_ADD_U64)
KERNEL64(k_lshl_b64, B_LSHL_B64)
KERNEL(k_cndmask_s, OP_CNDMASK_S, "v_cmp_lt_f32 s[20:21], %8, %9\n")
KERNEL(k_fma_clamp, OP_FMA_CLAMP, "")
KERNEL(k_mul_sgpr, OP_MUL_SGPR, "s_mov_b32 s20, 0x3f7fbe77\n")

// One evaluation of the forward pair loop (plain entry) with the record in registers, three
// ways of dropping alpha < alpha_min:
//   SELECT  -- v_cmp into an SGPR pair + v_cndmask (what the compiler makes of ?:)
//   EXEC    -- v_cmpx narrows exec for the three accumulating FMAs, s_mov restores it
//   ARITH   -- s = clamp((alpha - below(alpha_min)) * 2^60) in one v_fma with the clamp modifier,
//              w = alpha * s: FMA-class instructions only
// 13 / 12 / 13 vector instructions per evaluation; two evaluations per trip on two accumulator sets
#define EVAL_HEAD(dx, dy, q, t, ax) \
    "v_sub_f32 " dx ", " ax ", %10\n v_sub_f32 " dy ", %13, %11\n v_mul_f32 " q ", " dx ", %14\n" \
    "v_fmac_f32 " q ", %15, " dy "\n v_mul_f32 " t ", " dy ", %16\n v_fma_f32 " dx ", " q ", " dx ", %17\n" \
    "v_fmac_f32 " dx ", " t ", " dy "\n v_exp_f32 " dx ", " dx "\n"
#define EVAL_ACC(w, r, g, b) \
    "v_fmac_f32 " r ", %18, " w "\n v_fmac_f32 " g ", %19, " w "\n v_fmac_f32 " b ", %20, " w "\n"

template <int FORM>
__global__ void k_eval(float *out, int iters, float amin, float below)
{
    const int lane = threadIdx.x & 63;
    float r0 = 0.f, g0 = 0.f, b0 = 0.f, r1 = 0.f, g1 = 0.f, b1 = 0.f;
    float px = (float)(lane & 7), py = (float)(lane >> 3);
    float ax = 4.f + amin, ay = 3.f, az = -0.05f, aw = -0.01f, bx = -0.04f, by = below * 0.f,
          bz = 0.3f, bw = 0.4f, cb = 0.5f;
    float huge = 1.152921504606846976e18f, bias = -below * 1.152921504606846976e18f;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (FORM == 0)
            asm volatile(EVAL_HEAD("%6", "%7", "%8", "%9", "%12")
                         "s_nop 0\n v_cmp_ngt_f32_e64 s[20:21], %21, %6\n s_nop 1\n"
                         "v_cndmask_b32_e64 %6, 0, %6, s[20:21]\n"
                         EVAL_ACC("%6", "%0", "%1", "%2")
                         EVAL_HEAD("%6", "%7", "%8", "%9", "%12")
                         "s_nop 0\n v_cmp_ngt_f32_e64 s[20:21], %21, %6\n s_nop 1\n"
                         "v_cndmask_b32_e64 %6, 0, %6, s[20:21]\n"
                         EVAL_ACC("%6", "%3", "%4", "%5")
                         : "+v"(r0), "+v"(g0), "+v"(b0), "+v"(r1), "+v"(g1), "+v"(b1), "+v"(d0), "+v"(d1),
                           "+v"(d2), "+v"(d3)
                         : "v"(px), "v"(py), "v"(ax), "v"(ay), "v"(az), "v"(aw), "v"(bx), "v"(by),
                           "v"(bz), "v"(bw), "v"(cb), "v"(amin), "v"(huge), "v"(bias)
                         : "vcc", "s20", "s21");
        else if (FORM == 1)
            asm volatile(EVAL_HEAD("%6", "%7", "%8", "%9", "%12")
                         "s_nop 0\n v_cmpx_le_f32 vcc, %21, %6\n"
                         EVAL_ACC("%6", "%0", "%1", "%2")
                         "s_mov_b64 exec, -1\n"
                         EVAL_HEAD("%6", "%7", "%8", "%9", "%12")
                         "s_nop 0\n v_cmpx_le_f32 vcc, %21, %6\n"
                         EVAL_ACC("%6", "%3", "%4", "%5")
                         "s_mov_b64 exec, -1\n"
                         : "+v"(r0), "+v"(g0), "+v"(b0), "+v"(r1), "+v"(g1), "+v"(b1), "+v"(d0), "+v"(d1),
                           "+v"(d2), "+v"(d3)
                         : "v"(px), "v"(py), "v"(ax), "v"(ay), "v"(az), "v"(aw), "v"(bx), "v"(by),
                           "v"(bz), "v"(bw), "v"(cb), "v"(amin), "v"(huge), "v"(bias)
                         : "vcc", "s20", "s21");
        else
            asm volatile(EVAL_HEAD("%6", "%7", "%8", "%9", "%12")
                         "s_nop 0\n v_fma_f32 %7, %6, %22, %23 clamp\n v_mul_f32 %6, %6, %7\n"
                         EVAL_ACC("%6", "%0", "%1", "%2")
                         EVAL_HEAD("%6", "%7", "%8", "%9", "%12")
                         "s_nop 0\n v_fma_f32 %7, %6, %22, %23 clamp\n v_mul_f32 %6, %6, %7\n"
                         EVAL_ACC("%6", "%3", "%4", "%5")
                         : "+v"(r0), "+v"(g0), "+v"(b0), "+v"(r1), "+v"(g1), "+v"(b1), "+v"(d0), "+v"(d1),
                           "+v"(d2), "+v"(d3)
                         : "v"(px), "v"(py), "v"(ax), "v"(ay), "v"(az), "v"(aw), "v"(bx), "v"(by),
                           "v"(bz), "v"(bw), "v"(cb), "v"(amin), "v"(huge), "v"(bias)
                         : "vcc", "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + g0 + b0 + r1 + g1 + b1;
}

typedef void (*kern_t)(float *, int, float, float);

static int run(const char *name, kern_t k, float *d_out, int cus, int waves_per_simd, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = cus * waves_per_simd;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 0.001f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr = 32.0 * iters * blocks * 4;
    const double rate = instr / (best * 1e-3);
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4e, "
           "\"simd_cycles_per_instr_at_2.4GHz\": %.2f}\n",
           name, waves_per_simd, best, rate, (double)cus * 4 * 2.4e9 / rate);
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * cus * 8));
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 (sgpr operand)", k_fma_sgpr},
        {"v_mul_f32", k_mul}, {"v_mul_f32 (inline constant)", k_mul_lit}, {"v_add_f32", k_add},
        {"v_sub_f32", k_sub}, {"v_min_f32", k_min}, {"v_max_f32", k_max}, {"v_med3_f32", k_med3},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr pair", k_cmp_sgpr},
        {"v_mov_b32", k_mov}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl},
        {"v_cvt_f32_i32", k_cvt}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp},
        {"v_mov_b32_dpp", k_mov_dpp}, {"v_add_f32_dpp", k_add_dpp}, {"v_mad_u32_u24", k_mad_u24},
        {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask_s}, {"v_fma_f32 clamp", k_fma_clamp},
        {"v_mul_f32 (sgpr operand)", k_mul_sgpr}, {"v_fmamk_f32 (literal)", k_fmamk},
        {"v_fmac_f32 (literal)", k_fmac_lit}, {"v_and_b32 (literal)", k_and_lit},
        {"v_mul_f32_e64 (neg modifier)", k_mul_neg}, {"v_sub_f32_e64 clamp", k_sub_clamp},
        {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24},
        {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_add_co_u32", k_add_co},
        {"v_addc_co_u32", k_addc_co}, {"v_bfe_u32", k_bfe}, {"v_floor_f32", k_floor},
        {"v_sqrt_f32", k_sqrt}, {"v_log_f32", k_log}, {"v_max_i32", k_max_i32}, {"v_or_b32", k_or},
        {"v_sub_u32", k_sub_u32},
    };
    for (auto &e : ks)
        for (int w : {2, 8})
            if (run(e.name, e.k, d_out, cus, w, 4000)) return 1;
    struct { const char *name; kern_t k; } k64[] = {
        {"v_mad_u64_u32", k_mad_u64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshlrev_b64", k_lshl_b64}};
    for (auto &e : k64)
        for (int w : {2, 8}) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            const int blocks = cus * w, iters = 4000;
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 0.001f);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double rate = 16.0 * iters * blocks * 4 / (best * 1e-3);
            printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4e, "
                   "\"simd_cycles_per_instr_at_2.4GHz\": %.2f}\n",
                   e.name, w, best, rate, (double)cus * 4 * 2.4e9 / rate);
            fflush(stdout);
        }
    // (the evaluation kernels: 2 evaluations per trip; the rate is per 32 "instructions" of the
    // run() bookkeeping, i.e. multiply the cycles by 32 / 2 for cycles per evaluation)
    const float amin = 1.f / 255.f, below = nextafterf(amin, 0.f);
    struct { const char *name; kern_t k; } es[] = {
        {"eval: cmp + cndmask", (kern_t)k_eval<0>}, {"eval: cmpx / exec", (kern_t)k_eval<1>},
        {"eval: fma clamp + mul", (kern_t)k_eval<2>}};
    for (auto &e : es)
        for (int w : {2, 4, 8}) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            const int blocks = cus * w, iters = 20000;
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, iters, amin, below);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double evals = 2.0 * iters * blocks * 4;
            printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_evals_per_s\": %.4e, "
                   "\"simd_cycles_per_eval_at_2.4GHz\": %.1f}\n",
                   e.name, w, best, evals / (best * 1e-3), (double)cus * 4 * 2.4e9 / (evals / (best * 1e-3)));
            fflush(stdout);
        }
    return 0;
}
