// Hardware probe (dev tool): issue cost of single VALU instructions on gfx950, alone and next to MFMAs.
// 4 waves per SIMD (16-wave workgroup, one per CU), 8 independent register chains; prints shader cycles per instruction
// per SIMD (a) alone and (b) when every 6 of them are followed by one v_mfma_f32_32x32x16_f16 (cost per group - 6 * alone
// = what the MFMA adds: 32 = nothing hidden, ~8 = the MFMA rides along).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define KINDS(X)                                                                              \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1")                                            \
    X(1, "v_add_f32", "v_add_f32 %0, %0, %1")                                                \
    X(2, "v_rcp_f32", "v_rcp_f32 %0, %0")                                                    \
    X(3, "v_exp_f32", "v_exp_f32 %0, %0")                                                    \
    X(4, "v_fma_mix_f32", "v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]")                \
    X(5, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, %1, -1.0, %1 op_sel_hi:[1,0,0]")            \
    X(6, "v_fma_mixhi_f16", "v_fma_mixhi_f16 %0, %1, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]") \
    X(7, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %0, %1")                                  \
    X(8, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0")                                            \
    X(9, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0")                                            \
    X(10, "v_and_b32", "v_and_b32 %0, %0, %1")                                               \
    X(11, "v_sub_f32", "v_sub_f32 %0, %0, %1")                                               \
    X(12, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1")                                         \
    X(13, "v_mov_dpp bcast", "v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
    X(14, "v_add_f32 dpp", "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
    X(15, "v_mul_f32 dpp", "v_mul_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
    X(16, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %1, %1")                                     \
    X(17, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1")                                         \
    X(18, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc")                                  \
    X(19, "v_max_f32", "v_max_f32 %0, %0, %1")                                               \
    X(20, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                               \
    X(21, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %0, %1")                           \
    X(22, "v_sub_f32 sdwa", "v_sub_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
    X(23, "v_mad_mix (fma_mix f16 src x2)", "v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,1,0]")  \
    X(24, "v_log_f32", "v_log_f32 %0, %0")                                                   \
    X(25, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1")                                  \
    X(26, "v_fmac_f32", "v_fmac_f32 %0, %1, %1")                                             \
    X(27, "v_fmac_f32 dpp", "v_fmac_f32_dpp %0, %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf")

template <int KIND>
__device__ __forceinline__ void valu(float &x, float c) {
#define X(id, name, text) \
    if (KIND == id) asm volatile(text : "+v"(x) : "v"(c));
    KINDS(X)
#undef X
}

template <int KIND, bool MFMA>
__global__ __launch_bounds__(1024) void k(int iters, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 A, B;
    for (int e = 0; e < 8; ++e) A[e] = (_Float16)(0.01f * (lane + e)), B[e] = (_Float16)(0.02f * (lane - e));
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[8];
    for (int r = 0; r < 8; ++r) x[r] = 1.0f + 0.01f * (lane + r);
    const float c = 0.999f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
#pragma unroll
            for (int v = 0; v < 6; ++v) valu<KIND>(x[(blk * 6 + v) & 7], c);
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[blk & 3]) : "v"(A), "v"(B));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 8; ++r) s += x[r];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

static long long *g_out;
static float *g_sink;

template <int KIND, bool MFMA>
static double run() {
    const int iters = 300, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        k<KIND, MFMA><<<blocks, 1024>>>(iters, g_out, g_sink);
        (void)hipDeviceSynchronize();
    }
    long long h[16];
    (void)hipMemcpy(h, g_out + 100 * 16, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < 16; ++w) mx = h[w] > mx ? h[w] : mx;
    return (double)mx / (iters * 8.0) / 4.0;  // cycles per group of 6 (+ MFMA) per wave of the SIMD
}

template <int KIND>
static void row(const char *name) {
    const double alone = run<KIND, false>(), both = run<KIND, true>();
    printf("%-34s %5.2f cyc/instr alone | 6 + MFMA: %6.1f cyc  (MFMA adds %5.1f)\n", name, alone / 6.0, both, both - alone);
}

int main() {
    (void)hipMalloc(&g_out, 256 * 16 * sizeof(long long));
    (void)hipMalloc(&g_sink, 256 * 1024 * sizeof(float));
#define X(id, name, text) row<id>(name);
    KINDS(X)
#undef X
    return 0;
}
