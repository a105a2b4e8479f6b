// Hardware probe (dev tool, round 5): issue cost of the PACKED-f32 and 64-bit-DPP instructions on gfx950, alone and next to MFMAs
// -- the same harness as probe_rates.hip (4 waves per SIMD, 8 independent register chains; shader cycles per instruction per
// SIMD).  Question behind it: does v_pk_fma_f32 cost one issue slot or two (is replacing two v_fma_f32 by one a gain)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define KINDS(X)                                                                                   \
    X(0, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %1")                                             \
    X(1, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1")                                                 \
    X(2, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1")                                                 \
    X(3, "v_mov_b64_dpp row_newbcast", "v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
    X(4, "v_mov_b64", "v_mov_b64 %0, %1")                                                           \
    X(5, "v_pk_fma_f32 op_sel bcast", "v_pk_fma_f32 %0, %0, %1, 1.0 op_sel_hi:[1,0,0]")            \
    X(6, "v_fma_f32 x2 (reference)", "v_fma_f32 %0, %0, %1, %1")

template <int KIND>
__device__ __forceinline__ void valu(f2 &x, f2 c) {
#define X(id, name, text) \
    if (KIND == id) asm volatile(text : "+v"(x) : "v"(c));
    KINDS(X)
#undef X
}
template <>
__device__ __forceinline__ void valu<6>(f2 &x, f2 c) {
    float a = x[0], b = x[1];
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c[0]));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(b) : "v"(c[1]));
    x[0] = a, x[1] = b;
}

template <int KIND, bool MFMA>
__global__ __launch_bounds__(1024) void k(int iters, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 A, B;
    for (int e = 0; e < 8; ++e) A[e] = (_Float16)(0.01f * (lane + e)), B[e] = (_Float16)(0.02f * (lane - e));
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f2 x[8];
    for (int r = 0; r < 8; ++r) x[r] = (f2){1.0f + 0.01f * (lane + r), 1.0f - 0.01f * r};
    const f2 c = {0.999f, 0.998f};
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
    for (int r = 0; r < 8; ++r) s += x[r][0] + x[r][1];
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
