// Hardware probe (dev tool): how much VALU work hides beside v_mfma_f32_32x32x16_f16 on one SIMD, as a function of
//   NV     VALU instructions per MFMA,
//   KIND   which VALU instruction (0 v_fma_f32, 1 v_rcp_f32, 2 v_pk_fma_f32, 3 v_mov_b32 DPP row_newbcast, 4 v_cvt_pk_f16_f32 pair),
//   GROUP  MFMAs per block: an iteration is GROUP*NV VALU instructions followed by GROUP MFMAs (GROUP = 1: fine interleave),
//   waves per SIMD (1, 2, 3; workgroup of 4, 8, 12 waves, one workgroup per CU).
// Instruction order is pinned with asm volatile.  Prints shader cycles per MFMA (one MFMA alone: 32).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu(float &x, f2 &p, float c) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (KIND == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(c));
    if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
}

template <int NV, int KIND, int GROUP, bool MFMA>
__global__ __launch_bounds__(768) void k(int iters, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 A, B;
    for (int e = 0; e < 8; ++e) A[e] = (_Float16)(0.01f * (lane + e)), B[e] = (_Float16)(0.02f * (lane - e));
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[8];
    f2 pk[8];
    for (int r = 0; r < 8; ++r) x[r] = 1.0f + 0.01f * (lane + r), pk[r] = f2{x[r], x[r]};
    const float c = 0.999f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 8 / GROUP; ++blk) {
#pragma unroll
            for (int v = 0; v < NV * GROUP; ++v) valu<KIND>(x[(blk * NV * GROUP + v) & 7], pk[(blk * NV * GROUP + v) & 7], c);
            if (MFMA) {
#pragma unroll
                for (int g = 0; g < GROUP; ++g)
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(blk * GROUP + g) & 3]) : "v"(A), "v"(B));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 8; ++r) s += x[r] + pk[r][0] + pk[r][1];
    sink[blockIdx.x * 768 + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 12 + wave] = t1 - t0;
}

static long long *g_out;
static float *g_sink;

template <int NV, int KIND, int GROUP, bool MFMA>
static double run(int wps) {
    const int iters = 400, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        k<NV, KIND, GROUP, MFMA><<<blocks, 256 * wps>>>(iters, g_out, g_sink);
        hipDeviceSynchronize();
    }
    long long h[12];
    hipMemcpy(h, g_out + 100 * 12, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < 4 * wps; ++w) mx = h[w] > mx ? h[w] : mx;
    return (double)mx / (iters * 8.0) / wps;  // cycles per MFMA slot per wave-of-this-SIMD
}

template <int NV, int KIND, int GROUP>
static void row(const char *kind) {
    printf("%-8s NV=%2d GROUP=%d |", kind, NV, GROUP);
    for (int wps = 1; wps <= 3; ++wps) {
        const double both = run<NV, KIND, GROUP, true>(wps), alone = run<NV, KIND, GROUP, false>(wps);
        printf("  wps=%d: with mfma %6.1f  valu alone %6.1f  hidden %5.1f |", wps, both, alone, 32.0 + alone - both);
    }
    printf("\n");
}

template <int KIND>
static void kind_rows(const char *kind) {
    row<2, KIND, 1>(kind);
    row<4, KIND, 1>(kind);
    row<6, KIND, 1>(kind);
    row<8, KIND, 1>(kind);
    row<12, KIND, 1>(kind);
    row<16, KIND, 1>(kind);
    row<12, KIND, 4>(kind);
    row<12, KIND, 8>(kind);
}

int main() {
    hipMalloc(&g_out, 256 * 12 * sizeof(long long));
    hipMalloc(&g_sink, 256 * 768 * sizeof(float));
    printf("cycles per (MFMA + NV VALU) slot, per wave sharing the SIMD (the SIMD's cost per slot); 'hidden' = 32 + alone - with\n");
    row<0, 0, 1>("none");
    kind_rows<0>("fma");
    kind_rows<1>("rcp");
    kind_rows<2>("pk_fma");
    kind_rows<3>("dpp");
    kind_rows<4>("cvt_pk");
    return 0;
}
