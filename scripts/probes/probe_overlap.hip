// Hardware probe (dev tool): do MFMAs of one wave overlap with VALU work of the other wave on the same SIMD?
// 8 waves per workgroup (wave w and w+4 share a SIMD), one workgroup per CU.  Modes:
//   0: every wave: 40 MFMAs (two accumulator chains) per iteration
//   1: every wave: the VALU mix of the K1 epilogue (16 x {fma, sub, exp2, add} + max3s) per iteration
//   2: waves 0-3 MFMAs, waves 4-7 VALU mix  (perfect overlap -> max of 0 and 1; none -> their sum)
//   3: every wave MFMAs then VALU mix       (what K1 does per chunk, without the phase offset)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void do_mfma(f16v &a0, f16v &a1, const h8 (&A)[4], const h8 (&B)[4]) {
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i & 3], B[i & 3], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(i + 1) & 3], B[(i + 2) & 3], a1, 0, 0, 0);
    }
}
__device__ __forceinline__ void do_valu(float (&y)[16], float &s, float &m) {
    float mm = m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        y[r] = __builtin_fmaf(y[r], 1.0001f, 0.001f);
        mm = fmaxf(mm, y[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s += __builtin_amdgcn_exp2f(y[r] - m);
    m = mm * 0.5f;
}

__global__ __launch_bounds__(512) void k(int mode, int iters, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 A[4], B[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) A[i][e] = (_Float16)(0.01f * (lane + e + i)), B[i][e] = (_Float16)(0.02f * (lane - e + i));
    f16v a0, a1;
    for (int r = 0; r < 16; ++r) a0[r] = 0.f, a1[r] = 0.f;
    float y[16], s = 0.f, m = 0.f;
    for (int r = 0; r < 16; ++r) y[r] = 0.01f * (lane + r);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const bool mf = (mode == 0) || (mode == 3) || (mode == 2 && wave < 4);
        const bool va = (mode == 1) || (mode == 3) || (mode == 2 && wave >= 4);
        if (mf) do_mfma(a0, a1, A, B);
        if (va) do_valu(y, s, m);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = s + m;
    for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + y[r];
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    long long *out;
    float *sink;
    const int blocks = 256, iters = 200;
    hipMalloc(&out, blocks * 8 * sizeof(long long));
    hipMalloc(&sink, blocks * 512 * sizeof(float));
    long long h[8];
    for (int mode = 0; mode < 4; ++mode) {
        k<<<blocks, 512>>>(mode, iters, out, sink);
        hipDeviceSynchronize();
        k<<<blocks, 512>>>(mode, iters, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(h, out + 100 * 8, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d cycles per iteration, waves 0..7:", mode);
        for (int w = 0; w < 8; ++w) printf(" %.0f", (double)h[w] / iters);
        printf("\n");
    }
    return 0;
}
