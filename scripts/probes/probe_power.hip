// Hardware probe (dev tool): what the chip sustains on v_mfma_f32_32x32x16_f16 once it is power-limited.
// 256 workgroups x 8 waves (2 per SIMD), every wave a stream of MFMAs on four accumulators for ~60 ms per variant:
//   operands: zeros | random binary16;   beside every MFMA: nothing | one ds_read_b128 | one ds_read_b128 + four v_fma_f32
// Prints TFLOP/s (wall clock, HIP events).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const h8 *src, int iters, long long *cyc, float *sink) {
    __shared__ h8 lds[512];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = src[threadIdx.x + 512];
    h8 A = src[threadIdx.x], B = src[(threadIdx.x * 7 + 3) & 1023];
    __syncthreads();
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[4] = {1.f, 2.f, 3.f, 4.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE >= 1) {
                h8 t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((lane + u * 8) & 511) * 16u));
                asm volatile("s_waitcnt lgkmcnt(0)");
                A = t;
            }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(A), "v"(B));
            if (MODE >= 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v]) : "v"(0.999f));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, const h8 *src, long long *cyc, float *sink) {
    const int blocks = 256, iters = 60000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 512>>>(src, 2000, cyc, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(src, iters, cyc, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    (void)hipMemcpy(&c, cyc + 100, sizeof(c), hipMemcpyDeviceToHost);
    const double mfmas = (double)blocks * 8 * iters * 8;
    (void)c;  // (s_memtime ticks at a fixed rate on this part: no clock estimate from it)
    printf("%-44s %7.1f ms  %7.1f TFLOP/s\n", what, ms, mfmas * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    h8 *src;
    long long *cyc;
    float *sink;
    (void)hipMalloc(&src, 1024 * sizeof(h8));
    (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMalloc(&sink, 256 * 512 * 4);
    _Float16 h[8192];
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (int i = 0; i < 8192; ++i) h[i] = pass ? (_Float16)((rand() % 2001 - 1000) / 1000.0f) : (_Float16)0.f;
        (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
        printf("== operands: %s\n", pass ? "random binary16 in [-1, 1]" : "zeros");
        run<0>("MFMA only", src, cyc, sink);
        run<1>("MFMA + one ds_read_b128 each", src, cyc, sink);
        run<2>("MFMA + ds_read_b128 + 4 v_fma_f32 each", src, cyc, sink);
    }
    return 0;
}
