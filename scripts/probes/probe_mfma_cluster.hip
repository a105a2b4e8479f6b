// Hardware probe (dev tool, round 5): does it matter WHERE a wave's MFMAs sit in its instruction stream?  Each wave repeats a
// group of 36 independent v_fma_f32 + 6 v_mfma_f32_32x32x16_f16 (the mix of a row of joint_bwd_kernel's consumer, roughly):
//   mode 0  36 VALU, then the 6 MFMAs back to back on ONE accumulator (a dependent chain: what the joint kernels do)
//   mode 1  36 VALU, then the 6 MFMAs back to back on TWO alternating accumulators
//   mode 2  six times (6 VALU + 1 MFMA), one accumulator
//   mode 3  six times (6 VALU + 1 MFMA), two alternating accumulators
//   mode 4  36 VALU only          mode 5  the 6 chained MFMAs only
// and the row of the backward consumer as it is (144 VALU + 12 MFMAs on two chains), per ROW instead of per group:
//   mode 6  48 VALU | 6 chained MFMAs | 6 x (7 VALU + MFMA) | 54 VALU      (what the compiler emits today)
//   mode 7  12 x (12 VALU + 1 MFMA)                                        (everything interleaved)
//   mode 8  144 VALU only         mode 9  12 MFMAs only (two chains of six)
//   mode 10 128 v_fma_f32 + 16 v_rcp_f32 only      mode 11 = mode 7 with 16 of its VALU being v_rcp_f32
// W waves per SIMD (workgroup = 4 W waves, one per CU).  Prints shader clocks per group and SIMD-resident wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 A, B;
    for (int e = 0; e < 8; ++e) A[e] = (_Float16)(0.01f * (lane + e)), B[e] = (_Float16)(0.02f * (lane - e));
    f16v acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
    float x[8];
    for (int r = 0; r < 8; ++r) x[r] = 1.0f + 0.01f * (lane + r);
    const float c = 0.999f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#define VALU(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(i) & 7]) : "v"(c))
#define MF0 asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(A), "v"(B))
#define MF1 asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(A), "v"(B))
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 1 || MODE == 4) {
#pragma unroll
            for (int v = 0; v < 36; ++v) VALU(v);
        }
        if (MODE == 0 || MODE == 5) { MF0; MF0; MF0; MF0; MF0; MF0; }
        if (MODE == 1) { MF0; MF1; MF0; MF1; MF0; MF1; }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
#pragma unroll
                for (int v = 0; v < 6; ++v) VALU(g * 6 + v);
                if (MODE == 3 && (g & 1)) MF1; else MF0;
            }
        }
        if (MODE == 6) {
#pragma unroll
            for (int v = 0; v < 48; ++v) VALU(v);
            MF0; MF0; MF0; MF0; MF0; MF0;
#pragma unroll
            for (int g = 0; g < 6; ++g) {
#pragma unroll
                for (int v = 0; v < 7; ++v) VALU(g * 7 + v);
                MF1;
            }
#pragma unroll
            for (int v = 0; v < 54; ++v) VALU(v);
        }
        if (MODE == 7) {
#pragma unroll
            for (int g = 0; g < 12; ++g) {
#pragma unroll
                for (int v = 0; v < 12; ++v) VALU(g * 12 + v);
                if (g < 6) MF0; else MF1;
            }
        }
        if (MODE == 8) {
#pragma unroll
            for (int v = 0; v < 144; ++v) VALU(v);
        }
        if (MODE == 9) { MF0; MF0; MF0; MF0; MF0; MF0; MF1; MF1; MF1; MF1; MF1; MF1; }
        if (MODE == 10) {  // 128 fma + 16 rcp
#pragma unroll
            for (int v = 0; v < 144; ++v) {
                if (v % 9 == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[v & 7]));
                else VALU(v);
            }
        }
        if (MODE == 11) {  // the row with its 16 reciprocals, interleaved
#pragma unroll
            for (int g = 0; g < 12; ++g) {
#pragma unroll
                for (int v = 0; v < 12; ++v) {
                    if ((g * 12 + v) % 9 == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[v & 7]));
                    else VALU(g * 12 + v);
                }
                if (g < 6) MF0; else MF1;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int r = 0; r < 8; ++r) s += x[r];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

static long long *g_out;
static float *g_sink;

template <int MODE>
static double run(int waves_per_simd) {
    const int iters = 400, blocks = 256, threads = 256 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) {
        k<MODE><<<blocks, threads>>>(iters, g_out, g_sink);
        (void)hipDeviceSynchronize();
    }
    long long h[16];
    (void)hipMemcpy(h, g_out + 100 * 16, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < 4 * waves_per_simd; ++w) mx = h[w] > mx ? h[w] : mx;
    return (double)mx / iters / waves_per_simd;  // clocks per group per wave of the SIMD
}

int main() {
    (void)hipMalloc(&g_out, 256 * 16 * sizeof(long long));
    (void)hipMalloc(&g_sink, 256 * 1024 * sizeof(float));
    const char *names[12] = {"36 VALU + 6 chained MFMA (1 acc)", "36 VALU + 6 MFMA (2 accs)", "6 x (6 VALU + MFMA), 1 acc",
                            "6 x (6 VALU + MFMA), 2 accs", "36 VALU only", "6 chained MFMA only",
                            "row: 48 V | 6 M | 6 x (7 V + M) | 54 V", "row: 12 x (12 V + M)", "row: 144 VALU only", "row: 12 MFMA only", "row: 128 fma + 16 rcp only", "row: 12 x (12 V + M), 16 of the V are v_rcp_f32"};
    for (int w = 1; w <= 4; ++w) {
        const double r[12] = {run<0>(w), run<1>(w), run<2>(w), run<3>(w), run<4>(w), run<5>(w), run<6>(w), run<7>(w), run<8>(w), run<9>(w), run<10>(w), run<11>(w)};
        for (int m = 0; m < 12; ++m) printf("%d wave(s)/SIMD  %-36s %7.1f clocks per group and wave\n", w, names[m], r[m]);
    }
    return 0;
}
