// Hardware probe (dev tool, round 4): what a LONE wave pays per instruction on gfx950 -- dependent chains against independent
// streams -- for the instruction mix of the linear-domain sweep step (rnnt_lin_kernels.hip).  One workgroup of 64 threads per
// CU-sized slot; s_memtime counts at 100 MHz, so every figure is reported in ns per instruction (x 2.4 ~ shader clocks).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/probe_lat.hip -o scripts/probes/probe_lat && scripts/probes/probe_lat
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define PROBE(NAME, PER, BODY)                                                              \
    __global__ __launch_bounds__(64) void NAME(long long *out, float *sink, int iters) {   \
        __shared__ float lds[4096];                                                        \
        float a = 1.0f + threadIdx.x * 1e-3f, b = 0.999f, c = 1e-3f, d = 1.0f, e = 1.1f, f = 0.9f, g = 1.01f, h = 0.99f;  \
        int sh = 0;                                                                        \
        unsigned addr = threadIdx.x * 8; const unsigned addr3 = threadIdx.x * 24; f32x4 q4 = {1.f, 1.f, 1.f, 1.f};                                                   \
        f32x2 q = {1.f, 1.f}, q2 = {1.f, 1.f}, q3 = {1.f, 1.f}; const f32x2 qc = {b, c}; const f32x3 st3 = {g, h, b};                                                  \
        float *gp = sink + threadIdx.x * 4;                                                \
        for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0.f;                         \
        __syncthreads();                                                                   \
        const long long t0 = __builtin_amdgcn_s_memtime();                                 \
        for (int it = 0; it < iters; ++it) {                                               \
            REP64(BODY)                                                                    \
        }                                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        const long long t1 = __builtin_amdgcn_s_memtime();                                 \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                   \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + q.x + q.y + q2.x + q2.y + q3.x + q3.y + q4.x + q4.w + (float)sh + (float)addr; \
    }                                                                                      \
    static const int NAME##_per = PER;

// 1 dependent fma
PROBE(p_fma_dep, 1, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
// 2 / 4 independent chains
PROBE(p_fma_2, 2, asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));)
PROBE(p_fma_4, 4, asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
PROBE(p_mul_dep, 1, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
PROBE(p_ldexp_dep, 1, asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a) : "v"(sh));)
PROBE(p_frexp_dep, 1, asm volatile("v_frexp_exp_i32_f32 %0, %1\n\tv_cvt_f32_i32 %1, %0" : "+v"(sh), "+v"(a));)  // 2 instr
PROBE(p_max3_dep, 1, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
// mov_dpp chain (each needs 2 wait states after the VALU write of its source)
PROBE(p_dpp_dep, 1, asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
// the cross-lane part of the alpha step: ldexp -> mul -> dpp -> fma, dependent
PROBE(p_cross, 4, asm volatile("v_ldexp_f32 %1, %0, %3\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f32 %0, %0, %2, %1" : "+v"(a), "+v"(d) : "v"(b), "v"(sh));)
// add with a DPP operand instead of mov + fma
PROBE(p_cross_adddpp, 4, asm volatile("v_ldexp_f32 %1, %0, %3\n\tv_mul_f32 %1, %1, %2\n\tv_mul_f32 %0, %0, %2\n\ts_nop 1\n\tv_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(d) : "v"(b), "v"(sh));)
// packed f32
PROBE(p_pkfma_dep, 1, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q) : "v"(q));)
PROBE(p_pkmul_ind, 2, asm volatile("v_pk_mul_f32 %0, %2, %2\n\tv_pk_mul_f32 %1, %2, %2" : "=v"(q), "=v"(q2) : "v"(qc));)
// LDS: dependent ds_read_b64 (address from the data: 0) = latency; and issue cost of 3 independent reads + wait
PROBE(p_lds_lat, 1, asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %1, %1, %2" : "=v"(q), "+v"(addr) : "v"(0));)
PROBE(p_lds_3, 3, asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(q), "=v"(q2), "=v"(q3) : "v"(addr));)
// the sweep's three 8-byte reads per diagonal as one 16-byte + one 8-byte read at a 24-byte lane stride (8-byte aligned only)
PROBE(p_lds_b128_b64_unaligned, 2, asm volatile("ds_read_b128 %0, %2 offset:8\n\tds_read_b64 %1, %2 offset:24\n\ts_waitcnt lgkmcnt(0)" : "=v"(q4), "=v"(q2) : "v"(addr3));)
PROBE(p_lds_3x_b64_stride24, 3, asm volatile("ds_read_b64 %0, %3 offset:8\n\tds_read_b64 %1, %3 offset:16\n\tds_read_b64 %2, %3 offset:24\n\ts_waitcnt lgkmcnt(0)" : "=v"(q), "=v"(q2), "=v"(q3) : "v"(addr3));)
// store issue: one dwordx3 store + 4 independent fmas
PROBE(p_store_fma4, 5, asm volatile("global_store_dwordx3 %6, %4, off sc1\n\tv_fma_f32 %0, %0, %7, %8\n\tv_fma_f32 %1, %1, %7, %8\n\tv_fma_f32 %2, %2, %7, %8\n\tv_fma_f32 %3, %3, %7, %8" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(st3), "v"(0), "v"(gp), "v"(b), "v"(c) : "memory");)
PROBE(p_store_only, 1, asm volatile("global_store_dwordx3 %1, %0, off sc1" :: "v"(st3), "v"(gp) : "memory");)

// clocks ramp up under load only: keep every CU busy for ~0.3 s before and between the probes
__global__ void warm(float *sink, int n) {
    float a = threadIdx.x;
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    sink[blockIdx.x * 256 + threadIdx.x] = a;
}

template <typename F>
static void run(const char *name, F kern, int per) {
    long long *out;
    float *sink;
    hipMalloc(&out, 1024 * sizeof(long long));
    hipMalloc(&sink, 4 << 20);
    const int iters = 512;
    hipLaunchKernelGGL(warm, dim3(1024), dim3(256), 0, 0, sink, 4000000);
    // 1024 workgroups of one wave = one wave per SIMD on every CU (a lone wave per SIMD, and enough activity to keep the clocks up)
    hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, out, sink, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, out, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    static long long h[1024];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    long long best = h[0];
    for (int i = 1; i < 1024; ++i) best = h[i] < best ? h[i] : best;
    const double ns = best * 10.0 / ((double)iters * 64 * per);
    const double wall = ms * 1e6 / ((double)iters * 64 * per);  // whole kernel by HIP events (includes launch, ~5 us)
    printf("%-18s %8.2f s_memtime ticks per instruction | %7.2f ns per instruction by HIP events (~%5.1f clocks at 2.4 GHz)\n", name, ns / 10.0, wall, wall * 2.4);
    hipFree(out);
    hipFree(sink);
}
#define RUN(N) run(#N, N, N##_per)

int main() {
    RUN(p_fma_dep);
    RUN(p_fma_2);
    RUN(p_fma_4);
    RUN(p_mul_dep);
    RUN(p_ldexp_dep);
    RUN(p_frexp_dep);
    RUN(p_max3_dep);
    RUN(p_dpp_dep);
    RUN(p_cross);
    RUN(p_cross_adddpp);
    RUN(p_pkfma_dep);
    RUN(p_pkmul_ind);
    RUN(p_lds_lat);
    RUN(p_lds_3);
    RUN(p_lds_b128_b64_unaligned);
    RUN(p_lds_3x_b64_stride24);
    RUN(p_store_fma4);
    RUN(p_store_only);
    return 0;
}
