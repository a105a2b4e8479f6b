// Hardware probe (dev tool) for a linear-domain lattice sweep on a lone wave:
//  (1) are the sticky exception bits of TRAPSTS set by VALU underflow / overflow with traps disabled, and readable with
//      s_getreg_b32 (would make "run a chunk in linear arithmetic, redo it in the log domain if anything under/overflowed" free);
//  (2) shader cycles per anti-diagonal of a K = 3 linear-domain step (3 mul + 3 fma + DPP + scale, edge weights read from LDS,
//      log2 + offset of the three values for the store, renormalisation of the lane's exponent every second diagonal) against
//      the log-domain step (3 x {sub, exp2, max, add, log2, add} + packed adds), both as a lone wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

__device__ __forceinline__ unsigned trapsts() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(v));
    return v;
}
__device__ __forceinline__ void trapsts_clear() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0"); }

__global__ void k_flags(unsigned *out) {
    volatile float tiny = 1.0e-30f, big = 1.0e30f, one = 1.0f, zero = 0.0f;
    trapsts_clear();
    out[0] = trapsts();
    float a = one * one;  // exact
    asm volatile("" ::"v"(a));
    out[1] = trapsts();
    float b = tiny * tiny;  // underflow
    asm volatile("" ::"v"(b));
    out[2] = trapsts();
    trapsts_clear();
    float c = big * big;  // overflow
    asm volatile("" ::"v"(c));
    out[3] = trapsts();
    trapsts_clear();
    float d = zero * tiny;  // exact zero
    asm volatile("" ::"v"(d));
    out[4] = trapsts();
    trapsts_clear();
    float e = __builtin_amdgcn_logf(zero);  // log2(0) = -inf
    asm volatile("" ::"v"(e));
    out[5] = trapsts();
    trapsts_clear();
    float f = one * 0.3f;  // inexact only
    asm volatile("" ::"v"(f));
    out[6] = trapsts();
    trapsts_clear();
    float g = __builtin_amdgcn_ldexpf(tiny, -100);  // underflow in ldexp
    asm volatile("" ::"v"(g));
    out[7] = trapsts();
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dpp_shr(float x, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE 0: log domain (as the shipping sweep: stage-major), MODE 1: linear domain
template <int MODE>
__global__ __launch_bounds__(64) void k_sweep(int ndiag, long long *cyc, float *sink, float *gout) {
    __shared__ f2 w[16][192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * 192; i += 64) {
        const float pb = 0.3f + 0.001f * (i % 97), pl = 0.2f + 0.001f * (i % 53);
        w[i / 192][i % 192] = MODE ? f2{pb, pl} : f2{log2f(pb), log2f(pl)};
    }
    __syncthreads();
    float a[3] = {MODE ? 1.0f : 0.0f, MODE ? 0.5f : -1.f, MODE ? 0.25f : -2.f};
    float e = 0.f, scale_left = 1.0f, edge = 0.f, off = 0.f;
    float *row = gout + lane * 3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < ndiag; ++n) {
        const f2 *wr = &w[n & 15][lane * 3];
        const f2 w0 = wr[0], w1 = wr[1], w2 = wr[2];
        float o0, o1, o2;
        if (MODE == 0) {
            FENCE();
            const f2 d2 = f2{a[2], a[2]} + w2, d1 = f2{a[1], a[1]} + w1, d0 = f2{a[0], a[0]} + w0;
            FENCE();
            edge = dpp_shr(d2[1], edge);
            const float u[3] = {d0[0], d1[0], d2[0]}, l[3] = {edge, d0[1], d1[1]};
            float dd[3], ee[3], mm[3];
            FENCE();
            for (int j = 2; j >= 0; --j) dd[j] = u[j] - l[j];
            FENCE();
            for (int j = 2; j >= 0; --j) ee[j] = __builtin_amdgcn_exp2f(-fabsf(dd[j]));
            FENCE();
            for (int j = 2; j >= 0; --j) mm[j] = fmaxf(u[j], l[j]);
            FENCE();
            for (int j = 2; j >= 0; --j) ee[j] = 1.0f + ee[j];
            FENCE();
            for (int j = 2; j >= 0; --j) ee[j] = __builtin_amdgcn_logf(ee[j]);
            FENCE();
            for (int j = 2; j >= 0; --j) a[j] = mm[j] + ee[j];
            FENCE();
            if ((n & 7) == 7) {
                const float mi = rintf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[1]), 20)));
                a[0] -= mi, a[1] -= mi, a[2] -= mi, off += mi;
            }
            o0 = a[0], o1 = a[1], o2 = a[2];
        } else {
            FENCE();
            const float t2 = a[2] * w2[1], t1 = a[1] * w1[1], t0v = a[0] * w0[1];
            FENCE();
            edge = dpp_shr(t2, edge);
            const float x = edge * scale_left;
            FENCE();
            a[2] = fmaf(a[2], w2[0], t1);
            a[1] = fmaf(a[1], w1[0], t0v);
            a[0] = fmaf(a[0], w0[0], x);
            FENCE();
            // log2 of the three values for the store (off the dependency chain of the next step)
            float g0 = __builtin_amdgcn_logf(a[0]), g1 = __builtin_amdgcn_logf(a[1]), g2 = __builtin_amdgcn_logf(a[2]);
            FENCE();
            g0 += e, g1 += e, g2 += e;
            if (n & 1) {  // renormalise the lane: max -> 2^0
                const float mx = fmaxf(fmaxf(g0, g1), g2);
                const float sh = rintf(mx);
                const int ish = (int)sh - (int)e;  // exponent to take out of the mantissas
                (void)ish;
                const float de = e - sh;  // <= 0 typically
                a[0] = __builtin_amdgcn_ldexpf(a[0], (int)de), a[1] = __builtin_amdgcn_ldexpf(a[1], (int)de),
                a[2] = __builtin_amdgcn_ldexpf(a[2], (int)de);
                e = sh;
                const float el = dpp_shr(e, 0.f);
                scale_left = __builtin_amdgcn_ldexpf(1.0f, (int)(el - e));
            }
            if ((n & 7) == 7) {
                const float mi = rintf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(g1), 20)));
                g0 -= mi, g1 -= mi, g2 -= mi, e -= mi, off += mi;
            }
            o0 = fmaxf(g0, -1.0e30f), o1 = fmaxf(g1, -1.0e30f), o2 = fmaxf(g2, -1.0e30f);
        }
        typedef float f3 __attribute__((ext_vector_type(3)));
        *(f3 *)(row + (size_t)(n & 63) * 192) = f3{o0, o1, o2};
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = a[0] + a[1] + a[2] + e + off;
}

int main() {
    unsigned *out;
    hipMalloc(&out, 64);
    k_flags<<<1, 1>>>(out);
    unsigned h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"after clear", "1*1 (exact)", "tiny*tiny (underflow)", "big*big (overflow)", "0*tiny (exact zero)",
                            "log2(0)", "1*0.3 (inexact)", "ldexp(tiny,-100)"};
    for (int i = 0; i < 8; ++i) printf("TRAPSTS.EXCP %-26s 0x%03x\n", names[i], h[i]);
    long long *cyc;
    float *sink, *gout;
    hipMalloc(&cyc, 64 * 8);
    hipMalloc(&sink, 64 * 64 * 4);
    hipMalloc(&gout, 64 * 192 * 4 * 64);
    const int nd = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k_sweep<0><<<64, 64>>>(nd, cyc, sink, gout);
            else k_sweep<1><<<64, 64>>>(nd, cyc, sink, gout);
            hipDeviceSynchronize();
        }
        long long c[64];
        hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.1f cycles per diagonal (lone wave, 64 workgroups)\n", mode, mode ? "linear" : "log", (double)c[5] / nd);
    }
    return 0;
}
