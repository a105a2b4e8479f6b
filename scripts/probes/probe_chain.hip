// Hardware probe (dev tool): cycles per diagonal of a LONE wave for the two candidate forms of the lattice recurrence,
// K = 3 columns per lane, weights read from LDS one row ahead, one store per diagonal (as the sweep kernels do):
//   mode 0: log2 domain, f32:  a_j <- max(u, l) + log2(1 + 2^-|u - l|),  u = a_j + wb_j,  l = a_{j-1} + wl_{j-1}
//   mode 1: linear domain, f64: a_j <- a_j * pb_j + a_{j-1} * pl_{j-1}   (weights f32 in LDS, converted on the fly),
//           rescaled by a power of two every 8 diagonals
//   mode 2: linear domain, f32 (range-unsafe; only to see what the f64 arithmetic costs)
// s_memtime counts at 100 MHz on gfx950 (constant-rate counter), so the result is reported in ns per diagonal.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int K = 3, ROWS = 64;

__device__ __forceinline__ float dpp_shr(float x, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double dpp_shr64(double x, double old) {
    const long long xi = __double_as_longlong(x), oi = __double_as_longlong(old);
    const int lo = __builtin_amdgcn_update_dpp((int)oi, (int)xi, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(oi >> 32), (int)(xi >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__global__ __launch_bounds__(64) void k(int mode, int steps, long long *out, float *sink, double *sink64) {
    __shared__ f32x2 w[ROWS][64 * K];
    const int lane = threadIdx.x;
    for (int r = 0; r < ROWS; ++r)
        for (int j = 0; j < K; ++j) {
            const float x = 0.3f + 0.001f * ((lane * K + j + r) & 15);
            w[r][lane * K + j] = (mode == 0) ? f32x2{-x, -x - 0.5f} : f32x2{__builtin_amdgcn_exp2f(-x), __builtin_amdgcn_exp2f(-x - 0.5f)};
        }
    __syncthreads();
    float *dst = sink + (size_t)blockIdx.x * 64 * K * 2048;
    double *dst64 = sink64 + (size_t)blockIdx.x * 64 * K * 2048;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        float a[K], edge = -1e30f;
        for (int j = 0; j < K; ++j) a[j] = (lane == 0 && j == 0) ? 0.f : -1e30f;
        f32x2 cur[K], nxt[K];
        for (int j = 0; j < K; ++j) cur[j] = w[0][lane * K + j];
#pragma unroll 16
        for (int s = 0; s < steps; ++s) {
            for (int j = 0; j < K; ++j) nxt[j] = w[(s + 1) & (ROWS - 1)][lane * K + j];
            f32x2 de[K];
            for (int j = 0; j < K; ++j) de[j] = f32x2{a[j], a[j]} + cur[j];
            edge = dpp_shr(de[K - 1][1], edge);
            for (int j = 0; j < K; ++j) {
                const float u = de[j][0], l = (j == 0) ? edge : de[j - 1][1];
                a[j] = fmaxf(u, l) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(u - l)));
            }
            if ((s & 7) == 7) {
                const float m = rintf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[0]), (s >> 3) & 63)));
                if (m > -1e29f)
                    for (int j = 0; j < K; ++j) a[j] -= m;
            }
            for (int j = 0; j < K; ++j) __builtin_nontemporal_store(a[j], dst + ((size_t)(s & 2047) * 64 + lane) * K + j);
            for (int j = 0; j < K; ++j) cur[j] = nxt[j];
        }
        sink[lane] = a[0] + a[1] + a[2];
    } else if (mode == 1) {
        double a[K], edge = 0.0;
        for (int j = 0; j < K; ++j) a[j] = (lane == 0 && j == 0) ? 1.0 : 0.0;
        f32x2 cur[K], nxt[K];
        for (int j = 0; j < K; ++j) cur[j] = w[0][lane * K + j];
#pragma unroll 16
        for (int s = 0; s < steps; ++s) {
            for (int j = 0; j < K; ++j) nxt[j] = w[(s + 1) & (ROWS - 1)][lane * K + j];
            double e[K];
            for (int j = 0; j < K; ++j) e[j] = a[j] * (double)cur[j][1];
            edge = dpp_shr64(e[K - 1], edge);
            for (int j = 0; j < K; ++j) a[j] = __builtin_fma(a[j], (double)cur[j][0], (j == 0) ? edge : e[j - 1]);
            if ((s & 7) == 7) {
                const int src = (s >> 3) & 63;
                const long long bits = __double_as_longlong(a[0]);
                const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
                const int ex = ((hi >> 20) & 0x7ff) - 1023;  // wave-uniform
                if (ex > -1000)
                    for (int j = 0; j < K; ++j) a[j] = __builtin_ldexp(a[j], -ex);
            }
            for (int j = 0; j < K; ++j) __builtin_nontemporal_store(a[j], dst64 + ((size_t)(s & 2047) * 64 + lane) * K + j);
            for (int j = 0; j < K; ++j) cur[j] = nxt[j];
        }
        sink64[lane] = a[0] + a[1] + a[2];
    } else if (mode >= 3) {
        // linear f32 again with parts removed: mode 3 no LDS reads, 4 no stores, 5 neither, 6 neither and no DPP
        float a[K], edge = 0.f;
        for (int j = 0; j < K; ++j) a[j] = (lane == 0 && j == 0) ? 1.f : 0.f;
        f32x2 cur[K], nxt[K];
        for (int j = 0; j < K; ++j) cur[j] = w[0][lane * K + j], nxt[j] = cur[j];
        const bool rd = (mode == 4), st = (mode == 3), dp = (mode != 6);
#pragma unroll 16
        for (int s = 0; s < steps; ++s) {
            if (rd)
                for (int j = 0; j < K; ++j) nxt[j] = w[(s + 1) & (ROWS - 1)][lane * K + j];
            float e[K];
            for (int j = 0; j < K; ++j) e[j] = a[j] * cur[j][1];
            if (dp) edge = dpp_shr(e[K - 1], edge);
            else edge = e[K - 1] * 0.5f;
            for (int j = 0; j < K; ++j) a[j] = __builtin_fmaf(a[j], cur[j][0], (j == 0) ? edge : e[j - 1]);
            if ((s & 7) == 7) {
                const int bits = __builtin_amdgcn_readlane(__float_as_int(a[0]), (s >> 3) & 63);
                const int ex = ((bits >> 23) & 0xff) - 127;
                if (ex > -120)
                    for (int j = 0; j < K; ++j) a[j] = __builtin_ldexpf(a[j], -ex);
            }
            if (st)
                for (int j = 0; j < K; ++j) __builtin_nontemporal_store(a[j], dst + ((size_t)(s & 2047) * 64 + lane) * K + j);
            for (int j = 0; j < K; ++j) cur[j] = nxt[j];
        }
        sink[lane] = a[0] + a[1] + a[2];
    } else {
        float a[K], edge = 0.f;
        for (int j = 0; j < K; ++j) a[j] = (lane == 0 && j == 0) ? 1.f : 0.f;
        f32x2 cur[K], nxt[K];
        for (int j = 0; j < K; ++j) cur[j] = w[0][lane * K + j];
#pragma unroll 16
        for (int s = 0; s < steps; ++s) {
            for (int j = 0; j < K; ++j) nxt[j] = w[(s + 1) & (ROWS - 1)][lane * K + j];
            float e[K];
            for (int j = 0; j < K; ++j) e[j] = a[j] * cur[j][1];
            edge = dpp_shr(e[K - 1], edge);
            for (int j = 0; j < K; ++j) a[j] = __builtin_fmaf(a[j], cur[j][0], (j == 0) ? edge : e[j - 1]);
            if ((s & 7) == 7) {
                const int bits = __builtin_amdgcn_readlane(__float_as_int(a[0]), (s >> 3) & 63);
                const int ex = ((bits >> 23) & 0xff) - 127;
                if (ex > -120)
                    for (int j = 0; j < K; ++j) a[j] = __builtin_ldexpf(a[j], -ex);
            }
            for (int j = 0; j < K; ++j) __builtin_nontemporal_store(a[j], dst + ((size_t)(s & 2047) * 64 + lane) * K + j);
            for (int j = 0; j < K; ++j) cur[j] = nxt[j];
        }
        sink[lane] = a[0] + a[1] + a[2];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 64, steps = 2048;
    long long *out;
    float *sink;
    double *sink64;
    hipMalloc(&out, blocks * sizeof(long long));
    hipMalloc(&sink, (size_t)blocks * 64 * K * 2048 * sizeof(float));
    hipMalloc(&sink64, (size_t)blocks * 64 * K * 2048 * sizeof(double));
    const char *names[7] = {"log2-domain f32", "linear f64", "linear f32", "lin f32 -lds", "lin f32 -store", "lin f32 -lds-store", "lin f32 -lds-store-dpp"};
    for (int mode = 0; mode < 7; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, mode, steps, out, sink, sink64);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[64];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            if (rep == 1)
                printf("%-18s kernel %.1f us for %d diagonals = %.1f ns per diagonal (s_memtime ticks of wave 0: %lld)\n",
                       names[mode], ms * 1e3, steps, ms * 1e6 / steps, h[0]);
        }
    }
    return 0;
}
