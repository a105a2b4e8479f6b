// Hardware probe (dev tool): HBM streaming rates of hand-written kernels on 4 GiB (far beyond the 256 MiB Infinity Cache).
//   read-only: every lane keeps N x 16 bytes of plain global loads in flight (grid-stride), default / non-temporal policy
//   read-only through LDS-DMA (global_load_lds_dwordx4): 32 KB per workgroup and round, like the loss kernels' patch staging
//   read + write out of place and in place (x[i] *= 2)
// Prints TB/s (HIP events).  hipcc --offload-arch=gfx950 -O3 scripts/probes/probe_hbm.hip -o scripts/probes/probe_hbm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int N, bool NT>
__global__ __launch_bounds__(256) void read_k(const v4f *x, size_t n16, float *sink) {
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * N) {
        v4f v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const size_t j = i + k * stride;
            if (j < n16) v[k] = NT ? __builtin_nontemporal_load(x + j) : x[j];
            else v[k] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < N; ++k) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void dma_k(const v4f *x, size_t n16, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 32 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    const size_t per = 2048;  // 16-byte units per workgroup and round (32 KB)
    for (size_t base = (size_t)blockIdx.x * per; base + per <= n16; base += (size_t)gridDim.x * per) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int piece = wave * 8 + k;  // 32 pieces of 1 KB
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(x + base + piece * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(smem + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        s += ((const float *)smem)[threadIdx.x * 32];
        __syncthreads();
    }
    if (s == 12345.678f) sink[0] = s;
}

// The lsm pass's shape: a "patch" = 8 row segments of 3360 bytes (30 cells x 28 logits) at a pitch of 16,800 bytes (U = 150),
// staged by LDS-DMA, then 28 exp2 per lane on the staged values (one cell per lane).  PERSIST: workgroups loop over patches
// (grid-stride); otherwise one patch per workgroup, as the shipped kernel does.
template <bool PERSIST, int WORK>
__global__ __launch_bounds__(256) void patch_k(const float *x, int n_patches, int tiles_u, float *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 8 x 3360 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int pidx = blockIdx.x; pidx < n_patches; pidx += PERSIST ? gridDim.x : n_patches) {
        const int tu = pidx % tiles_u, trow = pidx / tiles_u;  // patch (row block trow, column block tu)
        const float *p0 = x + ((size_t)trow * 8 * 150 + (size_t)tu * 30) * 28;
        for (int r = wave; r < 8; r += 4) {
            const float *src = p0 + (size_t)r * 150 * 28;
            for (int q0 = 0; q0 < 210; q0 += 64) {
                const int q = q0 + lane;
                if (q < 210)
                    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(src + q * 4),
                                                     (__attribute__((address_space(3))) void *)(smem + r * 3360 + q0 * 16), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 240) {
            const v4f *c = (const v4f *)(smem + threadIdx.x * 112);
            float m = -1e30f, s = 0.f;
            v4f v[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                v[i] = c[i];
                m = fmaxf(fmaxf(fmaxf(m, v[i][0]), fmaxf(v[i][1], v[i][2])), v[i][3]);
            }
            if (WORK) {
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    s += __builtin_amdgcn_exp2f(v[i][0] - m) + __builtin_amdgcn_exp2f(v[i][1] - m) + __builtin_amdgcn_exp2f(v[i][2] - m) +
                         __builtin_amdgcn_exp2f(v[i][3] - m);
            }
            acc += s + m;
            if (WORK) out[(size_t)pidx * 240 + threadIdx.x] = acc;  // 4 bytes per cell out, like lse
        }
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <bool INPLACE, bool NT>
__global__ __launch_bounds__(256) void rw_k(const v4f *x, v4f *y, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 4) {
        v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + k * stride;
            if (j < n16) v[k] = NT ? __builtin_nontemporal_load(x + j) : x[j];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + k * stride;
            if (j < n16) {
                const v4f o = v[k] * 2.0f;
                if (NT) __builtin_nontemporal_store(o, (INPLACE ? (v4f *)x : y) + j);
                else ((INPLACE ? (v4f *)x : y))[j] = o;
            }
        }
    }
}

// read + write with one contiguous 16 KB span per workgroup (no grid stride: what torch's elementwise kernels do)
template <bool INPLACE, bool NT, int SPAN16>
__global__ __launch_bounds__(256) void rw_span_k(const v4f *x, v4f *y, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (256 * SPAN16) + threadIdx.x;
    v4f v[SPAN16];
#pragma unroll
    for (int k = 0; k < SPAN16; ++k) {
        const size_t j = base + k * 256;
        if (j < n16) v[k] = NT ? __builtin_nontemporal_load(x + j) : x[j];
    }
#pragma unroll
    for (int k = 0; k < SPAN16; ++k) {
        const size_t j = base + k * 256;
        if (j < n16) {
            const v4f o = v[k] * 1.0001f;
            if (NT) __builtin_nontemporal_store(o, (INPLACE ? (v4f *)x : y) + j);
            else ((INPLACE ? (v4f *)x : y))[j] = o;
        }
    }
}

template <typename F>
static double timed(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e-3;
}

int main() {
    const size_t bytes = 4ull << 30, n16 = bytes / 16;
    v4f *x, *y;
    float *sink;
    hipMalloc(&x, bytes), hipMalloc(&y, bytes), hipMalloc(&sink, 4);
    hipMemset(x, 0, bytes), hipMemset(y, 0, bytes);
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %5d  read x4 %.2f  read x8 %.2f  read x8 nt %.2f  read x16 %.2f TB/s\n", grid,
               bytes / timed([&] { hipLaunchKernelGGL((read_k<4, false>), dim3(grid), dim3(256), 0, 0, x, n16, sink); }) / 1e12,
               bytes / timed([&] { hipLaunchKernelGGL((read_k<8, false>), dim3(grid), dim3(256), 0, 0, x, n16, sink); }) / 1e12,
               bytes / timed([&] { hipLaunchKernelGGL((read_k<8, true>), dim3(grid), dim3(256), 0, 0, x, n16, sink); }) / 1e12,
               bytes / timed([&] { hipLaunchKernelGGL((read_k<16, false>), dim3(grid), dim3(256), 0, 0, x, n16, sink); }) / 1e12);
    }
    for (int grid : {1280, 2560, 8192})
        printf("grid %5d  read via LDS-DMA (32 KB rounds, 5 workgroups per CU fit) %.2f TB/s\n", grid,
               bytes / timed([&] { hipLaunchKernelGGL(dma_k, dim3(grid), dim3(256), 32768, 0, x, n16, sink); }) / 1e12);
    {
        // 32 utterances x 600 x 150 cells x 28 logits = 322.6 MB per "step", 10 steps' worth of distinct memory = 3.2 GB
        const int tiles_u = 5, n_patches = 10 * 32 * 75 * tiles_u;
        const double pb = (double)n_patches * 8 * 3360;
        float *out;
        hipMalloc(&out, (size_t)n_patches * 240 * 4);
        printf("patch-shaped reads (8 x 3360 B at pitch 16,800 B), one patch per workgroup: DMA only %.2f, + softmax work %.2f TB/s\n",
               pb / timed([&] { hipLaunchKernelGGL((patch_k<false, 0>), dim3(n_patches), dim3(256), 26880, 0, (const float *)x, n_patches, tiles_u, out); }) / 1e12,
               pb / timed([&] { hipLaunchKernelGGL((patch_k<false, 1>), dim3(n_patches), dim3(256), 26880, 0, (const float *)x, n_patches, tiles_u, out); }) / 1e12);
        for (int grid : {1280, 1536, 3072})
            printf("  persistent, grid %4d: DMA only %.2f, + softmax work %.2f TB/s\n", grid,
                   pb / timed([&] { hipLaunchKernelGGL((patch_k<true, 0>), dim3(grid), dim3(256), 26880, 0, (const float *)x, n_patches, tiles_u, out); }) / 1e12,
                   pb / timed([&] { hipLaunchKernelGGL((patch_k<true, 1>), dim3(grid), dim3(256), 26880, 0, (const float *)x, n_patches, tiles_u, out); }) / 1e12);
    }
    for (int grid : {2048, 8192})
        printf("grid %5d  r+w out of place %.2f (nt %.2f)   in place %.2f (nt %.2f) TB/s\n", grid,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_k<false, false>), dim3(grid), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_k<false, true>), dim3(grid), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_k<true, false>), dim3(grid), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_k<true, true>), dim3(grid), dim3(256), 0, 0, x, y, n16); }) / 1e12);
    {
        const unsigned g4 = (unsigned)((n16 + 1023) / 1024), g8 = (unsigned)((n16 + 2047) / 2048);
        printf("one 16 KB span per workgroup: r+w out of place %.2f (nt %.2f)   in place %.2f (nt %.2f) TB/s\n",
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<false, false, 4>), dim3(g4), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<false, true, 4>), dim3(g4), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<true, false, 4>), dim3(g4), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<true, true, 4>), dim3(g4), dim3(256), 0, 0, x, y, n16); }) / 1e12);
        printf("one 32 KB span per workgroup: r+w out of place %.2f (nt %.2f)   in place %.2f (nt %.2f) TB/s\n",
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<false, false, 8>), dim3(g8), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<false, true, 8>), dim3(g8), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<true, false, 8>), dim3(g8), dim3(256), 0, 0, x, y, n16); }) / 1e12,
               2 * bytes / timed([&] { hipLaunchKernelGGL((rw_span_k<true, true, 8>), dim3(g8), dim3(256), 0, 0, x, y, n16); }) / 1e12);
    }
    return 0;
}
