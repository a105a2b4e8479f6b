// Hardware probe (dev tool, not part of the library): prints the lane/element mapping of ds_read_b64_tr_b16 and checks the
// assumed A/B fragment layout of v_mfma_f32_32x32x16_f16 against a host product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s4 lds_s4;

__global__ void tr_kernel(const _Float16 *in, _Float16 *out, int mode) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    int off;
    if (mode == 0) off = l * 4;  // dense: lane l -> elements 4l..4l+3
    else {                       // image X[k][n] with row stride 64 elements: lane p of group g -> &X[(p>>2)][16 g + 4 (p&3)]
        const int g = l >> 4, p = l & 15;
        off = (p >> 2) * 64 + 16 * g + 4 * (p & 3);
    }
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(lds + off));
    ((s4 *)out)[l] = v;
}

__global__ void mfma_kernel(const _Float16 *A, const _Float16 *B, float *C) {
    // A [32][16] row-major, B [16][32] row-major (k rows), C [32][32]
    const int l = threadIdx.x, i = l & 31, half = l >> 5;
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = A[i * 16 + 8 * half + e];
        b[e] = B[(8 * half + e) * 32 + i];
    }
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + i] = c[r];
}

int main() {
    std::vector<_Float16> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (_Float16)(float)(i % 2048);
    _Float16 *din, *dout;
    hipMalloc(&din, 8192 * 2);
    hipMalloc(&dout, 256 * 2);
    hipMemcpy(din, h.data(), 8192 * 2, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        tr_kernel<<<1, 64>>>(din, dout, mode);
        std::vector<_Float16> o(256);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("tr mode %d\n", mode);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, (float)o[4 * l], (float)o[4 * l + 1], (float)o[4 * l + 2], (float)o[4 * l + 3]);
    }
    // MFMA layout check
    std::vector<_Float16> A(512), B(512);
    std::vector<float> C(1024), R(1024, 0.f);
    for (int i = 0; i < 512; ++i) A[i] = (_Float16)(float)((i * 7 + 3) % 13 - 6), B[i] = (_Float16)(float)((i * 5 + 1) % 11 - 5);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j)
            for (int k = 0; k < 16; ++k) R[i * 32 + j] += (float)A[i * 16 + k] * (float)B[k * 32 + j];
    _Float16 *dA, *dB;
    float *dC;
    hipMalloc(&dA, 1024), hipMalloc(&dB, 1024), hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += (C[i] != R[i]);
    printf("mfma_f32_32x32x16_f16 layout check: %d mismatches of 1024\n", bad);
    return 0;
}
