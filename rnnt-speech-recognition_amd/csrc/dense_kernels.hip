// dense_kernels.hip -- the joint network's FIRST Dense layer and its backward on the gfx950 matrix cores, f32-grade.
//
// Reference: model.py:158-163  (joint_inp = enc[:, :, None] + pred[:, None]; Dense(J, tanh)) and what TF autodiff does behind
// it (run_rnnt.py:284).  The layer is factored exactly, W1^T (e_t + p_u) + b1 = (W1^T e_t + b1) + W1^T p_u, so it is three
// GEMMs over the B (T + U) rows of [enc; pred] instead of B T U rows:
//     proj  = X . W1 (+ b1 on the enc rows)          "NT": A = X rows,     B = W1^T rows   (K = H)
//     dX    = dproj . W1^T                            "NT": A = dproj rows, B = W1 rows     (K = J)
//     dW1   = X^T . dproj,  db1 = column sums (enc)   "TN": K = the B (T + U) rows, split over workgroups
// where X = [enc; pred] and dproj = [d enc_proj; d pred_proj] come from / go to the fused joint kernels (joint_kernels.hip).
//
// Arithmetic: every f32 operand is multiplied by a power of two S (max |S x| in [2^13, 2^14), one S per tensor, found by an
// abs-max pass) and split into binary16 hi = RNE(S x) and lo = RNE(S x - hi): 22 significand bits relative to the tensor's
// largest element.  A product is hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_f16 (each exact in f32, f32 accumulation; the
// dropped lo.lo term is <= 2^-22 relative), and the result is scaled back by 1 / (S_A S_B), exact.  The same scheme as the
// joint's J x V products (joint_kernels.hip: mfma3), here as LDS-tiled GEMMs:
//   * operands are pre-split once per call into binary16 images (HBM-bound elementwise passes), so the GEMM tiles arrive by
//     LDS-DMA (global_load_lds, 16 bytes per lane) without passing through registers;
//   * NT: 256 x 128 tile, 8 waves of 64 x 64, K chunks of 32 in a 3-stage ring (48 KB per stage), 64-byte LDS rows with the
//     16-byte chunks XOR-swizzled by (row >> 2) & 3: the ds_read_b128 fragment reads are bank-conflict free;
//   * TN: 128 x 128 tile, 4 waves of 64 x 64, K chunks of 16 ROWS in a 3-stage ring (16 KB per stage, three workgroups per
//     CU), row-major [k][column] tiles read TRANSPOSED with ds_read_b64_tr_b16, 16-byte chunks XOR-swizzled by 4 (k & 3);
//     split-K partials are summed in a fixed order (deterministic, no floating-point atomics).
#include "rnnt_common.h"

#include <math.h>

namespace rnnt {

typedef _Float16 df16;
typedef _Float16 dh2 __attribute__((ext_vector_type(2)));
typedef _Float16 dh4 __attribute__((ext_vector_type(4)));
typedef _Float16 dh8 __attribute__((ext_vector_type(8)));
typedef short ds4 __attribute__((ext_vector_type(4)));
typedef float df32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void dlds_void;
typedef __attribute__((address_space(1))) const void dglb_cvoid;
typedef __attribute__((address_space(3))) ds4 dlds_s4;

__device__ __forceinline__ constexpr int dcd_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// XCD-aware bijective remap: workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); give every XCD a CONTIGUOUS range
// of logical work items, so that items which share operand tiles (neighbours in logical order) share one L2
__device__ __forceinline__ uint32_t dxcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t xcd = bid & 7u, idx = bid >> 3, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
}

constexpr int kDenseScaleLog2 = 14;  // max |S x| < 2^14 before the binary16 rounding (headroom for the f16 range: 65504)

// ---------------------------------------------------------------------------------------------
// abs-max of a tensor (bit pattern of the non-negative float: unsigned order = float order), NaN / inf propagate as "huge"
// ---------------------------------------------------------------------------------------------
constexpr int kAbsBlocks = 512;  // grid of the abs-max pass = entries per tensor it leaves for the split kernels
struct AbsmaxJob {
    const float *x[3];
    size_t n[3];
    unsigned *out[3];  // [kAbsBlocks] per tensor: every block stores its own maximum (no atomics, no zero-fill)
    int count;
};
__global__ __launch_bounds__(256) void dense_absmax_kernel(const AbsmaxJob job) {
    __shared__ unsigned red[4];
    for (int t = 0; t < job.count; ++t) {
        const float *x = job.x[t];
        const size_t n = job.n[t];
        unsigned m = 0u;
        const size_t n4 = ((((uintptr_t)x) & 15) == 0) ? (n >> 2) : 0;
        const size_t stride = (size_t)gridDim.x * 256;
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {  // four 16-byte loads in flight
            const uint4 a = ((const uint4 *)x)[i], b = ((const uint4 *)x)[i + stride], c = ((const uint4 *)x)[i + 2 * stride],
                        d = ((const uint4 *)x)[i + 3 * stride];
            m = max(m, max(max(max(a.x & 0x7fffffffu, a.y & 0x7fffffffu), max(a.z & 0x7fffffffu, a.w & 0x7fffffffu)),
                           max(max(b.x & 0x7fffffffu, b.y & 0x7fffffffu), max(b.z & 0x7fffffffu, b.w & 0x7fffffffu))));
            m = max(m, max(max(max(c.x & 0x7fffffffu, c.y & 0x7fffffffu), max(c.z & 0x7fffffffu, c.w & 0x7fffffffu)),
                           max(max(d.x & 0x7fffffffu, d.y & 0x7fffffffu), max(d.z & 0x7fffffffu, d.w & 0x7fffffffu))));
        }
        for (; i < n4; i += stride) {
            const uint4 q = ((const uint4 *)x)[i];
            m = max(max(m, q.x & 0x7fffffffu), max(q.y & 0x7fffffffu, max(q.z & 0x7fffffffu, q.w & 0x7fffffffu)));
        }
        for (size_t k = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride)
            m = max(m, __float_as_uint(x[k]) & 0x7fffffffu);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        // write-through store / L2-coherent loads for these few words (see dense_max_of)
        if (threadIdx.x == 0)
            __hip_atomic_store(job.out[t] + blockIdx.x, max(max(red[0], red[1]), max(red[2], red[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the tensor's maximum from its per-block entries (every block of a consumer kernel redoes this small reduction).
// These few KB are re-written by every call and read by every workgroup on every XCD: a clean copy of the PREVIOUS call's
// entries can still sit in a reader's L2 / L1 (observed under HIP-graph replay: the operand scales lagged one call behind,
// results stayed f32-grade but were not bit-identical to a direct call, and NaN when the lagging scale overflowed binary16).
// Agent-scope (sc1) accesses on both sides keep them coherent whatever a launch path does at kernel boundaries.
__device__ __forceinline__ unsigned dense_max_of(const unsigned *arr, const int count) {
    __shared__ unsigned red2[4];
    unsigned m = 0u;
    for (int i = threadIdx.x; i < count; i += 256) m = max(m, __hip_atomic_load(arr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = m;
    __syncthreads();
    return max(max(red2[0], red2[1]), max(red2[2], red2[3]));
}

// S = 2^(14 - e) with max|x| < 2^e (1 for an all-zero or non-finite tensor: the products then carry the inf / NaN through)
__device__ __forceinline__ float dense_scale_from_bits(const unsigned bits) {
    const float m = __uint_as_float(bits);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.0f;
    return ldexpf(1.0f, kDenseScaleLog2 - (ilogbf(m) + 1));
}

// four values -> their binary16 hi / lo parts
__device__ __forceinline__ void dense_split4(const float4 v, const float S, dh4 &hi, dh4 &lo) {
    const float x[4] = {v.x * S, v.y * S, v.z * S, v.w * S};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const df16 h = (df16)x[e];
        hi[e] = h;
        lo[e] = (df16)(x[e] - (float)h);
    }
}

// x [rows][cols] f32 (row-major, contiguous) -> hi / lo images at rows [row_off, row_off + rows) of [*][cols] binary16 arrays,
// scaled by the tensor's power of two; scal[slot] receives S.  Flat, 16-byte loads / 8-byte stores.
__device__ __forceinline__ void dense_split_body(const float *__restrict__ x, df16 *__restrict__ hi, df16 *__restrict__ lo, size_t n,
                                                 size_t elem_off, const unsigned *maxarr, int maxcount, float *scal, int slot,
                                                 const unsigned blk, const unsigned nblk) {
    const float S = dense_scale_from_bits(dense_max_of(maxarr, maxcount));
    if (blk == 0 && threadIdx.x == 0) __hip_atomic_store(scal + slot, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t n4 = n >> 2;  // cols % 8 == 0 (checked by the launcher): n % 4 == 0 and the images are 8-byte aligned
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < n4; i += (size_t)nblk * 256) {
        const float4 v = ((const float4 *)x)[i];
        dh4 h, l;
        dense_split4(v, S, h, l);
        *(dh4 *)(hi + elem_off + i * 4) = h;
        *(dh4 *)(lo + elem_off + i * 4) = l;
    }
}

// The same for d enc_proj, plus per-block column sums of the UNSCALED input (db1 = sum over the enc rows), written to
// colpart[block][cols] and summed in a fixed order afterwards.  A block owns kDbRows rows; a thread owns 4 columns.
constexpr int kDbRows = 16;
__device__ __forceinline__ void dense_split_colsum_body(const float *__restrict__ x, df16 *__restrict__ hi, df16 *__restrict__ lo, int rows,
                                                        int cols, const unsigned *maxarr, int maxcount, float *scal, int slot,
                                                        float *__restrict__ colpart, const unsigned blk) {
    const float S = dense_scale_from_bits(dense_max_of(maxarr, maxcount));
    if (blk == 0 && threadIdx.x == 0) __hip_atomic_store(scal + slot, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c4 = cols >> 2;
    const int r0 = (int)blk * kDbRows, nr = min(kDbRows, rows - r0);
    for (int cq = threadIdx.x; cq < c4; cq += 256) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int rb = 0; rb < nr; rb += 4) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)  // four independent loads in flight
                v[q] = (rb + q < nr) ? *(const float4 *)(x + (size_t)(r0 + rb + q) * cols + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (rb + q >= nr) break;
                sum.x += v[q].x, sum.y += v[q].y, sum.z += v[q].z, sum.w += v[q].w;
                dh4 h, l;
                dense_split4(v[q], S, h, l);
                const size_t o = (size_t)(r0 + rb + q) * cols + cq * 4;
                *(dh4 *)(hi + o) = h;
                *(dh4 *)(lo + o) = l;
            }
        }
        *(float4 *)(colpart + (size_t)blk * cols + cq * 4) = sum;
    }
}

// zero rows [r0, r1) of a hi / lo image pair (the padding rows of the TN product's K range)
__device__ __forceinline__ void dense_zero_rows_body(df16 *hi, df16 *lo, int r0, int r1, int cols, const unsigned blk, const unsigned nblk) {
    const size_t n = (size_t)(r1 - r0) * cols, base = (size_t)r0 * cols;
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < n; i += (size_t)nblk * 256) hi[base + i] = (df16)0.f, lo[base + i] = (df16)0.f;
}

// W [H][J] f32 -> split images of W ([H][J], for dX = dproj . W^T) and of W^T ([J][H], for proj = X . W); tiny (H J <= 0.5 M)
__device__ __forceinline__ void dense_split_w_body(const float *W, df16 *whi, df16 *wlo, df16 *thi, df16 *tlo, int H, int J,
                                                   const unsigned *maxarr, int maxcount, float *scal, int slot, const unsigned blk,
                                                   const unsigned nblk) {
    const float S = dense_scale_from_bits(dense_max_of(maxarr, maxcount));
    if (blk == 0 && threadIdx.x == 0) __hip_atomic_store(scal + slot, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const size_t n = (size_t)H * J;
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < n; i += (size_t)nblk * 256) {
        const int h = (int)(i / (size_t)J), j = (int)(i - (size_t)h * J);
        const float x = W[i] * S;
        const df16 a = (df16)x, b = (df16)(x - (float)a);
        whi[i] = a, wlo[i] = b;
        thi[(size_t)j * H + h] = a, tlo[(size_t)j * H + h] = b;
    }
}

// Everything a GEMM pass needs of its operands, in ONE launch (round 5; up to six launches before): the split images of the two
// row segments (enc / pred rows of X, or of d proj -- the latter with the column partials of db1), the zero padding rows of the
// image, and (forward) the two images of W1.  Blocks are dealt to the jobs by index ranges; every block redoes its tensor's
// 512-entry abs-max reduction, as before.
struct PrepJob {
    const float *x[2];            // the two row segments
    df16 *hi, *lo;                // their image pair
    size_t n[2], elem_off[2];
    const unsigned *maxarr[2];
    int maxcount, slot[2];
    float *scal;
    int rows0, cols;              // segment 0 as rows x cols ...
    float *colpart;               // ... with per-block column sums when non-null (then g[0] = ceil(rows0 / kDbRows))
    int pad[4];                   // zero rows [pad[0], pad[1]) and [pad[2], pad[3]) of the image
    const float *W;               // nullable
    df16 *whi, *wlo, *thi, *tlo;
    int H, J, wslot;
    const unsigned *wmaxarr;
    unsigned g[4];                // blocks of: segment 0, segment 1, the padding rows, W
};
__global__ __launch_bounds__(256) void dense_prepare_kernel(const PrepJob job) {
    unsigned blk = blockIdx.x;
    if (blk < job.g[0]) {
        if (job.colpart)
            dense_split_colsum_body(job.x[0], job.hi, job.lo, job.rows0, job.cols, job.maxarr[0], job.maxcount, job.scal, job.slot[0], job.colpart, blk);
        else
            dense_split_body(job.x[0], job.hi, job.lo, job.n[0], job.elem_off[0], job.maxarr[0], job.maxcount, job.scal, job.slot[0], blk, job.g[0]);
        return;
    }
    blk -= job.g[0];
    if (blk < job.g[1]) {
        dense_split_body(job.x[1], job.hi, job.lo, job.n[1], job.elem_off[1], job.maxarr[1], job.maxcount, job.scal, job.slot[1], blk, job.g[1]);
        return;
    }
    blk -= job.g[1];
    if (blk < job.g[2]) {
        const unsigned half = job.g[2] / 2;
        if (blk < half)
            dense_zero_rows_body(job.hi, job.lo, job.pad[0], job.pad[1], job.cols, blk, half);
        else
            dense_zero_rows_body(job.hi, job.lo, job.pad[2], job.pad[3], job.cols, blk - half, half);
        return;
    }
    blk -= job.g[2];
    dense_split_w_body(job.W, job.whi, job.wlo, job.thi, job.tlo, job.H, job.J, job.wmaxarr, kAbsBlocks, job.scal, job.wslot, blk, job.g[3]);
}

// ---------------------------------------------------------------------------------------------
// NT product:  C[m][n] = (1 / (S_A S_B)) sum_k A[m][k] B[n][k]  (+ bias[n] on the first segment's rows)
// A = hi / lo images [M][K] holding TWO row segments (enc rows [0, R0), pred rows [R0p, R0p + R1), padding elsewhere), each with
// its own scale and its own output array (row stride ldc); B = hi / lo images [N][K].
// ---------------------------------------------------------------------------------------------
struct DenseNT {
    const df16 *Ahi, *Alo, *Bhi, *Blo;
    float *C0, *C1;
    float *E0, *E1;    // nullable: e^{2 C} tables next to C0 / C1 (what the fused joint's prep kernel would compute from C)
    float *tflag;      // with E0: tflag[0] is raised when some |C| exceeds the tables' range (joint_kernels.hip kExpTabLimit)
    const float *bias;
    const float *scal;
    int sa0, sa1, sb;  // slots of the scales in `scal`: A segment 0, A segment 1, B
    int M, N, K, R0, R0p, R1, ldc;
    int tiles_n;
};

constexpr int kNtM = 256, kNtN = 128, kNtK = 32;
constexpr int kNtPartA = kNtM * 64, kNtPartB = kNtN * 64;      // bytes per hi (or lo) part of a stage: 64-byte rows
constexpr int kNtStage = 2 * kNtPartA + 2 * kNtPartB;          // 48 KB
constexpr int kNtStages = 3;

#ifndef DENSE_NT_LOADERS
#define DENSE_NT_LOADERS 0  // loader waves per workgroup; 0 (shipped): the compute waves issue the LDS-DMA themselves.  Measured at
                            // C2: 4 loaders 75.5 us, 0 loaders 73.7 us -- the fill is not issue-bound (PMC: matrix pipe 40 % busy,
                            // HBM fetch 94 MB for 451 MB of tile loads, L2 hit rate 83 %: the stages wait for L2 -> LDS data)
#endif
constexpr int kNtLoaders = DENSE_NT_LOADERS;
constexpr int kNtThreads = (8 + kNtLoaders) * 64;
constexpr int kNtIssuers = kNtLoaders ? kNtLoaders : 8;   // waves that issue DMA pieces
constexpr int kNtPieces = 48 / kNtIssuers;                // pieces per issuing wave and stage

__global__ __launch_bounds__(kNtThreads) void dense_gemm_nt_kernel(const DenseNT g) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = (wave >> 2) & 1;
    // the N tiles of one M tile are neighbours in ONE XCD's queue: the A rows are fetched from HBM once and shared through that L2
    const uint32_t lid = dxcd_remap(blockIdx.x, gridDim.x);
    const int nt = (int)(lid % (unsigned)g.tiles_n), mt = (int)(lid / (unsigned)g.tiles_n);
    const int m0 = mt * kNtM, n0 = nt * kNtN;
    const int K = g.K;
    const int nK = K / kNtK;
    const bool loader = kNtLoaders && wave >= 8;          // wave-uniform
    const int iw = kNtLoaders ? wave - 8 : wave;          // index among the issuing waves

    // LDS-DMA pieces: wave-instructions i = iw + kNtIssuers k of the 48 that fill one stage.
    //   i <  16: A hi rows 16 i .. ; i < 32: A lo ; i < 40: B hi rows 16 (i - 32) .. ; else B lo.
    // A lane moves 16 bytes: row = 16 (block) + lane / 4, LDS chunk position p = lane & 3 holds logical chunk p ^ ((row >> 2) & 3)
    // (Issuing a piece costs a wave 60-180 cycles of its in-order stream; moving the pieces to dedicated loader waves --
    // DENSE_NT_LOADERS -- did not change the kernel's time, see above.)
    const df16 *src[kNtPieces];
    int ldsoff[kNtPieces];
    if (!kNtLoaders || loader) {
#pragma unroll
        for (int k = 0; k < kNtPieces; ++k) {
            const int i = iw + kNtIssuers * k;
            const bool isA = i < 32;
            const int blk = isA ? (i & 15) : ((i - 32) & 7);
            const bool lo = isA ? (i >= 16) : (i >= 40);
            const int row = blk * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = isA ? min(m0 + row, g.M - 1) : min(n0 + row, g.N - 1);
            const df16 *base = isA ? (lo ? g.Alo : g.Ahi) : (lo ? g.Blo : g.Bhi);
            src[k] = base + (size_t)grow * K + c * 8;
            ldsoff[k] = (isA ? (lo ? kNtPartA : 0) : 2 * kNtPartA + (lo ? kNtPartB : 0)) + blk * 1024;
        }
    }
    auto dma_piece = [&](const int k, const int kc, const int stage) {
        lds_dma16(src[k] + kc * kNtK, dsm + stage * kNtStage + ldsoff[k]);
    };

    if (loader) {
        // ---- loader wave: stage kc + 2 goes out right after the barrier that frees its ring slot
#pragma unroll
        for (int k = 0; k < kNtPieces; ++k) dma_piece(k, 0, 0);
        if (nK > 1) {
#pragma unroll
            for (int k = 0; k < kNtPieces; ++k) dma_piece(k, 1, 1);
        }
        int sc = 0;
        for (int kc = 0; kc < nK; ++kc) {
            if (kc + 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kNtPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage kc complete in LDS; every compute wave is done with stage kc - 1
            if (kc + 2 < nK) {
                const int sp = (sc + 2) % kNtStages;
#pragma unroll
                for (int k = 0; k < kNtPieces; ++k) dma_piece(k, kc + 2, sp);
            }
            sc = (sc + 1) % kNtStages;
        }
        __builtin_amdgcn_s_barrier();  // the epilogue's barrier (the compute waves re-use the ring as staging tiles)
        return;
    }

    df32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (!kNtLoaders) {
#pragma unroll
        for (int k = 0; k < kNtPieces; ++k) dma_piece(k, 0, 0);
        if (nK > 1) {
#pragma unroll
            for (int k = 0; k < kNtPieces; ++k) dma_piece(k, 1, 1);
        }
    }
    // fragment addresses of this lane inside a stage (bytes): rows of its two A tiles and its two B tiles
    int arow[2], brow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) arow[q] = wm * 64 + q * 32 + n31, brow[q] = wn * 64 + q * 32 + n31;

    int sc = 0;
    for (int kc = 0; kc < nK; ++kc) {
        if (!kNtLoaders) {
            if (kc + 1 < nK) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // stage kc complete in LDS; the stage of kc - 1 (refilled next) is free
        asm volatile("" ::: "memory");
        const char *S = dsm + sc * kNtStage;
        const bool pf = !kNtLoaders && kc + 2 < nK;
        const int sp = (sc + 2) % kNtStages;
        // fragments of k-step ks in [ks & 1]: read one k-step ahead of their MFMAs
        dh8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto rd = [&](const int ks, dh8 (&a_h)[2], dh8 (&a_l)[2], dh8 (&b_h)[2], dh8 (&b_l)[2]) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pa = ((ks * 2 + half) ^ ((arow[q] >> 2) & 3)) * 16, pb = ((ks * 2 + half) ^ ((brow[q] >> 2) & 3)) * 16;
                a_h[q] = *(const dh8 *)(S + arow[q] * 64 + pa);
                a_l[q] = *(const dh8 *)(S + kNtPartA + arow[q] * 64 + pa);
                b_h[q] = *(const dh8 *)(S + 2 * kNtPartA + brow[q] * 64 + pb);
                b_l[q] = *(const dh8 *)(S + 2 * kNtPartA + kNtPartB + brow[q] * 64 + pb);
            }
        };
        rd(0, ah[0], al[0], bh[0], bl[0]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) rd(1, ah[1], al[1], bh[1], bl[1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kNtLoaders == 0) {
                if (pf) {  // three of the six pieces of stage kc + 2 per k-step, between the MFMA groups
                    dma_piece(3 * ks + 0, kc + 2, sp);
                    dma_piece(3 * ks + 1, kc + 2, sp);
                    dma_piece(3 * ks + 2, kc + 2, sp);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][mi], bh[ks][ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][mi], bh[ks][ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][mi], bl[ks][ni], acc[mi][ni], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        sc = (sc + 1) % kNtStages;
    }
    // epilogue: exact power-of-two rescale, bias, row segment.  The C/D layout gives a lane ONE column of 16 rows: stored from
    // there it is 64 four-byte store instructions per wave and output (store-issue bound: a 256 x 128 tile with its table took
    // as long to write as 8 of the 20 K stages took to multiply).  Each wave passes its 64 x 64 tile through LDS instead (the
    // stage ring is free now) and writes row segments of 256 bytes with 16-byte stores: 16 store instructions per output.
    const float sB = __hip_atomic_load(g.scal + g.sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float inv0 = 1.0f / (__hip_atomic_load(g.scal + g.sa0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * sB);
    const float inv1 = 1.0f / (__hip_atomic_load(g.scal + g.sa1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * sB);
    const bool tables = g.E0 != nullptr;
    __builtin_amdgcn_s_barrier();  // every compute wave is done reading the last stage (the loaders join this barrier, then leave)
    constexpr int kPitch = 68;     // floats per tile row in LDS: 16-byte aligned rows, conflict-free column writes
    float *tile = (float *)dsm + wave * 64 * kPitch;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(mi * 32 + dcd_row(r, half)) * kPitch + ni * 32 + n31] = acc[mi][ni][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a wave only reads its own tile: no barrier
    const int cq = lane & 15, rsub = lane >> 4;         // 16 lanes x 4 columns = one 64-column row segment; 4 rows per pass
    const int col = n0 + wn * 64 + cq * 4;
    bool big = false;
    if (col < g.N) {  // N % 4 == 0: a 4-column group is inside or outside as a whole
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) bv = *(const float4 *)(g.bias + col);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rl = it * 4 + rsub;
            const int row = m0 + wm * 64 + rl;
            const float4 a = *(const float4 *)(tile + rl * kPitch + cq * 4);
            float4 v;
            size_t o;
            float *C, *E;
            if (row < g.R0) {
                v = make_float4(fmaf(a.x, inv0, bv.x), fmaf(a.y, inv0, bv.y), fmaf(a.z, inv0, bv.z), fmaf(a.w, inv0, bv.w));
                o = (size_t)row * g.ldc + col, C = g.C0, E = g.E0;
            } else if (row >= g.R0p && row < g.R0p + g.R1) {
                v = make_float4(a.x * inv1, a.y * inv1, a.z * inv1, a.w * inv1);
                o = (size_t)(row - g.R0p) * g.ldc + col, C = g.C1, E = g.E1;
            } else {
                continue;
            }
            *(float4 *)(C + o) = v;
            if (tables) {
                big |= exp_tab_out_of_range(v.x) || exp_tab_out_of_range(v.y) || exp_tab_out_of_range(v.z) || exp_tab_out_of_range(v.w);  // (rnnt_common.h)
                *(float4 *)(E + o) = make_float4(exp_tab(v.x), exp_tab(v.y), exp_tab(v.z), exp_tab(v.w));
            }
        }
    }
    if (tables && __any(big) && lane == 0) g.tflag[0] = 1.0f;
}

// ---------------------------------------------------------------------------------------------
// TN product, split over K:  P[split][m][n] = (1 / (S_A S_B)) sum_{k in range(split)} A[k][m] B[k][n]
// A = hi / lo images [Kp][M], B = hi / lo images [Kp][N]; the K rows come in two ranges of whole 16-row chunks (enc rows,
// then pred rows; padding rows are zero), each with its own pair of scales: splits [0, ns0) cover chunks [0, c0), the rest
// cover [c0, Kchunks).
// ---------------------------------------------------------------------------------------------
struct DenseTN {
    const df16 *Ahi, *Alo, *Bhi, *Blo;
    float *P;
    const float *scal;
    int sa0, sb0, sa1, sb1;
    int M, N, Kchunks, c0;  // Kchunks = Kp / 16
    int tiles_m, tiles_n, nsplit, ns0;
};

constexpr int kTnT = 128, kTnK = 16;
constexpr int kTnPart = kTnK * kTnT * 2;      // bytes per hi (or lo) part: 16 rows x 256 B = 4 KB
constexpr int kTnStage = 4 * kTnPart;         // A hi, A lo, B hi, B lo: 16 KB
constexpr int kTnStages = 3;

__global__ __launch_bounds__(256) void dense_gemm_tn_kernel(const DenseTN g) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    // all tiles of one K split are neighbours in one XCD's queue: they share the split's rows of both operands through that L2
    int bid = (int)dxcd_remap(blockIdx.x, gridDim.x);
    const int ntile = g.tiles_m * g.tiles_n;
    const int split = bid / ntile;
    bid -= split * ntile;
    const int nt = bid % g.tiles_n, mt = bid / g.tiles_n;
    const int m0 = mt * kTnT, n0 = nt * kTnT;
    const bool first = split < g.ns0;
    const int rs = first ? split : split - g.ns0, rn = first ? g.ns0 : g.nsplit - g.ns0;  // split index / count inside its range
    const int rc0 = first ? 0 : g.c0, rcn = first ? g.c0 : g.Kchunks - g.c0;
    const int c_lo = rc0 + (int)((long long)rcn * rs / rn), c_hi = rc0 + (int)((long long)rcn * (rs + 1) / rn);

    // LDS-DMA pieces: wave-instructions i = wave + 4 k (k = 0..3) of the 16 per stage; part = i >> 2 (A hi, A lo, B hi, B lo),
    // k-rows 4 (i & 3) .. + 3.  A lane moves 16 bytes: k-row = 4 (i & 3) + lane / 16, LDS chunk position p = lane & 15 holds
    // logical chunk p ^ (4 (krow & 3)) of the row's 16 chunks (128 columns).
    const df16 *src[4];
    int ldsoff[4];
    size_t ld[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = wave + 4 * k;
        const int part = i >> 2, krow = 4 * (i & 3) + (lane >> 4);
        const int c = (lane & 15) ^ (4 * (krow & 3));
        const bool isA = part < 2;
        const int width = isA ? g.M : g.N;
        const int col = min((isA ? m0 : n0) + c * 8, width - 8);  // a tile may overhang the matrix: clamp (masked at the store)
        const df16 *base = part == 0 ? g.Ahi : part == 1 ? g.Alo : part == 2 ? g.Bhi : g.Blo;
        src[k] = base + (size_t)krow * width + col;
        ld[k] = (size_t)width * kTnK;
        ldsoff[k] = part * kTnPart + (i & 3) * 1024;
    }
    auto dma_piece = [&](const int k, const int chunk, const int stage) {
        lds_dma16(src[k] + (size_t)chunk * ld[k], dsm + stage * kTnStage + ldsoff[k]);
    };

    df32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nC = c_hi - c_lo;
    if (nC > 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_piece(k, c_lo, 0);
    }
    if (nC > 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_piece(k, c_lo + 1, 1);
    }
    // transposed fragment reads: 16-lane group g4, lane p of the group supplies the address of 4 consecutive columns of one
    // k-row and receives 4 consecutive k of one column (see joint_f16_kernels.hip header)
    const int g4 = lane >> 4, pl = lane & 15;
    const int krow0 = 8 * (g4 >> 1) + (pl >> 2);             // + 4 for the second read; (krow0 + 4) & 3 == krow0 & 3
    const int swz = 4 * (krow0 & 3);
    int aoff[2], boff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ca = wm * 64 + q * 32 + 16 * (g4 & 1) + 4 * (pl & 3), cb = wn * 64 + q * 32 + 16 * (g4 & 1) + 4 * (pl & 3);
        aoff[q] = krow0 * 256 + (((ca >> 3) ^ swz) * 16) + (ca & 7) * 2;
        boff[q] = krow0 * 256 + (((cb >> 3) ^ swz) * 16) + (cb & 7) * 2;
    }
    auto rd = [&](const char *p) {
        const dh4 k03 = __builtin_bit_cast(dh4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((dlds_s4 *)p));
        const dh4 k47 = __builtin_bit_cast(dh4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((dlds_s4 *)(p + 4 * 256)));
        return __builtin_shufflevector(k03, k47, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    int sc = 0;
    for (int ci = 0; ci < nC; ++ci) {
        if (ci + 1 < nC) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char *S = dsm + sc * kTnStage;
        dh8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ah[q] = rd(S + aoff[q]);
            al[q] = rd(S + kTnPart + aoff[q]);
            bh[q] = rd(S + 2 * kTnPart + boff[q]);
            bl[q] = rd(S + 3 * kTnPart + boff[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ci + 2 < nC) {
            const int sp = (sc + 2) % kTnStages;
#pragma unroll
            for (int k = 0; k < 4; ++k) dma_piece(k, c_lo + ci + 2, sp);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        sc = (sc + 1) % kTnStages;
    }
    const float inv = 1.0f / (__hip_atomic_load(g.scal + (first ? g.sa0 : g.sa1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) *
                              __hip_atomic_load(g.scal + (first ? g.sb0 : g.sb1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float *P = g.P + (size_t)split * g.M * g.N;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + ni * 32 + n31;
        if (col >= g.N) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + dcd_row(r, half);
                if (row < g.M) P[(size_t)row * g.N + col] = acc[mi][ni][r] * inv;
            }
    }
}

// out[i] = sum_q P[q][i] in a FIXED association (deterministic): G lanes share an output float4, lane g sums partials g, g + G,
// g + 2G, ... in order (four loads in flight), then a binary tree over the G lanes.  n % 4 == 0.
template <int G>
__device__ __forceinline__ void dense_reduce_body(float *out, const float *P, int nparts, size_t n, const unsigned blk) {
    const size_t n4 = n >> 2;
    const size_t i = ((size_t)blk * 256 + threadIdx.x) / G;
    const int grp = threadIdx.x & (G - 1);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4 *p4 = (const float4 *)P + i;
        int q = grp;
        for (; q + 3 * G < nparts; q += 4 * G) {
            const float4 a = p4[(size_t)q * n4], b = p4[(size_t)(q + G) * n4], c = p4[(size_t)(q + 2 * G) * n4], d = p4[(size_t)(q + 3 * G) * n4];
            s.x = (((s.x + a.x) + b.x) + c.x) + d.x, s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
            s.z = (((s.z + a.z) + b.z) + c.z) + d.z, s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
        }
        for (; q < nparts; q += G) {
            const float4 a = p4[(size_t)q * n4];
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        s.x += __shfl_down(s.x, off, G), s.y += __shfl_down(s.y, off, G);
        s.z += __shfl_down(s.z, off, G), s.w += __shfl_down(s.w, off, G);
    }
    if (i < n4 && grp == 0) ((float4 *)out)[i] = s;
}

// dW1 (G = 4) and db1 (G = 64) in one launch
__global__ __launch_bounds__(256) void dense_reduce2_kernel(float *out0, const float *P0, int nparts0, size_t n0, unsigned g0, float *out1,
                                                            const float *P1, int nparts1, size_t n1) {
    if (blockIdx.x < g0)
        dense_reduce_body<4>(out0, P0, nparts0, n0, blockIdx.x);
    else
        dense_reduce_body<64>(out1, P1, nparts1, n1, blockIdx.x - g0);
}
static hipError_t dense_reduce2(float *out0, const float *P0, int nparts0, size_t n0, float *out1, const float *P1, int nparts1, size_t n1,
                                hipStream_t s) {
    if (((n0 | n1) & 3) != 0 || ((((uintptr_t)out0) | ((uintptr_t)out1)) & 15) != 0) return hipErrorInvalidValue;
    const unsigned g0 = (unsigned)(((n0 >> 2) * 4 + 255) / 256), g1 = (unsigned)(((n1 >> 2) * 64 + 255) / 256);
    hipLaunchKernelGGL(dense_reduce2_kernel, dim3(g0 + g1), dim3(256), 0, s, out0, P0, nparts0, n0, g0, out1, P1, nparts1, n1);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct DenseLayout {
    size_t proj_e, proj_p, dproj_e, dproj_p, Xhi, Xlo, Dhi, Dlo, Whi, Wlo, WThi, WTlo, dWpart, dbpart, scal, total;
    int R0, R0p, R1, Rp, nsplit, ns0, db_blocks;
};

bool dense_supported(int H, int J) { return H >= 32 && J >= 64 && (H % 32) == 0 && (J % 64) == 0 && H <= 4096 && J <= 4096; }

// the layout does not depend on the device (the workspace query must not): the split count is fixed from the shape
static DenseLayout make_dense_layout(int B, int T, int U, int H, int J, size_t base) {
    DenseLayout L;
    L.R0 = B * T, L.R1 = B * U;
    L.R0p = (L.R0 + kTnK - 1) / kTnK * kTnK;                 // the pred rows start on a chunk boundary
    L.Rp = L.R0p + (L.R1 + kTnK - 1) / kTnK * kTnK;
    const int tiles = ((H + kTnT - 1) / kTnT) * ((J + kTnT - 1) / kTnT);
    const int c_all = L.Rp / kTnK, c0 = L.R0p / kTnK;
    int ns = (3 * 256 + tiles - 1) / tiles;                  // ~3 workgroups per CU of a 256-CU part
    if (ns > c_all / 8) ns = c_all / 8;                      // at least 8 chunks (128 rows) per split
    if (ns < 2) ns = 2;
    if (ns > 64) ns = 64;
    L.nsplit = ns;
    L.ns0 = (int)((long long)ns * c0 / c_all);
    if (L.ns0 < 1) L.ns0 = 1;
    if (L.ns0 > ns - 1) L.ns0 = ns - 1;
    L.db_blocks = (L.R0 + kDbRows - 1) / kDbRows;
    size_t off = base;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    L.proj_e = take((size_t)L.R0 * J * sizeof(float));
    L.proj_p = take((size_t)L.R1 * J * sizeof(float));
    L.dproj_e = take((size_t)L.R0 * J * sizeof(float));
    L.dproj_p = take((size_t)L.R1 * J * sizeof(float));
    L.Xhi = take((size_t)L.Rp * H * sizeof(df16));
    L.Xlo = take((size_t)L.Rp * H * sizeof(df16));
    L.Dhi = take((size_t)L.Rp * J * sizeof(df16));
    L.Dlo = take((size_t)L.Rp * J * sizeof(df16));
    L.Whi = take((size_t)H * J * sizeof(df16));
    L.Wlo = take((size_t)H * J * sizeof(df16));
    L.WThi = take((size_t)H * J * sizeof(df16));
    L.WTlo = take((size_t)H * J * sizeof(df16));
    L.dWpart = take((size_t)L.nsplit * H * J * sizeof(float));
    L.dbpart = take((size_t)L.db_blocks * J * sizeof(float));
    L.scal = take(256 + (size_t)(3 * kAbsBlocks + 2 * kHookBlocks) * sizeof(unsigned));  // scales + the per-block abs-max entries
    L.total = off;
    return L;
}

hipError_t dense_workspace_bytes(int B, int T, int U, int H, int J, size_t base, size_t *bytes) {
    if (!dense_supported(H, J)) return hipErrorInvalidValue;
    *bytes = make_dense_layout(B, T, U, H, J, base).total;
    return hipSuccess;
}

// pointers of the projections inside the workspace (the fused joint reads / writes them there)
void dense_proj_pointers(void *workspace, int B, int T, int U, int H, int J, size_t base, float **enc_proj, float **pred_proj,
                         float **d_enc_proj, float **d_pred_proj) {
    const DenseLayout L = make_dense_layout(B, T, U, H, J, base);
    char *ws = (char *)workspace;
    *enc_proj = (float *)(ws + L.proj_e), *pred_proj = (float *)(ws + L.proj_p);
    *d_enc_proj = (float *)(ws + L.dproj_e), *d_pred_proj = (float *)(ws + L.dproj_p);
}

enum { kSlotXe = 0, kSlotXp = 1, kSlotW = 2, kSlotDe = 3, kSlotDp = 4 };  // scal[slot] = S
// per-block abs-max entries behind the scales: X enc, X pred, W1 (kAbsBlocks each), then d enc_proj, d pred_proj (kHookBlocks each)
static unsigned *dense_maxarr(float *scal, int slot) {
    unsigned *base = (unsigned *)(scal + 64);
    return slot < 3 ? base + slot * kAbsBlocks : base + 3 * kAbsBlocks + (slot - 3) * kHookBlocks;
}
void dense_hook_pointers(void *workspace, int B, int T, int U, int H, int J, size_t base, unsigned **dmax_enc, unsigned **dmax_pred) {
    const DenseLayout L = make_dense_layout(B, T, U, H, J, base);
    float *scal = (float *)((char *)workspace + L.scal);
    *dmax_enc = dense_maxarr(scal, kSlotDe), *dmax_pred = dense_maxarr(scal, kSlotDp);
}

static unsigned dense_flat_grid(size_t n4) { return (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048); }

static void dense_absmax(const AbsmaxJob &job, hipStream_t s) {
    hipLaunchKernelGGL(dense_absmax_kernel, dim3(kAbsBlocks), dim3(256), 0, s, job);
}

static hipError_t dense_nt_launch(DenseNT &g, hipStream_t s) {
    const size_t shm = (size_t)kNtStages * kNtStage;
    hipError_t e = hipFuncSetAttribute((const void *)dense_gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    g.tiles_n = (g.N + kNtN - 1) / kNtN;
    hipLaunchKernelGGL(dense_gemm_nt_kernel, dim3((unsigned)(((g.M + kNtM - 1) / kNtM) * g.tiles_n)), dim3(kNtThreads), shm, s, g);
    return hipGetLastError();
}

// proj = [enc; pred] . W1 (+ b1 on the enc rows), into the workspace; the split image of X stays there for the backward
// expE / expP / tflag: nullable; when given (joint_dtype 0) the GEMM's epilogue also writes the fused joint's e^{2x} tables
hipError_t launch_dense_fwd(const float *enc, const float *pred, const float *W1, const float *b1, int B, int T, int U, int H, int J,
                            void *workspace, size_t base, float *expE, float *expP, float *tflag, hipStream_t s) {
    if (!dense_supported(H, J)) return hipErrorInvalidValue;
    if ((((uintptr_t)enc | (uintptr_t)pred | (uintptr_t)W1 | (uintptr_t)b1) & 15) != 0) return hipErrorInvalidValue;
    const DenseLayout L = make_dense_layout(B, T, U, H, J, base);
    char *ws = (char *)workspace;
    float *scal = (float *)(ws + L.scal);
    df16 *Xhi = (df16 *)(ws + L.Xhi), *Xlo = (df16 *)(ws + L.Xlo);
    const size_t ne = (size_t)L.R0 * H, np = (size_t)L.R1 * H, nw = (size_t)H * J;
    {
        AbsmaxJob job;
        job.x[0] = enc, job.n[0] = ne, job.out[0] = dense_maxarr(scal, kSlotXe);
        job.x[1] = pred, job.n[1] = np, job.out[1] = dense_maxarr(scal, kSlotXp);
        job.x[2] = W1, job.n[2] = nw, job.out[2] = dense_maxarr(scal, kSlotW);
        job.count = 3;
        dense_absmax(job, s);
    }
    {
        PrepJob pj;
        pj.x[0] = enc, pj.x[1] = pred, pj.hi = Xhi, pj.lo = Xlo;
        pj.n[0] = ne, pj.n[1] = np, pj.elem_off[0] = 0, pj.elem_off[1] = (size_t)L.R0p * H;
        pj.maxarr[0] = dense_maxarr(scal, kSlotXe), pj.maxarr[1] = dense_maxarr(scal, kSlotXp), pj.maxcount = kAbsBlocks;
        pj.slot[0] = kSlotXe, pj.slot[1] = kSlotXp, pj.scal = scal;
        pj.rows0 = L.R0, pj.cols = H, pj.colpart = nullptr;
        pj.pad[0] = L.R0, pj.pad[1] = L.R0p, pj.pad[2] = L.R0p + L.R1, pj.pad[3] = L.Rp;
        pj.W = W1, pj.whi = (df16 *)(ws + L.Whi), pj.wlo = (df16 *)(ws + L.Wlo), pj.thi = (df16 *)(ws + L.WThi), pj.tlo = (df16 *)(ws + L.WTlo);
        pj.H = H, pj.J = J, pj.wslot = kSlotW, pj.wmaxarr = dense_maxarr(scal, kSlotW);
        pj.g[0] = dense_flat_grid(ne / 4), pj.g[1] = dense_flat_grid(np / 4), pj.g[2] = 32, pj.g[3] = 512;
        hipLaunchKernelGGL(dense_prepare_kernel, dim3(pj.g[0] + pj.g[1] + pj.g[2] + pj.g[3]), dim3(256), 0, s, pj);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    DenseNT g;
    g.Ahi = Xhi, g.Alo = Xlo, g.Bhi = (const df16 *)(ws + L.WThi), g.Blo = (const df16 *)(ws + L.WTlo);
    g.C0 = (float *)(ws + L.proj_e), g.C1 = (float *)(ws + L.proj_p), g.bias = b1, g.scal = scal;
    g.E0 = expE, g.E1 = expP, g.tflag = tflag;
    g.sa0 = kSlotXe, g.sa1 = kSlotXp, g.sb = kSlotW;
    g.M = L.R0p + L.R1, g.N = J, g.K = H, g.R0 = L.R0, g.R0p = L.R0p, g.R1 = L.R1, g.ldc = J;
    return dense_nt_launch(g, s);
}

// d enc = d enc_proj . W1^T, d pred = d pred_proj . W1^T, dW1 = enc^T . d enc_proj + pred^T . d pred_proj, db1 = sum_rows d enc_proj
// (d enc_proj / d pred_proj are read from the workspace, where the fused joint's backward left them; X's split image and the
// weight images are the forward's)
hipError_t launch_dense_bwd(int B, int T, int U, int H, int J, float *d_enc, float *d_pred, float *dW1, float *db1, void *workspace,
                            size_t base, hipStream_t s) {
    if (!dense_supported(H, J)) return hipErrorInvalidValue;
    if ((((uintptr_t)d_enc | (uintptr_t)d_pred) & 15) != 0) return hipErrorInvalidValue;  // 16-byte row-segment stores
    const DenseLayout L = make_dense_layout(B, T, U, H, J, base);
    char *ws = (char *)workspace;
    float *scal = (float *)(ws + L.scal);
    hipError_t e;
    df16 *Dhi = (df16 *)(ws + L.Dhi), *Dlo = (df16 *)(ws + L.Dlo);
    const float *de = (const float *)(ws + L.dproj_e), *dp = (const float *)(ws + L.dproj_p);
    const size_t ne = (size_t)L.R0 * J, np = (size_t)L.R1 * J;
    // (the abs-max entries of d enc_proj / d pred_proj were left by the fused joint's reductions: JointHooks)
    {
        PrepJob pj;
        pj.x[0] = de, pj.x[1] = dp, pj.hi = Dhi, pj.lo = Dlo;
        pj.n[0] = ne, pj.n[1] = np, pj.elem_off[0] = 0, pj.elem_off[1] = (size_t)L.R0p * J;
        pj.maxarr[0] = dense_maxarr(scal, kSlotDe), pj.maxarr[1] = dense_maxarr(scal, kSlotDp), pj.maxcount = kHookBlocks;
        pj.slot[0] = kSlotDe, pj.slot[1] = kSlotDp, pj.scal = scal;
        pj.rows0 = L.R0, pj.cols = J, pj.colpart = (float *)(ws + L.dbpart);  // (db1's column partials ride on the enc rows)
        pj.pad[0] = L.R0, pj.pad[1] = L.R0p, pj.pad[2] = L.R0p + L.R1, pj.pad[3] = L.Rp;
        pj.W = nullptr, pj.whi = pj.wlo = pj.thi = pj.tlo = nullptr, pj.H = H, pj.J = J, pj.wslot = kSlotW, pj.wmaxarr = nullptr;
        pj.g[0] = (unsigned)L.db_blocks, pj.g[1] = dense_flat_grid(np / 4), pj.g[2] = 32, pj.g[3] = 0;
        hipLaunchKernelGGL(dense_prepare_kernel, dim3(pj.g[0] + pj.g[1] + pj.g[2]), dim3(256), 0, s, pj);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // dX = dproj . W1^T  (NT: A = dproj rows, B = W1 rows [H][J], K = J)
    {
        DenseNT g;
        g.Ahi = Dhi, g.Alo = Dlo, g.Bhi = (const df16 *)(ws + L.Whi), g.Blo = (const df16 *)(ws + L.Wlo);
        g.C0 = d_enc, g.C1 = d_pred, g.bias = nullptr, g.scal = scal, g.sa0 = kSlotDe, g.sa1 = kSlotDp, g.sb = kSlotW;
        g.E0 = g.E1 = g.tflag = nullptr;
        g.M = L.R0p + L.R1, g.N = H, g.K = J, g.R0 = L.R0, g.R0p = L.R0p, g.R1 = L.R1, g.ldc = H;
        if ((e = dense_nt_launch(g, s)) != hipSuccess) return e;
    }
    // dW1 = X^T . dproj over the enc rows and the pred rows (TN, split over K; each range with its own scales)
    {
        const size_t shm = (size_t)kTnStages * kTnStage;
        DenseTN g;
        g.Ahi = (const df16 *)(ws + L.Xhi), g.Alo = (const df16 *)(ws + L.Xlo), g.Bhi = Dhi, g.Blo = Dlo;
        g.P = (float *)(ws + L.dWpart), g.scal = scal;
        g.sa0 = kSlotXe, g.sb0 = kSlotDe, g.sa1 = kSlotXp, g.sb1 = kSlotDp;
        g.M = H, g.N = J, g.Kchunks = L.Rp / kTnK, g.c0 = L.R0p / kTnK;
        g.tiles_m = (H + kTnT - 1) / kTnT, g.tiles_n = (J + kTnT - 1) / kTnT, g.nsplit = L.nsplit, g.ns0 = L.ns0;
        hipLaunchKernelGGL(dense_gemm_tn_kernel, dim3((unsigned)(g.tiles_m * g.tiles_n * g.nsplit)), dim3(256), shm, s, g);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        return dense_reduce2(dW1, g.P, L.nsplit, (size_t)H * J, db1, (const float *)(ws + L.dbpart), L.db_blocks, (size_t)J, s);
    }
}

}  // namespace rnnt
