// joint_kernels.hip -- joint network fused with the transducer loss (gfx950): f32-GRADE products on the f16 MFMA units.
//
// Replaces, for the hot path, the tail of the reference model and everything TF autodiff does
// behind it:
//   model.py:158-160  z0 = enc[:,:,None,:] + pred[:,None,:,:]
//   model.py:162-163  h  = tanh(z0 @ W1 + b1)
//   model.py:165-166  y  = h @ W2 + b2                         -> logits [B,T,U,V]
//   run_rnnt.py:284   tape.gradient(...) through those three ops
// The first Dense layer is factored exactly (W1^T(e+p)+b1 = (W1^T e + b1) + W1^T p), so the kernels
// take the two small projections  enc_proj [B,T,J], pred_proj [B,U,J]  (dense_kernels.hip, or library GEMMs on the host
// side) and never materialise the [B,T,U,J] or [B,T,U,V] tensors:
//
//   forward   (joint_fwd_kernel, J <= 640; joint_phase1s_kernel for 640 < J <= 704)
//             logits tile = tanh(A_t + C_u) . W2 + b2 as a split-precision product on v_mfma_f32_32x32x16_f16 (both operands
//             as binary16 hi + lo parts, f32 accumulation; W2 scaled by a power of two into binary16's range first), softmax
//             epilogue in registers: the lattice's edge PROBABILITIES for the linear-domain sweeps (round 5: rnnt_lin.h,
//             lin_sweep_kernel -- the lattice the loss op runs on) and the parked logits tile (V <= 32 floats per cell)
//   backward  (joint_cellrec_kernel + joint_bwd_kernel; joint_dl_kernel + joint_phase2s_kernel for the wide joint)
//             per-cell gradient set-up from the lattice (mantissas + frames, range certificate; log-domain values for an
//             utterance that was handed back), then dlogits, dh = dl . W2^T, dz = dh * (1-h^2), d enc_proj = sum_u dz,
//             d pred_proj = sum_t dz, dW2 = h^T . dl in one persistent producer / consumer kernel.  h is recomputed in the
//             MFMA C/D register layout, which is at the same time a valid A-operand layout for dW2 (the K-slot <-> lattice
//             column assignment of a dot product is free), so no data moves between the three products.
//   hand-back (joint_redo_kernel) utterances the linear lattice cannot represent are redone in the log domain from the parked
//             logits by a team of workgroups (rnnt_redo.h), exactly as in the loss op.
//   reductions over u-tiles / row splits / blocks go through partial buffers summed in a fixed
//   order (deterministic; no floating-point atomics), one launch for all four outputs.
//
// Limits (checked at the boundary): V <= 32 (one MFMA column tile), J % 64 == 0, J <= 704.
#include "rnnt_common.h"
#include "rnnt_cell.h"
#include "rnnt_redo.h"

#include <math.h>
#ifdef JH_TRACE
#include <stdio.h>
#include <stdlib.h>
#endif

namespace rnnt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 jf16;
typedef _Float16 jh8 __attribute__((ext_vector_type(8)));
typedef float jf2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void jglb_cvoid;

__device__ __forceinline__ float jex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float jlg2(float x) { return __builtin_amdgcn_logf(x); }
// tanh(x) = 1 - 2/(1+e^{2x}); saturates correctly at +-inf, absolute error ~1e-7
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + jex2(x * 2.8853900817779268f));
}
// tanh(a + c) from the tabulated factors ea = e^{2a}, ec = e^{2c} (joint_prep_kernel): one multiply-add, one reciprocal,
// one multiply-add instead of an exponential and a reciprocal.  Exact to ~1e-7 while |a|, |c| <= kExpTabLimit (both
// factors are normal f32 numbers; an overflowing product gives +1, an underflowing one -1, like tanh).  Beyond that
// the prep kernel raises a flag and the kernels evaluate fast_tanh(a + c) on the raw projections.
__device__ __forceinline__ float tanh_from_exp(float ea, float ec) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fmaf(ea, ec, 1.0f));
}
// r = (1 - tanh(a + c)) / 2 = 1 / (1 + e^{2(a + c)}), the quantity joint_fwd_kernel puts on the matrix units (same tables, same
// saturation behaviour: 0 for an overflowing product, 1 for an underflowing one)
__device__ __forceinline__ float r_from_exp(float ea, float ec) { return __builtin_amdgcn_rcpf(fmaf(ea, ec, 1.0f)); }
__device__ __forceinline__ float fast_r(float x) { return __builtin_amdgcn_rcpf(1.0f + jex2(x * 2.8853900817779268f)); }
// row of the 32x32 MFMA C/D tile held in register `reg` of a lane in half `half` (= lane >> 5)
__device__ __forceinline__ constexpr int cd_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

#ifdef JH_TRACE
#define JT1(slot)                                                                   \
    do {                                                                           \
        if (trc && lane == 0) trc[(slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define JT1(slot) do { } while (0)
#endif

struct JointParams {
    LossParams lp;  // lattice workspace, labels, lengths, costs, cost_scale (acts/grads unused)
    const float *enc_proj, *pred_proj, *W2, *b2;
    float *dl;      // [cells][32] logits parked by the forward pass (read-only in every backward: a backward call may be repeated)
    float *dlg;     // [cells][32] dlogits of the two-kernel backward (joint_dl_kernel -> joint_phase2s / joint_phase2 kernels)
    float4 *rec;    // [cells] per-cell gradient set-up of the single-kernel backward (joint_cellrec_kernel)
    int *reclab;    // [cells] label of the cell, or -1
    float2 *xbl;    // [cells] blank / label logits of the cell, written by joint_fwd_kernel for joint_cellrec_kernel
    int *plan;       // joint_rowplan_kernel: [0] = target weight per workgroup, [1 + k] = first item of workgroup k (k = 0 .. nblk),
                     // [2 + kBwdMaxBlocks + i] = weight of the items before item i (i = 0 .. n_items), then {rows visited, rows inside the utterances}
    int visit_all;   // 1: every lattice row of an utterance counts as occupied (RNNT_VISIT_ALL: no occupancy floor)
    uint8_t *live8;  // [B][n_ut][4 n_tr32] one BIT per (lattice row, u-tile): some cell of the tile has occupancy above 2^-kOccFloor
                     // (joint_cellrec_kernel; joint_bwd_kernel skips the other rows -- see kOccFloor); n_tr32 = ceil(T / 32)
    float *dApart;  // [n_ut][B][T][J]
    float *dCpart;  // [n_ts][B][U][J]
    float *dWpart;  // [B*n_ut*n_ts][J][32]
    float *dbpart;  // [ceil(cells/256)][32]
    float *d_enc_proj, *d_pred_proj, *dW2, *db2;
    float *expE, *expP;  // [B][T][J], [B][U][J]  e^{2 x} tables of the two projections
    float *tflag;        // [0] != 0: some |projection| exceeds kExpTabLimit, use the raw projections + fast_tanh
                         // [1] != 0: some |W2| exceeds kRFormLimit: joint_fwd_kernel accumulates h = tanh(.) instead of r = (1 - h) / 2
                         // [2] = 1 / s2: W2 enters every product as s2 W2, s2 the power of two that puts max |W2| into
                         //       [2^13, 2^14) (binary16 hi + lo parts then carry 22 significand bits whatever W2's magnitude)
                         // [3] == 1: the workspace holds the state of a whole-network forward call (JointHooks::prep_mode)
    jf16 *W2s;           // [VT][J/16][2 (hi, lo)][64 lanes][8]: s2 W2 as binary16 hi + lo parts in MFMA fragment order, per vocabulary tile
    float *b2s;          // [VT][2][32]: b2[v] + sum_j W2[j][v] as an f32 hi + lo pair (-1e30 / 0 beyond V), for joint_fwd_kernel
#ifdef JH_TRACE
    long long *trace;    // dev builds only: s_memtime stamps of one workgroup of phase 1 and one of phase 2
#endif
    int J, n_ut, TR, n_tr, TS, n_ts;
    int VT, vt;       // vocabulary tiles of 32 symbols (1, or 2 for 32 < V <= 64: round 5) and the tile THIS launch works on; the
                      // parked logits are [cells][32 VT], W2s / b2s / dWpart / dbpart hold VT consecutive tile images
    int tables_ready; // the e^{2x} tables and tflag[0] were written by the caller (the dense layer's epilogue): prep does W2 only
    int logits_only;  // compute_rnnt_joint_logits: full lengths written by the prep kernel, only the parked logits are kept
    int nblk;         // workgroups per J group of joint_bwd_kernel (sizes what the reduction reads of the partial buffers)
    int need_state;   // backward-only whole-network call: the reductions return NaN unless tflag[3] says the state is there
};

constexpr int kMaxVT = 4;  // vocabulary tiles of 32 symbols the f32-grade joint takes (V <= 128)
constexpr float kRFormLimit = 4096.0f;  // largest |W2| the forward kernel's r = (1 - h) / 2 accumulation is used for (joint_prep_kernel)

// tables for tanh_from_exp + the overflow flag (zeroed before the launch)
__global__ __launch_bounds__(256) void joint_prep_kernel(const JointParams jp) {
    const LossParams &p = jp.lp;
    const size_t nE = (size_t)p.B * p.T * jp.J, nP = (size_t)p.B * p.U * jp.J;
    bool big = false;
    auto one = [&](const float x) {
        big |= exp_tab_out_of_range(x);  // also catches NaN
        return exp_tab(x);
    };
    if (jp.tables_ready) {
        // nothing to tabulate
    } else if (((nE | nP) & 3) == 0 && ((((uintptr_t)jp.enc_proj | (uintptr_t)jp.pred_proj) & 15) == 0)) {  // 16-byte accesses (J % 4 == 0)
        const size_t nE4 = nE >> 2, n4 = (nE + nP) >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const float4 x = (i < nE4) ? ((const float4 *)jp.enc_proj)[i] : ((const float4 *)jp.pred_proj)[i - nE4];
            const float4 ex = make_float4(one(x.x), one(x.y), one(x.z), one(x.w));
            if (i < nE4) ((float4 *)jp.expE)[i] = ex;
            else ((float4 *)jp.expP)[i - nE4] = ex;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nE + nP; i += (size_t)gridDim.x * 256) {
            const float ex = one((i < nE) ? jp.enc_proj[i] : jp.pred_proj[i - nE]);
            if (i < nE) jp.expE[i] = ex;
            else jp.expP[i - nE] = ex;
        }
    }
    if (__any(big) && (threadIdx.x & 63) == 0) jp.tflag[0] = 1.0f;
    // tflag[3]: "the workspace holds the state of a whole-network forward call" (tables by the dense layer's epilogue, W2 images
    // by this launch).  Any other call on the workspace zero-fills the flag words; a backward-only call that skips this kernel
    // (JointHooks::prep_mode 2) checks the word and returns NaN gradients instead of numbers from someone else's tables.
    if (jp.tables_ready && blockIdx.x == 0 && threadIdx.x == 0) jp.tflag[3] = 1.0f;
    if (jp.logits_only && blockIdx.x == 1)  // every lattice cell is wanted: full lengths (the arrays live in the workspace)
        for (int b = threadIdx.x; b < p.B; b += 256) {
            const_cast<int *>(p.input_lengths)[b] = p.T;
            const_cast<int *>(p.label_lengths)[b] = p.U - 1;
        }
    // W2 = hi + lo with both parts binary16, laid out as the fragments of v_mfma_f32_32x32x16_f16: lane l of k-step ks holds
    // s2 W2[16 ks + 8 (l >> 5) + 0..7][l & 31] (zero beyond V).  s2 = the power of two that puts max |W2| into [2^13, 2^14):
    // hi and lo are then both normal binary16 numbers for every weight within 2^-13 of the largest (22 significand bits
    // together; without the scale the lo part of a weight below 2^-3 sat in binary16's subnormals), and weights beyond
    // binary16's range (65504) need no other kernels.  A non-finite weight gives s2 = 1 and NaN results, as it should.
    // Block 0 reads W2 once, column by column, for the bias table AND max |W2| (it publishes 1 / s2 and the r / h choice);
    // blocks 1 .. build the fragment image and find max |W2| themselves first (J V <= 22,528 floats out of L2, all loads of a
    // thread in flight at once): the two run side by side, under the table pass of the other blocks.
    __shared__ float wred[4];
    auto scale_of = [&](float wmax, float &s2, float &inv2) {  // wave maxima -> the block's -> the scale
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
        if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = wmax;
        __syncthreads();
        wmax = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        s2 = 1.0f, inv2 = 1.0f;
        if (wmax > 0.f && wmax < 3.0e38f) {
            const int e = ilogbf(wmax);  // wmax in [2^e, 2^(e+1))
            s2 = ldexpf(1.0f, 13 - e), inv2 = ldexpf(1.0f, e - 13);
        }
        return wmax;
    };
    if (blockIdx.x == 0) {
        // joint_fwd_kernel accumulates r = (1 - h) / 2: logits = (b2 + sum_j W2) - 2 W2^T r.  That forms every logit as the
        // difference of two sums of the magnitude of the LARGEST weight: fine while those are of the order of the logits, a
        // cancellation otherwise (a unit with h = 0 and a weight of 1e5 costs 4e-3 of absolute logit error in f32 accumulators).
        // Beyond kRFormLimit the kernel accumulates h itself and the table holds b2 alone.
        __shared__ double part[kMaxVT][8][32];
        const int v = threadIdx.x & 31, q = threadIdx.x >> 5;
        double sum[kMaxVT] = {0.0, 0.0, 0.0, 0.0};
        float wmax = 0.f;
#pragma unroll
        for (int tile = 0; tile < kMaxVT; ++tile) {
            const int vv = 32 * tile + v;
            if (vv < p.V) {  // 32 loads in flight per thread: the block is alone on its CU and waits out a full memory latency per
                             // round (with 8 per round, ten rounds at J = 640, this block alone made the kernel 28 us long)
                constexpr int kRound = 32;
                float w[kRound];
                for (int j0 = q; j0 < jp.J; j0 += 8 * kRound) {
#pragma unroll
                    for (int k = 0; k < kRound; ++k) w[k] = (j0 + 8 * k < jp.J) ? jp.W2[(size_t)(j0 + 8 * k) * p.V + vv] : 0.f;
#pragma unroll
                    for (int k = 0; k < kRound; ++k) sum[tile] += (double)w[k], wmax = fmaxf(wmax, fabsf(w[k]));  // (a NaN operand is ignored by v_max)
                }
            }
        }
        float s2, inv2;
        wmax = scale_of(wmax, s2, inv2);
        const bool hform = !(wmax <= kRFormLimit);
        if (threadIdx.x == 0) jp.tflag[2] = inv2, jp.tflag[1] = hform ? 1.0f : 0.f;
#pragma unroll
        for (int tile = 0; tile < kMaxVT; ++tile) part[tile][q][v] = hform ? 0.0 : sum[tile];
        __syncthreads();
        if (threadIdx.x < 32 * jp.VT) {
            const int tile = threadIdx.x >> 5, vv = threadIdx.x;  // (v = threadIdx.x & 31)
            double t = (vv < p.V) ? (double)jp.b2[vv] : -1.0e30;
            for (int k = 0; k < 8; ++k) t += part[tile][k][v];
            const float hi = (float)t;
            jp.b2s[64 * tile + v] = hi, jp.b2s[64 * tile + 32 + v] = (vv < p.V) ? (float)(t - (double)hi) : 0.f;
        }
        return;
    }
    const int nfrag1 = (jp.J / 16) * 64, nfrag = nfrag1 * jp.VT, fblk = (int)blockIdx.x - 1;
    if (fblk * 256 < nfrag) {
        const int nW = jp.J * p.V;
        float wmax = 0.f;
        for (int i0 = 0; i0 < nW; i0 += 256 * 64) {  // 64 loads in flight per thread and round (J V <= 81,920 floats: five rounds at most)
            float w[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int i = i0 + 256 * k + (int)threadIdx.x;
                w[k] = (i < nW) ? fabsf(jp.W2[i]) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) wmax = fmaxf(wmax, w[k]);
        }
        float s2, inv2;
        scale_of(wmax, s2, inv2);
        for (int i = fblk * 256 + threadIdx.x; i < nfrag; i += ((int)gridDim.x - 1) * 256) {
            const int tile = i / nfrag1, i1 = i - tile * nfrag1;
            const int ks = i1 >> 6, l = i1 & 63, v = 32 * tile + (l & 31);
            jh8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 16 * ks + 8 * (l >> 5) + e;
                const float w = (v < p.V) ? jp.W2[(size_t)j * p.V + v] * s2 : 0.f;
                hi[e] = (jf16)w;
                lo[e] = (jf16)(w - (float)hi[e]);
            }
            jf16 *img = jp.W2s + (size_t)tile * jp.J * 64;  // a tile image: J / 16 k-steps x (hi, lo) x 64 lanes x 8 halves
            *(jh8 *)(img + ((size_t)(ks * 2 + 0) * 64 + l) * 8) = hi;
            *(jh8 *)(img + ((size_t)(ks * 2 + 1) * 64 + l) * 8) = lo;
        }
    }
}

constexpr int kP1Waves = 8;    // phase-1 workgroup = 8 waves (2 per SIMD: one wave's tanh VALU hides the other's MFMA issue)
constexpr int kStagePad = 33;  // row stride of the per-wave 32x32 staging tiles (bank-conflict free both ways)

typedef _Float16 jh2 __attribute__((ext_vector_type(2)));
// x -> binary16 hi (round to nearest even) and lo = binary16(x - hi).  The residual is one v_fma_mix_f32 per value: it reads
// the binary16 operand straight out of the packed register (no v_cvt_f32_f16), hi * -1 + x is exact in f32.
// (v_fma_mixlo_f16 / v_fma_mixhi_f16 would also do the final rounding, but they issue at half rate on gfx950 -- 8.2 against
// 4.3 cycles, profiles/r02_probe_rates.txt -- and lose to v_fma_mix_f32 + half a v_cvt_pk_f16_f32.)
// NOTE on hazards: the compiler does not see into inline asm, so it cannot insert the wait state gfx950 needs between a
// transcendental (v_rcp_f32, v_exp_f32 ...) and a VALU instruction that reads its result.  Both forms leave the hi conversion
// to the compiler, which therefore always sits between a reciprocal and the asm residuals (they depend on it); the _trans
// name marks the call sites whose inputs come straight out of v_rcp_f32.
__device__ __forceinline__ void split_pair(const float x0, const float x1, jh2 &hi, jh2 &lo) {
    hi[0] = (jf16)x0, hi[1] = (jf16)x1;
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
    lo[0] = (jf16)l0, lo[1] = (jf16)l1;
}
__device__ __forceinline__ void split_pair_trans(const float x0, const float x1, jh2 &hi, jh2 &lo) {
    hi[0] = (jf16)x0, hi[1] = (jf16)x1;
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
    lo[0] = (jf16)l0, lo[1] = (jf16)l1;
}
__device__ __forceinline__ void split_h8(const float (&x)[8], jh8 &hi, jh8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        jh2 h2, l2;
        split_pair(x[e], x[e + 1], h2, l2);
        hi[e] = h2[0], hi[e + 1] = h2[1];
        lo[e] = l2[0], lo[e + 1] = l2[1];
    }
}
__device__ __forceinline__ f32x16 mfma3(const jh8 ahi, const jh8 alo, const jh8 bhi, const jh8 blo, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------------
// phase 1, split-precision form (default): the same tile, epilogue and outputs as joint_phase1_kernel, but the J x V
// product runs on v_mfma_f32_32x32x16_f16 with BOTH operands split into binary16 hi + lo parts:
//     h . W2  =  h_hi . W_hi  +  h_lo . W_hi  +  h_hi . W_lo      (+ h_lo . W_lo ~ 2^-22, dropped)
// Products of binary16 numbers are exact in f32 and the accumulation is f32, so the logits carry f32-grade error (NumPy
// emulation at J = 640: rms 4.4e-7 against 3.2e-7 for an f32 matmul) while three of these MFMAs cover 16 joint units in 96
// matrix-pipe cycles instead of 512 for eight v_mfma_f32_32x32x2_f32.  The kernel becomes bound by the tanh generation.
// A lane (cell i = lane & 31, half = lane >> 5) builds the 8 consecutive joint units 16 ks + 8 half + 0..7 of its cell.
// LDS: Ct [J][32] | W2 hi/lo fragments of 4 k-steps, double-buffered by LDS-DMA [2][8 KB] | Arow [8][J] | stage [8][32][33]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kP1Waves * 64) void joint_phase1s_kernel(const JointParams jp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LossParams &p = jp.lp;
    const int J = jp.J, V = p.V;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *Ct = lds;                                  // [J][32]
    char *W2c = (char *)(Ct + J * 32);                // [2][8192]
    float *Arow = (float *)(W2c + 2 * 8192);          // [kP1Waves][J]
    float *stage = Arow + kP1Waves * J;               // [kP1Waves][32][kStagePad]
    float *my_arow = Arow + wave * J;
    float *my_stage = stage + wave * 32 * kStagePad;
    const bool slow = jp.tflag[0] != 0.f;  // kernel-uniform
    const float w2inv = jp.tflag[2];       // the W2 images hold s2 W2 (joint_prep_kernel)
    const float *Etab = slow ? jp.enc_proj : jp.expE, *Ptab = slow ? jp.pred_proj : jp.expP;

    int bid = blockIdx.x;
    const int tr = bid % jp.n_tr;
    bid /= jp.n_tr;
    const int ut = bid % jp.n_ut;
    const int b = bid / jp.n_ut;
    const int u0 = ut * 32;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int t_begin = tr * jp.TR, t_end = min(min(t_begin + jp.TR, p.T), Tb);
    const bool tile_live = (t_begin < t_end) && (u0 < Ub);

    if (tile_live) {
        for (int idx = tid; idx < 32 * (J / 4); idx += kP1Waves * 64) {
            const int u = idx & 31, j4 = idx >> 5;
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u0 + u < p.U) c4 = *(const float4 *)(Ptab + ((size_t)b * p.U + u0 + u) * J + j4 * 4);
            Ct[(j4 * 4 + 0) * 32 + u] = c4.x;
            Ct[(j4 * 4 + 1) * 32 + u] = c4.y;
            Ct[(j4 * 4 + 2) * 32 + u] = c4.z;
            Ct[(j4 * 4 + 3) * 32 + u] = c4.w;
        }
    }
    const int n_iter = tile_live ? (t_end - t_begin + kP1Waves - 1) / kP1Waves : 0;
    const int nchunk = J / 64;  // 4 k-steps (64 joint units, 8 KB of hi/lo fragments) per chunk; wave w copies KB w
    auto w2_dma = [&](const int jc) {
        const char *src = (const char *)jp.W2s + (size_t)jc * 8192 + wave * 1024 + lane * 16;
        lds_dma16(src, W2c + (jc & 1) * 8192 + wave * 1024);
    };
    for (int it = 0; it < n_iter; ++it) {
        const int t = t_begin + it * kP1Waves + wave;
        const bool active = t < t_end;  // wave-uniform
        if (active)
            for (int j = lane; j < J; j += 64) my_arow[j] = Etab[((size_t)b * p.T + t) * J + j];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        w2_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk 0 (and, first time round, Ct / Arow) visible
        for (int jc = 0; jc < nchunk; ++jc) {
            const char *wbuf = W2c + (jc & 1) * 8192;
            if (jc + 1 < nchunk) w2_dma(jc + 1);
            if (active) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int j0 = jc * 64 + k4 * 16 + 8 * half;
                    const float4 a0 = *(const float4 *)(my_arow + j0), a1 = *(const float4 *)(my_arow + j0 + 4);
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    float h[8];
                    if (!slow) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = tanh_from_exp(av[e], Ct[(j0 + e) * 32 + l31]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = fast_tanh(av[e] + Ct[(j0 + e) * 32 + l31]);
                    }
                    jh8 hi, lo;
                    split_h8(h, hi, lo);
                    const jh8 wh = *(const jh8 *)(wbuf + k4 * 2048 + lane * 16);
                    const jh8 wl = *(const jh8 *)(wbuf + k4 * 2048 + 1024 + lane * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, wl, acc, 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's piece of the next chunk has landed
            __syncthreads();  // next chunk visible; everyone is done with the buffer it will overwrite after that
        }
        // ---- epilogue: logits tile -> LDS, one lattice cell per lane (log2-domain edge weights + lse: the wide joint stays on
        //      the log-domain sweeps), park the tile
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) my_stage[cd_row(r, half) * kStagePad + l31] = acc[r];
        }
        __syncthreads();
        if (active && lane < 32) {
            Cell cl;
            cl.b = b, cl.t = t, cl.u = u0 + lane, cl.Tb = Tb, cl.Ub = Ub;
            cl.valid = cl.u < Ub;
            float *xs = my_stage + lane * kStagePad;
            if (cl.valid) {
                const uint32_t c = ((uint32_t)(b * p.T + t)) * (uint32_t)p.U + (uint32_t)cl.u;
                float m = -INFINITY;
                for (int v = 0; v < V; ++v) {
                    xs[v] = fmaf(xs[v], w2inv, jp.b2[v]);
                    m = fmaxf(m, xs[v]);
                }
                float ssum = 0.f;
                const float nml = -m * kLog2e;
                for (int v = 0; v < V; ++v) ssum += jex2(fmaf(xs[v], kLog2e, nml));
                const float lg2s = jlg2(ssum);
                const float lse = m + kLn2 * lg2s;
                const bool blank_stays = (cl.t < Tb - 1) || (cl.u == Ub - 1);
                const float ob = blank_stays ? fmaf(xs[p.blank] - m, kLog2e, -lg2s) : kNeg;
                float ol = kNeg;
                if (cl.u < Ub - 1) {
                    const int lab = min(max(p.labels[(size_t)b * (p.U - 1) + cl.u], 0), V - 1);
                    ol = fmaf(xs[lab] - m, kLog2e, -lg2s);
                }
                p.lse[c] = lse;
                const size_t wi = ((size_t)b * p.Nr + (cl.t + cl.u)) * p.Up + cl.u;
                ((float2 *)p.W)[wi] = make_float2(ob, ol);
            }
        }
        __syncthreads();
        if (active)
            for (int e = lane; e < 1024; e += 64) {
                const int uu = e >> 5, v = e & 31;
                if (u0 + uu < Ub) {
                    const size_t c = ((size_t)(b * p.T + t)) * p.U + u0 + uu;
                    jp.dl[c * 32 + v] = my_stage[uu * kStagePad + v];
                }
            }
        __syncthreads();  // staging tiles and Arow are rewritten by the next iteration
    }
}

// ---------------------------------------------------------------------------------------------
// forward, resident-table form (default for J <= 640): the same outputs as joint_phase1s_kernel -- softmax denominators, lattice
// edge weights, the parked logits tile -- from a kernel whose main loop has NO barrier and NO per-wave LDS staging:
//   * persistent 16-wave workgroups, one per CU, striding over (row tile, utterance, u-tile) items;
//   * LDS holds exactly two tables: the W2 hi/lo fragment image (128 J bytes, loaded once per workgroup) and the item's
//     e^{2 pred_proj} tile Ct [32 u][J] (128 J bytes, rows XOR-swizzled by u so that a lane's two 16-byte reads per k-step
//     are conflict-free), both brought in by LDS-DMA; 256 J bytes = all 160 KiB at J = 640;
//   * the product is formed TRANSPOSED, logits^T = W2^T . h^T (A = W2 fragments, B = h): in the 32x32 C/D layout a LANE then
//     owns one lattice cell and its 16 registers run over half of the vocabulary (the other half sits in lane ^ 32), so bias,
//     log-softmax and the blank / label picks are register arithmetic plus two v_permlane32_swap -- no staging tile;
//   * a wave carries TWO lattice rows at once (two accumulators): every Ct / W2 fragment read from LDS feeds both;
//   * the e^{2 enc_proj} addends of a row are wave-uniform per half: plain 16-byte loads (L1-served), no LDS copy.
// ---------------------------------------------------------------------------------------------
constexpr int kFwdWaves = 16;
constexpr int kFwdRows = 2 * kFwdWaves;  // lattice rows per item: one row pair per wave

// lane (16-lane row R, position i) <- lane (R, E): the enc-side addends of a k-step sit one per lane (see joint_fwd_kernel)
template <int E>
__device__ __forceinline__ float row_bcast(const float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + E /*row_newbcast:E*/, 0xf, 0xf, true));
}
// The two rows of a wave share every pred-side factor: their enc-side addends travel as a PAIR -- ONE 64-bit DPP broadcast per
// joint unit (v_mov_b64_dpp takes row_newbcast and overlaps with the matrix pipe like any VALU move: probe_pk.hip) instead of two
// 32-bit ones: -30 us of the kernel's 630 at B32 T600 U150 J640.  The multiply-adds stay scalar (see bwd_consumer on packed f32).
template <int E>
__device__ __forceinline__ jf2 row_bcast2(const jf2 x) {
    const long long v = __builtin_amdgcn_update_dpp((long long)0, __builtin_bit_cast(long long, x), 0x150 + E /*row_newbcast:E*/, 0xf, 0xf, true);
    return __builtin_bit_cast(jf2, v);
}
template <int E, bool SLOW, bool HFORM>
__device__ __forceinline__ void fwd_h_pair(const jf2 ea01, const float ec, float &h0, float &h1) {
    const jf2 a = row_bcast2<E>(ea01);
    if (!SLOW) {  // r = (1 - h) / 2, see fwd_row_epilogue
        h0 = r_from_exp(a[0], ec), h1 = r_from_exp(a[1], ec);
    } else {
        h0 = fast_r(a[0] + ec), h1 = fast_r(a[1] + ec);
    }
    if (HFORM) h0 = fmaf(h0, -2.0f, 1.0f), h1 = fmaf(h1, -2.0f, 1.0f);  // h itself (huge weights: see joint_prep_kernel)
}
// The J-long product of one row pair: acc += W2^T . r^T, A = W2 fragments (row = symbol), B = r (column = cell), with
// r = (1 - h) / 2 = 1 / (1 + e^{2(a + c)}) (see fwd_row_epilogue).  Measured alternatives (profiles/r02_notes.md): building
// step ks + 1's fragments between the MFMAs of step ks (one MFMA per ~13 VALU instructions, pinned with sched_barrier) is
// slower (621 vs 605 us): the MFMAs do not hide behind freshly written operands the way they do in an operand-invariant probe.
struct FwdFrags {
    jh8 hi0, lo0, hi1, lo1;  // B fragments of the two rows, binary16 hi + lo parts
};
// units E, E + 1 (already evaluated: row 0 in x0, row 1 in x1) -> one packed hi / lo dword per row
template <int E>
__device__ __forceinline__ void fwd_split_quad(const float (&x0)[2], const float (&x1)[2], FwdFrags &f) {
    jh2 h, l;
    split_pair_trans(x0[0], x0[1], h, l);
    f.hi0[E] = h[0], f.hi0[E + 1] = h[1], f.lo0[E] = l[0], f.lo0[E + 1] = l[1];
    split_pair_trans(x1[0], x1[1], h, l);
    f.hi1[E] = h[0], f.hi1[E + 1] = h[1], f.lo1[E] = l[0], f.lo1[E + 1] = l[1];
}
// The enc-side addends of FOUR k-steps arrive with ONE coalesced dword load per lane (lane (row R, position i) <- unit
// 16 (4 q + R) + i: 256 contiguous bytes) instead of one load per k-step (the knock-out of those loads: -11 % of the kernel).
// Every 16-lane row must then hold "its" k-step's 16 units -- rows 2, 3 (half 1) rotated by 8, see joint_fwd_kernel -- for
// row_newbcast to pick from: x -> (row_ror:8 copy) -> v_permlane32_swap -> 2 x v_permlane16_swap gives the four registers.
struct FwdAddends {
    float y[4];
};
__device__ __forceinline__ FwdAddends fwd_spread(const float x) {
    const uint32_t xi = __float_as_uint(x);
    const uint32_t xr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xi, 0x128 /*row_ror:8*/, 0xf, 0xf, false);
    const auto pq = __builtin_amdgcn_permlane32_swap(xi, xr, false, false);  // p = [x0 x1 x'0 x'1], q = [x2 x3 x'2 x'3] (rows)
    const auto y01 = __builtin_amdgcn_permlane16_swap(pq[0], pq[0], false, false);
    const auto y23 = __builtin_amdgcn_permlane16_swap(pq[1], pq[1], false, false);
    FwdAddends f;
    f.y[0] = __uint_as_float(y01[0]), f.y[1] = __uint_as_float(y01[1]);
    f.y[2] = __uint_as_float(y23[0]), f.y[3] = __uint_as_float(y23[1]);
    return f;
}
template <bool SLOW, bool HFORM>
__device__ __forceinline__ void fwd_row_pair(const int J, const float *e0, const float *e1, const char *ct_row, const uint32_t ct_swz,
                                             const char *wlane, const int half, const int lane, f32x16 &acc0, f32x16 &acc1) {
    const int n = J / 16;
    auto kstep = [&](const int ks, const float a0, const float a1) __attribute__((always_inline)) {
        const jf2 a01 = {a0, a1};
        const uint32_t c0 = (uint32_t)(4 * ks + 2 * half);
        const float4 c4a = *(const float4 *)(ct_row + (size_t)(c0 ^ ct_swz) * 16);
        const float4 c4b = *(const float4 *)(ct_row + (size_t)((c0 + 1u) ^ ct_swz) * 16);
        const jh8 wh = *(const jh8 *)(wlane + (size_t)(ks * 2 + 0) * 1024);
        const jh8 wl = *(const jh8 *)(wlane + (size_t)(ks * 2 + 1) * 1024);
        FwdFrags f;
        float x0[2], x1[2];
        fwd_h_pair<0, SLOW, HFORM>(a01, c4a.x, x0[0], x1[0]);
        fwd_h_pair<1, SLOW, HFORM>(a01, c4a.y, x0[1], x1[1]);
        fwd_split_quad<0>(x0, x1, f);
        fwd_h_pair<2, SLOW, HFORM>(a01, c4a.z, x0[0], x1[0]);
        fwd_h_pair<3, SLOW, HFORM>(a01, c4a.w, x0[1], x1[1]);
        fwd_split_quad<2>(x0, x1, f);
        fwd_h_pair<4, SLOW, HFORM>(a01, c4b.x, x0[0], x1[0]);
        fwd_h_pair<5, SLOW, HFORM>(a01, c4b.y, x0[1], x1[1]);
        fwd_split_quad<4>(x0, x1, f);
        fwd_h_pair<6, SLOW, HFORM>(a01, c4b.z, x0[0], x1[0]);
        fwd_h_pair<7, SLOW, HFORM>(a01, c4b.w, x0[1], x1[1]);
        fwd_split_quad<6>(x0, x1, f);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, f.hi0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, f.hi1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, f.lo0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, f.lo1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, f.hi0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, f.hi1, acc1, 0, 0, 0);
    };
    // this lane's unit within the group of 64 (the ea_off rotation of the one-k-step form lives in fwd_spread now)
    const int ngroups = (n + 3) >> 2;
    float xa = e0[min(lane, J - 1)], xb = e1[min(lane, J - 1)];
    for (int q = 0; q < ngroups; ++q) {
        const FwdAddends ya = fwd_spread(xa), yb = fwd_spread(xb);
        if (q + 1 < ngroups) {  // next group's addends: in flight during the four k-steps below
            const int idx = min(64 * (q + 1) + lane, J - 1);
            xa = e0[idx], xb = e1[idx];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < n) kstep(4 * q + r, ya.y[r], yb.y[r]);
    }
}

__device__ __forceinline__ float half_swap_max(const float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(const float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// lsm outputs + parked logits of one lattice row held in the transposed C/D layout (lane = cell u0 + l31, register r = symbol
// cd_row(r, half)).  `acc` holds (s2 W2)^T r with r = (1 - h) / 2 = 1 / (1 + e^{2(a + c)}): the main loop saves the multiply-add that
// turns the reciprocal into tanh, and logits = (b2 + sum_j W2[j]) - (2 / s2) acc with the first term tabulated by joint_prep_kernel
// as an f32 hi + lo pair (-1e30 for the pad symbols).  (With weights beyond kRFormLimit the kernel accumulates h itself and the
// table holds b2 alone: the r form would form the logits as the difference of two sums of the magnitude of the largest weight.)  Rounding: the split products bound the error of acc by ~2^-22 sum_j |W2[j][v]| r_j
// -- at most twice the bound of the h form, and of the order of an f32 matrix product's.  `rsel_b` / `rsel_l`: the register
// that holds this lane's blank / label logit, or -1 when it lives in the other half (or there is no label edge).
// Round 5: the lattice's edges leave as PROBABILITIES {p(blank), p(label)} = e_i / sum for the linear-domain sweeps (rnnt_lin.h:
// zero where an edge leaves the lattice, NaN -- the utterance is handed back -- where an edge the lattice owns is below 2^-100);
// lse is still stored (the backward's per-cell set-up needs the denominator and does not re-read the tile).
// Returns the cell's decay statistic -log2 max(p_blank, p_label) (rnnt_lin.h; 0 for lanes that own no cell, half 1 included).
// MODE 0: the whole vocabulary is this tile (V <= 32).  Several vocabulary tiles (32 < V <= 128, round 5): MODE 1 = every tile but
// the last, launched first: park the tile's 32 logits and nothing else; MODE 2 = the last tile: re-reads what the others parked
// for the cell (16 floats per half-lane and tile, the same symbols-per-register layout; once for the maximum and the blank /
// label picks, once more -- out of L2 -- for the sum) and runs the softmax over all of them.  rsel_*: registers of THIS tile.
template <int MODE>
__device__ __forceinline__ float fwd_row_epilogue(const JointParams &jp, const f32x16 &acc, const float m2inv,
                                                  const int rsel_b, const int rsel_l, const int lab,
                                                  const int b, const int t, const int u, const int Tb, const int Ub, const int half) {
    const LossParams &p = jp.lp;
    const float *b2t = jp.b2s + 64 * jp.vt;
    const int Vp = 32 * jp.VT;
    float x[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 are the four consecutive symbols 8g + 4 half + 0..3 of the tile
        const float4 bh = *(const float4 *)(b2t + 8 * g + 4 * half), bl = *(const float4 *)(b2t + 32 + 8 * g + 4 * half);
        x[4 * g + 0] = fmaf(acc[4 * g + 0], m2inv, bh.x) + bl.x;
        x[4 * g + 1] = fmaf(acc[4 * g + 1], m2inv, bh.y) + bl.y;
        x[4 * g + 2] = fmaf(acc[4 * g + 2], m2inv, bh.z) + bl.z;
        x[4 * g + 3] = fmaf(acc[4 * g + 3], m2inv, bh.w) + bl.w;
    }
    const size_t c = ((size_t)(b * p.T + t)) * p.U + min(u, p.U - 1);  // (clamped: MODE 2's loads are unconditional)
    if (MODE == 1) {
        if (u < Ub) {
            float *dst = jp.dl + c * Vp + 32 * jp.vt + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4 *)(dst + 8 * g) = make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
        }
        return 0.f;
    }
    float m = x[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, x[r]);
    float xb = -INFINITY, xl = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xb = (rsel_b == r) ? x[r] : xb;
        xl = (rsel_l == r) ? x[r] : xl;
    }
    auto reg_in = [&](const int v, const int tile) {  // register of symbol v in tile `tile` for this half-lane, or -1
        const int w = v - 32 * tile;
        return (w >= 0 && w < 32 && ((w >> 2) & 1) == half) ? (w & 3) + 4 * (w >> 3) : -1;
    };
    if (MODE == 2) {
        for (int ot = 0; ot < jp.VT - 1; ++ot) {
            const float *src = jp.dl + c * Vp + 32 * ot + 4 * half;
            const int rb = reg_in(p.blank, ot), rl = reg_in(lab, ot);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = *(const float4 *)(src + 8 * g);
                const float xo[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    m = fmaxf(m, xo[k]);
                    xb = (rb == 4 * g + k) ? xo[k] : xb;
                    xl = (rl == 4 * g + k) ? xo[k] : xl;
                }
            }
        }
    }
    m = half_swap_max(m);
    const float nml = -m * kLog2e;
    float ssum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum += jex2(fmaf(x[r], kLog2e, nml));
    if (MODE == 2) {
        for (int ot = 0; ot < jp.VT - 1; ++ot) {
            const float *src = jp.dl + c * Vp + 32 * ot + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = *(const float4 *)(src + 8 * g);
                ssum += (jex2(fmaf(q.x, kLog2e, nml)) + jex2(fmaf(q.y, kLog2e, nml))) + (jex2(fmaf(q.z, kLog2e, nml)) + jex2(fmaf(q.w, kLog2e, nml)));
            }
        }
    }
    ssum = half_swap_sum(ssum);
    xb = half_swap_max(xb);
    xl = half_swap_max(xl);
    float stat = 0.f;
    if (u < Ub) {
        if (half == 0) {
            const float inv = __builtin_amdgcn_rcpf(ssum);
            const bool blank_stays = (t < Tb - 1) || (u == Ub - 1);
            float pb = 0.f, pl = 0.f;
            if (blank_stays) {
                pb = jex2(fmaf(xb, kLog2e, nml)) * inv;
                if (!(pb >= kTinyEdge)) pb = NAN;  // (also a NaN logit): not representable on the linear lattice
            }
            if (u < Ub - 1) {
                pl = jex2(fmaf(xl, kLog2e, nml)) * inv;
                if (!(pl >= kTinyEdge)) pl = NAN;
            }
            p.lse[c] = m + kLn2 * jlg2(ssum);
            const size_t wi = ((size_t)b * p.Nr + (t + u)) * p.Up + u;
            ((float2 *)p.W)[wi] = make_float2(pb, pl);
            jp.xbl[c] = make_float2(xb, xl);  // what the backward's per-cell records need of the logits tile (8 of its 128 bytes)
            stat = (pb != pb || pl != pl) ? 200.f : -jlg2(fmaxf(fmaxf(pb, pl), 1.0e-37f));
        }
        // park the logits (bias included): registers 4g .. 4g+3 are the four consecutive symbols 8g + 4 half + 0..3
        float *dst = jp.dl + c * Vp + 32 * jp.vt + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4 *)(dst + 8 * g) = make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
    }
    return stat;
}

// {sum of the decay statistic, cells} of one wave's row pair -> its slot of the utterance's statistic table (what the linear
// sweeps choose their frame-block length from: rnnt_lin.h; slot = u-tile * ceil(T / 2) + row pair, LossParams::pstatStride 1).
// The sum over the 32 cell lanes runs on DPP row operations + two v_readlane (six __shfl_xor steps -- ds_bpermute round trips
// in a dependent chain -- cost the kernel 5 %); the cell count is known without looking at the lanes.
__device__ __forceinline__ void fwd_put_stat(const JointParams &jp, const int b, const int ut, const int t0, float stat, const float cnt,
                                             const int lane) {
    const LossParams &p = jp.lp;
    stat += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(stat), 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true));
    stat += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(stat), 0x4E /*quad_perm:[2,3,0,1]*/, 0xf, 0xf, true));
    stat += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(stat), 0x141 /*row_half_mirror*/, 0xf, 0xf, true));
    stat += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(stat), 0x140 /*row_mirror*/, 0xf, 0xf, true));
    // every lane of a 16-lane row now holds its row's sum; the cells sit in lanes 0..31 (half 1 contributes zeros)
    const float sum = __builtin_amdgcn_readlane(stat, 0) + __builtin_amdgcn_readlane(stat, 16);
    if (lane == 0 && t0 < p.T)
        p.pstat[(size_t)b * p.nPstat + (size_t)ut * ((p.T + 1) >> 1) + (t0 >> 1)] = make_float2(sum, cnt);
}

__global__ __launch_bounds__(kFwdWaves * 64) void joint_fwd_kernel(const JointParams jp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LossParams &p = jp.lp;
    const int J = jp.J, V = p.V;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool slow = jp.tflag[0] != 0.f;  // kernel-uniform
    // the W2 image holds s2 W2.  r form: logits = (b2 + sum_j W2) - (2 / s2) acc;  h form (huge weights): logits = b2 + acc / s2
    const bool hform = jp.tflag[1] != 0.f;  // kernel-uniform
    const float m2inv = hform ? jp.tflag[2] : -2.0f * jp.tflag[2];
    const float *Etab = slow ? jp.enc_proj : jp.expE, *Ptab = slow ? jp.pred_proj : jp.expP;
    char *Ct = (char *)lds;                 // [32 u][J] floats, 16-byte chunk c of row u stored at chunk c ^ (u & 15)
    char *Wimg = Ct + (size_t)J * 128;      // [J/16][hi, lo][64 lanes][16 B]
    const int cpr = J / 4;                  // 16-byte chunks per Ct row

    const char *w2img = (const char *)(jp.W2s + (size_t)jp.vt * J * 64);  // this launch's vocabulary tile
    for (int pc = wave; pc < J / 8; pc += kFwdWaves)  // W2 fragment image: J/8 pieces of 1 KB, once per workgroup
        lds_dma16(w2img + (size_t)pc * 1024 + lane * 16, Wimg + pc * 1024);
    // register that holds symbol v of a 32-symbol tile in this lane's half (or -1: the other half's, or not in this tile)
    auto reg_of = [&](const int v, const int tile) {
        const int w = v - 32 * tile;
        return (w >= 0 && w < 32 && ((w >> 2) & 1) == half) ? (w & 3) + 4 * (w >> 3) : -1;
    };
    const int mode = jp.VT == 1 ? 0 : (jp.vt < jp.VT - 1 ? 1 : 2);  // fwd_row_epilogue's MODE (kernel-uniform)
    const int rsel_b = reg_of(p.blank, jp.vt);
    const uint32_t ct_lane = (uint32_t)(l31 * cpr);          // this lane's Ct row, in chunks
    const uint32_t ct_swz = (uint32_t)(l31 & 15);
    // enc-side addends of a k-step (16 joint units), one per lane: positions 0..7 of every 16-lane row hold the 8 units of
    // the lane's half (units 8 half + 0..7), so row_newbcast:e hands every lane "its" unit e (fwd_spread builds those rows)

    // items = (utterance, u-tile, row tile) with the row tile fastest; every workgroup takes a CONTIGUOUS range of them, so
    // that the 128 J-byte Ct tile is reloaded only when (utterance, u-tile) changes
    const int n_tr = (p.T + kFwdRows - 1) / kFwdRows;
    const int n_items = n_tr * p.B * jp.n_ut;
    const int it_lo = (int)((long long)n_items * blockIdx.x / gridDim.x);
    const int it_hi = (int)((long long)n_items * (blockIdx.x + 1) / gridDim.x);
    int ct_owner = -1;
    int rsel_l = -1, lab_l = -1;  // this lane's label: its register in this tile (or -1), the symbol itself (or -1)
    for (int item = it_lo; item < it_hi; ++item) {
        const int bu = item / n_tr, tr = item - bu * n_tr;
        const int b = bu / jp.n_ut, ut = bu - b * jp.n_ut;
        const int u0 = ut * 32;
        const int Tb = length_T(p, b), Ub = length_U(p, b);
        const int t_begin = tr * kFwdRows, t_end = min(min(t_begin + kFwdRows, p.T), Tb);
        const int t0 = t_begin + 2 * wave;
        if (t_begin >= t_end || u0 >= Ub) {  // workgroup-uniform: no lattice cell here -- the statistic slots still get their zeros
            if (mode != 1) fwd_put_stat(jp, b, ut, t0, 0.f, 0.f, lane);
            continue;
        }
        const int u = u0 + l31;
        if (ct_owner != bu) {
            ct_owner = bu;
            __syncthreads();  // every wave is done with the previous Ct tile
            for (int pc = wave; pc < J / 8; pc += kFwdWaves) {
                const uint32_t q = (uint32_t)pc * 64u + (uint32_t)lane;  // LDS chunk this lane fills
                const uint32_t ur = q / (uint32_t)cpr, cpos = q - ur * (uint32_t)cpr;
                const uint32_t c = cpos ^ (ur & 15u);                    // ... with this logical chunk of row ur
                const float *src = Ptab + ((size_t)b * p.U + min(u0 + (int)ur, p.U - 1)) * J + c * 4u;
                lds_dma16(src, Ct + (size_t)pc * 1024);
            }
            rsel_l = lab_l = -1;
            if (u < Ub - 1) {
                lab_l = min(max(p.labels[(size_t)b * (p.U - 1) + u], 0), V - 1);
                rsel_l = reg_of(lab_l, jp.vt);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (t0 >= t_end) {  // wave-uniform (no barrier below)
            if (mode != 1) fwd_put_stat(jp, b, ut, t0, 0.f, 0.f, lane);
            continue;
        }
        const bool two = t0 + 1 < t_end;  // wave-uniform
        const float *e0 = Etab + ((size_t)b * p.T + t0) * J;
        const float *e1 = two ? e0 + J : e0;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
        const char *ct_row = Ct + (size_t)ct_lane * 16, *wlane = Wimg + lane * 16;
        if (hform) {  // (kernel-uniform branches: four straight-line instantiations of the product loop)
            if (!slow)
                fwd_row_pair<false, true>(J, e0, e1, ct_row, ct_swz, wlane, half, lane, acc0, acc1);
            else
                fwd_row_pair<true, true>(J, e0, e1, ct_row, ct_swz, wlane, half, lane, acc0, acc1);
        } else if (!slow) {
            fwd_row_pair<false, false>(J, e0, e1, ct_row, ct_swz, wlane, half, lane, acc0, acc1);
        } else {
            fwd_row_pair<true, false>(J, e0, e1, ct_row, ct_swz, wlane, half, lane, acc0, acc1);
        }
        if (mode == 0) {
            float stat = fwd_row_epilogue<0>(jp, acc0, m2inv, rsel_b, rsel_l, lab_l, b, t0, u, Tb, Ub, half);
            if (two) stat += fwd_row_epilogue<0>(jp, acc1, m2inv, rsel_b, rsel_l, lab_l, b, t0 + 1, u, Tb, Ub, half);
            fwd_put_stat(jp, b, ut, t0, stat, (float)(min(32, Ub - u0) * (two ? 2 : 1)), lane);
        } else if (mode == 1) {
            fwd_row_epilogue<1>(jp, acc0, m2inv, -1, -1, -1, b, t0, u, Tb, Ub, half);
            if (two) fwd_row_epilogue<1>(jp, acc1, m2inv, -1, -1, -1, b, t0 + 1, u, Tb, Ub, half);
        } else {
            float stat = fwd_row_epilogue<2>(jp, acc0, m2inv, rsel_b, rsel_l, lab_l, b, t0, u, Tb, Ub, half);
            if (two) stat += fwd_row_epilogue<2>(jp, acc1, m2inv, rsel_b, rsel_l, lab_l, b, t0 + 1, u, Tb, Ub, half);
            fwd_put_stat(jp, b, ut, t0, stat, (float)(min(32, Ub - u0) * (two ? 2 : 1)), lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward of the WIDE joint (640 < J <= 704), step 1: dlogits from the parked logits tile (one lattice cell per lane, 128 B
// in / 128 B out) plus this workgroup's share of db2 = sum_cells dl.  Replaces a second run of phase 1.
// ---------------------------------------------------------------------------------------------
constexpr int kDlChunks = 8;  // 256-cell chunks per workgroup (fewer, fatter db2 partials)

__global__ __launch_bounds__(256) void joint_dl_kernel(const JointParams jp) {
    __shared__ float red[256][33];
    const LossParams &p = jp.lp;
    const int V = p.V, tid = threadIdx.x;
    const unsigned block_id = blockIdx.x;
    float colsum[32];
#pragma unroll
    for (int v = 0; v < 32; ++v) colsum[v] = 0.f;
    const size_t total = (size_t)p.cells * 32;
    for (int ch = 0; ch < kDlChunks; ++ch) {
        const uint32_t c0 = (block_id * kDlChunks + ch) * 256u;
        if (c0 >= p.cells) break;  // workgroup-uniform
        const uint32_t c = c0 + tid;
        // the chunk's 256 tiles are 32 KB of contiguous memory: move them with fully coalesced 16-byte accesses and hand
        // every lane its own cell through LDS (a lane reading its 128 bytes directly uses a quarter of every request)
        const float4 *g4 = (const float4 *)(jp.dl + (size_t)c0 * 32);
        float4 *o4 = (float4 *)(jp.dlg + (size_t)c0 * 32);  // NOT in place: the parked logits serve a repeated backward call
        __syncthreads();  // the previous chunk's write-back is done with `red`
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = i * 256 + tid;  // float4 index inside the chunk: cell e >> 3, columns 4 (e & 7) ..
            if ((size_t)c0 * 32 + (size_t)e * 4 < total) {
                const float4 t4 = g4[e];
                float *r = &red[e >> 3][4 * (e & 7)];
                r[0] = t4.x, r[1] = t4.y, r[2] = t4.z, r[3] = t4.w;
            }
        }
        __syncthreads();
        const Cell cl = decode(p, c);
        float x[32];
#pragma unroll
        for (int v = 0; v < 32; ++v) x[v] = red[tid][v];
        if (cl.valid) {
            const CellGrad g = cell_grad_setup(p, cl, c);
            float xb = 0.f, xl = 0.f;
#pragma unroll
            for (int v = 0; v < 32; ++v) {
                xb = (v == p.blank) ? x[v] : xb;
                xl = (g.has_label && v == g.lab) ? x[v] : xl;
            }
            const float cb = g.has_blank_corr ? g.scale * jex2(fmaf(xb, kLog2e, g.nl) + g.cb) : 0.f;
            const float clb = g.has_label ? g.scale * jex2(fmaf(xl, kLog2e, g.nl) + g.cl) : 0.f;
#pragma unroll
            for (int v = 0; v < 32; ++v) {
                float gv = (v < V) ? g.scale * jex2(fmaf(x[v], kLog2e, g.c0)) : 0.f;
                gv -= (v == p.blank) ? cb : 0.f;
                gv -= (g.has_label && v == g.lab) ? clb : 0.f;
                x[v] = gv;
                colsum[v] += gv;
            }
        } else {
            // padded cell: phase 2 reads whole 32-column tiles of every valid row, so it must find exact zeros here
#pragma unroll
            for (int v = 0; v < 32; ++v) x[v] = 0.f;
        }
#pragma unroll
        for (int v = 0; v < 32; ++v) red[tid][v] = x[v];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = i * 256 + tid;
            if ((size_t)c0 * 32 + (size_t)e * 4 < total) {
                const float *r = &red[e >> 3][4 * (e & 7)];
                o4[e] = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
    }
    // db2 partial of this workgroup: fixed-order column sums through LDS
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 32; ++v) red[tid][v] = colsum[v];
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int r = 0; r < 256; ++r) s += red[r][tid];
        jp.dbpart[(size_t)block_id * 32 + tid] = s;
    }
}
// ---------------------------------------------------------------------------------------------
// phase 2 of the WIDE joint (640 < J <= 704; joint_bwd_kernel below is the default): block = (utterance, u-tile, 64-wide J slab,
// row split), one lattice row per wave per iteration; both products on v_mfma_f32_32x32x16_f16 with every operand as binary16
// hi + lo parts (three MFMAs per 16-wide k-step, see joint_phase1s_kernel):
//   dh[u][j]   = sum_v dl[u][v] W2[j][v]   A = dl rows (k = v, 8 consecutive per lane), B = W2^T fragments kept in registers
//   dW2[j][v] += sum_u h[u][j] dl[u][v]    A = h straight from the C/D layout of the h tile: the k-slots of a k-step are
//                                          assigned to the lattice columns cd_row(8 ks + e, half), and dl is gathered in the
//                                          same order for B, so the registers that hold h ARE the A fragments
// dl is multiplied by a power of two S_b (|S_b dl| <= 2^14, from this utterance's cost_scale) before it is split, so small
// upstream gradients do not fall into binary16's subnormals; dh and the dW2 partial are divided by S_b again (exact).
// LDS: Cs [64 j][36] | dlr [4][32 u][36] | red [4][32][33]
// ---------------------------------------------------------------------------------------------
#ifndef P2S_WG_PER_CU
#define P2S_WG_PER_CU 2  // 227 VGPRs, no spills: 3.18 ms per fused step at C2; 3 (168 VGPRs, 57 spilled) 5.37 ms; 1 (284) 4.42 ms
#endif
__global__ __launch_bounds__(256, P2S_WG_PER_CU) void joint_phase2s_kernel(const JointParams jp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LossParams &p = jp.lp;
    const int J = jp.J, V = p.V;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kCs = 36;                     // row stride of Cs and of the dl tiles: 16-byte aligned, conflict-free b128 reads
    float *Cs = lds;                            // [64][kCs]
    float *dlr = Cs + 64 * kCs;                 // [4][32][kCs]
    float *red = dlr + 4 * 32 * kCs;            // [4][32][kStagePad]
    jh8 *wfrag = (jh8 *)(red + 4 * 32 * kStagePad);  // [jt 2][ks 2][hi, lo][64 lanes]: B fragments of dh (8 KB)
    float *my_dl = dlr + wave * 32 * kCs;
    const bool slow = jp.tflag[0] != 0.f;  // kernel-uniform
    const float w2inv = jp.tflag[2], s2 = 1.0f / w2inv;  // W2 enters the products as s2 W2 (powers of two: exact)
    const float *Etab = slow ? jp.enc_proj : jp.expE, *Ptab = slow ? jp.pred_proj : jp.expP;

    int bid = blockIdx.x;
    const int ts = bid % jp.n_ts;
    bid /= jp.n_ts;
    const int js = bid % (J / 64);
    bid /= (J / 64);
    const int ut = bid % jp.n_ut;
    const int b = bid / jp.n_ut;
    const int u0 = ut * 32, j0 = js * 64;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int t_begin = ts * jp.TS, t_end = min(min(t_begin + jp.TS, p.T), Tb);
    const bool tile_live = (t_begin < t_end) && (u0 < Ub);

    // |dl| <= 2 |cost_scale[b]| (joint_dl_kernel): S = 2^(13 - e) with |cost_scale[b]| < 2^e
    float S = 1.0f, invS = 1.0f;
    {
        const float cs = p.cost_scale ? fabsf(p.cost_scale[b]) : 1.0f;
        if (cs > 0.f && cs < 3.0e38f) {
            const int e = ilogbf(cs) + 1;
            S = ldexpf(1.0f, 13 - e), invS = ldexpf(1.0f, e - 13);
        }
    }

    f32x16 accC[2], accW[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) accC[q][r] = 0.f, accW[q][r] = 0.f;

    if (tile_live) {
        for (int idx = tid; idx < 32 * 16; idx += 256) {  // C slab, transposed: Cs[j][u]
            const int u = idx & 31, j4 = idx >> 5;
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u0 + u < p.U) c4 = *(const float4 *)(Ptab + ((size_t)b * p.U + u0 + u) * J + j0 + j4 * 4);
            Cs[(j4 * 4 + 0) * kCs + u] = c4.x;
            Cs[(j4 * 4 + 1) * kCs + u] = c4.y;
            Cs[(j4 * 4 + 2) * kCs + u] = c4.z;
            Cs[(j4 * 4 + 3) * kCs + u] = c4.w;
        }
    }
    // B fragments of dh = dl . W2^T (row-independent): lane (j = jt*32 + l31, half), k-step ks: W2[j0 + j][16 ks + 8 half + 0..7];
    // built once per workgroup (wave w builds (jt, ks) = (w >> 1, w & 1)) and read back from LDS where they are used
    {
        const int jt = wave >> 1, ks = wave & 1;
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int v = 16 * ks + 8 * half + e;
            w[e] = (v < V) ? jp.W2[(size_t)(j0 + jt * 32 + l31) * V + v] * s2 : 0.f;
        }
        jh8 hi, lo;
        split_h8(w, hi, lo);
        wfrag[((jt * 2 + ks) * 2 + 0) * 64 + lane] = hi;
        wfrag[((jt * 2 + ks) * 2 + 1) * 64 + lane] = lo;
    }
    __syncthreads();

    const int n_iter = tile_live ? (t_end - t_begin + 3) / 4 : 0;
    auto dl_fetch = [&](const int t, float (&d)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = lane + q * 64, uu = e >> 5, v = e & 31;
            d[q] = (t < t_end && u0 + uu < p.U) ? jp.dlg[(((size_t)(b * p.T + t)) * p.U + u0 + uu) * 32 + v] : 0.f;
        }
    };
    float dnext[16];
    if (n_iter > 0) dl_fetch(t_begin + wave, dnext);
    for (int it = 0; it < n_iter; ++it) {
        const int t = t_begin + it * 4 + wave;
        if (t < t_end) {  // wave-uniform; no workgroup barrier inside: the dl row buffer is wave-private
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = lane + q * 64;
                my_dl[(e >> 5) * kCs + (e & 31)] = dnext[q] * S;
            }
            dl_fetch(t + 4, dnext);  // next row of this wave: latency hides under this row's work
            // fragments of the (scaled) dl tile: A of dh (row u = l31, k = v) and B of dW2 (k = lattice column, n = v = l31)
            jh8 dahi[2], dalo[2], dbhi[2], dblo[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float4 x0 = *(const float4 *)(my_dl + l31 * kCs + 16 * ks + 8 * half);
                const float4 x1 = *(const float4 *)(my_dl + l31 * kCs + 16 * ks + 8 * half + 4);
                const float xa[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                split_h8(xa, dahi[ks], dalo[ks]);
                float xb[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[e] = my_dl[cd_row(8 * ks + e, half) * kCs + l31];
                split_h8(xb, dbhi[ks], dblo[ks]);
            }
            const float *arow = Etab + ((size_t)b * p.T + t) * J + j0;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                // h tile in C/D layout: rows = lattice columns u, column = joint unit j = jt*32 + l31
                const float aj = arow[jt * 32 + l31];
                float h[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 c4 = *(const float4 *)(Cs + (jt * 32 + l31) * kCs + 8 * g + 4 * half);
                    if (!slow) {
                        h[4 * g + 0] = tanh_from_exp(aj, c4.x);
                        h[4 * g + 1] = tanh_from_exp(aj, c4.y);
                        h[4 * g + 2] = tanh_from_exp(aj, c4.z);
                        h[4 * g + 3] = tanh_from_exp(aj, c4.w);
                    } else {
                        h[4 * g + 0] = fast_tanh(aj + c4.x);
                        h[4 * g + 1] = fast_tanh(aj + c4.y);
                        h[4 * g + 2] = fast_tanh(aj + c4.z);
                        h[4 * g + 3] = fast_tanh(aj + c4.w);
                    }
                }
                // S dh[u][j] = sum_v (S dl[u][v]) W2[j][v]
                f32x16 dh;
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    dh = mfma3(dahi[ks], dalo[ks], wfrag[((jt * 2 + ks) * 2 + 0) * 64 + lane],
                               wfrag[((jt * 2 + ks) * 2 + 1) * 64 + lane], dh);
                // dz = dh * (1 - h^2);  sum over u -> d enc_proj partial;  running sum over t -> d pred_proj
                float colsum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dz = dh[r] * (invS * w2inv) * (1.0f - h[r] * h[r]);
                    accC[jt][r] += dz;
                    colsum += dz;
                }
                colsum += __shfl_xor(colsum, 32);
                if (lane < 32)
                    jp.dApart[(((size_t)ut * p.B + b) * p.T + t) * J + j0 + jt * 32 + lane] = colsum;
                // S dW2[j][v] += sum_u h[u][j] (S dl[u][v]): k-slot (ks, half, e) <-> lattice column cd_row(8 ks + e, half)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float hk[8] = {h[8 * ks], h[8 * ks + 1], h[8 * ks + 2], h[8 * ks + 3],
                                         h[8 * ks + 4], h[8 * ks + 5], h[8 * ks + 6], h[8 * ks + 7]};
                    jh8 hhi, hlo;
                    split_h8(hk, hhi, hlo);
                    accW[jt] = mfma3(hhi, hlo, dbhi[ks], dblo[ks], accW[jt]);
                }
            }
        }
    }
    // ---- deterministic cross-wave reductions, then the partial buffers
    if (!tile_live) return;
    const int wid = (b * jp.n_ut + ut) * jp.n_ts + ts;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + cd_row(r, half)) * kStagePad + l31] = accC[jt][r];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            const int uu = e >> 5, j = e & 31;
            float sum = 0.f;
            for (int w = 0; w < 4; ++w) sum += red[(w * 32 + uu) * kStagePad + j];
            if (u0 + uu < p.U) jp.dCpart[(((size_t)ts * p.B + b) * p.U + u0 + uu) * J + j0 + jt * 32 + j] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + cd_row(r, half)) * kStagePad + l31] = accW[jt][r] * invS;
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            const int jj = e >> 5, v = e & 31;
            float sum = 0.f;
            for (int w = 0; w < 4; ++w) sum += red[(w * 32 + jj) * kStagePad + v];
            jp.dWpart[((size_t)wid * J + j0 + jt * 32 + jj) * 32 + v] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, single-kernel form (default for J <= 640): dlogits + the whole scatter through the joint in ONE persistent
// launch, replacing joint_dl_kernel + joint_phase2s_kernel.  A workgroup = J/64 CONSUMER waves (one 64-unit J slab each,
// all working on the same lattice row) + 2 PRODUCER waves.
//   producers: per lattice row (32 cells of one u-tile) turn the parked logits into S.dlogits and publish them ONCE, already
//              split into binary16 hi/lo parts and laid out as the MFMA fragments both products need (A of dh: lane = cell,
//              k = symbol; B of dW2: lane = symbol, k-slots = the lattice columns of the C/D layout), 8 KB per row, into a
//              ring of kBwdRing slots in LDS; they also accumulate db2.  (Before: every J slab re-read the f32 tile from L2,
//              re-scaled, re-staged and re-split it: ~30 % of phase 2's VALU work, and a separate 0.25 ms kernel.)
//   consumers: h tile in C/D layout from register-resident e^{2 pred_proj} factors (the 16 lattice columns x 2 j-tiles a
//              lane needs never change within an (utterance, u-tile)), dh = dl.W2^T, dz = dh (1 - h^2), sum_u -> d enc_proj
//              partial, running sum_t -> d pred_proj, dW2 += h^T.dl with the h registers as A fragments -- as in
//              joint_phase2s_kernel; dW2 accumulators persist over ALL items of the workgroup (re-scaled by an exact power
//              of two when the utterance, hence S, changes).
//   hand-off:  per ring slot a sequence word (row + 1 once published) and a use counter (consumers that have read it);
//              LDS operations of a wave complete in order, so "data, wait, flag" suffices; all polls are bounded and a
//              timeout poisons the outputs with NaN instead of hanging.
// Items = (utterance, u-tile, row tile of kBwdRows rows), row tile fastest; every workgroup takes a contiguous range.
// ---------------------------------------------------------------------------------------------
constexpr int kBwdRing = 8;
constexpr int kBwdSlotBytes = 8192 + 2048;  // fragment image + the row's enc-side addends for the group's joint units
constexpr int kBwdRows = 32;
// Lattice rows the backward does not visit.  Every dlogits value of a cell is bounded by 2 |cost_scale| x the cell's occupancy
// alpha beta / L (the softmax term is occupancy x probability, the blank / label corrections are occupancy x an edge probability x
// beta' / beta <= occupancy).  A (row, u-tile) none of whose 32 cells has an occupancy above 2^-kOccFloor contributes less than
// 2^-(kOccFloor - 6) |cost_scale| per logit row to anything -- below the resolution of an f32 sum that holds one O(1) term -- and is
// skipped: its cells get exactly zero where the reference leaves 1e-15's.  On a 600 x 150 lattice that is 40 ... 52 % of the rows for
// unstructured logits (the mass sits in a band around the diagonal; hypergeometric tails), more for a trained model.  NaN counts as
// occupied.  An utterance the hand-back redoes is visited whole.
// Round 6, late: the floor is 2^-40 -- and below it nothing is lost at all.  The products take dlogits as binary16 hi + lo parts of
// S x dlogits with S |cost_scale| < 2^13 (bwd_scale), i.e. |S dl| <= 2^14 x occupancy: below an occupancy of 2^-40 that is under 2^-26, hi and
// lo both round to zero (the smallest binary16 subnormal is 2^-24) -- the rows between 2^-50 and 2^-40 were visited to multiply and add exact zeros.
#ifndef RNNT_OCC_FLOOR
#define RNNT_OCC_FLOOR 40  // (dev builds: 100000 = visit every row, for same-box timing of the pruning)
#endif
constexpr int kOccFloor = RNNT_OCC_FLOOR;
__device__ __forceinline__ uint32_t bwd_live_mask(const JointParams &jp, const int b, const int ut, const int tr, const int t_begin, const int t_end) {
    const int n_tr32 = (jp.lp.T + 31) >> 5;
    uint32_t m = ((const uint32_t *)jp.live8)[((size_t)b * jp.n_ut + ut) * n_tr32 + tr];
    const int n = t_end - t_begin;  // rows of this tile inside the utterance (1 .. 32); the tile starts at a multiple of 32
    return n >= 32 ? m : (m & ((1u << n) - 1u));
}
// Work distribution with the row pruning (round 5): the rows the pruning leaves form a band whose shape and width depend on the data
// (bench.py's input: an utterance keeps 36 ... 84 % of its rows, a (utterance, u-tile) column of it anything from 0 to 100 %), so equal
// numbers of ITEMS per workgroup meant 150 ... 690 rows per workgroup and the slowest one set the kernel's time; so did every static
// interleave of the items that was tried (by row-tile class, by utterance).  joint_rowplan_kernel therefore cuts the item list
// (utterance, u-tile, row tile -- columns stay contiguous) into ranges of equal WEIGHT, weight = rows visited + kBwdItemCost per
// item inside its utterance: prefix[i] = weight of the items before i, item i belongs to workgroup min(prefix[i] / target, nblk - 1).
// The target is at least a (kBwdSlots - 2)-th of a full column, so that a column spans fewer than kBwdSlots workgroups (its d pred_proj
// partial slabs); with little to do, fewer workgroups than the grid holds get any.  Deterministic: the cuts depend on the data, not on timing.
constexpr int kBwdSlots = 8;     // d pred_proj partial slabs per (utterance, u-tile)
constexpr int kBwdItemCost = 3;  // rows' worth of time an item costs before its first row (decode, run change)
constexpr int kBwdMaxBlocks = 256;  // workgroups per group at most (sizes the dW2 / db2 partial buffers)

// workgroup (of a J group) that owns an item under joint_rowplan_kernel's cut
__device__ __forceinline__ int bwd_blk_of(const int *plan, const int item, const int nblk) { return min(plan[2 + kBwdMaxBlocks + item] / plan[0], nblk - 1); }
struct BwdItem {
    int b, ut, u0, Tb, Ub, t_begin, t_end;
    int col_first;     // first item of this item's column (utterance, u-tile)
    uint32_t mask;     // bit r: row t_begin + r is visited (bwd_live_mask)
    bool live;
};
// What an item needs from memory -- the utterance's two lengths and the row bits of its tile: three dependent round trips when
// fetched item by item, and with the pruning a workgroup walks over items it has nothing to do in (first version: ~3 us per dead
// item, a 14 us hole in the ring at every run change).  joint_bwd_kernel fetches them for its whole range at once, a thread per
// item, into this table (static LDS, in front of the kernel's dynamic ring); ranges beyond kBwdItemCache items fall back to memory.
constexpr int kBwdItemCache = 2048;
struct BwdItemMem {
    uint32_t mask;
    int Tb, Ub;
};
__shared__ BwdItemMem bwd_item_cache[kBwdItemCache];
__shared__ int bwd_item_cache_lo, bwd_item_cache_n;

template <bool CACHED = true>
__device__ __forceinline__ BwdItem bwd_item(const JointParams &jp, const int item, const int n_tr) {
    const LossParams &p = jp.lp;
    BwdItem it;
    const int col = item / n_tr, tr = item - col * n_tr;
    it.b = col / jp.n_ut, it.ut = col - it.b * jp.n_ut;
    it.col_first = col * n_tr;
    it.u0 = it.ut * 32;
    it.t_begin = tr * kBwdRows;
    const int ci = item - bwd_item_cache_lo;
    const bool hit = CACHED && (unsigned)ci < (unsigned)bwd_item_cache_n;
    if (hit) {
        const BwdItemMem m = bwd_item_cache[ci];
        it.Tb = m.Tb, it.Ub = m.Ub, it.mask = m.mask;
    } else {
        it.Tb = length_T(p, it.b), it.Ub = length_U(p, it.b);
    }
    it.t_end = min(min(it.t_begin + kBwdRows, p.T), it.Tb);
    it.live = (it.t_begin < it.t_end) && (it.u0 < it.Ub);
    if (!hit) it.mask = it.live ? bwd_live_mask(jp, it.b, it.ut, tr, it.t_begin, it.t_end) : 0u;
    return it;
}
// power-of-two dlogits scale of an utterance: |S dl| <= 2^14 (|dl| <= 2 |cost_scale|)
__device__ __forceinline__ void bwd_scale(const LossParams &p, const int b, float &S, float &invS) {
    S = 1.0f, invS = 1.0f;
    const float cs = p.cost_scale ? fabsf(p.cost_scale[b]) : 1.0f;
    if (cs > 0.f && cs < 3.0e38f) {
        const int e = ilogbf(cs) + 1;
        S = ldexpf(1.0f, 13 - e), invS = ldexpf(1.0f, e - 13);
    }
}
__device__ __forceinline__ int lds_ld_i32(const uint32_t addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ bool lds_poll_ge(const uint32_t addr, const int need) {
    for (int spin = 0; spin < (1 << 20); ++spin) {
        if (lds_ld_i32(addr) >= need) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// one dlogits value (scaled by sS = cost_scale * S) of symbol v for a cell with the given set-up
__device__ __forceinline__ float bwd_dl(const float x, const int v, const int V, const int blank, const float c0, const float sS,
                                        const float corr_b, const float corr_l, const int lab) {
    float d = (v < V) ? sS * jex2(fmaf(x, kLog2e, c0)) : 0.f;
    d -= (v == blank) ? corr_b : 0.f;
    d -= (v == lab) ? corr_l : 0.f;
    return d;
}

template <bool SLOW>
__device__ void bwd_consumer(const JointParams &jp, char *ring, const uint32_t seq_a, const uint32_t use_a, const int cw,
                             const int j0, const int blk, const int nblk, const int lane, const int it_lo, const int it_hi,
                             const int n_tr) {
    const LossParams &p = jp.lp;
    const int J = jp.J, V = p.V;
    const int half = lane >> 5, l31 = lane & 31;
    const float *Ptab = SLOW ? jp.pred_proj : jp.expP;
    const float w2inv = jp.tflag[2], s2 = 1.0f / w2inv;  // W2 enters the products as s2 W2 (joint_prep_kernel; powers of two: exact)
    // B fragments of dh = dl . W2^T for this wave's 32 joint units: lane (j = j0 + l31, half), k-step ks holds
    // s2 W2[j][16 ks + 8 half + 0..7] as binary16 hi + lo parts; row-independent, kept in registers
    jh8 wf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int v = 32 * jp.vt + 16 * ks + 8 * half + e;  // (this launch's vocabulary tile)
            w[e] = (v < V) ? jp.W2[(size_t)(j0 + l31) * V + v] * s2 : 0.f;
        }
        split_h8(w, wf[ks][0], wf[ks][1]);
    }
    // The consumers of one SIMD (waves w, w + 4, w + 8) run identical instruction streams on the same rows; a one-off
    // stagger of about a third of a row keeps them from queueing for the same pipe at the same moment.
    for (int i = 0; i < (cw >> 2); ++i) __builtin_amdgcn_s_sleep(4);

    int rows_total = 0;  // rows this workgroup will visit (all waves count alike): an item per lane, one round trip for the masks
    for (int base = it_lo; base < it_hi; base += 64) {
        int c = (base + lane < it_hi) ? __popc(bwd_item(jp, base + lane, n_tr).mask) : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
        rows_total += __builtin_amdgcn_readfirstlane(c);
    }

    f32x16 accC, accW;
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[r] = 0.f, accW[r] = 0.f;
    float ec[16];
    float S = 1.0f, invS = 1.0f;
    int cur_bu = -1, cur_b = -1, cur_u0 = 0, cur_slot = 0;  // (cur_bu: the current column's first item)
    int row = 0;
    bool poisoned = false;

    auto flush_C = [&]() {  // d pred_proj partial of (cur_b, u-tile) accumulated by this workgroup so far
        if (cur_bu < 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = cur_u0 + cd_row(r, half);
            if (u < p.U) {
                float *dst = jp.dCpart + (((size_t)cur_slot * p.B + cur_b) * p.U + u) * J + j0 + l31;
                const float val = accC[r] * (invS * w2inv) + (jp.vt > 0 ? *dst : 0.f);  // (a later vocabulary tile adds to the first's)
                *dst = poisoned ? NAN : val;
            }
            accC[r] = 0.f;
        }
    };

    // software pipeline, one row deep: the A fragments of dh for the NEXT row are polled for and read while the current
    // row's products run, so that neither the sequence-word round trip nor the LDS latency sits in front of the MFMA chain
#ifdef JH_TRACE
    // dev builds: s_memtime stamps of consumers 0, 4, 8 (one SIMD) of workgroup 20, rows 40..71, 6 stamps per row
    long long *trc = (blockIdx.x == 20 && (cw & 3) == 0 && cw < 12) ? jp.trace + (cw >> 2) * 256 : nullptr;
    int rc_count = 0;
    const long long wg_t0 = (long long)__builtin_amdgcn_s_memtime();
#define BT(k)                                                                                          \
    do {                                                                                               \
        if (trc && lane == 0 && row >= 40 && row < 72) trc[(row - 40) * 6 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BT(k) do { } while (0)
#endif
    jh8 fa[2][2];
    float aj_next = 0.f;  // e^{2 enc_proj} (or enc_proj) of the next row for this lane's joint unit, delivered with the image
    if (rows_total > 0) {
        if (!lds_poll_ge(seq_a, 1)) poisoned = true;
        const jh8 *f0 = (const jh8 *)ring;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[ks][0] = f0[(ks * 2 + 0) * 64 + lane], fa[ks][1] = f0[(ks * 2 + 1) * 64 + lane];
        aj_next = *(const float *)(ring + 8192 + (cw * 32 + l31) * 4);
    }

    for (int item = it_lo; item < it_hi; ++item) {
        const BwdItem it = bwd_item(jp, item, n_tr);
        if (!it.live) continue;
        if (it.col_first != cur_bu) {
#ifdef JH_TRACE
            int rc_slot = -1;
            if (trc && lane == 0 && rc_count < 12) rc_slot = 192 + 5 * rc_count, trc[rc_slot] = (long long)__builtin_amdgcn_s_memtime(), trc[rc_slot + 4] = row;
            ++rc_count;
#endif
            flush_C();
#ifdef JH_TRACE
            if (rc_slot >= 0) trc[rc_slot + 1] = (long long)__builtin_amdgcn_s_memtime();
#endif
            cur_bu = it.col_first, cur_u0 = it.u0;
            // which of the workgroups of my group that share this column am I?  (its d pred_proj slab)
            cur_slot = blk - bwd_blk_of(jp.plan, it.col_first, nblk);
            if (it.b != cur_b) {  // S changes: re-scale the dW2 accumulator (exact, powers of two)
                float Sn, invSn;
                bwd_scale(p, it.b, Sn, invSn);
                const float f = Sn * invS;
#pragma unroll
                for (int r = 0; r < 16; ++r) accW[r] *= f;
                S = Sn, invS = invSn, cur_b = it.b;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = min(it.u0 + cd_row(r, half), p.U - 1);
                ec[r] = Ptab[((size_t)it.b * p.U + u) * J + j0 + l31];
            }
#ifdef JH_TRACE
            if (rc_slot >= 0) {
                trc[rc_slot + 2] = (long long)__builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                trc[rc_slot + 3] = (long long)__builtin_amdgcn_s_memtime();
            }
#endif
        }
        for (int t = it.t_begin; t < it.t_end; ++t) {
            if (!((it.mask >> (t - it.t_begin)) & 1u)) continue;  // (no cell of this row's tile carries mass: kOccFloor)
            const float aj = aj_next;
            const int slot = row % kBwdRing;
            const jh8 *frag = (const jh8 *)(ring + (size_t)slot * kBwdSlotBytes);
            BT(0);
            // B fragments of dW2 for this row: issued now, needed after the dh chain
            jh8 fb[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[ks][0] = frag[(4 + ks * 2 + 0) * 64 + lane], fb[ks][1] = frag[(4 + ks * 2 + 1) * 64 + lane];
            // (Packed f32 -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- takes 30 instructions per row out of this arithmetic.  Measured
            // twice: neutral while the kernel was producer-bound, -3.5 % of the step once it was consumer-bound -- the same as the
            // interleaved order below, and NOT additive with it (profiles/r05_notes.md).  A packed-f32 instruction does not overlap
            // with the matrix pipe the way a plain VALU instruction does (scripts/probes/probe_pk.hip), so the scalar form + the
            // interleave is what stays; this file is compiled with -fno-slp-vectorize, build.py.)
            float h[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = SLOW ? fast_tanh(aj + ec[r]) : tanh_from_exp(aj, ec[r]);
            BT(1);
            // S dh[u][j] = sum_v (S dl[u][v]) W2[j][v]
            f32x16 dh;
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dh = mfma3(fa[ks][0], fa[ks][1], wf[ks][0], wf[ks][1], dh);
            // Order within the row (scripts/probes/probe_mfma_cluster.hip: three waves per SIMD running this row's mix -- 144 VALU
            // + 12 MFMAs -- take 735 clocks per row and wave with the MFMAs in two clusters, 688 with one MFMA after every twelve
            // VALU; a SIMD gives a wave's MFMA little cover from the OTHER waves' VALU, the wave has to bring its own): the six
            // MFMAs of the dh chain go out between the h tile's multiply-adds and reciprocals, which are therefore pinned in front
            // of the poll (the compiler would sink them behind it, next to their first use).  -3.5 % of the fused step.
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x402, 8, 0);  // VALU | transcendental
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);    // MFMA
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(h[r]));
            BT(2);
            // next row's A fragments (its sequence word first)
            if (row + 1 < rows_total) {
                const int nslot = (row + 1) % kBwdRing;
                if (!lds_poll_ge(seq_a + 4u * nslot, row + 2)) poisoned = true;
                const jh8 *fn = (const jh8 *)(ring + (size_t)nslot * kBwdSlotBytes);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[ks][0] = fn[(ks * 2 + 0) * 64 + lane], fa[ks][1] = fn[(ks * 2 + 1) * 64 + lane];
                aj_next = *(const float *)((const char *)fn + 8192 + (cw * 32 + l31) * 4);
            }
            BT(3);
            // S dW2[j][v] += sum_u h[u][j] (S dl[u][v]): k-slot (ks, half, e) <-> lattice column cd_row(8 ks + e, half)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float hk[8] = {h[8 * ks], h[8 * ks + 1], h[8 * ks + 2], h[8 * ks + 3],
                                     h[8 * ks + 4], h[8 * ks + 5], h[8 * ks + 6], h[8 * ks + 7]};
                jh8 hhi, hlo;
                split_h8(hk, hhi, hlo);
                accW = mfma3(hhi, hlo, fb[ks][0], fb[ks][1], accW);
            }
            BT(4);
            float colsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dz = fmaf(-dh[r] * h[r], h[r], dh[r]);  // dh (1 - h^2)
                accC[r] += dz;
                colsum += dz;
            }
            colsum = half_swap_sum(colsum);
            // the dW2 chain likewise: split of the first eight columns | three MFMAs with the second split between them | three MFMAs
            // with the dz arithmetic between them (which is why the slot is handed back below and not in front of dz: the
            // lane-0 branch would end the scheduling region)
            __builtin_amdgcn_sched_group_barrier(0x402, 16, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 22, 0);
            }
            // every read of this row's slot has returned (fb above, fa one row earlier): hand it back to the loader
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(use_a + 4u * slot), "v"(1) : "memory");
            BT(5);
            if (lane < 32) {
                float *dst = jp.dApart + (((size_t)it.ut * p.B + it.b) * p.T + t) * J + j0 + lane;
                const float val = colsum * (invS * w2inv) + (jp.vt > 0 ? *dst : 0.f);
                *dst = poisoned ? NAN : val;
            }
            ++row;
        }
    }
    flush_C();
#ifdef JH_TRACE
    if (cw == 0 && lane == 0) {  // every workgroup: {clocks from the first poll to here, rows visited}
        jp.trace[1024 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memtime() - wg_t0;
        jp.trace[1024 + 2 * blockIdx.x + 1] = rows_total;
    }
#endif
    // this workgroup's dW2 partial: accW is [j rows][v cols]
#pragma unroll
    for (int r = 0; r < 16; ++r)
        jp.dWpart[((size_t)(jp.vt * kBwdMaxBlocks + blk) * J + j0 + cd_row(r, half)) * 32 + l31] = poisoned ? NAN : accW[r] * invS;
}

// Per-cell gradient set-up, once per lattice cell, by the whole chip in parallel (a chain of dependent gathers from the
// diagonal-major lattice arrays: latency-bound for a single wave, so it is NOT done by the backward kernel's producers):
//   rec[c] = { c0 = log2(alpha beta / L) - lse log2e (add x log2e: log2 of softmax x occupancy), sS = cost_scale S_b,
//              corr_b, corr_l = the blank / label corrections, already scaled by sS }    lab[c] = label of the cell or -1
// The producers of joint_bwd_kernel turn a row of these plus the parked logits into the dlogits fragments.
// Round 5: the lattice is the LINEAR one (mantissas + integer frames, rnnt_lin.h) unless the utterance was handed back
// (state word 2: log2 values + offsets, as before); on the linear lattice the kernel also evaluates the per-cell range
// certificate and raises the utterance's flag (joint_redo_kernel then redoes it and rewrites its records).
// Workgroup = a patch of 8 lattice rows x 32 columns.  The lattice state is diagonal-major (row n = t + u of the skewed
// arrays): the patch touches 40 diagonals and, on each, a window of at most 10 consecutive columns -- staged through LDS with
// window-contiguous loads (a lane-per-cell gather touches a different 64-byte sector for every lane: 4 x 64 sectors per wave).
__device__ __forceinline__ void rec_from_log(const JointParams &jp, const CellGrad &g, const float2 xx, const float S, float4 &rec,
                                             int &lab) {
    const float sS = g.scale * S;
    rec.x = g.c0, rec.y = sS;
    if (g.has_blank_corr) rec.z = sS * jex2(fmaf(xx.x, kLog2e, g.nl) + g.cb);
    if (g.has_label) lab = g.lab, rec.w = sS * jex2(fmaf(xx.y, kLog2e, g.nl) + g.cl);
}
// The same record from the linear lattice: ma = alpha^(t,u), mb = beta^(t,u), m_t1 = beta^(t+1,u), m_u1 = beta^(t,u+1) as
// mantissas relative to the frame tables (the last two are only looked at where that neighbour exists).  Returns false when
// the cell fails the range certificate (rnnt_lin.h lin_grad_setup: what a flush can have cost x the other side's mass, over the
// likelihood, must stay below 2^kCertBits).
__device__ __forceinline__ bool rec_from_lin(const LossParams &p, const Cell &cl, const uint32_t c, const float ma, const float mb,
                                             const float m_t1, const float m_u1, const float2 xx, const float S, float4 &rec,
                                             int &lab) {
    const int n = cl.t + cl.u;
    const int sh = p.lshift[cl.b];  // the block length the sweeps chose for this utterance
    const int kc = n >> sh, kc1 = (n + 1) >> sh;
    const int l0 = (int)fdiv((uint32_t)cl.u, p.divOG), l1 = (int)fdiv((uint32_t)cl.u + 1u, p.divOG);
    const size_t tb = (size_t)cl.b * p.NCl * 64;
    const int ea = p.EA[tb + (size_t)kc * 64 + l0], eb = p.EB[tb + (size_t)kc * 64 + l0];
    const float mL = p.lik[4 * cl.b];
    const int EL = ((const int *)p.lik)[4 * cl.b + 1];
    const float scale = p.cost_scale ? p.cost_scale[cl.b] : 1.0f;
    const float sS = scale * S;
    const float nl = -p.lse[c] * kLog2e;
    // alpha / L as (qa, base): mantissas may sit anywhere in the f32 range (a dragged frame), so split before multiplying
    const int xa = frexp_e(ma), xb = frexp_e(mb);
    const float qa = frexp_m(ma) * __builtin_amdgcn_rcpf(mL);
    const int base = ea + xa - EL;
    // log2 of the occupancy alpha beta / L: log2 of a mantissa product in [1/4, 2) plus an integer (-inf for a cell without mass)
    rec.x = (jlg2(qa * frexp_m(mb)) + (float)(base + eb + xb)) + nl;
    rec.y = sS;
    if (cl.t < cl.Tb - 1) {
        const float occ = ldexp_f(qa * frexp_m(m_t1), base + p.EB[tb + (size_t)kc1 * 64 + l0] + frexp_e(m_t1));
        rec.z = sS * occ * jex2(fmaf(xx.x, kLog2e, nl));
    } else if (cl.u == cl.Ub - 1) {
        rec.z = sS * ldexp_f(qa, base) * jex2(fmaf(xx.x, kLog2e, nl));  // the terminal transition: beta of the virtual end node is 1
    }
    if (cl.u < cl.Ub - 1) {
        lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], p.V);
        const float occ = ldexp_f(qa * frexp_m(m_u1), base + p.EB[tb + (size_t)kc1 * 64 + l1] + frexp_e(m_u1));
        rec.w = sS * occ * jex2(fmaf(xx.y, kLog2e, nl));
    }
    int worst = ea + eb - 252 - EL;
    if (mb != 0.f) worst = max(worst, ea - 126 + eb + xb - EL);
    if (ma != 0.f) worst = max(worst, eb - 126 + ea + xa - EL);
    return !(worst > kCertBits || !(ma <= FLT_MAX) || !(mb <= FLT_MAX) || !(ma >= 0.f) || !(mb >= 0.f));
}

constexpr int kRecRows = 8, kRecDiags = kRecRows + 32, kRecWin = 10;
__global__ __launch_bounds__(256) void joint_cellrec_kernel(const JointParams jp) {
    __shared__ float As[kRecDiags][kRecWin], Bs[kRecDiags][kRecWin];
    const LossParams &p = jp.lp;
    const int tid = threadIdx.x;
    const int n_tt = (p.T + kRecRows - 1) / kRecRows;
    int bid = blockIdx.x;
    const int ut = bid % jp.n_ut;
    bid /= jp.n_ut;
    const int tt = bid % n_tt;
    const int b = bid / n_tt;
    const int t0 = tt * kRecRows, u0 = ut * 32;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const bool live = (t0 < Tb) && (u0 < Ub);  // workgroup-uniform
    const bool lin = !lin_skip(p, b);          // the utterance's lattice is in the linear format (workgroup-uniform)
    if (live) {
        const size_t base = (size_t)b * p.Nr;
        for (int e = tid; e < 2 * kRecDiags * kRecWin; e += 256) {
            const int arr = e / (kRecDiags * kRecWin), q = e - arr * (kRecDiags * kRecWin);
            const int n = q / kRecWin, k = q - n * kRecWin;
            const int col = min(u0 + max(0, n - kRecRows) + k, p.Up - 1);  // clamped entries are never used
            const size_t idx = (base + min(t0 + u0 + n, p.Nr - 1)) * p.Up + col;
            (arr ? Bs : As)[n][k] = (arr ? p.Bt : p.A)[idx];
        }
    }
    __shared__ int rowbits;  // bit r: row t0 + r of this patch has a cell with occupancy above 2^-kOccFloor
    if (tid == 0) rowbits = 0;
    __syncthreads();
    const int r = tid >> 5, cu = tid & 31;
    const int t = t0 + r, u = u0 + cu;
    const bool inside = t < p.T && u < p.U;
    const uint32_t c = ((uint32_t)(b * p.T + min(t, p.T - 1))) * (uint32_t)p.U + (uint32_t)min(u, p.U - 1);
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    int lab = -1;
    bool occupied = false;
    if (inside && live && t < Tb && u < Ub) {
        Cell cl;
        cl.b = b, cl.t = t, cl.u = u, cl.Tb = Tb, cl.Ub = Ub, cl.valid = true;
        const int n = r + cu, n1 = n + 1;
        const int w0 = max(0, n - kRecRows), w1 = max(0, n1 - kRecRows);
        float S, invS;
        bwd_scale(p, b, S, invS);
        // blank / label logits from the forward kernel's compact copy (gathering them out of the parked tiles read two
        // sectors of every cell's 128-byte row: the whole 369 MB at C2 for 23 MB of payload)
        const float2 xx = jp.xbl[c];
        if (lin) {
            if (!rec_from_lin(p, cl, c, As[n][cu - w0], Bs[n][cu - w0], Bs[n1][cu - w1], Bs[n1][cu + 1 - w1], xx, S, rec, lab))
                atomicOr(p.flags + 4 * b + kFlagG, 1);  // (rare) the utterance is redone in the log domain
        } else {
            const CellGrad g = cell_grad_from(p, cl, c, As[n][cu - w0], Bs[n][cu - w0], Bs[n1][cu - w1], Bs[n1][cu + 1 - w1]);
            rec_from_log(jp, g, xx, S, rec, lab);
        }
        // rec.x = log2(occupancy) - lse log2 e: the occupancy alone decides whether the backward visits the cell's row (NaN: yes)
        occupied = jp.visit_all || !(fmaf(p.lse[c], kLog2e, rec.x) <= (float)-kOccFloor);
    }
    const unsigned long long occ = __ballot(occupied);
    if ((tid & 63) == 0) {
        const int bits = ((uint32_t)occ ? 1 : 0) | ((uint32_t)(occ >> 32) ? 2 : 0);
        if (bits) atomicOr(&rowbits, bits << (2 * (tid >> 6)));
    }
    // (a row the backward will not visit needs no records: its 32 cells are this half-wave's)
    const bool row_visited = ((tid & 32) ? (uint32_t)(occ >> 32) : (uint32_t)occ) != 0u;
    if (inside && row_visited) {
        jp.rec[c] = rec;
        jp.reclab[c] = lab;
    }
    __syncthreads();
    if (tid == 0) jp.live8[((size_t)b * jp.n_ut + ut) * (size_t)(4 * ((p.T + 31) >> 5)) + tt] = (uint8_t)rowbits;
}

// ---------------------------------------------------------------------------------------------
// The hand-back of the fused joint (rnnt_redo.h; the loss op's is lin_redo_kernel): grid = utterances x team, a workgroup whose
// utterance is fine reads its flag words and the two likelihoods and returns.  A flagged utterance (a sweep's flag, the two
// likelihoods apart, the certificate of joint_cellrec_kernel) is redone in the log domain from its PARKED logits tile -- `q` is
// the loss parameters with acts = the [cells][32] tile (pad symbols at -1e30: probability zero) -- and, when the backward
// wants them, its per-cell records are rewritten from the log-domain lattice.  Its state word then says "log-domain lattice".
// ---------------------------------------------------------------------------------------------
template <int K, int G, int NB>
__global__ __launch_bounds__(kRedoThreads) void joint_redo_kernel(const JointParams jp, const LossParams q, const int want_rec,
                                                                  const int team) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LossParams &p = jp.lp;
    const int tid = threadIdx.x;
    const int ub = (int)blockIdx.x / team;
    const int b = p.b0 + ub;
    RedoTeam tm;
    tm.k = (int)blockIdx.x - ub * team, tm.n = team, tm.bar = p.bar + kRedoCtr * b, tm.ok = true;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const i32x4 fw = __builtin_nontemporal_load((const i32x4 *)(p.flags + 4 * b));
    const f64x2 lw = __builtin_nontemporal_load((const f64x2 *)(p.ll + 2 * b));
    if (fw[kFlagState] == 2) return;  // already redone (its records come from joint_cellrec_kernel's log-domain branch)
    const bool agree = fabs(lw[0] - lw[1]) * 0.6931471805599453 <= 2e-5 + 1e-8 * fabs(lw[0]);
    if ((fw[kFlagA] | fw[kFlagB] | fw[kFlagG]) == 0 && agree) return;
    redo_lattice<K, G, NB>(q, b, tm, lds, tid);
    if (!tm.ok && q.costs && tm.k == 0 && tid == 0) st_f32_wt(q.costs + b, NAN);
    if (!want_rec) return;
    uint32_t lo, hi;
    redo_cell_range(p, b, tm, lo, hi);
    float S, invS;
    bwd_scale(p, b, S, invS);
    for (uint32_t c = lo + (uint32_t)tid; c < hi; c += kRedoThreads) {
        const Cell cl = decode(p, c);
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        int lab = -1;
        if (cl.valid) {
            const CellGrad g = cell_grad_setup<true>(p, cl, c);
            rec_from_log(jp, g, jp.xbl[c], S, rec, lab);
            if (!tm.ok) rec.x = NAN;
        }
        jp.rec[c] = rec;
        jp.reclab[c] = lab;
    }
    if (tm.k == 0) {  // the row bits joint_cellrec_kernel formed from the abandoned lattice: the backward visits this utterance whole
        const int nbytes = jp.n_ut * 4 * ((p.T + 31) >> 5);
        for (int i = tid; i < nbytes; i += kRedoThreads) jp.live8[(size_t)b * nbytes + i] = 0xff;
    }
}

// The cut of joint_bwd_kernel's item list into ranges of equal weight (see kBwdSlots).  One workgroup: a chunk of items per thread,
// a scan over the threads' sums, then each thread writes its items' prefix weights and the cuts that fall into its chunk.
constexpr int kPlanThreads = 1024;
__host__ __device__ inline int plan_stamp(int T, int U, int B, int J, int V) {
    return (int)(0x5eed0000u ^ ((unsigned)T * 73856093u) ^ ((unsigned)U * 19349663u) ^ ((unsigned)B * 83492791u) ^ ((unsigned)J * 2654435761u) ^ (unsigned)V);
}
__global__ __launch_bounds__(kPlanThreads) void joint_rowplan_kernel(const JointParams jp) {
    __shared__ int part[kPlanThreads];
    const LossParams &p = jp.lp;
    const int tid = threadIdx.x, nblk = jp.nblk;
    const int n_tr = (p.T + kBwdRows - 1) / kBwdRows, n_items = n_tr * p.B * jp.n_ut;
    const int chunk = (n_items + kPlanThreads - 1) / kPlanThreads, lo = min(tid * chunk, n_items), hi = min(lo + chunk, n_items);
    auto weight = [&](const int item) -> int {
        const BwdItem it = bwd_item<false>(jp, item, n_tr);
        return it.live ? __popc(it.mask) + kBwdItemCost : 0;
    };
    __shared__ int rows_visited, rows_inside;  // (for get_rnnt_joint_backward_rows: what the pruning left of the lattice rows)
    if (tid == 0) rows_visited = 0, rows_inside = 0;
    __syncthreads();
    // (the weights are fetched once -- lengths and row bits: two dependent round trips per item -- and kept in LDS for the second pass)
    constexpr int kPlanCache = 8192;
    __shared__ int wl[kPlanCache];
    int sum = 0, vis = 0, ins = 0;
    for (int i = lo; i < hi; ++i) {
        const BwdItem it = bwd_item<false>(jp, i, n_tr);
        const int wi = it.live ? __popc(it.mask) + kBwdItemCost : 0;
        if (it.live) sum += wi, vis += __popc(it.mask), ins += it.t_end - it.t_begin;
        if (i < kPlanCache) wl[i] = wi;
    }
    auto weight2 = [&](const int item) -> int { return item < kPlanCache ? wl[item] : weight(item); };
    if (vis) atomicAdd(&rows_visited, vis);
    if (ins) atomicAdd(&rows_inside, ins);
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        jp.plan[3 + kBwdMaxBlocks + n_items] = rows_visited, jp.plan[4 + kBwdMaxBlocks + n_items] = rows_inside;
        jp.plan[5 + kBwdMaxBlocks + n_items] = plan_stamp(p.T, p.U, p.B, jp.J, p.V);  // (joint_backward_rows: the counts are this shape's)
    }
    for (int d = 1; d < kPlanThreads; d <<= 1) {  // inclusive scan
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int total = part[kPlanThreads - 1], before = part[tid] - sum;
    const int col_w = n_tr * (kBwdRows + kBwdItemCost);  // the heaviest column there can be
    // (at least one item's weight: consecutive items then sit in the same or in neighbouring workgroups -- no workgroup inside a
    // column's span is left without an item of it, which is what the d pred_proj reduction assumes of the slabs)
    const int target = max(max((total + nblk - 1) / nblk, (col_w + kBwdSlots - 3) / (kBwdSlots - 2)), kBwdRows + kBwdItemCost);
    int *prefix = jp.plan + 2 + kBwdMaxBlocks, *first = jp.plan + 1;
    if (tid == 0) jp.plan[0] = target;
    int w = before;
    for (int i = lo; i < hi; ++i) {
        prefix[i] = w;
        const int k = min(w / target, nblk - 1);
        const int kp = i == 0 ? -1 : min((w - weight2(i - 1)) / target, nblk - 1);  // workgroup of the item before
        for (int q = kp + 1; q <= k; ++q) first[q] = i;  // workgroups kp + 1 .. k start here (all but the last of them empty)
        w += weight2(i);
    }
    if (hi == n_items && lo < hi) {  // (whoever holds the last item)
        prefix[n_items] = total;
        const int klast = min((total - weight2(n_items - 1)) / target, nblk - 1);
        for (int q = klast + 1; q <= nblk; ++q) first[q] = n_items;
    }
}

// dlogits fragments of one lattice row tile (32 cells x 32 symbols), 8 KB: pieces 0..3 = A fragments of dh ([ks][hi, lo],
// lane = cell, k = symbol 16 ks + 8 half + e), pieces 4..7 = B fragments of dW2 ([ks][hi, lo], lane = symbol, k-slot
// (ks, half, e) = lattice column cd_row(8 ks + e, half)); every value is S_b . dlogits split into binary16 hi + lo.
// Built ONCE per row by a producer wave of the backward workgroup, straight into the LDS ring -- they never exist in HBM.
// The two producers (alternating rows) sit on the two SIMDs that carry only two consumer waves.
// walks the live rows of a workgroup's item range in order (the same sequence every wave of the workgroup sees)
struct BwdRowIter {
    int item, it_hi, n_tr, t;
    BwdItem it;
    bool valid;
    __device__ __forceinline__ void settle(const JointParams &jp) {  // move to the first live row at or after (item, t)
        while (item < it_hi) {
            if (it.live && t < it.t_end) {
                const uint32_t rest = it.mask >> (t - it.t_begin);  // rows of this tile still to come, this one at bit 0
                if (rest) {
                    t += __builtin_ctz(rest);
                    valid = true;
                    return;
                }
            }
            ++item;
            if (item < it_hi) {
                it = bwd_item(jp, item, n_tr);
                t = it.t_begin;
            }
        }
        valid = false;
    }
    __device__ __forceinline__ void init(const JointParams &jp, int lo, int hi, int ntr) {
        item = lo, it_hi = hi, n_tr = ntr, valid = false;
        if (item < it_hi) {
            it = bwd_item(jp, item, n_tr);
            t = it.t_begin;
        }
        settle(jp);
    }
    __device__ __forceinline__ void next(const JointParams &jp) {
        ++t;
        settle(jp);
    }
};

// everything a producer needs from memory for one row; loaded one row (of its own) ahead
struct BwdRowLoads {
    float4 rc;        // per-cell set-up of this lane's cell
    int lab;
    float4 xa[2][2];  // this cell's logits: symbols 16 ks + 8 half + 0..7 of this launch's vocabulary tile
    float av[5];      // enc-side addends of the group's joint units
    int b, t, u0, Ub;
};
__device__ __forceinline__ void bwd_row_loads(const JointParams &jp, const BwdItem &it, const int t, const float *Etab,
                                              const int n_cons, const int group, const int lane, BwdRowLoads &L) {
    const LossParams &p = jp.lp;
    const int half = lane >> 5, l31 = lane & 31;
    const uint32_t cbase = ((uint32_t)(it.b * p.T + t)) * (uint32_t)p.U + (uint32_t)it.u0;
    // every load of a row is unconditional (addresses clamped into the tensor, values masked at use): the number of loads in
    // flight is then a compile-time constant and the wait for THIS row's data need not drain the next row's
    const uint32_t c = cbase + (uint32_t)min(l31, p.U - 1 - it.u0);
    L.b = it.b, L.t = t, L.u0 = it.u0, L.Ub = it.Ub;
    L.rc = jp.rec[c];
    L.lab = jp.reclab[c];
    const int Vp = 32 * jp.VT, vo = 32 * jp.vt;  // row stride of the parked logits, this launch's vocabulary tile
    const float *xrow = jp.dl + (size_t)c * Vp + vo;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        L.xa[ks][0] = *(const float4 *)(xrow + 16 * ks + 8 * half), L.xa[ks][1] = *(const float4 *)(xrow + 16 * ks + 8 * half + 4);
    const float *asrc = Etab + ((size_t)it.b * p.T + t) * jp.J + group * n_cons * 32;
#pragma unroll
    for (int k = 0; k < 5; ++k) L.av[k] = asrc[min(lane + 64 * k, n_cons * 32 - 1)];
}

// One 16-bit element of the row's A image (already in the ring slot), zero-extended.  (Not ds_read_u16_d16 / _d16_hi into the two
// halves of one register: with SRAM ECC on -- every MI300 / MI355 -- a d16 load clears the other half instead of keeping it.)
__device__ __forceinline__ uint32_t lds_ld_u16(const uint32_t addr, const int imm) {
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm) : "memory");
    return v;
}

// The two producers of a workgroup turn rows of parked logits + per-cell records into the dlogits operand images of the two
// products, alternating rows.  Round 5 (late): a s_memtime trace (-DJH_TRACE, scripts/probes/bwd_trace.py) showed the kernel
// PRODUCER-bound -- 7,470 clocks per own row against 3,736 per row for the consumers, which sat in their sequence-word poll for
// 600 ... 1,500 of them: a producer evaluated every dlogits value TWICE (once per operand layout: 16 more scattered loads of the
// logits, the per-cell records through an LDS scratch, 16 more exponentials) in ~1,100 instructions per row.  Now it evaluates
// them once, in the layout of the dh operand (lane = lattice column, 16 symbols), splits and stores that image, and GATHERS the
// dW2 operand (lane = symbol, 16 lattice columns: the transpose) from the image it has just written, 16-bit element by element
// (ds_read_u16: 32 reads per lane, immediate offsets, + 16 v_lshl_or_b32; LDS returns a wave's own writes in order).  The db2 sums
// moved to the first layout (16 per-lane accumulators, reduced over the lattice-column lanes once per workgroup).
__device__ void bwd_producer(const JointParams &jp, char *ring, const uint32_t seq_a, const uint32_t use_a,
                             const int pw, const int n_cons, const int group, const int blk, const bool want_db, const int lane,
                             const int it_lo, const int it_hi, const int n_tr) {
    const LossParams &p = jp.lp;
    const int vo = 32 * jp.vt;
    const int half = lane >> 5, l31 = lane & 31;
    const float *Etab = (jp.tflag[0] != 0.f) ? jp.enc_proj : jp.expE;
    float dbacc[2][8];  // db2[vo + 16 ks + 8 half + e] over the lattice columns this lane has seen
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dbacc[ks][e] = 0.f;
    float invS = 1.0f;
    int cur_b = -1;
    bool poisoned = false;
    // element (lattice column u, symbol w of the tile) of the A image sits at byte
    //   (w >> 4) * 2048 + hl * 1024 + (u + 32 * ((w >> 3) & 1)) * 16 + 2 * (w & 7)      (hl: 0 = hi parts, 1 = lo parts)
    // this lane gathers symbol l31 of the columns cd_row(8 ks + e, half) = (e & 3) + 8 (e >> 2) + 16 ks + 4 half
    const uint32_t gather_lane = (uint32_t)((l31 >> 4) * 2048 + ((l31 >> 3) & 1) * 512 + 2 * (l31 & 7) + 64 * half);
    const uint32_t ring_a = (uint32_t)(uintptr_t)((lds_void *)ring);
    // the producers are the head of the pipeline and by far the lighter role: they win issue arbitration on their SIMD
    __builtin_amdgcn_s_setprio(3);
    // the two producers alternate rows: this one takes rows pw, pw + 2, ...
    BwdRowIter iter;
    iter.init(jp, it_lo, it_hi, n_tr);
    if (pw == 1 && iter.valid) iter.next(jp);
    int row = pw;
#ifdef JH_TRACE
    // producer stamps of workgroup 20, its rows 40..71: 0 = top of the iteration, 1 = next row's loads issued, 2 = slot free,
    // 3 = row published
    long long *ptrc = (blk == 10 && group == 0) ? jp.trace + 768 + pw * 64 : nullptr;
#define PT(k)                                                                                                      \
    do {                                                                                                           \
        if (ptrc && lane == 0 && row >= 40 && row < 72) ptrc[((row - 40) >> 1) * 4 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PT(k) do { } while (0)
#endif
    BwdRowLoads L;
    if (iter.valid) bwd_row_loads(jp, iter.it, iter.t, Etab, n_cons, group, lane, L);
    while (iter.valid) {
        // ---- my next row's loads go out before this row's arithmetic: a producer never sits behind a memory round trip
        PT(0);
        iter.next(jp);
        if (iter.valid) iter.next(jp);
        BwdRowLoads Ln = L;
        if (iter.valid) bwd_row_loads(jp, iter.it, iter.t, Etab, n_cons, group, lane, Ln);
        PT(1);
        if (L.b != cur_b) {
            float S;
            bwd_scale(p, L.b, S, invS);
            cur_b = L.b;
        }
        const int slot = row % kBwdRing;
        const bool valid = L.u0 + l31 < L.Ub;
        // ---- the slot must be free: every consumer has finished the row that used it kBwdRing rows ago
        if (row >= kBwdRing && !lds_poll_ge(use_a + 4u * slot, n_cons * (row / kBwdRing))) poisoned = true;
        PT(2);
        char *slotp = ring + (size_t)slot * kBwdSlotBytes;
        jh8 *frag = (jh8 *)slotp;
        // ---- A fragments of dh (row = this lane's lattice column, k = symbol 16 ks + 8 half + e of the tile): the dlogits values
        // themselves.  Symbols beyond V need no test: their parked logit is -1e30 (the bias table of joint_prep_kernel), e^x = 0.
        const int lab_rel = (valid ? L.lab : -1) - vo - 8 * half, blank_rel = p.blank - vo - 8 * half;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float xs[8] = {L.xa[ks][0].x, L.xa[ks][0].y, L.xa[ks][0].z, L.xa[ks][0].w,
                                 L.xa[ks][1].x, L.xa[ks][1].y, L.xa[ks][1].z, L.xa[ks][1].w};
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = L.rc.y * jex2(fmaf(xs[e], kLog2e, L.rc.x));
                v -= (blank_rel == 16 * ks + e) ? L.rc.z : 0.f;
                v -= (lab_rel == 16 * ks + e) ? L.rc.w : 0.f;
                d[e] = valid ? v : 0.f;
                dbacc[ks][e] = fmaf(d[e], invS, dbacc[ks][e]);
            }
            if (poisoned) d[0] = NAN;
            jh8 hi, lo;
            split_h8(d, hi, lo);
            frag[(ks * 2 + 0) * 64 + lane] = hi;
            frag[(ks * 2 + 1) * 64 + lane] = lo;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (lane + 64 * k < n_cons * 32) *(float *)(slotp + 8192 + (lane + 64 * k) * 4) = L.av[k];
        // ---- B fragments of dW2 (column = symbol l31, k-slot (ks, half, e) = lattice column cd_row(8 ks + e, half)): gathered
        const uint32_t ga = ring_a + (uint32_t)slot * (uint32_t)kBwdSlotBytes + gather_lane;
        uint32_t w0[2][2][4], w1[2][2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // elements e = 2 q, 2 q + 1
                    w0[ks][hl][q] = lds_ld_u16(ga, hl * 1024 + (((2 * q) & 3) + 8 * ((2 * q) >> 2) + 16 * ks) * 16);
                    w1[ks][hl][q] = lds_ld_u16(ga, hl * 1024 + (((2 * q + 1) & 3) + 8 * ((2 * q + 1) >> 2) + 16 * ks) * 16);
                }
        // The compiler does not know that the asm reads above are loads: the wait must DEFINE the gathered registers, or their first
        // use is scheduled in front of it (it was: four v_lshl_or_b32 ahead of the s_waitcnt -- right by luck, the reads they used were
        // ~130 cycles old -- until the registers were tied to the wait; tests/test_isa_audit.py checks the order).
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 g0[2][2], g1[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                g0[ks][hl] = (u32x4){w0[ks][hl][0], w0[ks][hl][1], w0[ks][hl][2], w0[ks][hl][3]};
                g1[ks][hl] = (u32x4){w1[ks][hl][0], w1[ks][hl][1], w1[ks][hl][2], w1[ks][hl][3]};
            }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g0[0][0]), "+v"(g0[0][1]), "+v"(g0[1][0]), "+v"(g0[1][1]), "+v"(g1[0][0]), "+v"(g1[0][1]), "+v"(g1[1][0]), "+v"(g1[1][1])
                     :
                     : "memory");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                const u32x4 packed = g0[ks][hl] | (g1[ks][hl] << 16);
                frag[(4 + ks * 2 + hl) * 64 + lane] = __builtin_bit_cast(jh8, packed);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the row's fragments and addends are in LDS
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(seq_a + 4u * slot), "v"(row + 1) : "memory");
        PT(3);
        L = Ln;
        row += 2;
    }
    if (want_db) {
        // sum over the lattice-column lanes of each half; lanes 0 and 32 write their half's 16 symbols
        float *dst = jp.dbpart + ((size_t)(jp.vt * kBwdMaxBlocks + blk) * 2 + pw) * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = dbacc[ks][e];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
                if (l31 == 0) dst[16 * ks + 8 * half + e] = poisoned ? NAN : v;
            }
    }
}

// Consumers own 32 joint units each; a workgroup holds at most kBwdMaxCons of them (register budget: 12 waves per CU), so
// wider joints are covered by several GROUPS of workgroups, each group walking the whole item list for its share of J.
constexpr int kBwdMaxCons = 10;
inline int bwd_groups(int J) { return (J / 32 + kBwdMaxCons - 1) / kBwdMaxCons; }

__global__ __launch_bounds__(768) void joint_bwd_kernel(const JointParams jp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LossParams &p = jp.lp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_groups = (jp.J / 32 + kBwdMaxCons - 1) / kBwdMaxCons;
    const int n_cons = jp.J / 32 / n_groups;
    const bool slow = jp.tflag[0] != 0.f;  // kernel-uniform
    char *ring = (char *)lds;                                    // [kBwdRing][10 KB] dlogits fragments + enc addends of a row
    int *ctr = (int *)(ring + kBwdRing * kBwdSlotBytes);         // seq[kBwdRing], use[kBwdRing]
    if (tid < 2 * kBwdRing) ctr[tid] = 0;
    const uint32_t seq_a = (uint32_t)(uintptr_t)((lds_void *)ctr), use_a = seq_a + 4u * kBwdRing;

    // gridDim.x = n_groups * nblk: workgroup -> (group, index within the group); contiguous item range per workgroup
    const int nblk = gridDim.x / n_groups;
    const int group = blockIdx.x % n_groups, blk = blockIdx.x / n_groups;
    const int n_tr = (p.T + kBwdRows - 1) / kBwdRows;
    const int it_lo = jp.plan[1 + blk], it_hi = jp.plan[2 + blk];  // joint_rowplan_kernel: ranges of equal weight, not of equal length
    {   // the items' lengths and row bits, a thread per item (bwd_item_cache)
        const int n_c = min(it_hi - it_lo, kBwdItemCache);
        for (int i = tid; i < n_c; i += (int)blockDim.x) {
            const BwdItem it = bwd_item<false>(jp, it_lo + i, n_tr);
            bwd_item_cache[i].mask = it.mask, bwd_item_cache[i].Tb = it.Tb, bwd_item_cache[i].Ub = it.Ub;
#ifdef JH_TRACE
            if (blockIdx.x < 8 && (blockIdx.x & 1) == 0 && i < 24) jp.trace[1536 + 24 * (blockIdx.x >> 1) + i] = it.mask;
#endif
        }
        if (tid == 0) bwd_item_cache_lo = it_lo, bwd_item_cache_n = n_c;
    }
    __syncthreads();
    if (wave < n_cons) {
        const int j0 = (group * n_cons + wave) * 32;
        if (!slow)
            bwd_consumer<false>(jp, ring, seq_a, use_a, wave, j0, blk, nblk, lane, it_lo, it_hi, n_tr);
        else
            bwd_consumer<true>(jp, ring, seq_a, use_a, wave, j0, blk, nblk, lane, it_lo, it_hi, n_tr);
    } else {
        bwd_producer(jp, ring, seq_a, use_a, wave - n_cons, n_cons, group, blk, group == 0, lane,
                     it_lo, it_hi, n_tr);
    }
}

// abs-max (as the bit pattern of |x|) of what a block wrote, for the consumer of the output (dense_kernels.hip): every block
// stores its own entry -- no atomics, no zero-fill
__device__ __forceinline__ unsigned absbits4(const float4 v) {
    return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
               max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
}
__device__ __forceinline__ void store_block_max(unsigned m, unsigned *slot) {  // slot = this block's entry
    __shared__ unsigned red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // (write-through: the consumer's loads are agent-scope too, dense_kernels.hip dense_max_of)
    if (threadIdx.x == 0)
        __hip_atomic_store(slot, max(max(red[0], red[1]), max(red[2], red[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Block `blk` of `nblk` of: out[i] = poison + sum_q in[q*n + i]  (fixed order q = 0, 1, ...: deterministic; 16-byte accesses, up
// to four partials in flight per thread).  `bmslot` (nullable): where this block's abs-max bit pattern goes.
__device__ __forceinline__ void reduce_partials_body(float *out, const float *in, const int nparts, const size_t n, const unsigned blk,
                                                     const unsigned nblk, unsigned *bmslot, const float poison) {
    unsigned bm = 0u;
    if ((n & 3) == 0 && (((uintptr_t)out | (uintptr_t)in) & 15) == 0) {
        const size_t n4 = n >> 2;
        const float4 *in4 = (const float4 *)in;
        for (size_t i = (size_t)blk * 256 + threadIdx.x; i < n4; i += (size_t)nblk * 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int q = 0;
            for (; q + 4 <= nparts; q += 4) {
                const float4 a = in4[(size_t)q * n4 + i], b = in4[(size_t)(q + 1) * n4 + i];
                const float4 c = in4[(size_t)(q + 2) * n4 + i], d = in4[(size_t)(q + 3) * n4 + i];
                s.x = (((s.x + a.x) + b.x) + c.x) + d.x, s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
                s.z = (((s.z + a.z) + b.z) + c.z) + d.z, s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
            }
            for (; q < nparts; ++q) {
                const float4 a = in4[(size_t)q * n4 + i];
                s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            }
            s.x += poison, s.y += poison, s.z += poison, s.w += poison;
            ((float4 *)out)[i] = s;
            bm = max(bm, absbits4(s));
        }
        if (bmslot) store_block_max(bm, bmslot);
        return;
    }
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < n; i += (size_t)nblk * 256) {
        float s = 0.f;
        for (int q = 0; q < nparts; ++q) s += in[(size_t)q * n + i];
        s += poison;
        out[i] = s;
        bm = max(bm, __float_as_uint(s) & 0x7fffffffu);
    }
    if (bmslot) store_block_max(bm, bmslot);
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(float *out, const float *in, int nparts, size_t n, unsigned *blockmax) {
    reduce_partials_body(out, in, nparts, n, blockIdx.x, gridDim.x, blockmax ? blockmax + blockIdx.x : nullptr, 0.f);
}

// d enc_proj[b][t][:] = sum over the u-tiles of their partial rows, in u-tile order.  Every backward kernel writes the row
// (ut, b, t) exactly when the u-tile starts inside the utterance's label range and t < T_b, and never otherwise: the reduction
// reads only those rows (and writes zeros for t >= T_b), so the 4 n_ut B T J bytes of partials need no zero-fill.
// `live8` (the fused f32-grade joint; nullptr for the f16 joint): joint_bwd_kernel's row bits -- a partial row it skipped was never
// written and counts as zero.
__device__ __forceinline__ void reduce_enc_body(float *out, const float *in, const int n_ut, const LossParams &p, const int J,
                                                const unsigned blk, const unsigned nblk, unsigned *bmslot, const float poison,
                                                const uint8_t *live8) {
    const uint32_t J4 = (uint32_t)J >> 2, n4 = (uint32_t)p.B * (uint32_t)p.T * J4;
    const float4 *in4 = (const float4 *)in;
    unsigned bm = 0u;
    const uint32_t lrow = 4u * (((uint32_t)p.T + 31u) >> 5);  // bytes of row bits per (utterance, u-tile)
    auto visited = [&](const uint32_t b, const uint32_t t, const int q) -> bool {
        return !live8 || ((live8[((size_t)b * n_ut + q) * lrow + (t >> 3)] >> (t & 7u)) & 1u);
    };
    if (((uintptr_t)out & 15) != 0) {  // a caller's gradient buffer off the 16-byte grid (the partials are workspace: aligned)
        const uint32_t n = n4 * 4u;
        for (uint32_t i = blk * 256u + threadIdx.x; i < n; i += nblk * 256u) {
            const uint32_t row = i / (uint32_t)J, b = row / (uint32_t)p.T, t = row - b * (uint32_t)p.T;
            float s = 0.f;
            if ((int)t < length_T(p, (int)b)) {
                const int nv = min(n_ut, (length_U(p, (int)b) + 31) >> 5);
                for (int q = 0; q < nv; ++q)
                    if (visited(b, t, q)) s += in[(size_t)q * n + i];
            }
            s += poison;
            out[i] = s;
            bm = max(bm, __float_as_uint(s) & 0x7fffffffu);
        }
        if (bmslot) store_block_max(bm, bmslot);
        return;
    }
    for (uint32_t i = blk * 256u + threadIdx.x; i < n4; i += nblk * 256u) {
        const uint32_t row = i / J4, b = row / (uint32_t)p.T, t = row - b * (uint32_t)p.T;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)t < length_T(p, (int)b)) {
            const int nv = min(n_ut, (length_U(p, (int)b) + 31) >> 5);
            int q = 0;
            if (live8) {  // (the same left-to-right association as below, minus the rows that do not exist)
                uint32_t vis = 0u;  // all the row bits first (independent loads), then the rows
                for (int k = 0; k < min(nv, 32); ++k) vis |= (visited(b, t, k) ? 1u : 0u) << k;
                for (; q < nv; ++q)
                    if (q >= 32 ? visited(b, t, q) : ((vis >> q) & 1u)) {
                        const float4 a = in4[(size_t)q * n4 + i];
                        s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
                    }
            }
            for (; q + 4 <= nv; q += 4) {
                const float4 a = in4[(size_t)q * n4 + i], bb = in4[(size_t)(q + 1) * n4 + i];
                const float4 c = in4[(size_t)(q + 2) * n4 + i], d = in4[(size_t)(q + 3) * n4 + i];
                s.x = (((s.x + a.x) + bb.x) + c.x) + d.x, s.y = (((s.y + a.y) + bb.y) + c.y) + d.y;
                s.z = (((s.z + a.z) + bb.z) + c.z) + d.z, s.w = (((s.w + a.w) + bb.w) + c.w) + d.w;
            }
            for (; q < nv; ++q) {
                const float4 a = in4[(size_t)q * n4 + i];
                s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            }
        }
        s.x += poison, s.y += poison, s.z += poison, s.w += poison;
        ((float4 *)out)[i] = s;
        bm = max(bm, absbits4(s));
    }
    if (bmslot) store_block_max(bm, bmslot);
}
__global__ __launch_bounds__(256) void reduce_enc_kernel(float *out, const float *in, int n_ut, const LossParams p, int J,
                                                         unsigned *blockmax, const uint8_t *live8) {
    reduce_enc_body(out, in, n_ut, p, J, blockIdx.x, gridDim.x, blockmax ? blockmax + blockIdx.x : nullptr, 0.f, live8);
}

// d pred_proj[b][u][:] from the partial slabs of joint_bwd_kernel: an (utterance, u-tile) is written by the workgroups of a J
// group whose item ranges hold one of its live row tiles -- consecutive workgroups, slab index = distance from the first --
// and by nobody when the u-tile starts beyond the utterance's labels.  The reduction recomputes that count from the item
// partition (bwd_consumer's cur_slot) and reads only slabs that were written: the partial buffers need no zero-fill.
__device__ __forceinline__ void reduce_pred_body(float *out, const float *in, const JointParams &jp, const unsigned blk,
                                                 const unsigned nblk, unsigned *bmslot, const float poison) {
    const LossParams &p = jp.lp;
    const int J = jp.J;
    const uint32_t n = (uint32_t)p.B * (uint32_t)p.U * (uint32_t)J;
    const bool vec = (((uintptr_t)out) & 15) == 0;
    const uint32_t step = vec ? 4u : 1u, nq = n / step, Jq = (uint32_t)J / step;
    const int n_tr = (p.T + kBwdRows - 1) / kBwdRows;
    unsigned bm = 0u;
    for (uint32_t i = blk * 256u + threadIdx.x; i < nq; i += nblk * 256u) {
        const uint32_t row = i / Jq, b = row / (uint32_t)p.U, u = row - b * (uint32_t)p.U;
        const int ut = (int)(u >> 5);
        // the slabs that exist for this (utterance, u-tile): one per workgroup whose range (joint_rowplan_kernel) holds items of the
        // column inside the utterance
        int ns = 0;
        if (ut * 32 < length_U(p, (int)b)) {
            const int first = ((int)b * jp.n_ut + ut) * n_tr, n_live = (length_T(p, (int)b) + kBwdRows - 1) / kBwdRows;
            ns = min(bwd_blk_of(jp.plan, first + n_live - 1, jp.nblk) - bwd_blk_of(jp.plan, first, jp.nblk) + 1, kBwdSlots);
        }
        if (vec) {
            const float4 *in4 = (const float4 *)in;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < ns; ++q) {
                const float4 a = in4[(size_t)q * nq + i];
                s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            }
            s.x += poison, s.y += poison, s.z += poison, s.w += poison;
            ((float4 *)out)[i] = s;
            bm = max(bm, absbits4(s));
        } else {
            float s = 0.f;
            for (int q = 0; q < ns; ++q) s += in[(size_t)q * n + i];
            s += poison;
            out[i] = s;
            bm = max(bm, __float_as_uint(s) & 0x7fffffffu);
        }
    }
    if (bmslot) store_block_max(bm, bmslot);
}

// Deterministic tree: out[i] = poison + sum_p in[p*stride_p + map(i)], 256 threads = 32 outputs x 8 partial lanes.
// Each partial lane sums its strided share in a fixed order, then the 8 lanes are combined in order.
template <bool W2MAP>
__device__ __forceinline__ void reduce_small_body(float *out, const float *in, const int nparts, const int tparts, const int n, const int J,
                                                  const int V, const unsigned blk, const float poison) {  // tparts: partials ALLOCATED per vocabulary tile
    __shared__ float sm[8][33];
    const int o = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int i = (int)blk * 32 + o;
    float s = 0.f;
    if (i < n) {
        size_t base, stride;
        if (W2MAP) {  // in is [tile][tparts][J][32], out is [J][V]
            const int j = i / V, v = i - j * V;
            base = ((size_t)(v >> 5) * tparts * J + j) * 32 + (v & 31);
            stride = (size_t)J * 32;
        } else {  // in is [tile][tparts][32], out is [V]
            base = (size_t)(i >> 5) * tparts * 32 + (i & 31);
            stride = 32;
        }
        for (int q = pl; q < nparts; q += 8) s += in[(size_t)q * stride + base];
    }
    sm[pl][o] = s;
    __syncthreads();
    if (pl == 0 && i < n) {
        float t = poison;
        for (int k = 0; k < 8; ++k) t += sm[k][o];
        out[i] = t;
    }
}

// All four outputs of the fused joint's backward in ONE launch (round 5; four launches before): blocks [0, kHookBlocks) d enc_proj,
// [kHookBlocks, 2 kHookBlocks) d pred_proj, then ceil(J V / 32) blocks of dW2, then one block of db2.  `single`: the partials
// come from joint_bwd_kernel (jp.nblk workgroups per J group), else from the wide joint's two-kernel backward (nC / nW / nDb
// zero-filled partials).  A backward-only whole-network call on a workspace whose state word is not the forward's (need_state)
// returns NaN in every output, loudly, instead of numbers from someone else's tables.
__global__ __launch_bounds__(256) void joint_reduce_kernel(const JointParams jp, const int single, const int nC, const int nW,
                                                           const int nDb, unsigned *dmax_enc, unsigned *dmax_pred) {
    const LossParams &p = jp.lp;
    const float poison = (jp.need_state && jp.tflag[3] != 1.0f) ? NAN : 0.f;
    const unsigned blk = blockIdx.x;
    const unsigned nWblk = (unsigned)(jp.J * p.V + 31) / 32u;
    if (blk < (unsigned)kHookBlocks) {
        reduce_enc_body(jp.d_enc_proj, jp.dApart, jp.n_ut, p, jp.J, blk, kHookBlocks, dmax_enc ? dmax_enc + blk : nullptr, poison,
                        single ? jp.live8 : nullptr);  // (single: the partials are joint_bwd_kernel's)
    } else if (blk < 2u * kHookBlocks) {
        const unsigned k = blk - kHookBlocks;
        if (single)
            reduce_pred_body(jp.d_pred_proj, jp.dCpart, jp, k, kHookBlocks, dmax_pred ? dmax_pred + k : nullptr, poison);
        else
            reduce_partials_body(jp.d_pred_proj, jp.dCpart, nC, (size_t)p.B * p.U * jp.J, k, kHookBlocks,
                                 dmax_pred ? dmax_pred + k : nullptr, poison);
    } else if (blk < 2u * kHookBlocks + nWblk) {
        reduce_small_body<true>(jp.dW2, jp.dWpart, single ? jp.nblk : nW, single ? kBwdMaxBlocks : nW, jp.J * p.V, jp.J, p.V,
                                blk - 2u * kHookBlocks, poison);
    } else {  // (one or two blocks of 32 symbols)
        reduce_small_body<false>(jp.db2, jp.dbpart, single ? 2 * jp.nblk : nDb, single ? 2 * kBwdMaxBlocks : nDb, p.V, jp.J, p.V,
                                 blk - 2u * kHookBlocks - nWblk, poison);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct JointLayout {
    WsLayout w;
    size_t dl, dlg, rec, reclab, xbl, live8, plan, dApart, dCpart, dWpart, dbpart, expE, expP, tflag, W2s, pstat, total;
    int n_ut, TR, n_tr, TS, n_ts, nC, nW, nDb, nPstat, VT;
    bool wide;  // 640 < J <= 704: W2 streams through the LDS (joint_phase1s_kernel), two-kernel backward, log-domain sweeps
};

static JointLayout make_joint_layout(int T, int U, int B, int J, int V) {
    JointLayout L;
    L.w = make_layout(T, U, B);
    L.VT = (V + 31) / 32;  // vocabulary tiles of 32 symbols: 2 for 32 < V <= 64 (every [.][32] array below then holds two)
    L.wide = (size_t)J * 256 > 160 * 1024;  // the forward's two resident tables no longer fit the LDS together
    L.n_ut = (U + 31) / 32;
    L.TR = 40;  // rows per phase-1 block (amortises the 128*J-byte C^T tile)
    L.n_tr = (T + L.TR - 1) / L.TR;
    L.n_ts = (T >= 256) ? 4 : 1;  // row splits of phase 2 (parallelism vs partial-buffer count)
    L.TS = ((T + L.n_ts - 1) / L.n_ts + 3) / 4 * 4;
    L.n_ts = (T + L.TS - 1) / L.TS;
    size_t off = L.w.total;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    L.dl = take((size_t)B * T * U * 32 * L.VT * sizeof(float));
    L.dlg = take(L.wide ? (size_t)B * T * U * 32 * sizeof(float) : 0);  // dlogits of the wide joint's two-kernel backward
    L.rec = take((size_t)B * T * U * sizeof(float4));
    L.reclab = take((size_t)B * T * U * sizeof(int));
    L.xbl = take((size_t)B * T * U * sizeof(float2));
    L.live8 = take((size_t)B * L.n_ut * 4 * ((T + 31) / 32));  // one bit per (row, u-tile): joint_cellrec_kernel -> joint_bwd_kernel
    L.plan = take((size_t)(6 + kBwdMaxBlocks + (size_t)B * L.n_ut * ((T + kBwdRows - 1) / kBwdRows)) * sizeof(int));  // joint_rowplan_kernel
    L.dApart = take((size_t)L.n_ut * B * T * J * sizeof(float));
    // partial buffers: what joint_bwd_kernel writes (kBwdSlots slabs of d pred_proj, one dW2 / two db2 partials per workgroup of a
    // J group; no zero-fill: the reduction knows what exists), or what the wide joint's two-kernel backward writes (zero-filled)
    const size_t gdl = ((size_t)B * T * U + 256 * kDlChunks - 1) / (256 * kDlChunks);
    L.nC = L.wide ? L.n_ts : kBwdSlots;
    L.nW = L.wide ? B * L.n_ut * L.n_ts : kBwdMaxBlocks;
    L.nDb = L.wide ? (int)gdl : 2 * kBwdMaxBlocks;
    L.dCpart = take((size_t)L.nC * B * U * J * sizeof(float));
    L.dWpart = take((size_t)L.VT * L.nW * J * 32 * sizeof(float));
    L.dbpart = take((size_t)L.VT * L.nDb * 32 * sizeof(float));
    L.expE = take((size_t)B * T * J * sizeof(float));
    L.expP = take((size_t)B * U * J * sizeof(float));
    L.tflag = take(256 + (size_t)kMaxVT * 256);  // flags (64 words, zeroed per call) + the b2s tables (64 words per vocabulary tile)
    L.W2s = take((size_t)L.VT * J * 32 * 2 * sizeof(jf16));
    L.nPstat = L.n_ut * ((T + 1) / 2);  // decay statistic of the forward kernel: one slot per (u-tile, row pair) (fwd_put_stat)
    L.pstat = take((size_t)B * L.nPstat * sizeof(float2));
    L.total = off;
    return L;
}

static bool joint_supported(int J, int V) {
    // J <= 704: the streaming forward (640 < J) keeps the whole C^T tile (128 J bytes), a row of enc_proj per wave (32 J) and
    // 50 KB of staging in LDS: 162,816 of the 163,840 bytes at J = 704.
    // V <= 128 (round 5; 32 before): up to four vocabulary tiles of 32 symbols, each tile one pass of the forward / backward
    // kernels (J <= 640: the wide joint's kernels keep one tile).
    return V >= 1 && V <= (J <= 640 ? 32 * kMaxVT : 32) && J >= 64 && (J % 64) == 0 && J <= 704;
}

// joint_f16_kernels.hip (large vocabularies on the f16 MFMA units)
bool joint_f16_supported(int J, int V);
hipError_t joint_f16_workspace_bytes(int T, int U, int B, int J, int V, size_t *bytes);
hipError_t launch_joint_loss_f16(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                                 const int *labels, const int *label_lengths, const int *input_lengths,
                                 const float *cost_scale, int J, int V, int B, int T, int U, int blank, float *costs,
                                 float *d_enc_proj, float *d_pred_proj, float *dW2, float *db2, int phases,
                                 void *workspace, hipStream_t s, const JointHooks *hooks);

// d enc_proj from its [n_ut][B][T][J] partial rows (reduce_enc_kernel); shared with the f16 joint
hipError_t launch_reduce_enc(float *out, const float *in, int n_ut, const LossParams &lp, int J, hipStream_t s, unsigned *blockmax,
                             const uint8_t *live8) {
    if ((unsigned long long)lp.B * lp.T * J >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit element indices in the kernel
    hipLaunchKernelGGL(reduce_enc_kernel, dim3(kHookBlocks), dim3(256), 0, s, out, in, n_ut, lp, J, blockmax, live8);
    return hipGetLastError();
}

// The four reductions behind the f16 joint's backward in ONE launch (round 6; four launches before, 23 us + a launch gap apiece at
// the small end): blocks [0, kHookBlocks) d enc_proj from its u-tile partial rows, [kHookBlocks, 2 kHookBlocks) d pred_proj from its
// row-strip slabs, then gW blocks of dW2 and gB blocks of db2 from K4's range partials.  The same bodies, the same association per
// element as the single launches: bit-identical results.
__global__ __launch_bounds__(256) void reduce_f16_backward_kernel(float *d_enc, const float *dApart, const int n_ut, const LossParams p, const int J,
                                                                  unsigned *bm_enc, const uint8_t *live8, float *d_pred, const float *dCpart,
                                                                  const int nC, unsigned *bm_pred, float *dW2, const float *dWpart, float *db2,
                                                                  const float *dbpart, const int nR, const int V, const unsigned gW, const unsigned gB) {
    const unsigned blk = blockIdx.x, H = (unsigned)kHookBlocks;
    if (blk < H) reduce_enc_body(d_enc, dApart, n_ut, p, J, blk, H, bm_enc ? bm_enc + blk : nullptr, 0.f, live8);
    else if (blk < 2u * H) reduce_partials_body(d_pred, dCpart, nC, (size_t)p.B * p.U * J, blk - H, H, bm_pred ? bm_pred + (blk - H) : nullptr, 0.f);
    else if (blk < 2u * H + gW) reduce_partials_body(dW2, dWpart, nR, (size_t)J * V, blk - 2u * H, gW, nullptr, 0.f);
    else reduce_partials_body(db2, dbpart, nR, (size_t)V, blk - 2u * H - gW, gB, nullptr, 0.f);
}

hipError_t launch_reduce_f16_backward(float *d_enc, const float *dApart, int n_ut, const LossParams &lp, int J, unsigned *bm_enc, const uint8_t *live8,
                                      float *d_pred, const float *dCpart, int nC, unsigned *bm_pred, float *dW2, const float *dWpart, float *db2,
                                      const float *dbpart, int nR, int V, hipStream_t s) {
    if ((unsigned long long)lp.B * lp.T * J >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit element indices in reduce_enc_body
    const size_t nW = ((size_t)J * V + 255) / 256, nB = ((size_t)V + 255) / 256;
    const unsigned gW = (unsigned)(nW < 1024 ? nW : 1024), gB = (unsigned)(nB < 1024 ? nB : 1024);
    hipLaunchKernelGGL(reduce_f16_backward_kernel, dim3(2u * kHookBlocks + gW + gB), dim3(256), 0, s, d_enc, dApart, n_ut, lp, J, bm_enc, live8, d_pred,
                       dCpart, nC, bm_pred, dW2, dWpart, db2, dbpart, nR, V, gW, gB);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(float *out, const float *in, int nparts, size_t n, hipStream_t s, unsigned *blockmax) {
    // (a consumer of `blockmax` reads kHookBlocks entries: the grid is then exactly that, whatever n is)
    const unsigned grid = blockmax ? (unsigned)kHookBlocks : (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid), dim3(256), 0, s, out, in, nparts, n, blockmax);
    return hipGetLastError();
}

// {lattice rows (x u-tiles) the last backward on this workspace visited, rows inside the utterances}: device -> host, synchronous
hipError_t joint_backward_rows(void *workspace, int T, int U, int B, int J, int V, int rows[2], hipStream_t s) {
    rows[0] = rows[1] = -1;
    if (!joint_supported(J, V)) return hipErrorInvalidValue;
    const JointLayout L = make_joint_layout(T, U, B, J, V);
    if (L.wide) return hipSuccess;  // (the wide joint's two-kernel backward visits everything)
    const size_t n_items = (size_t)B * L.n_ut * ((T + kBwdRows - 1) / kBwdRows);
    int h[3] = {-1, -1, 0};
    hipError_t e = hipMemcpyAsync(h, (char *)workspace + L.plan + (3 + kBwdMaxBlocks + n_items) * sizeof(int), 3 * sizeof(int),
                                  hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    // the row plan stamps its counters with the shape it ran on: a workspace no f32-grade backward of THIS shape has written
    // (fresh, or used by another shape / arithmetic type since) answers {-1, -1}, not whatever the words hold
    if (e == hipSuccess && h[2] == plan_stamp(T, U, B, J, V)) rows[0] = h[0], rows[1] = h[1];
    return e;
}

// where the fused joint (joint_dtype 0) keeps the e^{2x} tables and its flags: for a caller that fills them itself (JointHooks)
hipError_t joint_aux_pointers(void *workspace, int T, int U, int B, int J, int V, float **expE, float **expP, float **tflag) {
    if (!joint_supported(J, V) || sweep_K(U) == 0) return hipErrorInvalidValue;
    const JointLayout L = make_joint_layout(T, U, B, J, V);
    char *ws = (char *)workspace;
    *expE = (float *)(ws + L.expE), *expP = (float *)(ws + L.expP), *tflag = (float *)(ws + L.tflag);
    return hipSuccess;
}

// What every forward call of the f32-grade joint needs in front of its kernels, in ONE launch: the edge array pre-filled
// (probability zero for the linear lattice; the finite "log zero" for the wide joint's log-domain sweeps) and the flag words
// zeroed.  A caller that raises the table-range flag itself (the dense layer's epilogue, JointHooks::prep_mode 1) calls this
// in front of its own launches and says so (JointHooks::prefilled).
hipError_t launch_joint_prefill(void *workspace, int T, int U, int B, int J, int V, hipStream_t s) {
    if (!joint_supported(J, V) || sweep_K(U) == 0) return hipErrorInvalidValue;
    const JointLayout L = make_joint_layout(T, U, B, J, V);
    char *ws = (char *)workspace;
    return launch_fill2(ws + L.w.W, L.wide ? kFillByte : 0, L.w.A - L.w.W, ws + L.tflag, 0, 256, s);
}

// does the joint of the requested arithmetic take this shape?  (checked before anything is enqueued: the workspace layout is
// chosen by the vocabulary -- the two domains are disjoint -- and must be the one the requested kernels expect)
bool joint_dtype_supported(int joint_dtype, int J, int V) {
    if (joint_dtype == 0) return joint_supported(J, V);
    if (joint_dtype == 1) return joint_f16_supported(J, V);
    return false;
}

// Workspace of the fused joint.  joint_dtype 0 / 1: that path's layout (invalid when it does not take the shape); -1: the larger of
// the two for shapes both take (V = 128 with J <= 640, round 5) -- what a caller that has not chosen yet must allocate, and the
// offset behind which the whole-network entry points keep their own arrays whatever the arithmetic.
hipError_t joint_workspace_bytes(int T, int U, int B, int J, int V, int joint_dtype, size_t *bytes) {
    size_t n32 = 0, n16 = 0;
    const bool ok32 = joint_dtype != 1 && joint_supported(J, V) && sweep_K(U) != 0;
    const bool ok16 = joint_dtype != 0 && joint_f16_supported(J, V) && joint_f16_workspace_bytes(T, U, B, J, V, &n16) == hipSuccess;
    if (ok32) n32 = make_joint_layout(T, U, B, J, V).total;
    if (!ok32 && !ok16) return hipErrorInvalidValue;
    *bytes = n32 > n16 ? n32 : n16;
    return hipSuccess;
}

// provided by rnnt_entrypoint.hip
bool fill_loss_params(LossParams &p, const float *acts, float *grads, const int *labels, const int *label_lengths,
                      const int *input_lengths, const float *cost_scale, int V, int B, float *costs, void *workspace,
                      int maxT, int maxU, int blank);

// compute units of the current device (persistent kernels launch one workgroup per CU)
static int device_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// logits[c][0..V) <- the parked tiles dl[c][0..Vp) (bias included; Vp = 32 or 64)
__global__ __launch_bounds__(256) void joint_logits_copy_kernel(float *out, const float *dl, const uint32_t cells, const int V, const int Vp) {
    const size_t n = (size_t)cells * (size_t)V;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t c = i / (size_t)V;
        out[i] = dl[c * Vp + (i - c * (size_t)V)];
    }
}

// everything of JointParams that depends only on the workspace layout
static void joint_bind(JointParams &jp, const JointLayout &L, char *ws) {
    jp.dl = (float *)(ws + L.dl);
    jp.dlg = (float *)(ws + L.dlg);
    jp.rec = (float4 *)(ws + L.rec);
    jp.reclab = (int *)(ws + L.reclab);
    jp.xbl = (float2 *)(ws + L.xbl);
    jp.live8 = (uint8_t *)(ws + L.live8);
    jp.plan = (int *)(ws + L.plan);
    jp.dApart = (float *)(ws + L.dApart);
    jp.dCpart = (float *)(ws + L.dCpart);
    jp.dWpart = (float *)(ws + L.dWpart);
    jp.dbpart = (float *)(ws + L.dbpart);
    jp.expE = (float *)(ws + L.expE), jp.expP = (float *)(ws + L.expP), jp.tflag = (float *)(ws + L.tflag);
    jp.b2s = jp.tflag + 64;
    jp.W2s = (jf16 *)(ws + L.W2s);
    jp.n_ut = L.n_ut, jp.TR = L.TR, jp.n_tr = L.n_tr, jp.TS = L.TS, jp.n_ts = L.n_ts;
    jp.VT = L.VT, jp.vt = 0;
    jp.nblk = 0, jp.need_state = 0;
    // the forward kernel's decay statistic (instead of the lsm launch's per-patch slots)
    jp.lp.pstat = (float2 *)(ws + L.pstat), jp.lp.nPstat = L.nPstat, jp.lp.pstatStride = 1;
}

static hipError_t launch_joint_fwd(const JointParams &jp0, const JointLayout &L, int B, int T, hipStream_t s) {
    const int J = jp0.J;
    hipError_t e;
    if (!L.wide) {
        const size_t shm_fwd = (size_t)J * 256;  // Ct tile + W2 fragment image, both resident
        if ((e = set_lds(joint_fwd_kernel, shm_fwd)) != hipSuccess) return e;
        const int n_items = ((T + kFwdRows - 1) / kFwdRows) * B * L.n_ut;
        const int ncu = device_cu_count();
        JointParams jp = jp0;
        for (jp.vt = 0; jp.vt < L.VT; ++jp.vt)  // one pass per vocabulary tile; the last one sees every logit of a cell (epilogue MODE 2)
            hipLaunchKernelGGL(joint_fwd_kernel, dim3(n_items < ncu ? n_items : ncu), dim3(kFwdWaves * 64), shm_fwd, s, jp);
    } else {
        const JointParams &jp = jp0;  // J > 640: the two tables do not fit the LDS together; W2 streams through it in chunks instead
        const size_t shm1s = (size_t)J * 32 * sizeof(float) + 2 * 8192 + (kP1Waves * (size_t)J + kP1Waves * 32 * kStagePad) * sizeof(float);
        if ((e = set_lds(joint_phase1s_kernel, shm1s)) != hipSuccess) return e;
        hipLaunchKernelGGL(joint_phase1s_kernel, dim3((unsigned)B * L.n_ut * L.n_tr), dim3(kP1Waves * 64), shm1s, s, jp);
    }
    return hipGetLastError();
}

// Joint logits only (decoding: utils/decoding.py:6-18): the forward kernel of launch_joint_loss -- the same tables, the same
// split-precision products -- with every lattice cell live; what it parks in the workspace is copied out as [B, T, U, V].
hipError_t launch_joint_logits(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2, int J, int V,
                               int B, int T, int U, float *logits, void *workspace, hipStream_t s) {
    if (!joint_supported(J, V) || sweep_K(U) == 0) return hipErrorInvalidValue;
    if (((uintptr_t)enc_proj & 15) || ((uintptr_t)pred_proj & 15)) return hipErrorInvalidValue;
    if ((unsigned long long)B * T * J >= (1ull << 32) || (unsigned long long)B * U * J >= (1ull << 32)) return hipErrorInvalidValue;
    const JointLayout L = make_joint_layout(T, U, B, J, V);
    char *ws = (char *)workspace;
    // lengths and labels of the "everything is live" lattice sit in workspace regions the forward kernels do not touch
    int *il = (int *)(ws + L.rec), *ll = il + B, *labels = (int *)(ws + L.reclab);
    JointParams jp;
    if (!fill_loss_params(jp.lp, nullptr, nullptr, labels, ll, il, nullptr, V, B, nullptr, workspace, T, U, 0))
        return hipErrorInvalidValue;
    jp.enc_proj = enc_proj, jp.pred_proj = pred_proj, jp.W2 = W2, jp.b2 = b2;
    jp.J = J;
    joint_bind(jp, L, ws);
    jp.d_enc_proj = jp.d_pred_proj = jp.dW2 = jp.db2 = nullptr;
#ifdef JH_TRACE
    jp.trace = nullptr;
#endif
    jp.logits_only = 1, jp.tables_ready = 0;
    hipError_t e;
    if ((e = launch_fill2(jp.tflag, 0, 256, labels, 0, U > 1 ? (size_t)B * (U - 1) * sizeof(int) : 0, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(joint_prep_kernel, dim3(1024), dim3(256), 0, s, jp);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = launch_joint_fwd(jp, L, B, T, s)) != hipSuccess) return e;
    const size_t n = (size_t)jp.lp.cells * V;
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(joint_logits_copy_kernel, dim3(grid), dim3(256), 0, s, logits, jp.dl, jp.lp.cells, V, 32 * L.VT);
    return hipGetLastError();
}

// the hand-back launch of the fused joint (joint_redo_kernel) with the chunk geometry of the utterance's sweeps
template <int K, int G>
static hipError_t launch_joint_redo_k(const JointParams &jp, const LossParams &q, const bool want_rec, hipStream_t s) {
    constexpr int NB = ((size_t)4 * G * 2 * 64 * K * sizeof(float) + 16 <= 128 * 1024) ? 4 : 3;
    constexpr size_t shm = (size_t)NB * G * 2 * 64 * K * sizeof(float) + 16;
    hipError_t e = set_lds(joint_redo_kernel<K, G, NB>, shm);
    if (e != hipSuccess) return e;
    const int team = redo_team_size(jp.lp.nb, jp.lp.T, jp.lp.U, device_cu_count());
    hipLaunchKernelGGL((joint_redo_kernel<K, G, NB>), dim3(jp.lp.nb * team), dim3(kRedoThreads), shm, s, jp, q, want_rec ? 1 : 0, team);
    return hipGetLastError();
}
static hipError_t launch_joint_redo(const JointParams &jp, const bool want_rec, hipStream_t s) {
    // the loss parameters of the redo: the parked tile [cells][32] as the logits (pad symbols hold -1e30: probability zero)
    LossParams q = jp.lp;
    q.acts = jp.dl, q.grads = nullptr, q.V = 32 * jp.VT;
    q.divV = make_fastdiv((uint32_t)q.V);
    switch (sweep_K(jp.lp.U)) {
        case 1: return launch_joint_redo_k<1, 16>(jp, q, want_rec, s);
        case 2: return launch_joint_redo_k<2, 16>(jp, q, want_rec, s);
        case 3: return launch_joint_redo_k<3, 16>(jp, q, want_rec, s);
        case 4: return launch_joint_redo_k<4, 16>(jp, q, want_rec, s);
        case 6: return launch_joint_redo_k<6, 8>(jp, q, want_rec, s);
        case 8: return launch_joint_redo_k<8, 8>(jp, q, want_rec, s);
        case 12: return launch_joint_redo_k<12, 4>(jp, q, want_rec, s);
        case 16: return launch_joint_redo_k<16, 4>(jp, q, want_rec, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_joint_loss(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                             const int *labels, const int *label_lengths, const int *input_lengths,
                             const float *cost_scale, int J, int V, int B, int T, int U, int blank, float *costs,
                             float *d_enc_proj, float *d_pred_proj, float *dW2, float *db2, int joint_dtype,
                             int phases, void *workspace, hipStream_t s, const JointHooks *hooks) {
    // phases: bit 0 = forward (costs + lattice state in the workspace), bit 1 = backward (needs that state),
    // bit 2 = a backward-only call will follow this forward-only one (the f16 joint parks its softmax numerators for it),
    // bit 3 = RNNT_VISIT_ALL: the backward visits every lattice row (no occupancy floor)
    if (joint_dtype == 1)
        return launch_joint_loss_f16(enc_proj, pred_proj, W2, b2, labels, label_lengths, input_lengths, cost_scale, J, V,
                                     B, T, U, blank, costs, d_enc_proj, d_pred_proj, dW2, db2, phases, workspace, s, hooks);
    if (!joint_supported(J, V) || joint_dtype != 0 || sweep_K(U) == 0) return hipErrorInvalidValue;
    if (((uintptr_t)enc_proj & 15) || ((uintptr_t)pred_proj & 15)) return hipErrorInvalidValue;
    // the reductions over the [B][T][J] / [B][U][J] arrays index with 32 bits (B*T*U < 2^31 alone does not bound B*T*J)
    if ((unsigned long long)B * T * J >= (1ull << 32) || (unsigned long long)B * U * J >= (1ull << 32)) return hipErrorInvalidValue;
    const JointLayout L = make_joint_layout(T, U, B, J, V);
    JointParams jp;
    if (!fill_loss_params(jp.lp, nullptr, nullptr, labels, label_lengths, input_lengths, cost_scale, V, B, costs,
                          workspace, T, U, blank))
        return hipErrorInvalidValue;
    char *ws = (char *)workspace;
    jp.enc_proj = enc_proj, jp.pred_proj = pred_proj, jp.W2 = W2, jp.b2 = b2;
    jp.J = J;
    joint_bind(jp, L, ws);
    jp.d_enc_proj = d_enc_proj, jp.d_pred_proj = d_pred_proj, jp.dW2 = dW2, jp.db2 = db2;
#ifdef JH_TRACE
    static long long *trace_dev = nullptr;
    const size_t trace_bytes = 2048 * sizeof(long long);
    if (!trace_dev) (void)hipMalloc(&trace_dev, trace_bytes);
    (void)hipMemsetAsync(trace_dev, 0, trace_bytes, s);
    jp.trace = trace_dev;
#endif
    jp.logits_only = 0;
    jp.visit_all = (phases & 8) ? 1 : 0;
    const int prep_mode = hooks ? hooks->prep_mode : 0;
    jp.tables_ready = prep_mode == 1;
    jp.need_state = prep_mode == 2;
    const bool fwd = (phases & 1) != 0, bwd = (phases & 2) != 0 && d_enc_proj != nullptr;

    hipError_t e;
    // flag words (rebuilt by whichever phase runs the prep kernel) and, for a forward, the edge array's pre-fill: one launch
    if (!(hooks && hooks->prefilled)) {
        if (fwd)
            e = launch_fill2(jp.lp.W, L.wide ? kFillByte : 0, L.w.A - L.w.W, jp.tflag, 0, prep_mode == 0 ? 256 : 0, s);
        else
            e = prep_mode == 0 ? launch_fill(jp.tflag, 0, 256, s) : hipSuccess;
        if (e != hipSuccess) return e;
    }
    // tanh tables + W2 images (rebuilt by whichever phase runs: the projections may have changed -- unless the caller vouches
    // for the workspace, JointHooks::prep_mode)
    if (prep_mode != 2) hipLaunchKernelGGL(joint_prep_kernel, dim3(jp.tables_ready ? 64 : 1024), dim3(256), 0, s, jp);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (fwd) {
        if ((e = launch_joint_fwd(jp, L, B, T, s)) != hipSuccess) return e;
        if (!L.wide) {
            // the linear-domain lattice of the loss op (rnnt_lin.h: multiply / add sweeps on mantissas x 2^frame, 43 us at
            // B32 T600 U150 where the float64 log-domain recurrence this path used before took 138), with its hand-back:
            // in a forward-only call right here (costs), otherwise behind the backward's per-cell pass (certificate)
            if ((e = launch_sweeps_lin(jp.lp, s)) != hipSuccess) return e;
            if (!bwd && (e = launch_joint_redo(jp, false, s)) != hipSuccess) return e;
        } else {
            LossParams lpp = jp.lp;
            lpp.precise = 1;  // f32-GRADE joint: the log-domain recurrence in float64 (rnnt_sweep.h alpha_sweep_pr)
            if ((e = launch_sweeps(lpp, s)) != hipSuccess) return e;
        }
    }
    if (!bwd) return hipSuccess;  // score only

    int nC = 0, nW = 0, nDb = 0;
    if (!L.wide) {
        const int n_groups = bwd_groups(J);
        const int n_cons = J / 32 / n_groups;
        const int n_items = ((T + kBwdRows - 1) / kBwdRows) * B * L.n_ut;
        int nblk = device_cu_count() / n_groups;
        if (nblk > n_items) nblk = n_items;
        if (nblk > kBwdMaxBlocks) nblk = kBwdMaxBlocks;
        if (nblk < 1) nblk = 1;
        jp.nblk = nblk;
        hipLaunchKernelGGL(joint_cellrec_kernel, dim3((unsigned)B * L.n_ut * ((T + kRecRows - 1) / kRecRows)), dim3(256), 0, s, jp);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = launch_joint_redo(jp, true, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(joint_rowplan_kernel, dim3(1), dim3(kPlanThreads), 0, s, jp);  // (after the hand-back: it turns whole utterances on)
        if ((e = hipGetLastError()) != hipSuccess) return e;
        const size_t shm_bwd = (size_t)kBwdRing * kBwdSlotBytes + 2 * kBwdRing * sizeof(int);
        if ((e = set_lds(joint_bwd_kernel, shm_bwd)) != hipSuccess) return e;
        for (jp.vt = 0; jp.vt < L.VT; ++jp.vt)  // one pass per vocabulary tile (a later one adds its d enc_proj / d pred_proj partials to the first's)
            hipLaunchKernelGGL(joint_bwd_kernel, dim3(nblk * n_groups), dim3((n_cons + 2) * 64), shm_bwd, s, jp);
        jp.vt = 0;
    } else {
        // the wide joint: dlogits by their own kernel, then the block-per-tile products; partial buffers zero-filled (rows /
        // workgroups that path does not write must read as zero; the d enc_proj partials need none)
        if (launch_fill(jp.dCpart, 0, (L.dbpart - L.dCpart) + (size_t)L.nDb * 32 * sizeof(float), s) != hipSuccess) return hipErrorUnknown;
        const unsigned gdl = (jp.lp.cells + 256u * kDlChunks - 1u) / (256u * kDlChunks);
        hipLaunchKernelGGL(joint_dl_kernel, dim3(gdl), dim3(256), 0, s, jp);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        const unsigned g2 = (unsigned)B * L.n_ut * (J / 64) * L.n_ts;
        const size_t shm2s = ((size_t)64 * 36 + 4 * 32 * 36 + 4 * 32 * kStagePad) * sizeof(float) + 8192;
        hipLaunchKernelGGL(joint_phase2s_kernel, dim3(g2), dim3(256), shm2s, s, jp);
        nC = L.n_ts, nW = B * L.n_ut * L.n_ts, nDb = (int)gdl;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const unsigned nWblk = (unsigned)(J * V + 31) / 32u;
    hipLaunchKernelGGL(joint_reduce_kernel, dim3(2u * kHookBlocks + nWblk + (unsigned)L.VT), dim3(256), 0, s, jp, L.wide ? 0 : 1, nC, nW, nDb,
                       hooks ? hooks->dmax_enc : (unsigned *)nullptr, hooks ? hooks->dmax_pred : (unsigned *)nullptr);
#ifdef JH_TRACE
    {
        (void)hipStreamSynchronize(s);
        static long long h[2048];
        (void)hipMemcpy(h, trace_dev, trace_bytes, hipMemcpyDeviceToHost);
        const char *path = getenv("JH_TRACE_FILE32");
        if (FILE *f = fopen(path ? path : "/tmp/j32_trace.bin", "wb")) {
            fwrite(h, 1, trace_bytes, f);
            fclose(f);
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace rnnt
