// joint_f16_kernels.hip -- large-vocabulary joint network fused with the transducer loss on the f16 MFMA units
// (gfx950, v_mfma_f32_32x32x16_f16).  BASELINE config 5 ("fp16 joint MFMA, fp32 lattice"): V = 1024, J = 640.
//
// Same path as joint_kernels.hip (model.py:158-166 + autodiff through it),
// but the [cells x V] logits no longer fit a per-cell register tile, so the work is three GEMM-shaped kernels around the
// unchanged alpha/beta sweeps.  Nothing of size [cells x J] is ever stored; the only [cells x V]
// array is binary16 (2 B per logit instead of the 4+4 B of the unfused path): the softmax numerators the forward pass parks,
// turned in place into the loss gradient w.r.t. the logits by the backward pass.
//
//   prep      W2 -> binary16 in two layouts (MFMA-fragment-packed W2^T, row-major W2); power-of-two dlogits scale;
//             tables e^{2 enc_proj}, e^{2 pred_proj}: tanh(a+c) = 1 - 2/(1 + e^{2a} e^{2c}) costs one reciprocal per
//             (cell, joint unit) instead of an exponential and a reciprocal (the kernels are VALU-bound on it)
//   K1 logits (jh_logits_kernel<KS, MODE, B2LDS>)  logits^T tile = W2^T . h^T with h = tanh(enc_proj_t + pred_proj_u) built
//             straight into the B-operand registers (a wave owns 32 lattice cells of one row t, h never leaves the
//             register file); W2^T streams through LDS by LDS-DMA, 32 vocabulary rows per step, shared by 8 waves.
//             In the transposed product a LANE owns a cell and its registers run over the vocabulary, so the
//             log-softmax is an in-register online reduction.  Out: lse, lattice edge weights W (-> sweeps), the
//             blank / label logits of every cell, and (MODE 1, when a backward pass follows) the PARKED softmax
//             numerators 2^(y - R) in binary16, R = the integer at or above the largest y of the cell's 32-symbol chunk.
//   K3 dh     (jh_dhx_kernel<J/128>, round 6)  the pass over the parked values AND dh = dl . W2^T in one kernel: a workgroup owns
//             128 cells x ALL J units, so every dl row is fetched, multiplied back with its cell's factor (dl = binary16(S scale
//             2^(R + c0) parked), blank / label columns from the f32 edge logits) and written back for K4 exactly ONCE, in the
//             loader of the product; epilogue dz = dh (1 - h^2), sum_u -> d enc_proj partials, sum_t -> d pred_proj.
//             (Rounds 3-5: a separate 29.5 GB streaming pass K2 + a dh kernel whose five J-tile workgroups each fetched dl.
//             Rounds 1-2 ran the forward product a second time -- that kernel, MODE 2 of jh_logits_kernel, still serves a
//             second backward call over one forward.)
//   K4 dW2    (jh_dw_kernel)  dW2 = h^T . dl, split over ranges of cells; h^T is generated in A-fragment layout,
//             dl rows are DMA'd row-major and read TRANSPOSED with ds_read_b64_tr_b16; db2 rides along (v_dot2).
//   rows      (jh_rowbits_kernel, jh_order_kernel, round 6)  which lattice rows the backward visits at all (a row whose cells all
//             have an occupancy alpha beta / L below 2^-50 contributes nothing a binary32 sum can see; RNNT_VISIT_ALL: every row),
//             and the order K3's strips / K4's units are dealt in (by visited rows: a function of the data, not of timing).
//
// MFMA fragment layouts used (checked on hardware by scripts/probes/probe_f16.hip):
//   A: lane l holds A[i = l&31][k = 8*(l>>5) + 0..7];  B: lane l holds B[k = 8*(l>>5) + 0..7][n = l&31];
//   C/D: lane l register r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//   ds_read_b64_tr_b16: lane p of 16-lane group g supplies &X[k0 + (p>>2)][n0 + 4*(p&3)] and receives X[k0+0..3][n0+p].
#include "rnnt_common.h"
#include "rnnt_cell.h"

#include <math.h>
#ifdef JH_TRACE
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#endif

namespace rnnt {

typedef _Float16 f16;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;
typedef __attribute__((address_space(3))) s4 lds_s4;

__device__ __forceinline__ float hex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hlg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float htanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + hex2(x * 2.8853900817779268f));
}
// tanh(a + c) from the tabulated factors ea = e^{2a}, ec = e^{2c}: one multiply-add, one reciprocal, one multiply-add.
// Exact to ~1e-7 absolute while |a|, |c| <= kExpTabLimit (both factors normal f32 numbers; an overflowing product gives +1,
// an underflowing one -1, as tanh does).  Beyond that limit the prep kernel raises a flag and the kernels use htanh(a + c).
__device__ __forceinline__ float htanh2(float ea, float ec) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fmaf(ea, ec, 1.0f));
}
__device__ __forceinline__ constexpr int cdrow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float dot8(const h8 a, const h8 b, float c) {
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c, false);
    return c;
}
// An opaque copy: what is computed from it cannot be hoisted out of the region it is made in (values derived from the thread
// index that a kernel's epilogue needs must not stay in registers through its main loop).
__device__ __forceinline__ int launder(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float lane_f32(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
// XCD-aware bijective remap: XCD (blockIdx % 8) owns a contiguous range of logical work items
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t xcd = bid & 7u, idx = bid >> 3, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
}

// LDS-DMA with a SCALAR base and a 32-bit lane offset, issued as inline assembly (jh_dhx_kernel): the builtin made the compiler
// form 64-bit lane addresses, hoist them out of the loop, spill them, and wait vmcnt(0) behind every reload.  `lds`: byte address in
// LDS (wave-uniform); lane l lands at lds + SIZE l.  The callers order the DMA with s_waitcnt vmcnt / barriers themselves.
__device__ __forceinline__ void lds_dma16_s(const void *base, uint32_t lane_off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds) : "m0");
}
__device__ __forceinline__ void lds_dma4_s(const void *base, uint32_t lane_off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane_off), "s"(base), "s"(lds) : "m0");
}
__device__ __forceinline__ uint32_t lds_addr(const void *q) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)q; }

constexpr int kDlScaleLog2 = 14;   // |dlogits * S| <= 2^14 before the binary16 rounding
constexpr int kStageStride = 36;   // halfs per row of the K1/K2 staging tile: one chunk of 32 columns + 8 bytes.  18 dwords per row:
                                   // the 32 rows a half-wave writes (8 bytes per lane) fall into 32 different bank pairs
                                   // (20 dwords per row: 2.5e8 conflict cycles per launch at config 5, rows n and n + 16 collide)
constexpr float kRefLimit = 30000.0f;  // chunk references are kept as int16
#ifndef JH_TQ
#define JH_TQ 128
#endif
constexpr int kTQ = JH_TQ;         // lattice rows per K4 work unit

struct JhParams {
    LossParams lp;  // lattice workspace, labels, lengths, costs, cost_scale (acts / grads unused)
    const float *enc_proj, *pred_proj, *W2, *b2;
    f16 *W2Tp;    // [V/32][J/16][2 halves][32 v][8 j]  W2^T packed so that lane l of an MFMA A fragment reads
                  // bytes [16 l, 16 l + 16) of a contiguous 1 KB block (conflict-free ds_read_b128)
    f16 *W2c;     // [V/32][J][32 v]  W2 in K-chunk-major order: the 64 J bytes K3 brings in per 32-symbol step are contiguous
    float *dCacc; // K3: running sum over a workgroup's lattice rows of its d pred_proj tile, [workgroup][wave][2][NT][64 lanes][4]
    f16 *dl;      // [cells][V]  parked softmax numerators 2^(y - ref) after K1 (park mode), dlogits * S after K2
    short *pref;  // [B T][V/32][32 n_ut]  the integer references of the parked values, one per (cell, 32-symbol chunk): row, chunk, column
    int *state;   // [0]: 1 = dl holds the parked values of the forward call with these inputs; anything else: it does not
    float *xbl;   // [cells][2]  blank / label logits, log2-scaled (x * log2 e)
    float *scal;  // [0] = S, [1] = 1/S, [2] != 0: some |enc_proj| or |pred_proj| exceeds kExpTabLimit (use htanh)
    float *expE;  // [B][T][J]  e^{2 enc_proj}
    float *expP;  // [B][U][J]  e^{2 pred_proj}
    float *b2l;   // [V]  b2 * log2 e
    float *dApart;  // [n_ut][B][T][J]
    float *dCpart;  // [n_ts][B][U][J]
    float *dWpart;  // [n_ranges][J][V]
    float *dbpart;  // [n_ranges][V]
    const f16 *zrow;  // 1 KB of zeros: stands in for dl rows beyond the tensor (u >= U) in K4
    uint8_t *live8;   // [B][n_ut][4 ceil(T/32)] one BIT per (lattice row, u-tile): the backward visits the row (jh_rowbits_kernel; the
                      // four rows of an aligned group share their bit: K3 works in iterations of four rows)
    int *rowcnt;      // [0] = (rows x u-tiles) the backward visits, [1] = those inside the utterances, [2] = shape stamp
    int *order;       // [B n_ut n_ts] K3's strips, the ones with the most visited rows first (jh_order_kernel)
    int *uorder;      // [n_units] K4's work units sorted by visited rows (descending, ties by index: deterministic); null-equivalent: identity
    int visit_all;    // RNNT_VISIT_ALL: no occupancy floor
    int J, n_ut, n_tt, n_ts, TS, n_tq, n_units, n_ranges;
    int b2_lds_off;  // K1/K2: byte offset of the bias table in LDS, -1 = read it from global memory (does not fit)
    float *logits_out;  // MODE 3 of K1 (compute_rnnt_joint_logits, decoding): f32 logits [cells][V]
    int logits_only;    // every lattice cell is wanted: the prep kernel writes full lengths + zero labels into the workspace
#ifdef JH_TRACE
    long long *trace;  // dev builds only (-DJH_TRACE): per-wave s_memtime stamps of a few workgroups of K1
#endif
};
#ifdef JH_TRACE
constexpr int kTraceSlots = 160, kTraceBlocks = 4, kTraceStride = 4096;
#define JT(slot)                                                                     \
    do {                                                                             \
        if (tr && lane == 0) tr[(slot)] = (long long)__builtin_amdgcn_s_memtime();   \
    } while (0)
#else
#define JT(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// prep: binary16 copies of W2 and the dlogits scale
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jh_prep_kernel(const JhParams jp) {
    const int J = jp.J, V = jp.lp.V;
    const size_t n = (size_t)J * V;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i / V), v = (int)(i - (size_t)j * V);
        const f16 w = (f16)jp.W2[i];
        jp.W2c[((size_t)(v >> 5) * J + j) * 32 + (v & 31)] = w;
        jp.W2Tp[((((size_t)(v >> 5) * (J >> 4) + (j >> 4)) * 2 + ((j >> 3) & 1)) * 32 + (v & 31)) * 8 + (j & 7)] = w;
    }
    {
        const LossParams &p = jp.lp;
        const size_t nE = (size_t)p.B * p.T * J, nP = (size_t)p.B * p.U * J;
        bool big = false;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nE + nP; i += (size_t)gridDim.x * 256) {
            const float x = (i < nE) ? jp.enc_proj[i] : jp.pred_proj[i - nE];
            big |= exp_tab_out_of_range(x);  // also catches NaN
            const float ex = exp_tab(x);
            if (i < nE) jp.expE[i] = ex;
            else jp.expP[i - nE] = ex;
        }
        if (__any(big) && (threadIdx.x & 63) == 0) jp.scal[2] = 1.0f;  // scal[2] is zeroed before the launch
        for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) jp.b2l[v] = jp.b2[v] * kLog2e;
    }
    if (jp.logits_only && blockIdx.x == 1) {  // (the three arrays live in workspace regions the forward kernel does not touch)
        const LossParams &p = jp.lp;
        for (int b = threadIdx.x; b < p.B; b += 256) {
            const_cast<int *>(p.input_lengths)[b] = p.T;
            const_cast<int *>(p.label_lengths)[b] = p.U - 1;
        }
        for (int i = threadIdx.x; i < p.B * (p.U - 1); i += 256) const_cast<int *>(p.labels)[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float m = 1.0f;
        if (jp.lp.cost_scale) {
            m = 0.f;
            for (int b = 0; b < jp.lp.B; ++b) m = fmaxf(m, fabsf(jp.lp.cost_scale[b]));
        }
        int e = 0;
        if (m > 0.f) {
            float fr = frexpf(m, &e);  // m = fr * 2^e, fr in [0.5, 1)
            if (fr == 0.5f) --e;       // exact power of two: ceil(log2 m) = e - 1
        }
        const float S = ldexpf(1.0f, kDlScaleLog2 - e);
        jp.scal[0] = S;
        jp.scal[1] = 1.0f / S;
    }
}

// ---------------------------------------------------------------------------------------------
// K1 / K2: workgroup = 8 waves = 8 lattice rows x 32 lattice columns; wave w owns row t0+w, lane (n, half) owns
// column u0+n and the joint units {16 ks + 8 half + 0..7} of its h row.
// LDS: W2^T chunk [2][J/16][2][32 v][8 j] (2 x 64 J bytes)  |  MODE 1, 2: staging [8 waves][32 cells][kStageStride]
// ---------------------------------------------------------------------------------------------
// MODE 0: forward (lse, edge weights, edge logits).  MODE 1: forward + PARK: the softmax numerators of every chunk are also
// written to dl as binary16, relative to the chunk's own integer reference, so that the backward pass is a streaming kernel
// (the loader of jh_dhx_kernel) instead of this product a second time.  MODE 2: the product again with the dlogits epilogue (the
// route a backward call takes when the parked values are not there any more: a second backward over one forward).
template <int KS, int MODE, bool B2LDS>
__global__ __launch_bounds__(512) void jh_logits_kernel(const JhParams jp) {
    constexpr bool BWD = MODE == 2, PARK = MODE == 1, STAGE = MODE == 1 || MODE == 2, LOGITS = MODE == 3;
    if (BWD && jp.state[0] == 1) return;  // the streaming kernel has the parked values: nothing to recompute
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LossParams &p = jp.lp;
    constexpr int J = KS * 16;
    const int V = p.V;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int chunk_bytes = 64 * J;
    char *wbuf0 = smem, *wbuf1 = smem + chunk_bytes;

    int bid = blockIdx.x;
    const int ut = bid % jp.n_ut;
    bid /= jp.n_ut;
    const int tt = bid % jp.n_tt;
    const int b = bid / jp.n_tt;
    const int u0 = ut * 32, t0 = tt * 8;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    if (t0 >= Tb || u0 >= Ub) return;  // dead tile: nothing downstream reads it
    const int t = t0 + wave;
    const bool wave_live = t < Tb;  // wave-uniform
    const int tc = min(t, Tb - 1), u = u0 + n, uc = min(u, p.U - 1);
    const bool cell_valid = wave_live && (u < Ub);
    const uint32_t c = ((uint32_t)(b * p.T + tc)) * (uint32_t)p.U + (uint32_t)uc;

#ifdef JH_TRACE
    long long *tr = nullptr;
    if (!BWD && (blockIdx.x % kTraceStride) == kTraceStride / 2 && blockIdx.x / kTraceStride < kTraceBlocks)
        tr = jp.trace + ((blockIdx.x / kTraceStride) * 8 + wave) * kTraceSlots;
#endif
    JT(0);
    // one 1 KB piece (wave-instruction) of the LDS-DMA that brings W2^T chunk vc in; a wave owns pieces wave, wave+8, ..
    constexpr int kPieces = (KS + 7) / 8;
    auto dma_piece = [&](const int vc, char *dst, const int k) {
        const int i = wave + 8 * k;
        static_assert(KS % 8 == 0, "every wave owns the same number of pieces");
        lds_dma16((const char *)jp.W2Tp + (size_t)vc * chunk_bytes + i * 1024 + lane * 16, dst + i * 1024);
    };
    // bias table (FWD: b2 * log2 e) in LDS behind everything else when it fits.  A compile-time switch: with both routes in
    // one kernel the epilogue's wait for the table values became `vmcnt(0)` as well -- a wait for the next chunk's LDS-DMA.
    constexpr bool b2_in_lds = B2LDS;
    const float *b2tab = BWD ? jp.b2 : jp.b2l;
    const float *b2img = (const float *)(smem + (b2_in_lds ? jp.b2_lds_off : 0));

    // ---- h row of this lane's cell, rounded to binary16, in MFMA B-fragment order.
    // The tile's enc_proj / pred_proj rows (or their e^{2x} tables) are first brought into LDS by LDS-DMA, all pieces in
    // flight at once: fetching them lane by lane from global memory (32-byte row pieces behind a dependent tanh chain)
    // took 60-70k of a tile's 220k cycles.  Images (16-byte chunks; chunk c holds joint units 4c..4c+3):
    //   Pimg [4 KS chunks][32 columns]  -- lane n of a half-wave reads consecutive chunks: conflict-free
    //   Eimg [8 waves][4 KS chunks]     -- one row per wave, broadcast reads
    // They overlay the W2^T buffers (and the BWD staging area), which are not in use yet.
    h8 hf[KS];
    {
        const bool slow = jp.scal[2] != 0.f;  // kernel-uniform
        const float *Etab = slow ? jp.enc_proj : jp.expE, *Ptab = slow ? jp.pred_proj : jp.expP;
        char *Pimg = smem, *Eimg = smem + KS * 4 * 32 * 16;
        {
            const float *prow = Ptab + ((size_t)b * p.U + uc) * J + 4 * (lane >> 5);
            for (int i = wave; i < KS * 2; i += 8)  // piece i = chunks 2i, 2i+1 of all 32 columns
                lds_dma16(prow + 8 * i, Pimg + i * 1024);
            const float *erow = Etab + ((size_t)b * p.T + tc) * J;
            for (int k = 0; k * 64 < KS * 4; ++k)
                if (k * 64 + lane < KS * 4)
                    lds_dma16(erow + 4 * (k * 64 + lane), Eimg + wave * (KS * 64) + k * 1024);
            if (b2_in_lds)
                for (int i = wave; i < ((V + 255) >> 8); i += 8)
                    if (i * 256 + lane * 4 < V)  // (V is a multiple of 128: the last piece may be half full)
                        lds_dma16(b2tab + i * 256 + lane * 4, smem + jp.b2_lds_off + i * 1024);
        }
        wait_vm();
        __syncthreads();
        const char *pl = Pimg + (2 * half * 32 + n) * 16, *el = Eimg + wave * (KS * 64) + 2 * half * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 e0 = *(const float4 *)(el + ks * 64), e1 = *(const float4 *)(el + ks * 64 + 16);
            const float4 c0 = *(const float4 *)(pl + ks * 2048), c1 = *(const float4 *)(pl + ks * 2048 + 512);
            h8 v;
            if (!slow) {
                v[0] = (f16)htanh2(e0.x, c0.x), v[1] = (f16)htanh2(e0.y, c0.y);
                v[2] = (f16)htanh2(e0.z, c0.z), v[3] = (f16)htanh2(e0.w, c0.w);
                v[4] = (f16)htanh2(e1.x, c1.x), v[5] = (f16)htanh2(e1.y, c1.y);
                v[6] = (f16)htanh2(e1.z, c1.z), v[7] = (f16)htanh2(e1.w, c1.w);
            } else {
                v[0] = (f16)htanh(e0.x + c0.x), v[1] = (f16)htanh(e0.y + c0.y);
                v[2] = (f16)htanh(e0.z + c0.z), v[3] = (f16)htanh(e0.w + c0.w);
                v[4] = (f16)htanh(e1.x + c1.x), v[5] = (f16)htanh(e1.y + c1.y);
                v[6] = (f16)htanh(e1.z + c1.z), v[7] = (f16)htanh(e1.w + c1.w);
            }
            hf[ks] = v;
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the read-ahead (registers)
        }
        __syncthreads();  // every wave is done with the images: the W2^T buffers may be filled
    }
#pragma unroll
    for (int k = 0; k < kPieces; ++k) dma_piece(0, wbuf0, k);

    JT(1);
    // ---- per-cell scalars
    float xb = 0.f, xl = 0.f;  // blank / label logits
    int lab = 0;
    const bool has_label = cell_valid && (u < Ub - 1);
    if (has_label) lab = clamp_label(p.labels[(size_t)b * (p.U - 1) + u], V);
    float mref = -1.0e30f, ssum = 0.f;  // FWD: online log2-sum-exp2 state of this lane's share of the vocabulary
    CellGrad g;                         // BWD
    float scaleS = 0.f, c0 = -1.0e30f;
    f16 *my_stage = (f16 *)(smem + 2 * chunk_bytes) + wave * (32 * kStageStride);
    // FWD: the blank / label logits are picked out of the MFMA tiles (log2-scaled: y = x * log2 e).  Column v sits in
    // chunk v >> 5, in the half-lane ((v & 31) >> 2) & 1, register (v & 3) + 4 * ((v & 31) >> 3).
    const int vcb = p.blank >> 5, hb = ((p.blank & 31) >> 2) & 1, rb = (p.blank & 3) + 4 * ((p.blank & 31) >> 3);
    const int vcl = lab >> 5, hl = ((lab & 31) >> 2) & 1, rl = (lab & 3) + 4 * ((lab & 31) >> 3);
    if (BWD) {
        if (cell_valid) {
            Cell cl;
            cl.b = b, cl.t = t, cl.u = u, cl.Tb = Tb, cl.Ub = Ub, cl.valid = true;
            g = cell_grad_setup(p, cl, c);
            scaleS = g.scale * jp.scal[0];
            c0 = g.c0;
            xb = jp.xbl[2 * (size_t)c], xl = jp.xbl[2 * (size_t)c + 1];  // log2-scaled
        }
    }

    // ---- the wave's staged chunk ([32 cells][32 columns], written by the epilogue) leaves as two 1 KB store instructions
    // (16 rows x 64 bytes each), issued BETWEEN the MFMA groups of the next chunk.  The first version staged 128 columns and
    // stored 8 KB per wave in one burst behind every fourth epilogue: that burst held the wave for 1200-2100 shader clocks
    // (s_memtime timeline, with or without the LDS reads in front of it); spread out, the stores cost no visible cycles and
    // the staging tile is 2.5 KB per wave instead of 8.7.  (Run time: the same within noise -- the kernel is power-limited.)
    const size_t cell0 = (size_t)(b * p.T + t) * p.U + u0;
    const f16 *const st_src = my_stage + (lane >> 2) * kStageStride + (lane & 3) * 8;
    const uint32_t st_off = (uint32_t)(lane >> 2) * (uint32_t)(2 * V) + (uint32_t)(lane & 3) * 16u;
    auto stage_read = [&](const int j) -> h8 {  // rows are 8-byte aligned: two 8-byte reads
        const h4 a = *(const h4 *)(st_src + 16 * j * kStageStride), c = *(const h4 *)(st_src + 16 * j * kStageStride + 4);
        return __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto stage_store = [&](const int j, const int vcf, const h8 v) {
        char *const base = (char *)(jp.dl + cell0 * V + vcf * 32);  // wave-uniform
        if (u0 + 16 * j + (lane >> 2) < p.U) *(h8 *)(base + st_off + (uint32_t)(32 * j * V)) = v;
    };
    // PARK: the chunk references go out one short per cell and chunk, [lattice row][chunk][column (row pitch 32 n_ut)]: K3 brings the
    // references of a (4 rows x 32 columns) tile and one chunk in as four 64-byte segments (round 6; before: [cell][chunk])
    short *const pref_row = jp.pref + (size_t)(b * p.T + t) * (size_t)(V >> 5) * (size_t)(jp.n_ut * 32) + u0;

    // ---- epilogue of one 32-column chunk: acc[r] = logit (without bias) of this lane's cell at v = 32 vc + cdrow(r, half)
    auto epilogue = [&](const f32x16 &acc, const int vc) {
        // the four float4 of bias values this lane needs for chunk vc (typed LDS / global loads: a merged pointer would
        // turn them into flat loads, whose wait also covers the LDS-DMA pieces in flight)
        float4 bqs[4];
        typedef float vf4 __attribute__((ext_vector_type(4)));
        if (b2_in_lds) {
            const __attribute__((address_space(3))) vf4 *q3 =
                (const __attribute__((address_space(3))) vf4 *)(b2img + vc * 32 + 4 * half);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const vf4 v = q3[2 * q];
                bqs[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
            asm volatile("" ::: "memory");  // keep the two paths apart
            const __attribute__((address_space(1))) vf4 *q1 =
                (const __attribute__((address_space(1))) vf4 *)(b2tab + vc * 32 + 4 * half);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const vf4 v = q1[2 * q];
                bqs[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (!BWD) {
            // y = log2-scaled logits.  The running reference mref is only moved when a value exceeds it by more than
            // 2^64 (wave-uniform rare path): s = sum 2^(y - mref) stays in f32 range and keeps full precision because
            // mref never trails the running maximum by more than 64.
            float y[16];
            float m = -1.0e30f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = bqs[q];
                y[4 * q + 0] = fmaf(acc[4 * q + 0], kLog2e, bq.x);
                y[4 * q + 1] = fmaf(acc[4 * q + 1], kLog2e, bq.y);
                y[4 * q + 2] = fmaf(acc[4 * q + 2], kLog2e, bq.z);
                y[4 * q + 3] = fmaf(acc[4 * q + 3], kLog2e, bq.w);
                m = fmaxf(fmaxf(m, y[4 * q]), fmaxf(fmaxf(y[4 * q + 1], y[4 * q + 2]), y[4 * q + 3]));
            }
            // PARK: the chunk's reference = the integer at or above the largest of the cell's 32 values (both half-lanes
            // agree on it), so the parked 2^(y - ref) lie in (0, 1] whatever the rest of the vocabulary holds
            float ref = 0.f;
            if (PARK) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                ref = __builtin_amdgcn_fmed3f(ceilf(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]))), -kRefLimit, kRefLimit);
            }
            const float top = PARK ? ref : m;
            if (__any(top > mref + 64.0f)) {
                const float nr = fmaxf(mref, top);
                ssum *= hex2(mref - nr);
                mref = nr;
            }
            if (!PARK) {
                float s = ssum;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += hex2(y[r] - mref);
                ssum = s;
            } else {
                f16 *row = my_stage + n * kStageStride + 4 * half;
                float sc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float p0 = hex2(y[4 * q + 0] - ref), p1 = hex2(y[4 * q + 1] - ref);
                    const float p2 = hex2(y[4 * q + 2] - ref), p3 = hex2(y[4 * q + 3] - ref);
                    sc += (p0 + p1) + (p2 + p3);
                    h4 d;
                    d[0] = (f16)p0, d[1] = (f16)p1, d[2] = (f16)p2, d[3] = (f16)p3;
                    *(h4 *)(row + 8 * q) = d;
                }
                ssum = fmaf(sc, hex2(ref - mref), ssum);
                if (lane < 32) pref_row[(uint32_t)vc * (uint32_t)(jp.n_ut * 32) + (uint32_t)lane] = (short)(int)ref;
            }
            if (LOGITS && cell_valid) {  // decoding: the chunk's 16 logits of this half-lane, natural scale, straight to the caller
                float *o = jp.logits_out + (size_t)c * V + vc * 32 + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4 *)(o + 8 * q) = make_float4(y[4 * q] * kLn2, y[4 * q + 1] * kLn2, y[4 * q + 2] * kLn2, y[4 * q + 3] * kLn2);
            }
            if (vc == vcb) {  // wave-uniform
                asm volatile("" ::: "memory");  // keep this a branch (if-converted, it costs 17 selects per chunk)
                float v = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) v = (r == rb) ? y[r] : v;
                xb = (half == hb) ? v : xb;
            }
            const bool mine = has_label && (vc == vcl);
            if (__any(mine)) {
                asm volatile("" ::: "memory");
                float v = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) v = (r == rl) ? y[r] : v;
                xl = (mine && half == hl) ? v : xl;
            }
        } else {
            f16 *row = my_stage + n * kStageStride + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = bqs[q];
                h4 d;
                d[0] = (f16)(scaleS * hex2(fmaf(acc[4 * q + 0] + bq.x, kLog2e, c0)));
                d[1] = (f16)(scaleS * hex2(fmaf(acc[4 * q + 1] + bq.y, kLog2e, c0)));
                d[2] = (f16)(scaleS * hex2(fmaf(acc[4 * q + 2] + bq.z, kLog2e, c0)));
                d[3] = (f16)(scaleS * hex2(fmaf(acc[4 * q + 3] + bq.w, kLog2e, c0)));
                *(h4 *)(row + 8 * q) = d;
            }
            // the two edge columns, where this chunk holds them (LDS writes of one wave land in program order)
            const int vbase = vc * 32;
            if (cell_valid && half == 0) {
                const bool same = g.has_label && (g.lab == p.blank);
                const bool pb = (unsigned)(p.blank - vbase) < 32u, pl = g.has_label && !same && (unsigned)(g.lab - vbase) < 32u;
                if (pb || pl) {
                    const float cb = g.has_blank_corr ? hex2(xb + g.nl + g.cb) : 0.f;
                    const float clb = g.has_label ? hex2(xl + g.nl + g.cl) : 0.f;
                    if (pb) my_stage[n * kStageStride + p.blank - vbase] = (f16)(scaleS * (hex2(xb + c0) - cb - (same ? clb : 0.f)));
                    if (pl) my_stage[n * kStageStride + g.lab - vbase] = (f16)(scaleS * (hex2(xl + c0) - clb));
                }
            }
        }
    };

    // Waves w and w+4 share a SIMD.  Waves 4..7 run their epilogue one chunk late (before the next chunk's MFMAs instead
    // of after their own), so that on every SIMD one wave is in its VALU phase while the other feeds the matrix pipe.
    // Either way the chunk a wave staged last is chunk vc - 1 when it enters the MFMA groups of chunk vc: that is where its
    // two store instructions are issued (stage_store).
    const bool late = wave >= 4;  // wave-uniform
    const int NC = V >> 5;
    f32x16 acc;
    for (int vc = 0; vc < NC; ++vc) {
        JT(2 + 4 * vc);
        wait_vm();
        __syncthreads();  // chunk vc is in LDS; every wave is done with the other buffer
        char *nbuf = ((vc + 1) & 1) ? wbuf1 : wbuf0;
        const bool more = vc + 1 < NC;
        if (!wave_live && more) {
#pragma unroll
            for (int k = 0; k < kPieces; ++k) dma_piece(vc + 1, nbuf, k);
        }
        JT(3 + 4 * vc);
        if (!wave_live) continue;
        if (late && vc > 0) epilogue(acc, vc - 1);
        JT(4 + 4 * vc);
        const char *wb = ((vc & 1) ? wbuf1 : wbuf0) + lane * 16;
        constexpr bool kTwoChains = !(STAGE && KS > 32);  // two accumulation chains unless registers are short
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
        {
            // A fragments one group of k-steps ahead of their MFMAs (bounded: the compiler would otherwise hoist all KS
            // reads); groups of 4, or of 2 where registers are short
            constexpr int G = kTwoChains ? 4 : 2;
            constexpr int NG = KS / G;
            // groups after which the staged chunk's pieces are read / stored (kE2 == NG: behind the loop)
            constexpr int kE1 = NG >= 3 ? NG / 3 : 1, kE2 = NG >= 3 ? 2 * NG / 3 : NG;
            const bool staged = STAGE && vc > 0;
            h8 sp;
            h8 acur[G], anxt[G];
#pragma unroll
            for (int q = 0; q < G; ++q) acur[q] = *(const h8 *)(wb + q * 1024);
#pragma unroll
            for (int g4 = 0; g4 < KS / G; ++g4) {
                if (g4 + 1 < KS / G) {
#pragma unroll
                    for (int q = 0; q < G; ++q) anxt[q] = *(const h8 *)(wb + (G * (g4 + 1) + q) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the next chunk's LDS-DMA pieces are issued between MFMA groups (issuing all of them at the top of the
                // chunk cost ~150 cycles apiece with the matrix pipe idle)
#pragma unroll
                for (int k = 0; k < kPieces; ++k)
                    if ((k * (KS / G)) / kPieces == g4 && more) dma_piece(vc + 1, nbuf, k);
                if (kTwoChains) {
#pragma unroll
                    for (int q = 0; q < G; q += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(acur[q], hf[G * g4 + q], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(acur[q + 1], hf[G * g4 + q + 1], acc1, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(acur[q], hf[G * g4 + q], acc0, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (STAGE && staged) {
                    if (g4 == 0) sp = stage_read(0);
                    if (g4 == kE1) {
                        stage_store(0, vc - 1, sp);
                        sp = stage_read(1);
                    }
                    if (g4 == kE2) {
                        stage_store(1, vc - 1, sp);
                    }
                }
#pragma unroll
                for (int q = 0; q < G; ++q) acur[q] = anxt[q];
            }
            if (STAGE && staged && kE2 >= NG) {
                stage_store(1, vc - 1, sp);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = kTwoChains ? acc0[r] + acc1[r] : acc0[r];
        JT(5 + 4 * vc);
        if (!late) epilogue(acc, vc);
    }
    JT(2 + 4 * 32);
    if (late && wave_live) epilogue(acc, NC - 1);
    if (STAGE && wave_live) {  // the last chunk's pieces
        stage_store(0, NC - 1, stage_read(0));
        stage_store(1, NC - 1, stage_read(1));
    }

    if (!BWD && wave_live) {
        // merge the two half-lanes of a cell, then the same outputs as the lsm pass of rnnt_kernels.hip
        const float m2 = __shfl_xor(mref, 32), s2 = __shfl_xor(ssum, 32);
        const float M = fmaxf(mref, m2);
        const float S = ssum * hex2(mref - M) + s2 * hex2(m2 - M);
        const float lgS = hlg2(S);
        const float lse2 = M + lgS;
        // the captured edge logits live in one of the two half-lanes of the cell
        const float xb_o = __shfl_xor(xb, 32), xl_o = __shfl_xor(xl, 32);
        xb = (half == hb) ? xb : xb_o;
        xl = (half == hl) ? xl : xl_o;
        if (cell_valid && half == 0) {
            const bool blank_stays = (t < Tb - 1) || (u == Ub - 1);
            const float ob = blank_stays ? (xb - M) - lgS : kNeg;
            const float ol = has_label ? (xl - M) - lgS : kNeg;
            p.lse[c] = lse2 * kLn2;
            ((float2 *)p.W)[((size_t)b * p.Nr + (t + u)) * p.Up + u] = make_float2(ob, ol);
            ((float2 *)jp.xbl)[c] = make_float2(xb, xl);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Lattice rows the backward does not visit (round 6; the f32-grade joint has had this since round 5: joint_kernels.hip kOccFloor).
// Every dlogits value of a cell is bounded by 2 |cost_scale| x the cell's occupancy alpha.beta / L.  One workgroup per (utterance,
// 8 rows, 32 columns): a row's bit is set when some cell of it has an occupancy above 2^-kOccFloorH (NaN counts as occupied), and
// the four rows of an aligned group then share the OR of their bits -- K3 works in iterations of four rows, K4 row by row, the
// d enc_proj reduction reads the rows K3 wrote: all three follow these bits.  RNNT_VISIT_ALL (include/rnnt.h): every row inside
// the utterance.  How many rows that is depends on the data: counted for get_rnnt_joint_backward_rows.
// The floor of THIS engine is 2^-40, not the 2^-50 of the binary32 paths, and it costs nothing at all: dl is stored as binary16 of
// S x dlogits with S |cost_scale| <= 2^14 (jh_prep_kernel), so every entry of a cell with an occupancy below 2^-40 is below 2^-26 in
// magnitude and ROUNDS TO ZERO (the smallest binary16 subnormal is 2^-24) -- the rows between the two floors were visited to multiply
// and add exact zeros.  (N(0,1) projections at config 5: 34.8 % -> 31 % of the rows.)
// ---------------------------------------------------------------------------------------------
constexpr int kOccFloorH = 40;
#ifndef JH_DHX_WGS
#define JH_DHX_WGS 2560  // (measured at config 5: 1280 / 2560 / 3840 workgroups: pruned N(0,1) 5.5 / 5.3 ms, trained-like 2.5 / 1.9 ms, all rows the same)
#endif
__host__ __device__ inline int rowcnt_stamp(int T, int U, int B, int J, int V) {
    return (int)(0x16f16000u ^ ((unsigned)T * 73856093u) ^ ((unsigned)U * 19349663u) ^ ((unsigned)B * 83492791u) ^ ((unsigned)J * 2654435761u) ^ (unsigned)V);
}
constexpr int kRowbitsRows = 64;  // lattice rows per workgroup of jh_rowbits_kernel
__global__ __launch_bounds__(256) void jh_rowbits_kernel(const JhParams jp) {
    // One workgroup per (utterance, 64 rows, 32 columns), walked DIAGONAL by diagonal: the lattice state is stored diagonal-major
    // (a diagonal's 32 columns of the tile are 128 consecutive bytes; a row's would be 32 different cache lines: 0.69 ms at config 5)
    const LossParams &p = jp.lp;
    const int tid = threadIdx.x, n_tb = (p.T + kRowbitsRows - 1) / kRowbitsRows;
    int bid = blockIdx.x;
    const int ut = bid % jp.n_ut;
    bid /= jp.n_ut;
    const int tb = bid % n_tb, b = bid / n_tb;
    const int t0 = tb * kRowbitsRows, u0 = ut * 32;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    __shared__ unsigned long long rowmask;  // bit r: row t0 + r has a cell with occupancy above the floor
    if (tid == 0) rowmask = 0ull;
    __syncthreads();
    const bool tile_live = t0 < Tb && u0 < Ub;
    if (tile_live) {
        const int ul = tid & 31, u = u0 + ul;
        const double ll2 = p.ll[2 * b];
        const size_t ob = (size_t)b * p.NC * p.NG + fdiv((uint32_t)u, p.divOG);
        unsigned long long mine = 0ull;
        for (int d = tid >> 5; d < kRowbitsRows + 31; d += 8) {  // diagonal n = t0 + u0 + d of the tile
            const int n = t0 + u0 + d, t = n - u;
            if (t >= t0 && t < t0 + kRowbitsRows && t < Tb && u < Ub) {
                const size_t sk = ((size_t)b * p.Nr + n) * p.Up + u, ok = ob + (size_t)(n / kRebase) * p.NG;
                const double l2occ = ((double)p.A[sk] + (double)p.offA[ok]) + ((double)p.Bt[sk] + (double)p.offB[ok]) - ll2;
                if (jp.visit_all || !(l2occ <= (double)-kOccFloorH)) mine |= 1ull << (t - t0);
            }
        }
        if (mine) atomicOr(&rowmask, mine);
    }
    __syncthreads();
    if (tid < kRowbitsRows / 8) {  // one byte of bits per 8 rows
        int m = (int)((rowmask >> (8 * tid)) & 0xffull);
        m = ((m & 0x0f) ? 0x0f : 0) | ((m & 0xf0) ? 0xf0 : 0);  // the groups of four rows
        // (rows of a group beyond T_b carry the group's bit: K3 / K4 / the reduction clip rows at T_b themselves)
        const int tt = tb * (kRowbitsRows / 8) + tid;
        if (tt < 4 * ((p.T + 31) >> 5)) jp.live8[((size_t)b * jp.n_ut + ut) * (size_t)(4 * ((p.T + 31) >> 5)) + tt] = (uint8_t)m;
        int vis = 0, ins = 0;
        for (int q = 0; q < 8; ++q) {
            const bool in = tile_live && (tt * 8 + q < Tb);
            ins += in, vis += in && ((m >> q) & 1);
        }
        if (vis) atomicAdd(jp.rowcnt, vis);
        if (ins) atomicAdd(jp.rowcnt + 1, ins);
    }
    if (blockIdx.x == 0 && tid == 0) jp.rowcnt[2] = rowcnt_stamp(p.T, p.U, p.B, jp.J, p.V);
}

// K3's strips (utterance, u-tile, row split) in the order of their work, heaviest first: one workgroup counts each strip's visited
// rows and sorts by counting (<= 256 buckets).  With the pruning the strips' work follows the alignment band -- about a third of
// them do everything on unstructured inputs -- and dealt in lattice order some CUs drew three or four full strips while others
// drew none (config 5: K3 at 35 % of the rows took 71 % of its all-rows time).  The order inside a bucket depends on timing; the
// results do not depend on the order (every strip writes its own slabs).
__global__ __launch_bounds__(1024) void jh_order_kernel(const JhParams jp, const int n_strips) {
    __shared__ int hist[257], cursor[257];
    const LossParams &p = jp.lp;
    const int tid = threadIdx.x;
    for (int i = tid; i < 257; i += 1024) hist[i] = 0;
    __syncthreads();
    auto weight = [&](const int i) -> int {
        int q = i;
        const int ts = q % jp.n_ts;
        q /= jp.n_ts;
        const int ut = q % jp.n_ut, b = p.b0 + q / jp.n_ut;
        const int Tb = length_T(p, b), Ub = length_U(p, b);
        const int t0 = ts * jp.TS, t1 = min(min(t0 + jp.TS, p.T), Tb);
        if (t0 >= t1 || ut * 32 >= Ub) return 0;
        const uint32_t *bits = (const uint32_t *)(jp.live8 + ((size_t)b * jp.n_ut + ut) * (size_t)(4 * ((p.T + 31) >> 5)));
        int w = 0;
        for (int t = t0; t < t1;) {  // (t0 is a multiple of 4; whole words where possible)
            const int e = min(t1, (t | 31) + 1);
            uint32_t m = bits[t >> 5] >> (t & 31);
            if (e - t < 32) m &= (1u << (e - t)) - 1u;
            w += __popc(m);
            t = e;
        }
        return min((w + 3) >> 2, 256);  // groups of four rows
    };
    constexpr int kMaxStrips = 8192;
    __shared__ short sw[kMaxStrips];  // (a strip's weight is a dozen dependent loads: computed once)
    for (int i = tid; i < n_strips; i += 1024) {
        const int w = weight(i);
        if (i < kMaxStrips) sw[i] = (short)w;
        atomicAdd(&hist[w], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int w = 256; w >= 0; --w) cursor[w] = pos, pos += hist[w];
    }
    __syncthreads();
    for (int i = tid; i < n_strips; i += 1024) jp.order[atomicAdd(&cursor[i < kMaxStrips ? (int)sw[i] : weight(i)], 1)] = i;
    // K4's work units (utterance, u-tile, kTQ rows) by visited rows, descending, ties by index.  K4's workgroups keep their dW2
    // accumulators over all of their units, so the units are DEALT (range r takes sorted positions r, 2R-1-r, 2R+r, ...: a snake), and
    // the deal must not depend on timing: a stable counting sort (a unit's visited rows: 0 .. kTQ).
    constexpr int kMaxUnits = 6144;
    __shared__ short uw[kMaxUnits];
    const int nu = jp.n_units;
    if (nu > kMaxUnits) {  // (identity: K4 then deals the units in lattice order)
        for (int i = tid; i < nu; i += 1024) jp.uorder[i] = i;
        return;
    }
    for (int i = tid; i < nu; i += 1024) {
        int q = i;
        const int tq = q % jp.n_tq;
        q /= jp.n_tq;
        const int ut = q % jp.n_ut, b = p.b0 + q / jp.n_ut;
        const int Tb = length_T(p, b), Ub = length_U(p, b);
        const int t0 = tq * kTQ, t1 = min(min(t0 + kTQ, p.T), Tb);
        int w = 0;
        if (t0 < t1 && ut * 32 < Ub) {
            const uint32_t *bits = (const uint32_t *)(jp.live8 + ((size_t)b * jp.n_ut + ut) * (size_t)(4 * ((p.T + 31) >> 5)));
            for (int t = t0; t < t1;) {
                const int e = min(t1, (t | 31) + 1);
                uint32_t m = bits[t >> 5] >> (t & 31);
                if (e - t < 32) m &= (1u << (e - t)) - 1u;
                w += __popc(m);
                t = e;
            }
        }
        uw[i] = (short)w;
    }
    __syncthreads();
    // stable counting sort, weight descending (a unit's visited rows: 0 .. kTQ): wave v takes the weights v, v + 16, ...; for each it
    // walks the list 64 units at a time (ballot + prefix count), first to count, then to place
    __shared__ int ustart[kTQ + 2];
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int w = wv; w <= kTQ; w += 16) {
        int c = 0;
        for (int j0 = 0; j0 < nu; j0 += 64) c += __builtin_popcountll(__ballot(j0 + lane < nu && uw[j0 + lane] == w));
        if (lane == 0) ustart[w] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int w = kTQ; w >= 0; --w) {
            const int c = ustart[w];
            ustart[w] = pos, pos += c;
        }
    }
    __syncthreads();
    for (int w = wv; w <= kTQ; w += 16) {
        int pos = ustart[w];
        for (int j0 = 0; j0 < nu; j0 += 64) {
            const bool mine = j0 + lane < nu && uw[j0 + lane] == w;
            const unsigned long long m = __ballot(mine);
            if (mine) jp.uorder[pos + __builtin_popcountll(m & lt)] = j0 + lane;
            pos += __builtin_popcountll(m);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3, round 6 (jh_dhx_kernel<NT>, J = 128 NT): the streaming pass K2 and the dh product in ONE kernel, every dl element touched once.
//   workgroup = (utterance, u-tile of 32 columns, row split); per iteration 4 lattice rows x 32 columns = 128 cells x ALL J units
//   (the old K3 gave a workgroup 128 units: five workgroups fetched -- and would have had to convert -- every dl row).
//   A operand: the cell rows of dl.  A thread owns one 16-byte piece of one cell row per 32-symbol step: it loads the PARKED
//   numerators two steps ahead (registers), multiplies them back with the cell's factor (exactly K2's arithmetic: one f32 factor per
//   (cell, chunk), the blank / label columns from the f32 edge terms), writes the binary16 result into the LDS stage AND back to
//   dl for K4 -- in place, once.  B operand: W2, chunk-major (W2c), 64 J bytes per step by LDS-DMA.
//   8 waves = 2 (column halves of the tile) x 4 (J quarters); a wave multiplies 64 cells x 32 NT units: 2 + NT fragment reads for
//   2 NT MFMAs per k-step (0.7 reads per MFMA at J = 640; the old 2 x 2 wave tile: 1.0).
//   An MFMA row tile holds 4 lattice rows x 8 columns (not 1 x 32), so that in the accumulator layout a lane's 16 registers are
//   4 rows x 4 columns: sum_t collapses to 4 values per tile in registers.  Those running column sums (d pred_proj) do not fit the
//   register file beside 32 NT accumulators: they live in a workspace slab the SAME wave re-reads and re-writes once per iteration
//   (16 bytes per lane and tile: L2 traffic of 160 KB per 640 MFMAs).  sum_u (d enc_proj): in registers + one pass through LDS
//   between the two column halves.
// LDS: 3 stages x (A 128 rows x 64 B + B J rows x 64 B), 16-byte chunks XOR-swizzled by (row >> 2) & 3 (conflict-free b128 reads),
//      + [4][J] floats for the d enc_proj hand-over = 157,696 bytes at J = 640.
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512) void jh_dhx_kernel(const JhParams jp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LossParams &p = jp.lp;
    constexpr int J = 128 * NT, JW = 32 * NT;
    constexpr uint32_t kBS = J * 64;                    // bytes of a W2 stage
    constexpr uint32_t kAoff = 3 * kBS;                 // four A stages of 8 KB behind the three W2 stages
    constexpr uint32_t kRoff = kAoff + 4 * 8192;        // four reference tiles [4 rows][32 columns] of shorts
    constexpr uint32_t kCoff = kRoff + 4 * 256;         // 2 x [4 rows][128 units] floats: row sums of column half 1 on their way to half 0
    constexpr uint32_t kLoff = kCoff + 4096;            // [32] ints: the tile's labels
    const int V = p.V, NK = V >> 5;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    float *const comb = (float *)(smem + kCoff);
    const uint32_t smem0 = lds_addr(smem);

    const uint32_t wg = (uint32_t)jp.order[blockIdx.x];  // the strips with the most visited rows first (jh_order_kernel)
    uint32_t bid = wg;
    const int ts = (int)(bid % (uint32_t)jp.n_ts);
    bid /= (uint32_t)jp.n_ts;
    const int ut = (int)(bid % (uint32_t)jp.n_ut);
    const int b = p.b0 + (int)(bid / (uint32_t)jp.n_ut);
    const int u0 = ut * 32;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int t_begin = ts * jp.TS, t_end = min(min(t_begin + jp.TS, p.T), Tb);
    if (t_begin >= t_end || u0 >= Ub) return;
    const bool parked = jp.state[0] == 1;  // kernel-uniform: dl holds parked numerators (else: dlogits from the recompute kernel)
    const bool slow = jp.scal[2] != 0.f;   // kernel-uniform: tabulated e^{2x} factors unusable, fall back to tanh(a + c)
    const float *Etab = slow ? jp.enc_proj : jp.expE, *Ptab = slow ? jp.pred_proj : jp.expP;
    const float c4 = 4.0f * jp.scal[1];    // 4 / S: (1 - h^2) is formed as 4 r (1 - r), S is the dlogits scale
    const int Upad = jp.n_ut * 32;

    // ---- this thread's piece of the A operand: LDS row `slot` (MFMA row tile slot >> 5; row m of a tile = lattice row m >> 3, column
    // m & 7), 16-byte position tid & 3 of it -- which holds the row's chunk lc (swizzle).  The lane that brings the piece in (LDS-DMA:
    // lane-linear image) is the lane that converts it.  Everything here is RECOMPUTED from the thread index where it is needed (Geo
    // of a laundered copy): kept in registers through the main loop, beside 32 NT accumulators, it was spilled, and the compiler
    // waits vmcnt(0) -- for every LDS-DMA piece in flight -- behind a scratch reload.  The column's label sits in LDS for the same reason.
    struct Geo {
        int slot, lc, crow, ccol, cu;
    };
    auto geo_of = [&](const int t2) -> Geo {
        Geo g;
        g.slot = t2 >> 2, g.lc = (t2 & 3) ^ ((g.slot >> 2) & 3);
        g.crow = (g.slot & 31) >> 3, g.ccol = 8 * (g.slot >> 5) + (g.slot & 7);
        g.cu = u0 + g.ccol;
        return g;
    };
    int *const labtab = (int *)(smem + kLoff);  // [32 columns]: the label that leaves the column's cells (clamped), -1 where there is none
    if (tid < 32) labtab[tid] = (u0 + tid < Ub - 1) ? clamp_label(p.labels[(size_t)b * (p.U - 1) + u0 + tid], V) : -1;
    __syncthreads();
    // (the main loop's copy, in a register: read from LDS in a_store -- between a step's fragment reads and its first MFMA -- the value was
    // consumed by the store's patch tests, and the wait for it, lgkmcnt(0), was a wait for EVERY fragment read of the step in front of the first MFMA)
    const int mylab = labtab[geo_of(tid).ccol];
    const double ll2 = p.ll[2 * b];
    const float cscale = (p.cost_scale ? p.cost_scale[b] : 1.0f) * jp.scal[0];
    // wave-uniform bases (scalar registers); everything per lane is a 32-bit offset from one of them
    const float *const offA_b = p.offA + (size_t)b * p.NC * p.NG, *const offB_b = p.offB + (size_t)b * p.NC * p.NG;
    const float *const A_b = p.A + (size_t)b * p.Nr * p.Up, *const Bt_b = p.Bt + (size_t)b * p.Nr * p.Up;
    auto row_block = [&](const int t) -> size_t { return ((size_t)(b * p.T + t) * p.U + u0); };  // first cell of the tile's row t
    // the strip's first dl row block / reference row, and the byte pitch of a lattice row in both
    const char *const dl_s = (const char *)(jp.dl + row_block(t_begin) * V);
    const char *const rf_s = (const char *)(jp.pref + ((size_t)(b * p.T + t_begin) * NK) * Upad + u0);
    const uint32_t dl_pitch = (uint32_t)p.U * (uint32_t)V * 2u, rf_pitch = (uint32_t)NK * (uint32_t)Upad * 2u;

    struct CellSt {    // what the conversion of this thread's cell row needs, per iteration
        uint32_t off;  // byte offset of the thread's piece from the iteration's dl row block (row and column clamped into what K1 wrote)
        int flags;     // 1: the cell row exists in the tensor (store it); 2: it is a lattice cell of the utterance (else: zeros)
        float c0, eb, el;
    };
    auto cell_place = [&](CellSt &c, const int t_it) {
        const Geo g = geo_of(launder(tid));
        const int ct = t_it + g.crow;
        const bool row_live = ct < t_end;
        c.flags = ((row_live && g.cu < p.U) ? 1 : 0) | ((row_live && g.cu < Ub) ? 2 : 0);
        const int ccol_c = min(g.cu, p.U - 1) - u0;  // (the column this thread LOADS: clamped into the tensor)
        c.off = ((uint32_t)((min(ct, t_end - 1) - t_it) * p.U + ccol_c) * (uint32_t)V + (uint32_t)g.lc * 8u) * 2u;
        c.c0 = kNeg, c.eb = 0.f, c.el = 0.f;
    };
    // cell_grad_from + K2's edge terms (ordinary loads: called where no LDS-DMA needs to stay in flight -- prologue and epilogue)
    struct CellRaw {  // the lattice values behind the factors: loaded early in the epilogue, used at its end
        float a, bt, b_t1, b_u1, oa, obt, ob_t1, ob_u1, lse;
        float2 x;
    };
    auto cell_fetch = [&](CellRaw &r, const CellSt &c, const int t_it) {
        if (!((c.flags & 2) && parked)) return;
        const Geo g = geo_of(launder(tid));
        const int cu = g.cu, lab = labtab[g.ccol];
        const bool has_label = lab >= 0;
        const uint32_t og0 = fdiv((uint32_t)cu, p.divOG), og1 = fdiv((uint32_t)cu + 1u, p.divOG);
        const int ct = t_it + g.crow, nd = ct + cu;
        const uint32_t kc_ = (uint32_t)nd / kRebase, kc1 = (uint32_t)(nd + 1) / kRebase;
        const uint32_t sk = (uint32_t)nd * (uint32_t)p.Up + (uint32_t)cu;
        r.a = A_b[sk], r.bt = Bt_b[sk];
        r.b_t1 = (ct < Tb - 1) ? Bt_b[sk + p.Up] : 0.f;
        r.b_u1 = has_label ? Bt_b[sk + p.Up + 1] : 0.f;
        r.oa = offA_b[kc_ * p.NG + og0], r.obt = offB_b[kc_ * p.NG + og0];
        r.ob_t1 = offB_b[kc1 * p.NG + og0], r.ob_u1 = offB_b[kc1 * p.NG + og1];
        const size_t c0w = row_block(t_it);  // uniform
        const uint32_t co = (uint32_t)(g.crow * p.U + g.ccol);
        r.lse = (p.lse + c0w)[co];
        r.x = ((const float2 *)jp.xbl + c0w)[co];
    };
    auto cell_finish = [&](CellSt &c, const CellRaw &r, const int t_it) {
        if (!((c.flags & 2) && parked)) return;
        const Geo g = geo_of(launder(tid));
        const int cu = g.cu, lab = labtab[g.ccol];
        const bool has_label = lab >= 0, same = lab == p.blank;
        const int ct = t_it + g.crow;
        const double da = (double)r.a + ((double)r.oa - ll2);
        const float nl = -r.lse * kLog2e;
        const float c0 = (float)(da + ((double)r.bt + (double)r.obt)) + nl;
        float cb = 0.f;
        bool has_bc = true;
        if (ct < Tb - 1) cb = (float)(da + ((double)r.b_t1 + (double)r.ob_t1));
        else if (cu == Ub - 1) cb = (float)da;
        else has_bc = false;
        const float cl = has_label ? (float)(da + ((double)r.b_u1 + (double)r.ob_u1)) : 0.f;
        const float cbv = has_bc ? hex2(r.x.x + nl + cb) : 0.f;
        const float clb = has_label ? hex2(r.x.y + nl + cl) : 0.f;
        c.c0 = c0;
        c.eb = cscale * (hex2(r.x.x + c0) - cbv - (same ? clb : 0.f));
        c.el = cscale * (hex2(r.x.y + c0) - clb);
    };
    auto cell_factors = [&](CellSt &c, const int t_it) {
        CellRaw r;
        cell_fetch(r, c, t_it);
        cell_finish(c, r, t_it);
    };
    // LDS-DMA of one step, per wave: its 1 KB of the A stage (16 cell rows x 64 B of parked values), the step's reference tile (every
    // wave brings the same 256 bytes: the waves' counts of outstanding pieces stay equal), NT pieces (16 W2 rows each) of the W2 stage
    auto dma_a = [&](const CellSt &c, const int t, const int kc, const int sa) {  // t = first row of the chunk's iteration: uniform
        const char *base = dl_s + (size_t)(uint32_t)(t - t_begin) * dl_pitch + (uint32_t)kc * 64u;
        lds_dma16_s(base, c.off, smem0 + kAoff + sa * 8192 + wave * 1024);
        const int rmax = t_end - 1 - t;  // rows of the tile beyond the strip read the last one's references (unused)
        const char *rb = rf_s + (size_t)(uint32_t)(t - t_begin) * rf_pitch + (uint32_t)kc * (uint32_t)(Upad * 2);
        const int ln = launder(tid) & 63;
        const uint32_t ro = (uint32_t)min(ln >> 4, rmax) * rf_pitch + (uint32_t)(ln & 15) * 4u;
        lds_dma4_s(rb, ro, smem0 + kRoff + sa * 256);
    };
    const uint32_t boff = (uint32_t)(16 * wave + (lane >> 2)) * 64u + (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);  // bytes; piece k: + 8192 k
    auto dma_b = [&](const int kc, const int sb, const int k) {
        const char *src = (const char *)(jp.W2c + (size_t)kc * (J * 32)) + k * 8192;  // uniform
        lds_dma16_s(src, boff, smem0 + sb * kBS + (wave + 8 * k) * 1024);
    };
    // the A piece in stage sa: parked values -> dlogits (K2's arithmetic), in place in LDS and, for K4, in dl.  In two halves: what it
    // reads from LDS (the raw piece, the cell's chunk reference, the column's label) is fetched with the step's fragment reads; the
    // arithmetic and the LDS writes sit among the MFMAs of the step's second k-step.
    struct ARaw {
        h8 v;
        short rf;
        int lab;
    };
    auto a_fetch = [&](ARaw &r, const int sa) {
        const int t2 = launder(tid);
        const Geo g = geo_of(t2);
        r.v = *(const h8 *)(smem + kAoff + sa * 8192 + t2 * 16);
        r.rf = *(const short *)(smem + kRoff + sa * 256 + (g.crow * 32 + g.ccol) * 2);
        r.lab = mylab;
    };
    auto a_finish = [&](const ARaw &r, const CellSt &c, const int kc, const int sa) -> h8 {
        const int t2 = launder(tid);
        const Geo g = geo_of(t2);
        h8 *const pa = (h8 *)(smem + kAoff + sa * 8192 + t2 * 16);
        // Straight-line (it is scheduled among the MFMAs): one factor per piece.  A cell outside the utterance -- a padded column of a
        // live tile, a row beyond the strip -- has c0 = "log zero", i.e. the factor 0, and what it multiplies is a value K1 wrote (rows
        // and columns are clamped where they are loaded): exact zeros without a select.  Where dl already holds dlogits (the
        // recompute kernel ran) the factor is 1 -- the round trip through f32 is exact -- or 0.
        const float mult = parked ? cscale * hex2((float)r.rf + c.c0) : ((c.flags & 2) ? 1.0f : 0.f);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(mult * (float)r.v[e]);
        *pa = o;
        // the blank / label column, where this piece holds it: one 2-byte LDS write over the piece (the same wave's LDS operations
        // land in program order); a_store does the same to dl
        const int vb = 32 * kc + 8 * g.lc, ib = p.blank - vb, il = ((r.lab >= 0 && r.lab != p.blank) ? r.lab : -1) - vb;
        const bool live = (c.flags & 2) && parked;
        if (live && (unsigned)ib < 8u) ((f16 *)pa)[ib] = (f16)c.eb;
        if (live && (unsigned)il < 8u) ((f16 *)pa)[il] = (f16)c.el;
        return o;
    };
    auto a_convert = [&](const CellSt &c, const int kc, const int sa) -> h8 {
        ARaw r;
        a_fetch(r, sa);
        return a_finish(r, c, kc, sa);
    };
    // ... and its way back to dl.  Issued at the START of the next step, in front of that step's LDS-DMA: the wait at the top of a
    // step allows the NT + 2 youngest operations to be outstanding, and those must be exactly the last step's DMA pieces.
    auto a_store = [&](const h8 o, const CellSt &c, const int t, const int kc) {
        if ((c.flags & 1) && parked) {
            char *base = (char *)const_cast<char *>(dl_s) + (size_t)(uint32_t)(t - t_begin) * dl_pitch + (uint32_t)kc * 64u;
            const uint32_t off = (uint32_t)launder((int)c.off);
            __builtin_nontemporal_store(o, (h8 *)(base + off));
            if (c.flags & 2) {  // the blank / label column over it (the same lane's stores to one address stay in order)
                const Geo g = geo_of(launder(tid));
                const int lab = mylab;
                const int vb = 32 * kc + 8 * g.lc, ib = p.blank - vb, il = ((lab >= 0 && lab != p.blank) ? lab : -1) - vb;
                if ((unsigned)ib < 8u) ((f16 *)(base + off))[ib] = (f16)c.eb;
                if ((unsigned)il < 8u) ((f16 *)(base + off))[il] = (f16)c.el;
            }
        }
    };

    // fragment read offsets of this lane (row n of a 32-row tile, k-step ks)
    const int sw = (n >> 2) & 3;
    const uint32_t foff0 = (uint32_t)(n * 64 + ((half ^ sw) * 16)), foff1 = (uint32_t)(n * 64 + (((2 + half) ^ sw) * 16));
    const uint32_t a_rd = kAoff + (uint32_t)((2 * wm) * 2048);  // + stage * 8192 + mi * 2048
    const uint32_t b_rd = (uint32_t)((wn * JW) * 64);           // + stage * kBS + ni * 2048

    // The iterations of this workgroup: the groups of four rows of its strip that the backward visits (jh_rowbits_kernel: some cell of
    // the group's 4 x 32 tile carries mass, or RNNT_VISIT_ALL).  next_live(t) = first row of the first such group at or after row t
    // (a multiple of four); anything >= t_end: none.  Scalar code: a 32-row word of bits per look.
    const uint32_t *const rowbits = (const uint32_t *)(jp.live8 + ((size_t)b * jp.n_ut + ut) * (size_t)(4 * ((p.T + 31) >> 5)));
    auto next_live = [&](int t) -> int {
        while (t < t_end) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowbits[t >> 5]) >> (t & 31);
            if (w) return t + __builtin_ctz(w);
            t = (t | 31) + 1;
        }
        return t_end;
    };
    const int t_first = next_live(t_begin);
    if (t_first >= t_end) return;  // nothing of this strip carries mass (its d enc_proj rows read as zero: the reduction follows the same bits)
    int t_n1 = next_live(t_first + 4);
    // ---- prologue: factors of the first two iterations; chunks 0..2 of A, 0..1 of W2 under way; chunk 0 converted
    CellSt cur, nxt;
    cell_place(cur, t_first);
    cell_factors(cur, t_first);
    nxt = cur;
    if (t_n1 < t_end) {
        cell_place(nxt, t_n1);
        cell_factors(nxt, t_n1);
    }
    dma_a(cur, t_first, 0, 0);
#pragma unroll
    for (int k = 0; k < NT; ++k) dma_b(0, 0, k);
    dma_a(cur, t_first, 1, 1);
#pragma unroll
    for (int k = 0; k < NT; ++k) dma_b(1, 1, k);
    dma_a(cur, t_first, 2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    h8 o_pend = a_convert(cur, 0, 0);  // the converted piece whose store is pending (chunk pk of the iteration at row pt, state pst)
    a_store(o_pend, cur, t_first, 0);
    bool pend = false;
    bool pend_next = false;
    int pk = 0;
    int sb = 0;  // W2 stage of the chunk being multiplied (the A stage is kc & 3)
    bool first = true;
#ifdef JH_TRACE
    long long *tr = (wg == 200) ? jp.trace + (size_t)(kTraceBlocks * 8 + 8 + wave) * kTraceSlots : nullptr;
    int tstep = 0;
#define DT(k)                                                                                          \
    do {                                                                                               \
        if (tr && tstep < 22 && lane == 0) tr[7 * tstep + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define DT(k) do { } while (0)
#endif

    for (int t_it = t_first; t_it < t_end;) {
        const bool has_next = t_n1 < t_end;                            // (t_n1: the next visited group, t_n2: the one after)
        const int t_n2 = has_next ? next_live(t_n1 + 4) : t_end;
        f32x16 acc[2][NT];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        // What a row tile of a 32-unit tile reads -- the enc-side factors of the 4 rows (per unit tile), the pred-side factors of
        // this lane's 4 columns, the running column sums -- is loaded one tile AHEAD of its arithmetic (loaded where it was used,
        // every tile paid a full memory latency three times: 6.9 of the kernel's 18.7 ms at config 5); the row sums are handed from
        // column half 1 to half 0 per unit tile (two 2 KB buffers, one barrier each), so that 4, not 4 NT, of them are alive.
        struct EpiT {
            float pr[4];
            float4 cs;
        };
        // (lane offsets from a laundered copy of the thread index, beside scalar bases: hoisted out of the row loop as 64-bit lane
        // addresses they were spilled, and every reload -- a wait for vmcnt(0) -- drained the tile loads prefetched just before)
        auto load_ej = [&](float (&ej)[4], const int ni) {
            const int n3 = launder(tid) & 31;
            const int jw = wn * JW + ni * 32;  // uniform; this lane's unit: jw + n
#pragma unroll
            for (int rw = 0; rw < 4; ++rw) ej[rw] = (Etab + ((size_t)b * p.T + min(t_it + rw, t_end - 1)) * J + jw)[(uint32_t)n3];
        };
        auto load_tile = [&](EpiT &in, const int ni, const int mi) {
            const int ln3 = launder(tid) & 63, n3 = ln3 & 31, half3 = ln3 >> 5;
            const int jw = wn * JW + ni * 32, ubw = u0 + 16 * wm + 8 * mi;  // uniform; this lane's columns: ubw + 4 half + 0..3
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
                in.pr[cq] = (Ptab + (size_t)b * p.U * J + jw)[(uint32_t)min(ubw + 4 * half3 + cq, p.U - 1) * (uint32_t)J + (uint32_t)n3];
            // (read even in the first iteration, where nothing has been written yet: the value is dropped below)
            in.cs = ((const float4 *)jp.dCacc + ((((size_t)wg * 8 + wave) * 2 + mi) * NT + ni) * 64)[(uint32_t)ln3];
        };
        float ejp[2][4];
        EpiT tin[3];
        auto load_idx = [&](const int idx) {  // the loads of row tile idx & 1 of unit tile idx >> 1 (idx: compile-time constant)
            if ((idx & 1) == 0) load_ej(ejp[(idx >> 1) & 1], idx >> 1);
            load_tile(tin[idx % 3], idx >> 1, idx & 1);
        };
        for (int kg = 0; kg < NK; kg += 4) {
            const bool last_group = kg + 4 >= NK;  // uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kc = kg + q;
                // Chunk (t_it, kc) must be complete: its W2 pieces and the raw A piece of the chunk converted in THIS step were issued
                // two steps ago; the last step issued NT + 2 pieces behind them (its store, if counted at all, in front of them), so
                // "at most NT + 2 outstanding" means everything up to two steps ago has landed.  The converted A pieces of the chunk:
                // LDS writes of the last step.  The first step of an iteration follows an epilogue with ordinary loads and stores, the
                // last steps of a workgroup issue less: drain there.
                DT(0);
                if (kc == 0 || (!has_next && last_group)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT + 2) : "memory");
                wait_lgkm();
                DT(1);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                DT(2);
                const int sb1 = (sb == 2) ? 0 : sb + 1, sb2 = (sb1 == 2) ? 0 : sb1 + 1;
                // the epilogue's first tile of inputs: a whole step ahead of their use (issued at the epilogue's start, their memory
                // latency sat in front of the first tile; measured: no difference in the kernel's time, kept because the allocation it
                // leads to has no spilled register)
                if (last_group && q == 3) load_idx(0);
                // what this step prepares: the converted A piece of chunk +1, W2 chunk +2, the raw A piece of chunk +3
                const bool x1 = last_group && q + 1 >= 4, x2 = last_group && q + 2 >= 4, x3 = last_group && q + 3 >= 4;  // ... lies in the next iteration
                const bool e1 = !x1 || has_next, e2 = !x2 || has_next, e3 = !x3 || has_next;                            // ... exists
                const int k1 = x1 ? kc + 1 - NK : kc + 1, k2 = x2 ? kc + 2 - NK : kc + 2, k3 = x3 ? kc + 3 - NK : kc + 3;
                // (Measured and dropped, config 5: waves 0..3 converting FIRST and their SIMD partners 4..7 last 14.8 against 14.2 ms;
                // `s_setprio 1` for waves 4..7, the reference tile brought in by one wave instead of all eight, the A pieces issued
                // in front of the W2 pieces (+1.1 ms), non-temporal A loads (+0.7 ms): profiles/r06_notes.md.)
                // fragment reads: k-step 0 and the A side of k-step 1 first; the W2 fragments of k-step 1 follow their k-step-0
                // counterparts into the registers those leave (56 registers of fragments would not fit beside 32 NT accumulators)
                h8 a[2][2], bf[NT];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[0][mi] = *(const h8 *)(smem + a_rd + q * 8192 + mi * 2048 + foff0);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) bf[ni] = *(const h8 *)(smem + b_rd + sb * kBS + ni * 2048 + foff0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[1][mi] = *(const h8 *)(smem + a_rd + q * 8192 + mi * 2048 + foff1);
                ARaw araw;
                a_fetch(araw, (q + 1) & 3);  // (also in the workgroup's last step, where nothing reads the result)
                __builtin_amdgcn_sched_barrier(0);
                if (pend) {
                    if (pend_next) a_store(o_pend, nxt, t_n1, pk);
                    else a_store(o_pend, cur, t_it, pk);
                }
                __builtin_amdgcn_sched_barrier(0);
                DT(5);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    if (e2) dma_b(k2, sb2, ni);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mi], bf[ni], acc[mi][ni], 0, 0, 0);
                    bf[ni] = *(const h8 *)(smem + b_rd + sb * kBS + ni * 2048 + foff1);
                }
                if (e3) {
                    if (x3) dma_a(nxt, t_n1, k3, (q + 3) & 3);
                    else dma_a(cur, t_it, k3, (q + 3) & 3);
                }
                __builtin_amdgcn_sched_barrier(0);
                DT(6);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][mi], bf[ni], acc[mi][ni], 0, 0, 0);
                }
                DT(3);
                // the conversion of the next chunk's A piece shares its scheduling region with the MFMAs of the second k-step (its
                // LDS reads were issued with the fragment reads)
                pend = e1, pend_next = x1, pk = k1;
                {
                    CellSt cc = cur;  // (a select, not a branch: the block stays one scheduling region)
                    cc.off = x1 ? nxt.off : cur.off, cc.flags = x1 ? nxt.flags : cur.flags;
                    cc.c0 = x1 ? nxt.c0 : cur.c0, cc.eb = x1 ? nxt.eb : cur.eb, cc.el = x1 ? nxt.el : cur.el;
                    o_pend = a_finish(araw, cc, k1, (q + 1) & 3);
                }
                __builtin_amdgcn_sched_barrier(0);
                DT(4);
#ifdef JH_TRACE
                ++tstep;
#endif
                sb = sb1;
            }
        }
        // ---- epilogue: acc[mi][ni][r] = S dh[row t_it + (r >> 2)][column u0 + 16 wm + 8 mi + 4 half + (r & 3)][unit JW wn + 32 ni + n]
        const bool last = !has_next;
#ifdef JH_TRACE
#define ET(k)                                                                                                   \
    do {                                                                                                        \
        if (tr && !first && tstep >= 22 && tstep < 70 && lane == 0) tr[154 + (k)] = (long long)__builtin_amdgcn_s_memtime();    \
    } while (0)
#else
#define ET(k) do { } while (0)
#endif
        ET(0);
        const int ln3 = launder(tid) & 63, n3 = ln3 & 31, half3 = ln3 >> 5;
        load_idx(1);  // (tile 0: at the top of the iteration's last step)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int jw = wn * JW + ni * 32;
            float rs[4] = {0.f, 0.f, 0.f, 0.f};  // per row of the iteration: sum over this wave's 16 columns of dz, this lane's unit
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int idx = 2 * ni + mi;
                if (idx + 2 < 2 * NT) load_idx(idx + 2);  // two tiles ahead (one ahead left ~1,000 clocks of latency exposed per tile)
                const EpiT &in = tin[idx % 3];
                const float(&ej)[4] = ejp[ni & 1];
                const int ubw = u0 + 16 * wm + 8 * mi;
                float4 *const cslot = (float4 *)jp.dCacc + ((((size_t)wg * 8 + wave) * 2 + mi) * NT + ni) * 64;  // uniform; + lane
                float4 cs = first ? make_float4(0.f, 0.f, 0.f, 0.f) : in.cs;
                float dz[16];
                // q = (1 - h^2) / 4 = r (1 - r) with r = 1 / (1 + e^{2(a + c)}); the factor 4 / S is applied once per output.  (The
                // branch AROUND the loops: inside, the compiler kept it per element -- 66 branches per unit tile, 52,000 clocks per
                // epilogue instead of ~15,000.)
                if (!slow) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float rr = __builtin_amdgcn_rcpf(fmaf(ej[r >> 2], in.pr[r & 3], 1.0f));
                        dz[r] = acc[mi][ni][r] * fmaf(-rr, rr, rr);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float h = htanh(ej[r >> 2] + in.pr[r & 3]);
                        dz[r] = acc[mi][ni][r] * (0.25f * fmaf(-h, h, 1.0f));
                    }
                }
                cs.x += (dz[0] + dz[4]) + (dz[8] + dz[12]);
                cs.y += (dz[1] + dz[5]) + (dz[9] + dz[13]);
                cs.z += (dz[2] + dz[6]) + (dz[10] + dz[14]);
                cs.w += (dz[3] + dz[7]) + (dz[11] + dz[15]);
#pragma unroll
                for (int rw = 0; rw < 4; ++rw) rs[rw] += (dz[4 * rw] + dz[4 * rw + 1]) + (dz[4 * rw + 2] + dz[4 * rw + 3]);
                if (!last) {
                    cslot[(uint32_t)ln3] = cs;
                } else {  // the strip is done: the d pred_proj partial of this row split
                    const float cv[4] = {cs.x, cs.y, cs.z, cs.w};
                    float *const crow_out = jp.dCpart + (((size_t)ts * p.B + b) * p.U) * J + jw;  // uniform
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq) {
                        const int uu = ubw + 4 * half3 + cq;
                        if (uu < p.U) crow_out[(uint32_t)uu * (uint32_t)J + (uint32_t)n3] = cv[cq] * c4;
                    }
                }
            }
            // d enc_proj partial rows of the unit tile: the two half-lanes of a unit, then the two column halves
            float *const cb = comb + (ni & 1) * 512;  // [4 rows][4 wn x 32 units]
#pragma unroll
            for (int rw = 0; rw < 4; ++rw) rs[rw] += __shfl_xor(rs[rw], 32);
            if (wm == 1 && ln3 < 32) {
#pragma unroll
                for (int rw = 0; rw < 4; ++rw) cb[rw * 128 + wn * 32 + n3] = rs[rw];
            }
            __syncthreads();
            if (wm == 0 && ln3 < 32) {
#pragma unroll
                for (int rw = 0; rw < 4; ++rw)
                    if (t_it + rw < t_end)
                        (jp.dApart + (((size_t)ut * p.B + b) * p.T + t_it + rw) * J + jw)[(uint32_t)n3] = (rs[rw] + cb[rw * 128 + wn * 32 + n3]) * c4;
            }
            if (ni == 0) ET(1);
            if (ni == 1) ET(2);
            if (ni == NT - 1) ET(3);
        }
        // the factors of the iteration after the next (its first chunk is converted in the last step of the next one).  (Their loads
        // issued two unit tiles earlier, the arithmetic here: 11 more live registers, spilled -- 13.7 against 13.1 ms at config 5.)
        CellSt nn = nxt;
        if (t_n2 < t_end) {
            cell_place(nn, t_n2);
            cell_factors(nn, t_n2);
        }
        ET(4);
        first = false;
        cur = nxt, nxt = nn;
        pend_next = false;  // (the piece converted in the last step belongs to what is now the current iteration)
        t_it = t_n1, t_n1 = t_n2;
    }
}

// ---------------------------------------------------------------------------------------------
// K4: dW2 = h^T . dl / S (and db2 = sum dl / S).  workgroup = (range of work units, 128-wide J tile, 512-wide V tile);
// 8 waves as 2 (64 joint units) x 4 (128 vocabulary columns), 2 x 4 accumulator tiles each.
// A work unit is (utterance, u-tile of 32, kTQ lattice rows); one lattice row (32 cells = 2 MFMA k-steps) per step.
// LDS per stage (three stages): dl rows [32 cells][2 VT + 64 B] (row-major, read transposed) | h^T fragments
// [2 ks][4 jb][2][32][8]; plus four 512-byte enc_proj row slices
// PARTIAL: V is not a multiple of 512 -- the last V tile holds 128, 256 or 384 columns; the waves of the empty 128-column
// groups keep the workgroup's barriers, DMA pieces and h^T staging and skip their products (a separate instantiation: the
// schedule of the full-tile kernel is left as it was tuned).
// ---------------------------------------------------------------------------------------------
#ifndef JH_K4_W256
#define JH_K4_W256 4  // waves per SIMD the 256-column instantiation is compiled for (4 = two workgroups per CU)
#endif
template <bool PARTIAL, int VT>
__global__ __launch_bounds__(512, VT == 512 ? 1 : JH_K4_W256) void jh_dw_kernel(const JhParams jp) {
    // VT = columns of a workgroup's V tile: 512 (one workgroup per CU, 126 KB of LDS), 256 or -- V = 128 only -- 128 (TWO per CU at 80 / 57 KB each: while one
    // sits in its barrier / fragment-read phase the other has the matrix pipe)
    constexpr int VB = VT / 128;       // 32-column accumulator blocks per wave
    constexpr int WCOL = VT / 4;       // columns per wave
    constexpr int kRow = 2 * VT + 64;  // bytes per dl row in LDS (+64: conflict-free transposed reads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LossParams &p = jp.lp;
    const int J = jp.J, V = p.V;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wj = wave & 1, wv = wave >> 1;
    constexpr int kDBytes = 32 * kRow, kHBytes = 2 * 4 * 32 * 32, kStage = kDBytes + kHBytes;

    uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_vt = (V + VT - 1) / VT, n_tiles = (J >> 7) * n_vt;
    const int tile = (int)(bid % (uint32_t)n_tiles), range = (int)(bid / (uint32_t)n_tiles);
    const int vt = tile % n_vt, jt = tile / n_vt;
    const int j0 = jt * 128, v0 = vt * VT;
    const int vw = PARTIAL ? min(VT, V - v0) : VT;     // columns of this V tile (a multiple of 128)
    const bool wv_live = !PARTIAL || wv * WCOL < vw;   // wave-uniform: this wave's columns exist
    constexpr bool kAllLanes = !PARTIAL && VT == 512;  // a dl row piece fills all 64 lanes of its LDS-DMA instruction
    // (round 6: how a range's units are picked -- see the unit loop below)

    f32x16 acc[2][VB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < VB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][q][r] = 0.f;
    float dbacc[VB];
#pragma unroll
    for (int q = 0; q < VB; ++q) dbacc[q] = 0.f;
    const bool do_db = (jt == 0) && (wj == 0);
    const int jl = tid & 127, cg = tid >> 7;  // h generation: this thread's joint unit and group of 8 cells
    const h2 ones = {(f16)1.0f, (f16)1.0f};

    const bool slow = jp.scal[2] != 0.f;  // kernel-uniform
    const float *Etab = slow ? jp.enc_proj : jp.expE, *Ptab = slow ? jp.pred_proj : jp.expP;
    char *ebuf = smem + 3 * kStage;  // [4][128] enc_proj values of the workgroup's joint units, one lattice row each
#ifdef JH_TRACE
    long long *tr = (blockIdx.x == 100) ? jp.trace + (size_t)(kTraceBlocks * 8 + wave) * kTraceSlots : nullptr;
    int tstep = 0;
#endif
    // the visited rows of the current unit, as byte offsets from its first row: [kTQ] bytes + a count, kept in the 64 padding bytes behind the
    // LAST dl row of the three stages (no DMA piece or h^T store touches them) -- NOT in extra LDS: the 256-column instantiation runs two
    // workgroups per CU at exactly 2 x 81,920 B, and 1 KB more per workgroup halved its occupancy (V = 256: 2.08 instead of 1.55 ms)
    static_assert(kTQ == 128, "the row list fills the padding of two stages, its count sits in the third's");
    auto rowlist = [&](const int k) -> unsigned char * { return (unsigned char *)(smem + (k >> 6) * kStage + 31 * kRow + 2 * VT + (k & 63)); };
    int *const rowcount = (int *)(smem + 2 * kStage + 31 * kRow + 2 * VT);
    for (int k = 0;; ++k) {
        // round 6: the units are dealt by WORK -- sorted by visited rows (jh_order_kernel), range r takes sorted positions r, 2R-1-r,
        // 2R+r, ... -- with the backward's row pruning a unit's work follows the alignment band (dealt in lattice order, a range's
        // consecutive units were busy or idle together: K4 9.3 instead of 5.3 ms at config 5; dealt every R-th in lattice order, a
        // range could draw only the short last units of the columns: 5.5 instead of 4.4 ms at V = 4096)
        int unit;
        if (jp.visit_all) {  // every row visited: the units weigh the same -- contiguous ranges, in lattice order (consecutive units are
                             // consecutive memory; PMC: the sorted deal fetched dl 2.0x instead of 1.3x, +0.7 ms at config 5)
            const int lo = (int)((long long)jp.n_units * range / jp.n_ranges), hi = (int)((long long)jp.n_units * (range + 1) / jp.n_ranges);
            if (lo + k >= hi) break;
            unit = lo + k;
        } else {
            const int pos = k * jp.n_ranges + ((k & 1) ? jp.n_ranges - 1 - range : range);
            if (pos >= jp.n_units) break;
            unit = jp.uorder[pos];
        }
        int q = unit;
        const int tq = q % jp.n_tq;
        q /= jp.n_tq;
        const int ut = q % jp.n_ut;
        const int b = q / jp.n_ut;
        const int u0 = ut * 32;
        const int Tb = length_T(p, b), Ub = length_U(p, b);
        const int t_begin = tq * kTQ, t_end = min(min(t_begin + kTQ, p.T), Tb);
        if (t_begin >= t_end || u0 >= Ub) continue;  // workgroup-uniform
        // the rows of the unit the backward visits (jh_rowbits_kernel), in order: wave 0 compacts the unit's 128 row bits
        // (every row visited: the list is the identity and is not built -- no bit fetch, no barriers in front of the unit)
        int nsteps = t_end - t_begin;
        if (!jp.visit_all) {
            wait_lgkm();
            __builtin_amdgcn_s_barrier();  // (the previous unit's last step has read its row list)
            if (wave == 0) {
                const uint32_t *bits = (const uint32_t *)(jp.live8 + ((size_t)b * jp.n_ut + ut) * (size_t)(4 * ((p.T + 31) >> 5))) + (t_begin >> 5);
                static_assert(kTQ == 128, "two ballots cover a unit");
                const bool a0 = t_begin + lane < t_end && ((bits[lane >> 5] >> (lane & 31)) & 1u);
                const bool a1 = t_begin + 64 + lane < t_end && ((bits[2 + (lane >> 5)] >> (lane & 31)) & 1u);
                const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int n0 = __builtin_popcountll(m0);
                if (a0) *rowlist(__builtin_popcountll(m0 & lt)) = (unsigned char)lane;
                if (a1) *rowlist(n0 + __builtin_popcountll(m1 & lt)) = (unsigned char)(64 + lane);
                if (lane == 0) rowcount[0] = n0 + __builtin_popcountll(m1);
            }
            wait_lgkm();
            __builtin_amdgcn_s_barrier();
            nsteps = __builtin_amdgcn_readfirstlane(rowcount[0]);
        }
        if (nsteps == 0) continue;  // workgroup-uniform
        // lattice row of step st (scalar: one LDS look per step in the main loop, not one per DMA piece)
        // (every row visited: the list is the identity -- plain arithmetic, no LDS round trip at the top of a step)
        auto row_of = [&](const int st) -> int {
            const int k = min(st, nsteps - 1);
            return t_begin + (jp.visit_all ? k : __builtin_amdgcn_readfirstlane((int)*rowlist(k)));
        };
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            pv[e] = Ptab[((size_t)b * p.U + min(u0 + 8 * cg + e, p.U - 1)) * J + j0 + jl];
        // Three stages of (dl rows + h^T fragments), enc_proj rows one step further ahead (four small buffers): at step s
        // the DMA of row s+2's dl and of row s+3's enc_proj slice are issued, row s+2's h^T is built, row s is multiplied.
        auto dma_e = [&](const int s, const int t) {  // wave 0, lanes 0..31: 128 floats of lattice row t (= row_of(s))
            if (wave == 0 && lane < 32) {
                const float *src = Etab + ((size_t)b * p.T + t) * J + j0 + lane * 4;
                lds_dma16(src, ebuf + (s & 3) * 512);
            }
        };
        auto dma_d = [&](const int t, char *st) {  // the dl rows of lattice row t
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // one dl row (512 columns = 1 KB) per wave-instruction
                const int i = wave + 8 * k;
                const f16 *src = (u0 + i < p.U) ? jp.dl + ((size_t)(b * p.T + t) * p.U + u0 + i) * V + v0 + lane * 8
                                                : jp.zrow + lane * 8;
                if (kAllLanes || lane * 8 < vw) lds_dma16(src, st + i * kRow);
            }
        };
        auto dma_d_piece = [&](const int t, char *st, const int k) {  // piece k (0..3) of the same
            const int i = wave + 8 * k;
            const f16 *src = (u0 + i < p.U) ? jp.dl + ((size_t)(b * p.T + t) * p.U + u0 + i) * V + v0 + lane * 8
                                            : jp.zrow + lane * 8;
            if (kAllLanes || lane * 8 < vw) lds_dma16(src, st + i * kRow);
        };
        auto build_h = [&](const int s, char *st) {
            const float ej = ((const float *)(ebuf + (s & 3) * 512))[jl];
            float hh[8];
            if (!slow) {
#pragma unroll
                for (int e = 0; e < 8; ++e) hh[e] = htanh2(ej, pv[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) hh[e] = htanh(ej + pv[e]);
            }
            h8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (u0 + 8 * cg + e < Ub) ? (f16)hh[e] : (f16)0.f;  // beyond U_b: no gradient
            // fragment image: [ks = cg >> 1][jb = jl >> 5][half = cg & 1][i = jl & 31][8 cells]  (lane-linear reads)
            *(h8 *)(st + kDBytes + ((((cg >> 1) * 4 + (jl >> 5)) * 2 + (cg & 1)) * 32 + (jl & 31)) * 16) = hv;
        };
        wait_lgkm();
        __builtin_amdgcn_s_barrier();  // the previous unit is done with every stage
        asm volatile("" ::: "memory");
        dma_e(0, row_of(0)), dma_e(1, row_of(1)), dma_e(2, row_of(2));
        wait_vm();
        __builtin_amdgcn_s_barrier();  // enc_proj rows 0..2 visible
        asm volatile("" ::: "memory");
        dma_d(row_of(0), smem);
        build_h(0, smem);
        if (nsteps > 1) {
            dma_d(row_of(1), smem + kStage);
            build_h(1, smem + kStage);
        }
        int c_s2 = row_of(2), c_s3 = row_of(3);
        for (int s = 0; s < nsteps; ++s) {
#ifdef JH_TRACE
            const bool tron = tr && tstep < 39;
            if (tron && lane == 0) tr[4 * tstep] = (long long)__builtin_amdgcn_s_memtime();
#endif
            if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // all but the newest stage's dl rows
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wait_lgkm();
            __builtin_amdgcn_s_barrier();  // stage s complete (dl rows + h^T); stage (s+2)%3 free; enc_proj row s+2 visible
            asm volatile("" ::: "memory");
#ifdef JH_TRACE
            if (tron && lane == 0) tr[4 * tstep + 1] = (long long)__builtin_amdgcn_s_memtime();
#endif
            // Stage s+2 is prepared WHILE stage s is multiplied: its five LDS-DMA pieces go out between the MFMA pairs of
            // the first k-step, its h^T values (8 tanh per thread) are computed between those of the second, and are
            // written to LDS at the end.  (Done up front, as a block, they cost 1100-1500 cycles per 32-cell step with
            // the matrix pipe idle: all eight waves are in the same phase after every barrier.)
            const bool pf = s + 2 < nsteps;
            const int t_s2 = c_s2, t_s3 = c_s3;  // (the rows whose pieces this step issues: looked up one step ago)
            char *stn = smem + ((s + 2) % 3) * kStage;
            float ejn = 0.f, hh[8];
            if (pf) ejn = ((const float *)(ebuf + ((s + 2) & 3) * 512))[jl];
#ifdef JH_TRACE
            if (tron && lane == 0) tr[4 * tstep + 2] = (long long)__builtin_amdgcn_s_memtime();
            if (tron && lane == 0) tr[4 * tstep + 3] = (long long)__builtin_amdgcn_s_memtime();
            ++tstep;
#endif
            const char *D = smem + (s % 3) * kStage, *H = D + kDBytes;
            const int g4 = lane >> 4, pl = lane & 15;
            // all fragment reads of the step first (2 x (2 A + 8 transposed B)), then MFMA pairs with the staging work
            // of stage s+2 between them
            h8 a[2][2];
            h4 blo[2][VB], bhi[2][VB];
            if (wv_live) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
                        a[ks][jb] = *(const h8 *)(H + ((ks * 4 + wj * 2 + jb) * 64 + lane) * 16);
                    const int row0 = ks * 16 + 8 * (g4 >> 1) + (pl >> 2);
#pragma unroll
                    for (int vb = 0; vb < VB; ++vb) {
                        const int col = wv * WCOL + vb * 32 + 16 * (g4 & 1) + 4 * (pl & 3);
                        const char *ad = D + row0 * kRow + col * 2;
                        blo[ks][vb] = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)ad));
                        bhi[ks][vb] = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(ad + 4 * kRow)));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int vb = 0; vb < VB; ++vb) {
                    const h4 lo4 = blo[ks][vb], hi4 = bhi[ks][vb];
                    if (wv_live) {
                        const h8 bf = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[0][vb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], bf, acc[0][vb], 0, 0, 0);
                        acc[1][vb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][1], bf, acc[1][vb], 0, 0, 0);
                    }
                    if (pf) {
                        constexpr int PP = 4 / VB, HP = 8 / VB;  // DMA pieces / h values per MFMA pair
                        if (ks == 0) {
                            if (vb == 0 && s + 3 < nsteps) dma_e(s + 3, t_s3);
#pragma unroll
                            for (int k = 0; k < PP; ++k) dma_d_piece(t_s2, stn, vb * PP + k);
                        } else {
                            // this thread's eight h values of row s+2, spread over the MFMA pairs of the second k-step
                            if (!slow) {
#pragma unroll
                                for (int e = vb * HP; e < (vb + 1) * HP; ++e) hh[e] = htanh2(ejn, pv[e]);
                            } else {
#pragma unroll
                                for (int e = vb * HP; e < (vb + 1) * HP; ++e) hh[e] = htanh(ejn + pv[e]);
                            }
                        }
                    }
                    if (do_db && wv_live) {
                        float sdb = dbacc[vb];
                        sdb = __builtin_amdgcn_fdot2(__builtin_shufflevector(lo4, lo4, 0, 1), ones, sdb, false);
                        sdb = __builtin_amdgcn_fdot2(__builtin_shufflevector(lo4, lo4, 2, 3), ones, sdb, false);
                        sdb = __builtin_amdgcn_fdot2(__builtin_shufflevector(hi4, hi4, 0, 1), ones, sdb, false);
                        sdb = __builtin_amdgcn_fdot2(__builtin_shufflevector(hi4, hi4, 2, 3), ones, sdb, false);
                        dbacc[vb] = sdb;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pf) {
                h8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (u0 + 8 * cg + e < Ub) ? (f16)hh[e] : (f16)0.f;  // beyond U_b: no gradient
                *(h8 *)(stn + kDBytes + ((((cg >> 1) * 4 + (jl >> 5)) * 2 + (cg & 1)) * 32 + (jl & 31)) * 16) = hv;
            }
            c_s2 = c_s3, c_s3 = row_of(s + 4);  // (behind the step's MFMAs: the look-up's LDS latency is not in front of the next step)
        }
    }
    const float invS = jp.scal[1];
    float *out = jp.dWpart + (size_t)range * J * V;
    if (!wv_live) return;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int vb = 0; vb < VB; ++vb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(size_t)(j0 + wj * 64 + jb * 32 + cdrow(r, half)) * V + v0 + wv * WCOL + vb * 32 + n] = acc[jb][vb][r] * invS;
    if (do_db) {
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
            float s = dbacc[vb];
            s += __shfl_xor(s, 32);
            if (lane < 32) jp.dbpart[(size_t)range * V + v0 + wv * WCOL + vb * 32 + n] = s * invS;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct JhLayout {
    WsLayout w;
    size_t W2Tp, W2c, dCacc, dl, pref, xbl, scal, expE, expP, b2l, dApart, dCpart, zrow, rowcnt, live8, order, uorder, dWpart, dbpart, total;
    int n_ut, n_tt, n_ts, TS, n_tq, n_units, n_ranges;
};

// Columns of K4's V tile: 512 = one workgroup per CU, 256 / 128 = two (see jh_dw_kernel).  Measured (profiles/r04_notes.md): 256-column
// tiles win while they are the only tile (V = 128: 4.97 -> 4.31 ms, V = 256: 4.40 -> 3.83 ms at B32 T600 U150) and lose beyond
// (V = 384: 7.89 -> 8.77 ms; config 5, V = 1024: K4 11.3 -> 15.5 ms -- every tile rebuilds the h^T image).  Round 5: V = 128 has its own
// 128-column instantiation -- all eight waves multiply (32 columns each) instead of four of a half-empty 256-column tile, 97 registers
// instead of 128 + 71 spilled: 4.32 -> 3.97 ms per step at the headline lattice (two 128-column tiles at V = 256: 5.32 -> 6.31 ms, not used).
static int k4_vt(int V) { return V <= 128 ? 128 : V <= 256 ? 256 : 512; }

bool joint_f16_supported(int J, int V) {
    // J: whole 128-unit tiles of K3 / K4 (K1 is instantiated per J); V: whole 128-column groups (four 32-column chunks share one
    // 8-byte reference word in K1; a wave of K4 owns 128 columns -- its 512-wide tile may be partly empty)
    return (J == 128 || J == 256 || J == 384 || J == 512 || J == 640) && V >= 128 && (V % 128) == 0 && V <= 8192;
}

static JhLayout make_jh_layout(int T, int U, int B, int J, int V) {
    JhLayout L;
    L.w = make_layout(T, U, B);
    L.n_ut = (U + 31) / 32;
    L.n_tt = (T + 7) / 8;
    // row splits of K3: its workgroups own (utterance, 32 columns, TS rows) x ALL joint units (round 6) -- as many splits as give about
    // JH_DHX_WGS workgroups (ten rounds of one per CU; config 5: 16), strips of at least 16 rows (four iterations of four)
    L.n_ts = (JH_DHX_WGS + B * L.n_ut - 1) / (B * L.n_ut);
    if (L.n_ts > (T + 15) / 16) L.n_ts = (T + 15) / 16;
    if (L.n_ts < 1) L.n_ts = 1;
    L.TS = ((T + L.n_ts - 1) / L.n_ts + 3) / 4 * 4;
    L.n_ts = (T + L.TS - 1) / L.TS;
    L.n_tq = (T + kTQ - 1) / kTQ;
    L.n_units = B * L.n_ut * L.n_tq;
    const int vt = k4_vt(V), n_tiles = (J / 128) * ((V + vt - 1) / vt);
    // K4: one workgroup per CU (two with 256-column tiles) -- as many ranges of cell units as fill the 256 CUs with n_tiles workgroups each
    // (config 5: 25 ranges x 10 tiles = 250 workgroups; 24 x 10, XCD-aligned, measured 4 % slower)
    L.n_ranges = (vt == 512 ? 256 : 512) / n_tiles;
    if (L.n_ranges < 1) L.n_ranges = 1;
    if (L.n_ranges > L.n_units) L.n_ranges = L.n_units;
    size_t off = L.w.total;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const size_t cells = (size_t)B * T * U;
    L.W2Tp = take((size_t)J * V * 2);
    L.W2c = take((size_t)J * V * 2);
    L.dCacc = take((size_t)B * L.n_ut * L.n_ts * 8 * 2 * (J / 128) * 64 * 16);
    L.dl = take(cells * V * 2);
    L.pref = take((size_t)B * T * (size_t)(V / 32) * (size_t)(L.n_ut * 32) * sizeof(short));
    L.xbl = take(cells * 2 * sizeof(float));
    L.scal = take(64);
    L.expE = take((size_t)B * T * J * sizeof(float));
    L.expP = take((size_t)B * U * J * sizeof(float));
    L.b2l = take((size_t)V * sizeof(float));
    L.dApart = take((size_t)L.n_ut * B * T * J * sizeof(float));
    L.dCpart = take((size_t)L.n_ts * B * U * J * sizeof(float));
    L.zrow = take(1024);
    L.rowcnt = take(256);  // (inside the region the backward zero-fills in front of its kernels: dCpart .. dWpart)
    L.live8 = take((size_t)B * L.n_ut * 4 * ((T + 31) / 32));
    L.order = take((size_t)B * L.n_ut * L.n_ts * sizeof(int));
    L.uorder = take((size_t)L.n_units * sizeof(int));
    L.dWpart = take((size_t)L.n_ranges * J * V * sizeof(float));
    L.dbpart = take((size_t)L.n_ranges * V * sizeof(float));
    L.total = off;
    return L;
}

hipError_t joint_f16_workspace_bytes(int T, int U, int B, int J, int V, size_t *bytes) {
    if (!joint_f16_supported(J, V) || sweep_K(U) == 0) return hipErrorInvalidValue;
    *bytes = make_jh_layout(T, U, B, J, V).total;
    return hipSuccess;
}

bool fill_loss_params(LossParams &p, const float *acts, float *grads, const int *labels, const int *label_lengths,
                      const int *input_lengths, const float *cost_scale, int V, int B, float *costs, void *workspace,
                      int maxT, int maxU, int blank);
hipError_t launch_reduce_partials(float *out, const float *in, int nparts, size_t n, hipStream_t s, unsigned *blockmax);
hipError_t launch_reduce_f16_backward(float *d_enc, const float *dApart, int n_ut, const LossParams &lp, int J, unsigned *bm_enc, const uint8_t *live8,
                                      float *d_pred, const float *dCpart, int nC, unsigned *bm_pred, float *dW2, const float *dWpart, float *db2,
                                      const float *dbpart, int nR, int V, hipStream_t s);
hipError_t launch_reduce_enc(float *out, const float *in, int n_ut, const LossParams &lp, int J, hipStream_t s, unsigned *blockmax,
                             const uint8_t *live8);

template <typename K>
static hipError_t set_lds_f16(K kernel, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int KS, int MODE>
static hipError_t launch_logits_mode(const JhParams &jp, unsigned grid, hipStream_t s) {
    // W2^T double buffer (+ staging tile when the epilogue writes [cells][V] rows), overlaid during the prologue by the
    // enc_proj / pred_proj images; bias table last
    size_t shm = 2 * (size_t)64 * (KS * 16) + ((MODE == 1 || MODE == 2) ? (size_t)8 * 32 * kStageStride * sizeof(f16) : 0);
    const size_t images = (size_t)KS * 4 * 32 * 16 + (size_t)8 * KS * 64;
    if (shm < images) shm = images;
    JhParams jq = jp;
    jq.b2_lds_off = -1;
    const size_t b2_bytes = (size_t)((jp.lp.V + 255) / 256 * 256) * 4;  // whole 1 KB pieces
    if (shm + b2_bytes <= 160 * 1024) {
        jq.b2_lds_off = (int)shm;
        shm += b2_bytes;
    }
    hipError_t e;
    if (jq.b2_lds_off >= 0) {
        if ((e = set_lds_f16(jh_logits_kernel<KS, MODE, true>, shm)) != hipSuccess) return e;
        hipLaunchKernelGGL((jh_logits_kernel<KS, MODE, true>), dim3(grid), dim3(512), shm, s, jq);
    } else {
        if ((e = set_lds_f16(jh_logits_kernel<KS, MODE, false>, shm)) != hipSuccess) return e;
        hipLaunchKernelGGL((jh_logits_kernel<KS, MODE, false>), dim3(grid), dim3(512), shm, s, jq);
    }
    return hipGetLastError();
}
template <int KS>
static hipError_t launch_logits(const JhParams &jp, int mode, unsigned grid, hipStream_t s);
template <int DUMMY = 0>
static hipError_t jh_launch_logits_for_J(const JhParams &jp, int J, int mode, unsigned tiles, hipStream_t s) {
    switch (J) {
        case 128: return launch_logits<8>(jp, mode, tiles, s);
        case 256: return launch_logits<16>(jp, mode, tiles, s);
        case 384: return launch_logits<24>(jp, mode, tiles, s);
        case 512: return launch_logits<32>(jp, mode, tiles, s);
        case 640: return launch_logits<40>(jp, mode, tiles, s);
    }
    return hipErrorInvalidValue;
}
template <int KS>
static hipError_t launch_logits(const JhParams &jp, int mode, unsigned grid, hipStream_t s) {
    switch (mode) {
        case 0: return launch_logits_mode<KS, 0>(jp, grid, s);
        case 1: return launch_logits_mode<KS, 1>(jp, grid, s);
        case 3: return launch_logits_mode<KS, 3>(jp, grid, s);
        default: return launch_logits_mode<KS, 2>(jp, grid, s);
    }
}

// {rows x 32-column tiles the last backward on this workspace visited, rows inside the utterances} of the f16 joint; {-1, -1} when
// the workspace does not hold the counts of a backward of this shape
hipError_t joint_f16_backward_rows(void *workspace, int T, int U, int B, int J, int V, int rows[2], hipStream_t s) {
    rows[0] = rows[1] = -1;
    if (!joint_f16_supported(J, V) || sweep_K(U) == 0) return hipErrorInvalidValue;
    const JhLayout L = make_jh_layout(T, U, B, J, V);
    int h[3] = {-1, -1, 0};
    hipError_t e = hipMemcpyAsync(h, (char *)workspace + L.rowcnt, 3 * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && h[2] == rowcnt_stamp(T, U, B, J, V)) rows[0] = h[0], rows[1] = h[1];
    return e;
}

// one word of workspace state, set on the stream between the kernels that depend on it
__global__ void jh_set_state_kernel(int *state, int value) { state[0] = value; }

static hipError_t jh_fill_params(JhParams &jp, const JhLayout &L, const float *enc_proj, const float *pred_proj, const float *W2,
                                 const float *b2, const int *labels, const int *label_lengths, const int *input_lengths,
                                 const float *cost_scale, int J, int V, int B, int T, int U, int blank, float *costs, void *workspace) {
    if (!fill_loss_params(jp.lp, nullptr, nullptr, labels, label_lengths, input_lengths, cost_scale, V, B, costs,
                          workspace, T, U, blank))
        return hipErrorInvalidValue;
    char *ws = (char *)workspace;
    jp.enc_proj = enc_proj, jp.pred_proj = pred_proj, jp.W2 = W2, jp.b2 = b2;
    jp.W2Tp = (f16 *)(ws + L.W2Tp), jp.W2c = (f16 *)(ws + L.W2c), jp.dCacc = (float *)(ws + L.dCacc), jp.dl = (f16 *)(ws + L.dl);
    jp.pref = (short *)(ws + L.pref);
    jp.xbl = (float *)(ws + L.xbl), jp.scal = (float *)(ws + L.scal);
    jp.state = (int *)(ws + L.scal) + 8;  // behind the words the prep kernel owns (zeroed before it runs, 32 bytes)
    jp.expE = (float *)(ws + L.expE), jp.expP = (float *)(ws + L.expP), jp.b2l = (float *)(ws + L.b2l);
    jp.dApart = (float *)(ws + L.dApart), jp.dCpart = (float *)(ws + L.dCpart);
    jp.dWpart = (float *)(ws + L.dWpart), jp.dbpart = (float *)(ws + L.dbpart);
    jp.zrow = (const f16 *)(ws + L.zrow);
    jp.live8 = (uint8_t *)(ws + L.live8), jp.rowcnt = (int *)(ws + L.rowcnt), jp.visit_all = 0;
    jp.order = (int *)(ws + L.order), jp.uorder = (int *)(ws + L.uorder);
    jp.J = J, jp.n_ut = L.n_ut, jp.n_tt = L.n_tt, jp.n_ts = L.n_ts, jp.TS = L.TS, jp.n_tq = L.n_tq;
    jp.n_units = L.n_units, jp.n_ranges = L.n_ranges;
    jp.logits_out = nullptr, jp.logits_only = 0;
    jp.b2_lds_off = -1;
#ifdef JH_TRACE
    jp.trace = nullptr;
#endif
    return hipSuccess;
}

// The joint alone on the f16 MFMA units, for decoding at large vocabularies (utils/decoding.py:6-18 evaluates dense_1 /
// dense_2 on one lattice cell per step; hparams.py:4: 4096 word pieces): K1 with every cell live and the f32 logits written out.
hipError_t launch_joint_logits_f16(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2, int J, int V,
                                   int B, int T, int U, float *logits, void *workspace, hipStream_t s) {
    if (!joint_f16_supported(J, V) || sweep_K(U) == 0) return hipErrorInvalidValue;
    if (((uintptr_t)enc_proj & 15) || ((uintptr_t)pred_proj & 15) || ((uintptr_t)b2 & 15) || ((uintptr_t)logits & 15)) return hipErrorInvalidValue;
    const JhLayout L = make_jh_layout(T, U, B, J, V);
    char *ws = (char *)workspace;
    // lengths and labels of the "everything is live" lattice: workspace regions the forward kernel does not touch
    int *il = (int *)(ws + L.dApart), *ll = il + B, *labels = (int *)(ws + L.dCpart);
    JhParams jp;
    hipError_t e = jh_fill_params(jp, L, enc_proj, pred_proj, W2, b2, labels, ll, il, nullptr, J, V, B, T, U, 0, nullptr, workspace);
    if (e != hipSuccess) return e;
    jp.logits_out = logits, jp.logits_only = 1;
    if (launch_fill(jp.scal, 0, 64, s) != hipSuccess) return hipErrorUnknown;  // prep words + the state word: nothing is parked
    hipLaunchKernelGGL(jh_prep_kernel, dim3(1024), dim3(256), 0, s, jp);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (launch_fill(jp.lp.W, kFillByte, L.w.A - L.w.W, s) != hipSuccess) return hipErrorUnknown;
    return jh_launch_logits_for_J(jp, J, 3, (unsigned)B * L.n_tt * L.n_ut, s);
}

hipError_t launch_joint_loss_f16(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                                 const int *labels, const int *label_lengths, const int *input_lengths,
                                 const float *cost_scale, int J, int V, int B, int T, int U, int blank, float *costs,
                                 float *d_enc_proj, float *d_pred_proj, float *dW2, float *db2, int phases,
                                 void *workspace, hipStream_t s, const JointHooks *hooks) {
    if (!joint_f16_supported(J, V)) return hipErrorInvalidValue;
    if (((uintptr_t)enc_proj & 15) || ((uintptr_t)pred_proj & 15) || ((uintptr_t)b2 & 15)) return hipErrorInvalidValue;
    if ((unsigned long long)B * T * J >= (1ull << 32) || (unsigned long long)B * U * J >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit indices in the reductions
    const JhLayout L = make_jh_layout(T, U, B, J, V);
    JhParams jp;
    hipError_t e = jh_fill_params(jp, L, enc_proj, pred_proj, W2, b2, labels, label_lengths, input_lengths, cost_scale, J, V, B, T, U,
                                  blank, costs, workspace);
    if (e != hipSuccess) return e;

#ifdef JH_TRACE
    static long long *trace_dev = nullptr;
    const size_t trace_bytes = (size_t)(kTraceBlocks + 2) * 8 * kTraceSlots * sizeof(long long);
    if (!trace_dev) hipMalloc(&trace_dev, trace_bytes);
    hipMemsetAsync(trace_dev, 0, trace_bytes, s);
    jp.trace = trace_dev;
#endif
    const unsigned tiles = (unsigned)B * L.n_tt * L.n_ut;
    auto logits = [&](int mode) -> hipError_t { return jh_launch_logits_for_J(jp, J, mode, tiles, s); };
    auto set_state = [&](int value) -> hipError_t {
        hipLaunchKernelGGL(jh_set_state_kernel, dim3(1), dim3(1), 0, s, jp.state, value);
        return hipGetLastError();
    };
    // Park the softmax numerators in the forward pass when a backward pass will use them: this call's own, or (phases bit 2,
    // the _fwd entry points) a later backward-only call on the same workspace.
    const bool want_bwd = (phases & 2) && d_enc_proj;
    const bool park = (phases & 1) && (want_bwd || (phases & 4));
    // the binary16 weight copies and the scale are rebuilt by whichever phase runs (cheap; W2 or cost_scale may differ)
    // (a forward phase also clears the state word behind the prep words -- dl is about to be overwritten --: one fill, not a launch of its own)
    if (launch_fill(jp.scal, 0, (phases & 1) ? 64 : 32, s) != hipSuccess) return hipErrorUnknown;
    hipLaunchKernelGGL(jh_prep_kernel, dim3(1024), dim3(256), 0, s, jp);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (phases & 1) {
        if (launch_fill(jp.lp.W, kFillByte, L.w.A - L.w.W, s) != hipSuccess) return hipErrorUnknown;
        if ((e = logits(park ? 1 : 0)) != hipSuccess) return e;
        if (park && (e = set_state(1)) != hipSuccess) return e;
#ifdef JH_TRACE
        {
            hipStreamSynchronize(s);
            std::vector<long long> h(trace_bytes / sizeof(long long));
            hipMemcpy(h.data(), trace_dev, trace_bytes, hipMemcpyDeviceToHost);
            const char *path = getenv("JH_TRACE_FILE");
            if (FILE *f = fopen(path ? path : "/tmp/jh_trace.bin", "wb")) {
                fwrite(h.data(), 1, trace_bytes, f);
                fclose(f);
            }
        }
#endif
        if ((e = launch_sweeps(jp.lp, s)) != hipSuccess) return e;
    }
    if (!(phases & 2) || !d_enc_proj) return hipSuccess;

    // dlogits: one streaming pass over the parked values, or -- when a backward call finds none (state != 1: they were
    // consumed by an earlier backward call, or the forward call did not park) -- the J x V product again.  Both kernels
    // are enqueued; the one whose precondition does not hold returns at once.
    // (Cutting the batch into utterance ranges and converting range q + 1 on a second stream beside the dh kernel of range q
    // was measured at config 5: the two kernels do overlap, and slow each other down by as much as the overlap hides.)
    // (a call that parked in its own forward phase knows the answer: the recompute kernel would read the state word and return)
    if (!((phases & 1) && park) && (e = logits(2)) != hipSuccess) return e;
    // dC partials + the zero row + the row counters (the dA partials need no zero-fill: launch_reduce_enc reads only the rows K3 writes)
    if (launch_fill(jp.dCpart, 0, L.live8 - L.dCpart, s) != hipSuccess) return hipErrorUnknown;
    // which lattice rows (x 32-column tiles) the backward visits: K3, K4 and the d enc_proj reduction follow these bits
    jp.visit_all = (phases & 8) ? 1 : 0;
    hipLaunchKernelGGL(jh_rowbits_kernel, dim3((unsigned)B * L.n_ut * ((T + kRowbitsRows - 1) / kRowbitsRows)), dim3(256), 0, s, jp);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(jh_order_kernel, dim3(1), dim3(1024), 0, s, jp, B * L.n_ut * L.n_ts);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    {
        const size_t shm = 3 * (size_t)(J * 64) + 4 * 8192 + 4 * 256 + 4096 + 128;  // W2 stages, A stages, reference tiles, hand-over rows, labels
        const unsigned grid = (unsigned)B * L.n_ut * L.n_ts;
        auto go = [&](auto kernel) -> hipError_t {
            hipError_t e2 = set_lds_f16(kernel, shm);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), shm, s, jp);
            return hipGetLastError();
        };
        switch (J / 128) {
            case 1: e = go(jh_dhx_kernel<1>); break;
            case 2: e = go(jh_dhx_kernel<2>); break;
            case 3: e = go(jh_dhx_kernel<3>); break;
            case 4: e = go(jh_dhx_kernel<4>); break;
            default: e = go(jh_dhx_kernel<5>); break;
        }
        if (e != hipSuccess) return e;
    }
    if ((e = set_state(2)) != hipSuccess) return e;
    {
        const int vt = k4_vt(V);
        const size_t shm = 3 * (size_t)(32 * (2 * vt + 64) + 2 * 4 * 32 * 32) + 4 * 512;  // stages (the unit's row list in their padding), enc row slices
        const unsigned grid = (unsigned)L.n_ranges * (J / 128) * ((V + vt - 1) / vt);
        auto go = [&](auto kernel) -> hipError_t {
            hipError_t e2 = set_lds_f16(kernel, shm);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), shm, s, jp);
            return hipGetLastError();
        };
        if (vt == 128) e = go(jh_dw_kernel<false, 128>);
        else if (vt == 256) e = go(jh_dw_kernel<false, 256>);  // (V = 256: the tile is full)
        else e = (V % 512 == 0) ? go(jh_dw_kernel<false, 512>) : go(jh_dw_kernel<true, 512>);
        if (e != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    // d enc_proj, d pred_proj, dW2, db2 from their partials: one launch (joint_kernels.hip)
    e = launch_reduce_f16_backward(d_enc_proj, jp.dApart, L.n_ut, jp.lp, J, hooks ? hooks->dmax_enc : nullptr, jp.live8, d_pred_proj, jp.dCpart, L.n_ts,
                                   hooks ? hooks->dmax_pred : nullptr, dW2, jp.dWpart, db2, jp.dbpart, L.n_ranges, V, s);
#ifdef JH_TRACE
    {
        hipStreamSynchronize(s);
        std::vector<long long> h(trace_bytes / sizeof(long long));
        hipMemcpy(h.data(), trace_dev, trace_bytes, hipMemcpyDeviceToHost);
        const char *path = getenv("JH_TRACE_FILE");
        if (FILE *f = fopen(path ? path : "/tmp/jh_trace.bin", "wb")) {
            fwrite(h.data(), 1, trace_bytes, f);
            fclose(f);
        }
    }
#endif
    return e;
}

}  // namespace rnnt
