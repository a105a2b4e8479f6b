// rnnt_lin.h -- the LINEAR-domain lattice of the small-vocabulary loss path (round 4): shared definitions and the per-cell
// device code of the two cell passes.  Replaces the same stages of warp-transducer's GPU path as rnnt_kernels.hip does
// (SURVEY.md 2.1, 8a-6 ... a-9; call site utils/loss.py:34-35), with the transcendentals taken off the sweeps' serial chain:
//
//   lsm pass    edge PROBABILITIES {p(blank), p(label)} per cell (zeros where an edge leaves the lattice), no lse store
//   sweeps      alpha^ / beta^ as float32 mantissas x 2^frame, one integer frame per sweep lane (K lattice columns) and block
//               of kLinR diagonals: the step is multiply / add only, the renormalisation is exact (powers of two)
//   grad pass   softmax numerators recomputed from the logits it reads anyway; occupancies formed from mantissas + frames
//
// Fixed frames can lose mass that falls more than 126 bits below its lane's frame.  Whether that mattered is decided per cell
// by the gradient pass (lin_certificate below: what a flush can have cost, times the other side's mass, over the likelihood)
// and per utterance by the sweeps (non-finite / zero likelihood, alpha-side vs beta-side likelihood).  An utterance that fails
// is redone in the log domain by lin_redo_kernel (rnnt_lin_kernels.hip) -- log2 values, recurrence in float64 (rnnt_sweep.h alpha_sweep_pr), any range -- so the
// results never depend on the shortcut.  tests/tools/emulate_linear.py restates all of it in NumPy.
#pragma once
#include "rnnt_sweep.h"

namespace rnnt {

// Diagonals per frame block: a lane renormalises (and the frame tables get a row) every 2^shift diagonals.  The renormalisation
// is ~22 instructions of the sweeping wave (4 us of the sweep at B32 T600 U150 with blocks of four), so blocks are EIGHT diagonals
// long where the mass decays slowly and FOUR where it decays fast -- chosen per utterance by the sweeps from a statistic the lsm
// pass leaves behind: the mean of -log2 max(p_blank, p_label) over the utterance's cells (bits of mass lost per diagonal along the
// better edge).  N(0,1) logits: 4.7 bits at 28 symbols, 5.8 at 60 (certificate margin -84 / -72 bits with blocks of eight);
// trained-like posteriors 0 ... 3; 3 x N(0,1): 7.3 (margin -43 with blocks of eight, -69 with four); 4 x N(0,1): 9.1 (-45 with four).
// K = 1, 12, 16: always four (K = 1: the frame look-back would be 8 lanes deep; 12, 16: their LDS chunks are four diagonals long).
__host__ __device__ constexpr int lin_shift_max(int K) { return (K == 1 || K >= 12) ? 2 : 3; }
constexpr float kLinDecayBits = 6.2f;  // mean bits per diagonal beyond which an utterance gets blocks of four
constexpr int kLinDrag = 118;            // a lane's frame is at most this far below the lanes mass can reach it from within a block
constexpr int kFrameNone = -(1 << 28);   // frame of a lane without mass and without a neighbour to copy from
constexpr int kCertBits = -40;           // per-cell bound (bits) on flush loss x other side / likelihood
constexpr float kTinyEdge = 7.8886091e-31f;  // 2^-100: an edge the lattice owns below this goes to the log-domain path (NaN in W)

// per-utterance words in LossParams::flags
enum { kFlagA = 0, kFlagB = 1, kFlagG = 2, kFlagState = 3 };  // state: 0 linear lattice, 2 log-domain lattice ready (after a redo)

__device__ __forceinline__ void st_i32_wt(int *q, int v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_i32_sc1(const int *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float frexp_m(float x) { return __builtin_amdgcn_frexp_mantf(x); }  // in [0.5, 1); 0 -> 0
__device__ __forceinline__ int frexp_e(float x) { return __builtin_amdgcn_frexp_expf(x); }      // 0 for x == 0
__device__ __forceinline__ float ldexp_f(float x, int e) { return __builtin_ldexpf(x, e); }     // v_ldexp_f32

// True when the utterance's lattice is not (or no longer) in the linear format: the fast gradient pass leaves it alone.
__device__ __forceinline__ bool lin_skip(const LossParams &p, const int b) {
    const int *fl = p.flags + 4 * b;
    return (fl[kFlagA] | fl[kFlagB]) != 0 || fl[kFlagState] == 2;
}

// ---- lsm: one valid cell, its V logits in x[] (registers) and at xs (its slot of the patch image in LDS) ----
// Returns -log2 max(p_blank, p_label) of the cell (the decay statistic; 200 where an edge had to be refused).
template <int VP>
__device__ __forceinline__ float lin_cell_lsm(const LossParams &p, const Cell &cl, const float (&x)[VP], const float *xs) {
    float m = x[0];
#pragma unroll
    for (int i = 1; i < VP; ++i) m = fmaxf(m, x[i]);
    float s = 0.f;
    const float nml = -m * kLog2e;
#pragma unroll
    for (int i = 0; i < VP; ++i) s += ex2(fmaf(x[i], kLog2e, nml));
    const float inv = __builtin_amdgcn_rcpf(s);
    // a blank from the last frame leaves the lattice unless it is THE terminal transition
    const bool blank_stays = (cl.t < cl.Tb - 1) || (cl.u == cl.Ub - 1);
    float pb = 0.f, pl = 0.f;
    if (blank_stays) {
        pb = ex2(fmaf(xs[p.blank], kLog2e, nml)) * inv;
        if (!(pb >= kTinyEdge)) pb = NAN;  // (also a NaN logit): not representable here -> the sweeps hand the utterance back
    }
    if (cl.u < cl.Ub - 1) {
        const int lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], p.V);
        pl = ex2(fmaf(xs[lab], kLog2e, nml)) * inv;
        if (!(pl >= kTinyEdge)) pl = NAN;
    }
    // The two edge probabilities go into the first two floats of the cell's own LDS slot (nobody else reads it); the patch kernel
    // writes them to the skewed edge array diagonal by diagonal afterwards (rnnt_kernels.hip: consecutive lanes then store
    // consecutive positions of one diagonal's row -- written from here, a lane per cell, every store of a wave went to 64 rows).
    float *const xo = const_cast<float *>(xs);
    xo[0] = pb, xo[1] = pl;
    const float best = fmaxf(pb, pl);  // (v_max ignores a NaN operand)
    return (pb != pb || pl != pl) ? 200.f : -lg2(fmaxf(best, 1.0e-37f));
}

// ---- gradient set-up of one valid cell from mantissas + frames ----
struct LinGrad {
    float h0;  // cost_scale * occupancy            (x e_v / s = the softmax term)
    float hb;  // cost_scale * alpha p(..) beta(t+1,u) / L without the p: multiplies e_blank / s
    float hl;  // the same for the label edge
    bool has_blank_corr, has_label, bad;
    int lab;
};

__device__ __forceinline__ LinGrad lin_grad_setup(const LossParams &p, const Cell &cl) {
    LinGrad g;
    const int n = cl.t + cl.u;
    const size_t sk = ((size_t)cl.b * p.Nr + n) * p.Up + cl.u;
    const float ma = p.A[sk], mb = p.Bt[sk];
    const int sh = p.lshift[cl.b];  // the block length the sweeps chose for this utterance
    const int kc = n >> sh, kc1 = (n + 1) >> sh;
    const int l0 = (int)fdiv((uint32_t)cl.u, p.divOG), l1 = (int)fdiv((uint32_t)cl.u + 1u, p.divOG);
    const size_t tb = (size_t)cl.b * p.NCl * 64;
    const int ea = p.EA[tb + (size_t)kc * 64 + l0], eb = p.EB[tb + (size_t)kc * 64 + l0];
    const float mL = p.lik[4 * cl.b];
    const int EL = ((const int *)p.lik)[4 * cl.b + 1];
    const float scale = p.cost_scale ? p.cost_scale[cl.b] : 1.0f;
    // alpha / L as (qa, base): mantissas may sit anywhere in the f32 range (a dragged frame), so split before multiplying
    const int xa = frexp_e(ma), xb = frexp_e(mb);
    const float qa = scale * frexp_m(ma) * __builtin_amdgcn_rcpf(mL);
    const int base = ea + xa - EL;
    g.h0 = ldexp_f(qa * frexp_m(mb), base + eb + xb);
    g.has_blank_corr = true;
    g.hb = 0.f;
    if (cl.t < cl.Tb - 1) {
        const float m1 = p.Bt[sk + p.Up];
        g.hb = ldexp_f(qa * frexp_m(m1), base + p.EB[tb + (size_t)kc1 * 64 + l0] + frexp_e(m1));
    } else if (cl.u == cl.Ub - 1) {
        g.hb = ldexp_f(qa, base);  // the terminal transition: beta of the virtual end node is 1
    } else {
        g.has_blank_corr = false;
    }
    g.has_label = cl.u < cl.Ub - 1;
    g.lab = 0;
    g.hl = 0.f;
    if (g.has_label) {
        g.lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], p.V);
        const float m1 = p.Bt[sk + p.Up + 1];
        g.hl = ldexp_f(qa * frexp_m(m1), base + p.EB[tb + (size_t)kc1 * 64 + l1] + frexp_e(m1));
    }
    // Certificate.  A product or sum below 2^-126 of its frame is flushed (or loses bits as a denormal): alpha^ of this cell is
    // short of alpha by at most ~2^(ea-126), which can reach the likelihood through at most beta(cell); the same for beta^; and
    // where both sides are zero, through 2^(ea-126) 2^(eb-126).  All three must be negligible against the likelihood.
    int worst = ea + eb - 252 - EL;
    if (mb != 0.f) worst = max(worst, ea - 126 + eb + xb - EL);
    if (ma != 0.f) worst = max(worst, eb - 126 + ea + xa - EL);
    g.bad = worst > kCertBits || !(ma <= FLT_MAX) || !(mb <= FLT_MAX) || !(ma >= 0.f) || !(mb >= 0.f);
    return g;
}

}  // namespace rnnt
