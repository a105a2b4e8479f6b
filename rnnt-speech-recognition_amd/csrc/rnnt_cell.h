// rnnt_cell.h -- device helpers shared by the loss kernels (rnnt_kernels.hip) and the fused joint
// kernels (joint_kernels.hip): lattice-cell decoding and the per-cell gradient set-up.
#pragma once
#include "rnnt_common.h"

namespace rnnt {

// Lengths as the kernels use them: clamped into the tensor, so that a caller's out-of-range T_b / L_b can never index
// outside acts / grads / the workspace slabs.  An utterance whose lengths had to be clamped is reported with a NaN cost
// and NaN gradients (lengths_invalid, consumed by the alpha sweep) instead of a plausible number.
__device__ __forceinline__ int length_T(const LossParams &p, const int b) { return min(max(p.input_lengths[b], 1), p.T); }
__device__ __forceinline__ int length_U(const LossParams &p, const int b) { return min(max(p.label_lengths[b] + 1, 1), p.U); }
__device__ __forceinline__ bool lengths_invalid(const LossParams &p, const int b) {
    const int t = p.input_lengths[b], l = p.label_lengths[b];
    return t < 1 || t > p.T || l < 0 || l > p.U - 1;
}

struct Cell {
    int b, t, u, Tb, Ub;
    bool valid;
};

__device__ __forceinline__ Cell decode(const LossParams &p, uint32_t c) {
    Cell r;
    r.b = r.t = r.u = r.Tb = r.Ub = 0;
    r.valid = false;
    if (c < p.cells) {
        const uint32_t row = fdiv(c, p.divU);
        r.u = (int)(c - row * (uint32_t)p.U);
        const uint32_t b = fdiv(row, p.divT);
        r.t = (int)(row - b * (uint32_t)p.T);
        r.b = (int)b;
        r.Tb = length_T(p, (int)b);
        r.Ub = length_U(p, (int)b);
        r.valid = (r.t < r.Tb) && (r.u < r.Ub);
    }
    return r;
}

__device__ __forceinline__ int clamp_label(int lab, int V) { return min(max(lab, 0), V - 1); }

// ---- write-through (sc1) stores / L2-served (sc1) loads (MI355X guide G16) ----
// The sweeps store the lattice state write-through: nothing on their XCD re-reads it, and the gradient pass runs on all XCDs.
template <bool SC1>
__device__ __forceinline__ float ld_f32(const float *q) {
    return SC1 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
}
template <bool SC1>
__device__ __forceinline__ double ld_f64(const double *q) {
    return SC1 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
}
__device__ __forceinline__ void st_f32_wt(float *q, float v) {
    __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_f64_wt(double *q, double v) {
    __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// What one valid cell needs from the lattice to form its gradient (all log2 domain).
struct CellGrad {
    float c0;     // alpha + beta - ll - lse*log2e  (add x*log2e -> log2 of softmax*occupancy)
    float nl;     // -lse*log2e
    float cb;     // alpha + beta(t+1,u) - ll   (blank correction exponent base) or terminal
    float cl;     // alpha + beta(t,u+1) - ll   (label correction exponent base)
    bool has_blank_corr, has_label;
    int lab;
    float scale;
};

// The set-up from the cell's lattice values: a = alpha~(t,u), bt = beta~(t,u), b_t1 = beta~(t+1,u), b_u1 = beta~(t,u+1) (the
// last two are only looked at where that neighbour exists), all relative to the offset tables.
template <bool SC1 = false>
__device__ __forceinline__ CellGrad cell_grad_from(const LossParams &p, const Cell &cl, uint32_t c, const float a, const float bt,
                                                   const float b_t1, const float b_u1) {
    CellGrad g;
    const int n = cl.t + cl.u;
    // offsets are kept per (block of kRebase diagonals, group of OG lattice columns: one sweep lane)
    const int kc = n / kRebase, kc1 = (n + 1) / kRebase;
    const int g0 = (int)fdiv((uint32_t)cl.u, p.divOG), g1 = (int)fdiv((uint32_t)cl.u + 1u, p.divOG);
    const size_t ob = (size_t)cl.b * p.NC * p.NG;
    const double oa = (double)ld_f32<SC1>(p.offA + ob + (size_t)kc * p.NG + g0);
    const double ll2 = ld_f64<SC1>(p.ll + 2 * cl.b);
    const double da = (double)a + (oa - ll2);  // alpha(t,u) - ll: residue + offset, formed in f64 and rounded ONCE below
    g.scale = p.cost_scale ? p.cost_scale[cl.b] : 1.0f;
    g.nl = -p.lse[c] * kLog2e;
    g.c0 = (float)(da + ((double)bt + (double)ld_f32<SC1>(p.offB + ob + (size_t)kc * p.NG + g0))) + g.nl;
    g.has_blank_corr = true;
    if (cl.t < cl.Tb - 1)
        g.cb = (float)(da + ((double)b_t1 + (double)ld_f32<SC1>(p.offB + ob + (size_t)kc1 * p.NG + g0)));
    else if (cl.u == cl.Ub - 1)
        g.cb = (float)da;
    else {
        g.cb = 0.f;
        g.has_blank_corr = false;
    }
    g.has_label = cl.u < cl.Ub - 1;
    g.lab = 0;
    g.cl = 0.f;
    if (g.has_label) {
        g.lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], p.V);
        g.cl = (float)(da + ((double)b_u1 + (double)ld_f32<SC1>(p.offB + ob + (size_t)kc1 * p.NG + g1)));
    }
    return g;
}

template <bool SC1 = false>
__device__ __forceinline__ CellGrad cell_grad_setup(const LossParams &p, const Cell &cl, uint32_t c) {
    const int n = cl.t + cl.u;
    const size_t sk = ((size_t)cl.b * p.Nr + n) * p.Up + cl.u;
    const float a = ld_f32<SC1>(p.A + sk);
    const float bt = ld_f32<SC1>(p.Bt + sk);
    const float b_t1 = (cl.t < cl.Tb - 1) ? ld_f32<SC1>(p.Bt + sk + p.Up) : 0.f;
    const float b_u1 = (cl.u < cl.Ub - 1) ? ld_f32<SC1>(p.Bt + sk + p.Up + 1) : 0.f;
    return cell_grad_from<SC1>(p, cl, c, a, bt, b_t1, b_u1);
}

}  // namespace rnnt
