// rnnt_cell.h -- device helpers shared by the loss kernels (rnnt_kernels.hip) and the fused joint
// kernels (joint_kernels.hip): lattice-cell decoding and the per-cell gradient set-up.
#pragma once
#include "rnnt_common.h"

namespace rnnt {

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
        r.Tb = p.input_lengths[b];
        r.Ub = p.label_lengths[b] + 1;
        r.valid = (r.t < r.Tb) && (r.u < r.Ub);
    }
    return r;
}

__device__ __forceinline__ int clamp_label(int lab, int V) { return min(max(lab, 0), V - 1); }

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

__device__ __forceinline__ CellGrad cell_grad_setup(const LossParams &p, const Cell &cl, uint32_t c) {
    CellGrad g;
    const int n = cl.t + cl.u;
    const size_t sk = ((size_t)cl.b * p.Nr + n) * p.Up + cl.u;
    const float a = p.A[sk];
    const float bt = p.Bt[sk];
    // offsets are kept per (block of kRebase diagonals, group of 64 lattice columns)
    const int kc = n / kRebase, kc1 = (n + 1) / kRebase;
    const int g0 = cl.u >> 6, g1 = (cl.u + 1) >> 6;
    const size_t ob = (size_t)cl.b * p.NC * p.NG;
    const double oa = p.offA[ob + (size_t)kc * p.NG + g0];
    const double ll2 = p.ll[2 * cl.b];
    const float E0 = (float)(oa + p.offB[ob + (size_t)kc * p.NG + g0] - ll2);
    g.scale = p.cost_scale ? p.cost_scale[cl.b] : 1.0f;
    g.nl = -p.lse[c] * kLog2e;
    g.c0 = (a + bt) + E0 + g.nl;
    g.has_blank_corr = true;
    if (cl.t < cl.Tb - 1)
        g.cb = a + p.Bt[sk + p.Up] + (float)(oa + p.offB[ob + (size_t)kc1 * p.NG + g0] - ll2);
    else if (cl.u == cl.Ub - 1)
        g.cb = a + (float)(oa - ll2);
    else {
        g.cb = 0.f;
        g.has_blank_corr = false;
    }
    g.has_label = cl.u < cl.Ub - 1;
    g.lab = 0;
    g.cl = 0.f;
    if (g.has_label) {
        g.lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], p.V);
        g.cl = a + p.Bt[sk + p.Up + 1] + (float)(oa + p.offB[ob + (size_t)kc1 * p.NG + g1] - ll2);
    }
    return g;
}

}  // namespace rnnt
