// rnnt_cellwave.h -- one lattice cell per WAVE (any vocabulary size, any alignment): the log-softmax denominator + edge
// weights ("lsm") and the fused-softmax gradient of a range of cells.  Shared by rnnt_kernels.hip (cell_wave_kernel) and
// rnnt_lin_kernels.hip (the log-domain redo of an utterance).  Replaces warp-transducer's log_softmax / compute_grad_kernel
// (SURVEY.md 2.1, 8a-6 / a-9).
#pragma once
#include "rnnt_sweep.h"

namespace rnnt {

// ---------------------------------------------------------------------------------------------
// General path (any V, any alignment): one lattice cell per WAVE, lanes stride over V.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void online_upd(float &m, float &s, float xv) {
    const float mn = fmaxf(m, xv);
    s = s * ex2((m - mn) * kLog2e) + ex2((xv - mn) * kLog2e);
    m = mn;
}

// consecutive cells one wave of cell_wave_kernel walks: about 4 KB of logits
__host__ __device__ inline int wave_cells(int V) { return V >= 1024 ? 1 : 1024 / V; }

// The cells [c_lo, c_hi) by ONE wave, lanes striding over V.  SC1: read the lattice state with agent-scope loads (the caller
// wrote it itself earlier in the same kernel: rnnt_lin_kernels.hip's log-domain redo).
constexpr int kOccFloorWave = 50;  // (the fused joint's backward uses the same floor per row tile: joint_kernels.hip kOccFloor)
template <bool V4, bool GRAD, bool SC1 = false>
__device__ __forceinline__ void cell_wave_range(const LossParams &p, const uint32_t c_lo, const uint32_t c_hi, const int lane) {
    const int V = p.V;
    for (uint32_t c = c_lo; c < c_hi; ++c) {
        const Cell cl = decode(p, c);
        const float *x = p.acts + (size_t)c * V;
        if (!GRAD) {
            if (!cl.valid) continue;
            float m = -FLT_MAX, s = 0.f;
            if (V4 && V <= 2048) {
                // the lane's share of the row fits 8 float4 registers: one pass for the maximum, one exponential per
                // logit (the online update below costs two) -- this pass was VALU-bound at V = 1024
                float4 q[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = lane * 4 + k * 256;
                    q[k] = (i < V) ? *(const float4 *)(x + i) : make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
                    m = fmaxf(fmaxf(m, fmaxf(q[k].x, q[k].y)), fmaxf(q[k].z, q[k].w));
                }
                const float nm = -m * kLog2e;  // lanes beyond the row keep m = -FLT_MAX, s = 0 (merged below like any other)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (lane * 4 + k * 256 < V)
                        s += (ex2(fmaf(q[k].x, kLog2e, nm)) + ex2(fmaf(q[k].y, kLog2e, nm))) +
                             (ex2(fmaf(q[k].z, kLog2e, nm)) + ex2(fmaf(q[k].w, kLog2e, nm)));
            } else if (V4) {
                for (int i = lane * 4; i < V; i += 256) {
                    const float4 q = *(const float4 *)(x + i);
                    online_upd(m, s, q.x), online_upd(m, s, q.y), online_upd(m, s, q.z), online_upd(m, s, q.w);
                }
            } else {
                for (int i = lane; i < V; i += 64) online_upd(m, s, x[i]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float mo = __shfl_xor(m, off), so = __shfl_xor(s, off);
                const float M = fmaxf(m, mo);
                s = s * ex2((m - M) * kLog2e) + so * ex2((mo - M) * kLog2e);
                m = M;
            }
            if (lane == 0) {
                const float lg2s = lg2(s);
            const float lse = m + kLn2 * lg2s;
                const bool blank_stays = (cl.t < cl.Tb - 1) || (cl.u == cl.Ub - 1);
                const float ob = blank_stays ? fmaf(x[p.blank] - m, kLog2e, -lg2s) : kNeg;
                float ol = kNeg;
                if (cl.u < cl.Ub - 1) {
                    const int lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], V);
                    ol = fmaf(x[lab] - m, kLog2e, -lg2s);
                }
                p.lse[c] = lse;
                const size_t wi = ((size_t)cl.b * p.Nr + (cl.t + cl.u)) * p.Up + cl.u;
                ((float2 *)p.W)[wi] = make_float2(ob, ol);
            }
        } else {
            float *gd = p.grads + (size_t)c * V;
            if (!cl.valid) {
                if (V4) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int i = lane * 4; i < V; i += 256) *(float4 *)(gd + i) = z;
                } else {
                    for (int i = lane; i < V; i += 64) gd[i] = 0.f;
                }
                continue;
            }
            const CellGrad g = cell_grad_setup<SC1>(p, cl, c);
            // A cell without mass: every gradient of it is bounded by 2 |cost_scale| x its occupancy alpha beta / L = 2^(c0 - nl); below
            // 2^-kOccFloorWave that is nothing an f32 sum holds, and the cell's V logits need not be read: zeros are written (the
            // reference leaves ~1e-15 there).  On an unstructured 1500 x 300 lattice that is most of the cells.  NaN counts as occupied.
            if (!p.visit_all && g.c0 - g.nl <= (float)-kOccFloorWave) {
                if (V4) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f z = {0.f, 0.f, 0.f, 0.f};
                    for (int i = lane * 4; i < V; i += 256) __builtin_nontemporal_store(z, (v4f *)(gd + i));
                } else {
                    for (int i = lane; i < V; i += 64) gd[i] = 0.f;
                }
                continue;
            }
            const float corr_b = g.has_blank_corr ? g.scale * ex2(fmaf(x[p.blank], kLog2e, g.nl) + g.cb) : 0.f;
            const float corr_l = g.has_label ? g.scale * ex2(fmaf(x[g.lab], kLog2e, g.nl) + g.cl) : 0.f;
            const int lab = g.has_label ? g.lab : -1;
            if (V4) {
                for (int i = lane * 4; i < V; i += 256) {
                    const float4 q = *(const float4 *)(x + i);
                    float r[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float gv = g.scale * ex2(fmaf(r[k], kLog2e, g.c0));
                        if (i + k == p.blank) gv -= corr_b;
                        if (i + k == lab) gv -= corr_l;
                        r[k] = gv;
                    }
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f out = {r[0], r[1], r[2], r[3]};
                    __builtin_nontemporal_store(out, (v4f *)(gd + i));  // written once, never re-read by this op
                }
            } else {
                for (int i = lane; i < V; i += 64) {
                    float gv = g.scale * ex2(fmaf(x[i], kLog2e, g.c0));
                    if (i == p.blank) gv -= corr_b;
                    if (i == lab) gv -= corr_l;
                    gd[i] = gv;
                }
            }
        }
    }
}



}  // namespace rnnt
