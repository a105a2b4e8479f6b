// rnnt_cellbody.h -- everything ONE LANE does for its lattice cell in the two cell passes of the small-vocabulary loss
// (V <= 64 logits in registers): the log-softmax denominator + edge weights ("lsm") and the fused-softmax gradient, in the
// log2-domain and in the linear-domain format (rnnt_lin.h).  Used by the patch kernels (rnnt_kernels.hip: the logits sit in
// LDS and the gradients replace them there) and by the log-domain redo of an utterance (rnnt_lin_kernels.hip: straight from /
// to global memory).  Replaces warp-transducer's log_softmax / compute_grad_kernel.
#pragma once
#include "rnnt_lin.h"

namespace rnnt {

// Everything one lane does for its lattice cell once the cell's V logits sit in LDS at `xs`:
// GRAD=false: softmax denominator + the two lattice edge weights;  GRAD=true: the V gradients
// (written to `out`, zeros for padded cells; the patch kernels pass out = xs: the LDS image is rewritten in place).
// LIN: the linear-domain lattice (rnnt_lin.h): edge probabilities instead of log2 weights and no lse store in the lsm pass;
// in the gradient pass the softmax numerators are recomputed and scaled by occupancies formed from mantissas + frames.
template <int VP, bool V4, bool GRAD, bool LIN = false, bool SC1 = false>
__device__ __forceinline__ float cell_body(const LossParams &p, const Cell &cl, const uint32_t c, const float *xs, float *out) {
    float stat = 0.f;  // LIN lsm: the cell's decay statistic (rnnt_lin.h), else unused
    const int V = p.V;
    if (cl.valid) {
        float x[VP];
        if (V4) {
#pragma unroll
            for (int i = 0; i < VP / 4; ++i) {
                if (i * 4 < V) {
                    const float4 q = ((const float4 *)xs)[i];
                    x[4 * i] = q.x, x[4 * i + 1] = q.y, x[4 * i + 2] = q.z, x[4 * i + 3] = q.w;
                } else {
                    x[4 * i] = x[4 * i + 1] = x[4 * i + 2] = x[4 * i + 3] = -INFINITY;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VP; ++i) x[i] = (i < V) ? xs[i] : -INFINITY;
        }

        if (!GRAD && LIN) {
            stat = lin_cell_lsm<VP>(p, cl, x, xs);
        } else if (GRAD && LIN) {
            const LinGrad g = lin_grad_setup(p, cl);
            if (g.bad) atomicOr(p.flags + 4 * cl.b + kFlagG, 1);  // (rare) the utterance is redone in the log domain
            float m = x[0];
#pragma unroll
            for (int i = 1; i < VP; ++i) m = fmaxf(m, x[i]);
            const float nml = -m * kLog2e;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VP; ++i) x[i] = ex2(fmaf(x[i], kLog2e, nml)), s += x[i];
            const float inv = __builtin_amdgcn_rcpf(s);
            const float cb = g.has_blank_corr ? g.hb * inv * ex2(fmaf(xs[p.blank], kLog2e, nml)) : 0.f;
            const float cl2 = g.has_label ? g.hl * inv * ex2(fmaf(xs[g.lab], kLog2e, nml)) : 0.f;
            const float h0 = g.h0 * inv;
            if (V4) {
#pragma unroll
                for (int i = 0; i < VP / 4; ++i)
                    if (i * 4 < V) ((float4 *)out)[i] = make_float4(h0 * x[4 * i], h0 * x[4 * i + 1], h0 * x[4 * i + 2], h0 * x[4 * i + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < VP; ++i)
                    if (i < V) out[i] = h0 * x[i];
            }
            if (g.has_blank_corr) out[p.blank] -= cb;
            if (g.has_label) out[g.lab] -= cl2;
        } else if (!GRAD) {
            float m = x[0];
#pragma unroll
            for (int i = 1; i < VP; ++i) m = fmaxf(m, x[i]);
            float s = 0.f;
            const float nml = -m * kLog2e;
#pragma unroll
            for (int i = 0; i < VP; ++i) s += ex2(fmaf(x[i], kLog2e, nml));
            const float lg2s = lg2(s);
            const float lse = m + kLn2 * lg2s;
            // a blank from the last frame leaves the lattice unless it is THE terminal transition
            const bool blank_stays = (cl.t < cl.Tb - 1) || (cl.u == cl.Ub - 1);
            const float ob = blank_stays ? fmaf(xs[p.blank] - m, kLog2e, -lg2s) : kNeg;
            float ol = kNeg;
            if (cl.u < cl.Ub - 1) {
                const int lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], V);
                ol = fmaf(xs[lab] - m, kLog2e, -lg2s);
            }
            const size_t wi = ((size_t)cl.b * p.Nr + (cl.t + cl.u)) * p.Up + cl.u;
            if (SC1) {  // the hand-back team (rnnt_redo.h): other workgroups read these within the same launch -- write-through
                st_f32_wt(p.lse + c, lse);
                st_f32_wt(p.W + 2 * wi, ob), st_f32_wt(p.W + 2 * wi + 1, ol);
            } else {
                p.lse[c] = lse;
                ((float2 *)p.W)[wi] = make_float2(ob, ol);
            }
        } else {
            const CellGrad g = cell_grad_setup<SC1>(p, cl, c);
            const float xb = xs[p.blank];
            const float xl = g.has_label ? xs[g.lab] : 0.f;
            if (V4) {
#pragma unroll
                for (int i = 0; i < VP / 4; ++i)
                    if (i * 4 < V) {
                        float4 q;
                        q.x = g.scale * ex2(fmaf(x[4 * i], kLog2e, g.c0));
                        q.y = g.scale * ex2(fmaf(x[4 * i + 1], kLog2e, g.c0));
                        q.z = g.scale * ex2(fmaf(x[4 * i + 2], kLog2e, g.c0));
                        q.w = g.scale * ex2(fmaf(x[4 * i + 3], kLog2e, g.c0));
                        ((float4 *)out)[i] = q;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < VP; ++i)
                    if (i < V) out[i] = g.scale * ex2(fmaf(x[i], kLog2e, g.c0));
            }
            if (g.has_blank_corr) out[p.blank] -= g.scale * ex2(fmaf(xb, kLog2e, g.nl) + g.cb);
            if (g.has_label) out[g.lab] -= g.scale * ex2(fmaf(xl, kLog2e, g.nl) + g.cl);
        }
    } else if (GRAD) {
        for (int i = 0; i < V; ++i) out[i] = 0.f;
    }

    return stat;
}

}  // namespace rnnt
