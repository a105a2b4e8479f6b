// rnnt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the transducer loss.
//
// Replaces the three GPU stages of the reference's native op:
//   a-6  log-softmax denominator            -> cell_tile_kernel / cell_wave_kernel <GRAD=false>   ("lsm" pass)
//   a-7/a-8 alpha / beta lattice recurrences -> sweep_ld_kernel (U <= 1024), sweep_wide_kernel (U <= 8192)
//   a-9  fused-softmax gradient              -> cell_tile_kernel / cell_wave_kernel <GRAD=true>    ("grad" pass)
// Call site in the reference: utils/loss.py:34-35 (rnnt_loss) via run_rnnt.py:272.
//
// Design (DESIGN.md has the full account):
//  * lsm pass reads every logit once (coalesced 16-B LDS-DMA), reduces each lattice cell's
//    V logits inside ONE lane (no cross-lane traffic for small V), and emits 3 scalars per
//    cell: lse, and the two lattice edge weights in log2 domain, written DIAGONAL-MAJOR so
//    that the sweeps' per-step loads are contiguous.
//  * the sweeps run one wave64 per (utterance, direction): the live anti-diagonal stays in
//    VGPRs (K consecutive u per lane), the only cross-lane traffic per step is ONE DPP
//    wave-shift, no LDS exchange and no s_barrier; edge weights stream HBM -> LDS by
//    LDS-DMA (issued by a loader wave of the same workgroup) through a ring of chunks of G
//    diagonals.  alpha~/beta~ are re-based every kRebase = 8 diagonals by INTEGER amounts (exact in
//    f32; the offsets are kept aside) so f32 log-space values stay O(10) instead of O(T+U).
//  * grad pass re-reads the logits once, forms all V gradients of a cell in one lane from
//    alpha~, beta~, lse, applies the blank/label corrections, and stores through LDS so the
//    HBM writes are full 16-B coalesced lines.
#include "rnnt_common.h"
#include "rnnt_cell.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

namespace rnnt {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Dev variants (scripts/build_variant.sh): cache policy of the gradient pass's 16-byte stores / of its logit loads.
#ifndef RNNT_GSTORE
#define RNNT_GSTORE 0  // 0 nt (shipped), 1 plain, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
#endif
#ifndef RNNT_GLOAD
#define RNNT_GLOAD 0   // aux bits of the gradient pass's global_load_lds (0 default policy, 2 nt)
#endif
typedef float gs_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void grad_store16(gs_v4f *dst, const gs_v4f v) {
#if RNNT_GSTORE == 0
    __builtin_nontemporal_store(v, dst);
#elif RNNT_GSTORE == 1
    *dst = v;
#elif RNNT_GSTORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#elif RNNT_GSTORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#elif RNNT_GSTORE == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
#endif
}

// Everything one lane does for its lattice cell once the cell's V logits sit in LDS at `xs`:
// GRAD=false: softmax denominator + the two lattice edge weights;  GRAD=true: the V gradients
// (written back into `xs`, zeros for padded cells).
template <int VP, bool V4, bool GRAD>
__device__ __forceinline__ void cell_body(const LossParams &p, const Cell &cl, const uint32_t c, float *xs) {
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

        if (!GRAD) {
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
            p.lse[c] = lse;
            const size_t wi = ((size_t)cl.b * p.Nr + (cl.t + cl.u)) * p.Up + cl.u;
            ((float2 *)p.W)[wi] = make_float2(ob, ol);
        } else {
            const CellGrad g = cell_grad_setup(p, cl, c);
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
                        ((float4 *)xs)[i] = q;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < VP; ++i)
                    if (i < V) xs[i] = g.scale * ex2(fmaf(x[i], kLog2e, g.c0));
            }
            if (g.has_blank_corr) xs[p.blank] -= g.scale * ex2(fmaf(xb, kLog2e, g.nl) + g.cb);
            if (g.has_label) xs[g.lab] -= g.scale * ex2(fmaf(xl, kLog2e, g.nl) + g.cl);
        }
    } else if (GRAD) {
        for (int i = 0; i < V; ++i) xs[i] = 0.f;
    }

}

// ---------------------------------------------------------------------------------------------
// Small-vocabulary TILE path (V % 4 == 0): a workgroup owns a TT x UU patch of one utterance's
// lattice instead of 256 consecutive cells.  HBM reads are TT row segments of UU*V*4 contiguous
// bytes; the diagonal-major (skewed) W / alpha~ / beta~ accesses of a patch fall into runs of up to
// min(TT,UU) consecutive words per diagonal, all issued from ONE CU (one XCD L2), so lines are
// merged on chip instead of being touched by 16 different workgroups on 8 different L2s.
// ---------------------------------------------------------------------------------------------
// AL = false: vocabularies that are not a multiple of 4 (the reference's 31-symbol character set): a patch row then starts
// at an arbitrary 4-byte offset.  The row is staged from the enclosing 16-byte-aligned span, so its image sits `a` floats
// (a = start & 3, per row) into a 16-byte-aligned LDS row; cells read their logits with scalar LDS reads, and the gradient
// rows go back with float4 stores for the aligned interior and single floats at the two ragged ends.  Needs B*T*U*V % 4 == 0
// (then no aligned span reaches past the tensor).
#ifndef RNNT_TILE_LDS_PAD
#define RNNT_TILE_LDS_PAD 0  // dev builds: extra LDS bytes per patch workgroup (occupancy experiments)
#endif
constexpr int kFillRows = 16;  // W diagonals per fill workgroup of the lsm launch

template <int VP, bool GRAD, bool AL = true>
__global__ __launch_bounds__(256) void cell_tile_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V;
    const TileGeom &tg = p.tile;
    // The lsm launch carries extra workgroups behind its patches that write "log zero" into the W positions no lattice cell
    // writes (the sweeps read whole rows of the skewed array): kFillRows diagonals of one utterance each.  This replaces a
    // 37 MB memset in front of the launch (7.5 us at C2) by ~14 MB of stores that overlap with the patches.
    const uint32_t n_patch_wg = (uint32_t)p.nb * (uint32_t)tg.tiles_t * (uint32_t)tg.tiles_u;
    if (!GRAD && blockIdx.x >= n_patch_wg) {
        const uint32_t f = blockIdx.x - n_patch_wg, per = (uint32_t)p.Nr / kFillRows;
        const int fb = p.b0 + (int)(f / per), chunk = (int)(f % per);
        const int Tf = length_T(p, fb), Uf = length_U(p, fb);
        const int Nf = Tf + Uf - 1;
        float2 *Wb = (float2 *)p.W + (size_t)fb * p.Nr * p.Up;
        const float2 z = make_float2(kNeg, kNeg);
        for (int idx = tid; idx < kFillRows * p.Up; idx += 256) {
            const int rr = idx / p.Up, u = idx - rr * p.Up;
            const int n = chunk * kFillRows + rr;
            // lattice cells of diagonal n: u in [lo, hi]; diagonals past the lattice are never read by the sweeps
            if (n < Nf && (u < max(0, n - Tf + 1) || u > min(n, Uf - 1))) Wb[(size_t)n * p.Up + u] = z;
        }
        return;
    }
    // XCD-aware remap: hand each XCD (blockIdx % 8) a contiguous range of patches (bijective form)
    uint32_t bid;
    {
        const uint32_t nwg = n_patch_wg, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
        const uint32_t q = nwg >> 3, r = nwg & 7u;
        bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
        // The gradient pass walks each XCD's range backwards: the logits the lsm pass read LAST are the ones most
        // likely still in the 256 MiB Infinity Cache.
        if (GRAD) bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + ((xcd < r ? q + 1u : q) - 1u - idx);
    }
    const uint32_t q1 = fdiv(bid, tg.div_tu);
    const uint32_t tu = bid - q1 * (uint32_t)tg.tiles_u;
    const uint32_t bb = fdiv(q1, tg.div_tt);
    const uint32_t tt = q1 - bb * (uint32_t)tg.tiles_t;
    const int b = p.b0 + (int)bb;
    const int t0 = (int)tt * tg.TT, u0 = (int)tu * tg.UU;
    const int Tb = length_T(p, b), Ub = length_U(p, b);

    // The valid part of a patch is a rectangle known to the whole workgroup: no per-cell bookkeeping.
    const int rows_valid = max(0, min(tg.TT, Tb - t0));         // lattice rows with t < T_b
    const int cols_valid = max(0, min(tg.UU, Ub - u0));         // lattice columns with u < U_b
    const int rows_in = max(0, min(tg.TT, p.T - t0));           // rows that exist in the tensor
    const int cols_in = max(0, min(tg.UU, p.U - u0));
    const int q_valid = cols_valid * V / 4;                     // 16-byte chunks per row that carry valid cells
    const int q_in = cols_in * V / 4;
    const int row_lds = AL ? tg.UU * V : ((tg.UU * V + 3 + 3) & ~3);  // floats per patch row in LDS (16-byte aligned rows)
    const size_t row_f = (size_t)p.U * V;                       // floats per lattice row in HBM
    const size_t patch0 = ((size_t)(b * p.T + t0) * p.U + u0) * V;

    // gradient row `r` of the patch back to HBM: from the LDS image (src != nullptr) or zeros
    auto store_row = [&](const int r, const float *src) {
        const size_t s0 = patch0 + r * row_f;
        if (AL) {
            for (int q = lane; q < q_in; q += 64) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f v = src ? ((const v4f *)src)[q] : (v4f){0.f, 0.f, 0.f, 0.f};
                grad_store16((v4f *)(p.grads + s0 + q * 4), v);
            }
        } else {
            const int a = (int)(s0 & 3), len = cols_in * V;
            float *base = p.grads + (s0 - a);
            for (int q = lane; q * 4 < a + len; q += 64) {
                const int e0 = q * 4 - a;  // element of the row segment held by the chunk's first float
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f vv = src ? ((const v4f *)src)[q] : (v4f){0.f, 0.f, 0.f, 0.f};  // (LDS rows are 16-byte aligned)
                const float v[4] = {vv[0], vv[1], vv[2], vv[3]};
                if (e0 >= 0 && e0 + 3 < len) {
                    grad_store16((v4f *)(base + q * 4), vv);  // non-temporal, like the aligned path
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e0 + k >= 0 && e0 + k < len) base[q * 4 + k] = v[k];
                }
            }
        }
    };

    if (rows_valid == 0 || cols_valid == 0) {
        if (GRAD) {  // an all-padding patch: exact zeros, no reads
            for (int r = wave; r < rows_in; r += 4) store_row(r, nullptr);
        }
        return;
    }

    // ---- stage: each wave streams whole row segments HBM -> LDS with 16-byte LDS-DMA ----
    for (int r = wave; r < rows_valid; r += 4) {
        const size_t s0 = patch0 + r * row_f;
        const int a = AL ? 0 : (int)(s0 & 3);
        const float *src = p.acts + (s0 - a);
        float *dst = lds + r * row_lds;
        const int nq = AL ? q_valid : (a + cols_valid * V + 3) / 4;
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const int q = q0 + lane;
            if (q < nq) {
                // default cache policy: non-temporal loads lose the Infinity-Cache reuse between the two cell passes
                __builtin_amdgcn_global_load_lds((glb_void *)(src + q * 4), (lds_void *)(dst + q0 * 4), 16, 0, GRAD ? RNNT_GLOAD : 0);
            }
        }
    }
    wait_vm0();
    __syncthreads();

    const uint32_t r = fdiv((uint32_t)tid, tg.divUU);
    const int cu = tid - (int)r * tg.UU;
    Cell cl;
    cl.b = b, cl.t = t0 + (int)r, cl.u = u0 + cu, cl.Tb = Tb, cl.Ub = Ub;
    cl.valid = ((int)r < rows_valid) && (cu < cols_valid);
    const uint32_t c = ((uint32_t)(b * p.T + cl.t)) * (uint32_t)p.U + (uint32_t)cl.u;
    if (AL) {
        if ((GRAD && tid < tg.TT * tg.UU) || cl.valid) cell_body<VP, true, GRAD>(p, cl, c, lds + tid * V);  // (lanes beyond the patch own no LDS)
    } else if ((int)r < tg.TT) {
        const int a = (int)((patch0 + r * row_f) & 3);
        if (GRAD || cl.valid) cell_body<VP, false, GRAD>(p, cl, c, lds + r * row_lds + a + cu * V);
    }
    if (GRAD && !AL) {
        __syncthreads();
        for (int rr = wave; rr < rows_in; rr += 4) store_row(rr, lds + rr * row_lds);
    } else if (GRAD) {
        __syncthreads();
        for (int rr = wave; rr < rows_in; rr += 4) {
            const float4 *srcl = (const float4 *)(lds + rr * row_lds);
            float *dstg = p.grads + patch0 + rr * row_f;
            // gradients are written once and not re-read by this op: keep them out of L2 / Infinity Cache
            for (int q = lane; q < q_in; q += 64) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                grad_store16((v4f *)(dstg + q * 4), ((const v4f *)srcl)[q]);
            }
        }
    }
}

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

template <bool V4, bool GRAD>
__global__ __launch_bounds__(256) void cell_wave_kernel(const LossParams p) {
    const int lane = threadIdx.x & 63;
    const int V = p.V;
    // No grid stride: a workgroup owns ONE contiguous span of cells (about 16 KB of logits: 4 waves x wave_cells(V) cells), as the
    // streaming kernels that reach 6 TB/s on this part do (scripts/probes/probe_hbm.hip; grid-stride loops: 4.7-5.0 TB/s).
    const uint32_t cpw = (uint32_t)wave_cells(V);
    const uint32_t c_lo = (blockIdx.x * 4u + (threadIdx.x >> 6)) * cpw;
    const uint32_t c_hi = min(c_lo + cpw, p.cells);
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
            const CellGrad g = cell_grad_setup(p, cl, c);
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

// ---------------------------------------------------------------------------------------------
// alpha / beta anti-diagonal sweeps: one wave64 per (utterance, direction).
//
// Lane l owns the K consecutive lattice columns u = l*K .. l*K+K-1 (row stride Up = 64*K, so every
// lane is always inside its row).  There are NO validity masks in the step: "log zero" is carried
// by the data.  The W workspace is pre-filled with a finite log-zero bit pattern, the lsm pass
// overwrites only real lattice cells and writes log-zero for edges that leave the lattice, hence
// any node outside [0,T_b) x [0,U_b) stays at log zero by construction.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_from_lower_lane(float x, float fill) {  // lane i <- lane i-1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138 /*wave_shr:1*/,
                                                      0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float x, float fill) {  // lane i <- lane i+1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130 /*wave_shl:1*/,
                                                      0xf, 0xf, false));
}
// log2(2^a + 2^b); log zero is any value <= kNeg (finite), so this never forms inf-inf.
__device__ __forceinline__ float lse2(float a, float b) {
    const float d = a - b;
    return fmaxf(a, b) + lg2(1.0f + ex2(-fabsf(d)));
}

// Stream `n16` 16-byte units global -> LDS (destination lane-linear, as LDS-DMA requires).
__device__ __forceinline__ void dma_rows(const float *g, float *l, int n16, int lane) {
    for (int i0 = 0; i0 < n16; i0 += 64) {
        const int k = i0 + lane;
        if (k < n16) __builtin_amdgcn_global_load_lds((glb_void *)(g + (size_t)k * 4), (lds_void *)(l + i0 * 4), 16, 0, 0);
    }
}

// Re-basing reference of the WIDE sweep (1024 < U <= 8192; the register-resident sweeps re-base per lane, see rebase_lane):
// the lattice cell on the straight line (0,0)->(T_b-1,U_b-1) -- NOT the row maximum: for near-uniform posteriors the
// alpha-maximum of a diagonal sits at the binomial centre, ~e^(0.19 n) above the cells that matter.
struct RidgeLine {
    uint32_t slope_fx;  // (U_b-1)/(N_b-1) in 16.16 fixed point
    __device__ __forceinline__ int u_at(int n) const { return (int)(((uint32_t)n * slope_fx + 32768u) >> 16); }
};
__device__ __forceinline__ RidgeLine make_ridge(int Ub, int Nb) {
    RidgeLine r;
    r.slope_fx = (Nb > 1) ? (((uint32_t)(Ub - 1) << 16) / (uint32_t)(Nb - 1)) : 0u;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Precision control, per LANE.  Every lane keeps its own cumulative INTEGER offset (exact in f32) for the K lattice columns
// it owns: true value = stored value + off[lane].  Every kRebase diagonals a lane re-bases against its own maximum, so the
// f32 values that carry probability mass stay O(K x edge weight) whatever the logits look like -- one offset per diagonal
// (the previous scheme, against the straight-line "ridge" cell) left the mass-carrying cells at |value| ~ 10^2..10^3 whenever
// the posterior strays from the straight line: 2e-4 cost error for trained-like late alignments, 2-3e-4 gradient error for
// 8 x N(0,1) logits (tests/tools/emulate_sweep.py; now 1e-6 / <1e-4).
// The only values that cross a lane boundary are the label-edge terms of a lane's last column; the (integer) offset
// difference of the two lanes is folded into that edge WEIGHT (`dlt`, off the dependent chain), so the step itself is unchanged.
// A lane that holds no lattice node yet copies the offset of the neighbour the mass will arrive from (R rounds: up to
// ceil(kRebase / K) lanes wake up within one block), so a first arrival is never rounded at the magnitude of the total offset.
// The offsets go to the table [block of kRebase diagonals][64 lanes] (one coalesced 256-byte store per block).
// ---------------------------------------------------------------------------------------------
struct SweepState {
    float off;       // this lane's cumulative (integer-valued) offset
    float dlt;       // alpha: off - off[lane + 1], beta: off[lane + 1] - off -- added to the label-edge weight of column K - 1
    float *tab;      // this utterance's offset table [NC][64], already advanced by `lane`
    float *row;      // wave-uniform base of the output row of the NEXT diagonal to be stored
    float edge;      // what DPP shifted in last (edge lane: log zero, see alpha_step_c)
};

template <int K, bool BETA>
__device__ __forceinline__ void rebase_lane(float (&v)[K], SweepState &st, const int kc) {
    float m = v[0];
#pragma unroll
    for (int j = 1; j < K; ++j) m = fmaxf(m, v[j]);
    const bool fin = m > kNegTest;
    const float mi = fin ? rintf(m) : 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] -= mi;  // log zeros stay log zeros: |mi| << 1e30
    float off = st.off + mi;
    constexpr int R = (kRebase + K - 1) / K;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float nb = BETA ? dpp_from_upper_lane(off, off) : dpp_from_lower_lane(off, off);  // edge lanes see themselves
        off = fin ? off : nb;
    }
    st.off = off;
    const float nr = dpp_from_upper_lane(off, off);
    st.dlt = BETA ? nr - off : off - nr;
    st_f32_wt(st.tab + (size_t)kc * 64, off);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// number of store instructions store_diag<K, true> issues (pieces of 4 dwords, then one of 3, 2 or 1)
constexpr int store_pieces(int K) { return K / 4 + ((K % 4) ? 1 : 0); }

// Write one diagonal's K values of this lane: `row` is the wave-uniform row base (SGPR pair), `voff`
// the lane's byte offset.  COUNTED: explicit instructions so that the number of VMEM operations per
// step is known exactly (for the counted s_waitcnt at chunk boundaries).
typedef float f32x3 __attribute__((ext_vector_type(3)));
// OFF: compile-time byte offset added to `row` (the 13-bit signed immediate of the store: |OFF| + 4 K <= 4096), so that
// consecutive diagonals can share one SGPR row base.
template <int K, bool COUNTED, int OFF = 0>
__device__ __forceinline__ void store_diag(float *row, const int voff, const int lane, const float (&v)[K]) {
    // All lattice stores are write-through (sc1): the gradient pass may run on another XCD while this
    // kernel is still alive, and nothing on this XCD re-reads them anyway.
    if (!COUNTED) {
        float *dst = row + lane * K + OFF / 4;
#pragma unroll
        for (int j = 0; j < K; ++j) st_f32_wt(dst + j, v[j]);
    } else {
        static_assert(OFF + 4 * K <= 4096 && OFF >= -4096, "store offset outside the immediate range");
        int j = 0;
#pragma unroll
        for (; j + 4 <= K; j += 4) {
            const f32x4 q = {v[j], v[j + 1], v[j + 2], v[j + 3]};
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 1" ::"v"(voff), "v"(q), "s"(row), "n"(OFF + j * 4));
        }
        if (K % 4 == 3) {
            const f32x3 q = {v[j], v[j + 1], v[j + 2]};
            asm volatile("global_store_dwordx3 %0, %1, %2 offset:%3 sc1" ::"v"(voff), "v"(q), "s"(row), "n"(OFF + j * 4));
        } else if (K % 4 == 2) {
            const f32x2 q = {v[j], v[j + 1]};
            asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 sc1" ::"v"(voff), "v"(q), "s"(row), "n"(OFF + j * 4));
        } else if (K % 4 == 1) {
            asm volatile("global_store_dword %0, %1, %2 offset:%3 sc1" ::"v"(voff), "v"(v[j]), "s"(row), "n"(OFF + j * 4));
        }
    }
}
// diagonals that can share one row base through the store immediate (forward: offsets 0 .. (R-1) Up 4)
constexpr int rows_per_base(int K) { return (4096 - 4 * K) / (64 * K * 4) + 1 > 16 ? 16 : (4096 - 4 * K) / (64 * K * 4) + 1; }

template <int N>
__device__ __forceinline__ void wait_vm_counted() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Edge weights of one diagonal for this lane: w[j] = {blank edge, label edge} of column u0 + j.
// LDS row layout = HBM row layout = [Up][2] (blank, label interleaved per column): K 8-byte reads.
template <int K>
__device__ __forceinline__ void load_w(f32x2 (&w)[K], const float *wrow) {
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = ((const f32x2 *)wrow)[j];
}

// Explicitly scheduled variant of load_w for the counted sweep: K ds_read_b64 whose completion the
// compiler does NOT track -- the caller waits with lds_wait<N>() (LDS returns in order, so waiting
// for "<= N outstanding" retires everything older than the newest N reads).  `addr` is the lane's
// LDS byte address of row 0 of the chunk buffer; the row/column offsets are immediates.
template <int K, int ROW>
__device__ __forceinline__ void lds_issue_row(f32x2 (&q)[K], const uint32_t addr) {
#pragma unroll
    for (int j = 0; j < K; ++j)
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[j]) : "v"(addr), "n"(ROW * 2 * 64 * K * 4 + j * 8));
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);  // nothing that consumes the rows may be hoisted above the wait
}

// One alpha step: diagonal r -> r+1 using the outgoing edge weights `w` of diagonal r.
// {d_j, e_j} = {a_j, a_j} + {blank_j, label_j} is ONE packed add per column.
template <int K>
__device__ __forceinline__ void alpha_step(float (&a)[K], const f32x2 (&w)[K], const float dlt) {
    f32x2 de[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const f32x2 aa = {a[j], a[j]};
        de[j] = aa + w[j];  // .x blank: (t-1,u) -> (t,u);  .y label: (t,u) -> (t,u+1)
    }
    de[K - 1][1] += dlt;  // into the next lane's frame (integer offset difference)
    const float from_left = dpp_from_lower_lane(de[K - 1][1], kNeg);
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = lse2(de[j][0], (j == 0) ? from_left : de[j - 1][1]);
}

// The same steps for the unrolled chunks.  Two instructions less per diagonal:
//  * the value DPP shifts into the edge lane (lane 0 / lane 63) is log zero; instead of re-materialising that
//    constant every step (the DPP move overwrites its `old` operand), the previous step's shifted register is passed
//    as `old`: its edge lane still holds log zero (the move never writes it);
//  * fmaxf on a DPP result makes the compiler canonicalise it first (v_max x, x); v_max_f32 itself quiets NaNs.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Stage-major log-add of K independent (u, l) pairs: a lone wave stalls on every instruction that consumes the result of
// the one just before it, so the K chains are issued stage by stage (sub x K, exp2 x K, max x K, add x K, log2 x K, add x K)
// with scheduling fences in between -- every consumer is K instructions behind its producer.
#define SWEEP_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int K>
__device__ __forceinline__ void lse2_staged(float (&out)[K], const float (&u)[K], const float (&l)[K]) {
    float d[K], e[K], m[K];
#pragma unroll
    for (int j = K - 1; j >= 0; --j) d[j] = u[j] - l[j];  // column 0 last: its l comes out of the DPP move
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) e[j] = ex2(-fabsf(d[j]));
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) m[j] = vmax(u[j], l[j]);
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) e[j] = 1.0f + e[j];
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) e[j] = lg2(e[j]);
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) out[j] = m[j] + e[j];
    SWEEP_FENCE();
}
template <int K>
__device__ __forceinline__ void alpha_step_c(float (&a)[K], const f32x2 (&w)[K], float &edge, const float dlt) {
    f32x2 de[K];
    // the value that leaves this lane goes into the next lane's frame: the integer offset difference rides on the edge
    // weight (an add that depends on the LDS read only, not on the previous diagonal)
    f32x2 wl = w[K - 1];
    wl[1] += dlt;
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {  // last column first: the DPP move waits for it
        const f32x2 aa = {a[j], a[j]};
        de[j] = aa + ((j == K - 1) ? wl : w[j]);
    }
    SWEEP_FENCE();
    edge = dpp_from_lower_lane(de[K - 1][1], edge);
    float u[K], l[K];
#pragma unroll
    for (int j = 0; j < K; ++j) u[j] = de[j][0], l[j] = (j == 0) ? edge : de[j - 1][1];
    SWEEP_FENCE();
    lse2_staged<K>(a, u, l);
}
template <int K>
__device__ __forceinline__ void beta_step_c(float (&bv)[K], const f32x2 (&w)[K], float &edge, const float dlt) {
    f32x2 wl = w[K - 1];
    wl[1] += dlt;  // the value arriving from the next lane is in THAT lane's frame
    SWEEP_FENCE();
    edge = dpp_from_upper_lane(bv[0], edge);
    f32x2 s2[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {  // last column last: it waits for the DPP move
        const f32x2 br = {bv[j], (j == K - 1) ? edge : bv[j + 1]};
        s2[j] = br + ((j == K - 1) ? wl : w[j]);
    }
    SWEEP_FENCE();
    float u[K], l[K];
#pragma unroll
    for (int j = 0; j < K; ++j) u[K - 1 - j] = s2[j][0], l[K - 1 - j] = s2[j][1];  // reversed: lse2_staged takes its index 0 last
    float nv[K];
    lse2_staged<K>(nv, u, l);
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = nv[K - 1 - j];
}

// One beta step: diagonal n+1 -> n using the outgoing edge weights `w` of diagonal n.
template <int K>
__device__ __forceinline__ void beta_step(float (&bv)[K], const f32x2 (&w)[K], const float dlt) {
    const float from_right = dpp_from_upper_lane(bv[0], kNeg);
    float nv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const f32x2 br = {bv[j], (j == K - 1) ? from_right + dlt : bv[j + 1]};
        const f32x2 s2 = br + w[j];
        nv[j] = lse2(s2[0], s2[1]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = nv[j];
}

// Fully unrolled, explicitly pipelined steps of one chunk (compile-time recursion over the step index
// II so that every LDS offset is an immediate and the two weight register sets ping-pong by name).
template <int K, int G, int II>
__device__ __forceinline__ void alpha_fast_steps(float (&a)[K], f32x2 (&wq)[2][K], const uint32_t abase, SweepState &st,
                                                 const int voff, const int lane, const int r0) {
    if constexpr (II < G) {
        constexpr int cur = II & 1, nxt = cur ^ 1;
        if constexpr (II + 1 < G) {
            lds_issue_row<K, II + 1>(wq[nxt], abase);
            lds_wait<K>();  // row II has landed, row II+1 stays in flight
        } else {
            lds_wait<0>();
        }
        const int n = r0 + II + 1;
        alpha_step_c<K>(a, wq[cur], st.edge, st.dlt);
        if ((n & (kRebase - 1)) == 0) rebase_lane<K, false>(a, st, n / kRebase);
        constexpr int R = rows_per_base(K);
        store_diag<K, true, (II % R) * 64 * K * 4>(st.row, voff, lane, a);
        if constexpr (II % R == R - 1 || II == G - 1) st.row += (II % R + 1) * 64 * K;
        alpha_fast_steps<K, G, II + 1>(a, wq, abase, st, voff, lane, r0);
    }
}

template <int K, int G, int II>
__device__ __forceinline__ void beta_fast_steps(float (&bv)[K], f32x2 (&wq)[2][K], const uint32_t abase, SweepState &st,
                                                const int voff, const int lane, const int r0) {
    if constexpr (II < G) {
        constexpr int cur = II & 1, nxt = cur ^ 1;
        constexpr int i = G - 1 - II;  // row inside the chunk (descending)
        if constexpr (i > 0) {
            lds_issue_row<K, i - 1>(wq[nxt], abase);
            lds_wait<K>();
        } else {
            lds_wait<0>();
        }
        const int n = r0 + i;
        beta_step_c<K>(bv, wq[cur], st.edge, st.dlt);
        if ((n & (kRebase - 1)) == kRebase - 1) rebase_lane<K, true>(bv, st, n / kRebase);
        constexpr int R = rows_per_base(K);
        store_diag<K, true, -(II % R) * 64 * K * 4>(st.row, voff, lane, bv);
        if constexpr (II % R == R - 1 || II == G - 1) st.row -= (II % R + 1) * 64 * K;
        beta_fast_steps<K, G, II + 1>(bv, wq, abase, st, voff, lane, r0);
    }
}

// ---------------------------------------------------------------------------------------------
// fills (see launch_fill in rnnt_common.h)
// ---------------------------------------------------------------------------------------------
// One contiguous 16 KB span per workgroup (four 16-byte stores per thread), no grid stride.
__global__ __launch_bounds__(256) void fill_kernel(uint32_t *dst, const uint32_t word, const size_t nwords) {
    if ((((uintptr_t)dst) & 15) == 0) {
        const size_t n4 = nwords >> 2;
        const uint4 q = make_uint4(word, word, word, word);
        const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + 256 * k < n4) ((uint4 *)dst)[base + 256 * k] = q;
        if (blockIdx.x == 0 && threadIdx.x < (nwords & 3)) dst[n4 * 4 + threadIdx.x] = word;
    } else {
        const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (base + 256 * k < nwords) dst[base + 256 * k] = word;
    }
}

hipError_t launch_fill(void *dst, int byte, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) != 0 || (((uintptr_t)dst) & 3) != 0) return hipErrorInvalidValue;
    const uint32_t b = (uint32_t)(byte & 0xff), word = b | (b << 8) | (b << 16) | (b << 24);
    const size_t nwords = bytes >> 2;
    const size_t grid = (nwords + 4095) / 4096;  // 16 KB per workgroup
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)grid), dim3(256), 0, s, (uint32_t *)dst, word, nwords);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
bool tile_path_ok(const LossParams &p, bool grad) {
    if (p.V > 60) return false;
    // rows that start off a 16-byte boundary (V % 4 != 0) are staged from the enclosing aligned span: the whole tensor
    // must end on a 16-byte boundary for that span never to leave it
    if ((p.V % 4) != 0 && (((size_t)p.cells * (size_t)p.V) % 4) != 0) return false;
    if (((uintptr_t)p.acts & 15) != 0) return false;
    if (grad && ((uintptr_t)p.grads & 15) != 0) return false;
    return true;
}

// Patch kernels (one lattice cell per lane) for V <= 60 and 16-byte-aligned tensors; otherwise one cell per wave.
template <bool GRAD>
static hipError_t launch_cell(const LossParams &p, hipStream_t s) {
    if (tile_path_ok(p, GRAD)) {
        // (the lsm launch fills the log-zero part of W itself: see cell_tile_kernel)
        const unsigned blocks = (unsigned)p.nb * p.tile.tiles_t * p.tile.tiles_u + (GRAD ? 0u : (unsigned)p.nb * (unsigned)(p.Nr / kFillRows));
        // the patch image: TT x UU cells (<= 256) of V floats, to the byte.  Sized by the patch, not by the 256 lanes, and without
        // padding: at V = 28 that is 26,880 B -- six workgroups per CU instead of five (gradient pass 115 -> 110.5 us); at V = 32
        // 30,720 B instead of 32,832 B -- five instead of four (146 -> 140 us)
        const size_t shm = (size_t)p.tile.TT * p.tile.UU * p.V * sizeof(float) + RNNT_TILE_LDS_PAD;
        if ((p.V % 4) != 0) {
            const size_t pitch = (size_t)((p.tile.UU * p.V + 3 + 3) & ~3);
            size_t shmu = (size_t)p.tile.TT * pitch * sizeof(float);
            if (shmu < shm) shmu = shm;
            if (p.V <= 32)
                hipLaunchKernelGGL((cell_tile_kernel<32, GRAD, false>), dim3(blocks), dim3(256), shmu, s, p);
            else
                hipLaunchKernelGGL((cell_tile_kernel<64, GRAD, false>), dim3(blocks), dim3(256), shmu, s, p);
        } else if (p.V <= 32)
            hipLaunchKernelGGL((cell_tile_kernel<32, GRAD>), dim3(blocks), dim3(256), shm, s, p);
        else
            hipLaunchKernelGGL((cell_tile_kernel<64, GRAD>), dim3(blocks), dim3(256), shm, s, p);
    } else {
        const bool v4 = (p.V % 4) == 0 && ((uintptr_t)p.acts & 15) == 0 && (!GRAD || ((uintptr_t)p.grads & 15) == 0);
        const unsigned per_wg = 4u * (unsigned)wave_cells(p.V);
        const unsigned blocks = (p.cells + per_wg - 1u) / per_wg;
        if (v4)
            hipLaunchKernelGGL((cell_wave_kernel<true, GRAD>), dim3(blocks), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((cell_wave_kernel<false, GRAD>), dim3(blocks), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_lsm(const LossParams &p, hipStream_t s) { return launch_cell<false>(p, s); }
hipError_t launch_grad(const LossParams &p, hipStream_t s) { return launch_cell<true>(p, s); }

// ---------------------------------------------------------------------------------------------
// LDS progress counters shared by the waves of one sweep workgroup.  A wave's LDS operations complete in order, so
// "data written, then counter written" is all the ordering a hand-off needs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_peek(const uint32_t addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
// Bounded: a protocol error must end in wrong numbers (caught by the parity tests), never in a hung GPU.
__device__ __forceinline__ int lds_wait_ge(const uint32_t addr, const int need) {  // returns the value it saw
    int v = 0;
    for (int spin = 0; spin < (1 << 18); ++spin) {
        v = lds_peek(addr);
        if (v >= need) return v;
        __builtin_amdgcn_s_sleep(1);
    }
    return v;
}
__device__ __forceinline__ void lds_post(const uint32_t addr, const int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// ---------------------------------------------------------------------------------------------
// Default sweep: the single-wave sweep above with the LDS-DMA moved to a LOADER wave of the same workgroup.
// Issuing the 24 `global_load_lds` pieces of a chunk from the sweeping wave itself cost it ~1/4 of its time (a piece takes
// 60-100 issue cycles and the wave issues in order: -23 us of 85 with the DMA knocked out); the loader has nothing else to
// do.  NB chunk buffers form a ring; two LDS counters: `landed` (chunks complete in LDS, loader -> sweeper) and `consumed`
// (chunks the sweeper is done with, sweeper -> loader).  The loader waits on `consumed` only when it is NB - 1 chunks ahead,
// the sweeper on `landed` only when the loader is behind: no cycle.  All polls are bounded; a sweeper whose poll gives up
// reports NaN as the utterance's cost (never a plausible number).
// ---------------------------------------------------------------------------------------------
struct LdLink {
    uint32_t landed, consumed;  // LDS byte addresses of the two counters
};

template <int K, int G, int NB, bool BETA>
__device__ void sweep_loader(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up, n16 = chunkf / 4, pieces = n16 / 64;
    static_assert(n16 % 64 == 0 && pieces <= 63, "chunk must be whole wave-instructions within the vmcnt range");
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int nchunks = (Tb + Ub - 2) / G + 1;
    const float *Wb = p.W + (size_t)b * p.Nr * 2 * Up;
    for (int i = 0; i < nchunks; ++i) {
        const int ck = BETA ? nchunks - 1 - i : i;
        if (i >= NB) lds_wait_ge(lk.consumed, i - NB + 1);  // ring slot i % NB is free again
        dma_rows(Wb + (size_t)ck * chunkf, bufs + (i % NB) * chunkf, n16, lane);
        if (i > 0) {
            wait_vm_counted<pieces>();  // loads return in order: everything but the chunk just issued has landed
            if (lane == 0) lds_post(lk.landed, i);
        }
    }
    wait_vm0();
    if (lane == 0) lds_post(lk.landed, nchunks);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int K, int G, int NB>
__device__ void alpha_sweep_ld(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1;
    float *out = p.A + (size_t)b * p.Nr * Up;
    const int voff = lane * K * 4;
    const int u0 = lane * K;

    float a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (u0 + j == 0) ? 0.f : kNeg;
    store_diag<K, false>(out, voff, lane, a);
    SweepState st;
    st.off = 0.f, st.dlt = 0.f;
    st.edge = kNeg;
    st.tab = p.offA + (size_t)b * p.NC * p.NG + lane;  // NG = 64 for the register-resident sweeps
    st_f32_wt(st.tab, 0.f);  // block 0
    st.row = out + Up;
    const int last_row = Nb - 1;
    const int nchunks = last_row / G + 1;

    int have = 0;  // chunks known to have landed (the loader runs up to NB - 1 ahead: most chunks need no look at the counter)
    bool timed_out = lengths_invalid(p, b);  // a bounded poll gave up, or the caller's lengths were out of range: the result must not look valid
    for (int ck = 0; ck < nchunks; ++ck) {
        if (have < ck + 1) {
            have = lds_wait_ge(lk.landed, ck + 1);
            timed_out |= have < ck + 1;
        }
        const float *cur = bufs + (ck % NB) * chunkf + 2 * u0;
        const int r0 = ck * G;
        if (K <= 15 && r0 + G <= last_row) {
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            f32x2 wq[2][K];
            lds_issue_row<K, 0>(wq[0], abase);
            alpha_fast_steps<K, G, 0>(a, wq, abase, st, voff, lane, r0);
        } else {
            for (int i = 0; i < G; ++i) {
                const int n = r0 + i + 1;
                if (n > last_row) break;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Up);
                alpha_step<K>(a, wc, st.dlt);
                if ((n & (kRebase - 1)) == 0) rebase_lane<K, false>(a, st, n / kRebase);
                store_diag<K, false>(st.row, voff, lane, a);
                st.row += Up;
            }
        }
        if (ck + 1 < nchunks) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of this chunk's buffer has returned
            if (lane == 0) lds_post(lk.consumed, ck + 1);
        }
    }
    {
        const float *wrow = bufs + ((nchunks - 1) % NB) * chunkf + (last_row % G) * 2 * Up + 2 * u0;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (u0 + j == Ub - 1) {
                const double ll2 = timed_out ? (double)NAN : (double)st.off + (double)a[j] + (double)wrow[2 * j];
                st_f64_wt(p.ll + 2 * b, ll2);
                st_f32_wt(p.costs + b, (float)(-ll2 * 0.6931471805599453));
            }
    }
}

template <int K, int G, int NB>
__device__ void beta_sweep_ld(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1;
    float *out = p.Bt + (size_t)b * p.Nr * Up;
    const int voff = lane * K * 4;
    const int u0 = lane * K;

    float bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = (u0 + j == Ub - 1) ? 0.f : kNeg;
    const int last = Nb - 1;
    const int ckl = last / G;
    SweepState st;
    st.off = 0.f, st.dlt = 0.f;
    st.edge = kNeg;
    st.tab = p.offB + (size_t)b * p.NC * p.NG + lane;
    st.row = out + (size_t)last * Up;

    int have = 0;
    bool timed_out = lengths_invalid(p, b);
    for (int ck = ckl; ck >= 0; --ck) {
        const int i_ring = ckl - ck;  // the loader's chunk index
        if (have < i_ring + 1) {
            have = lds_wait_ge(lk.landed, i_ring + 1);
            timed_out |= have < i_ring + 1;
        }
        const float *cur = bufs + (i_ring % NB) * chunkf + 2 * u0;
        const int r0 = ck * G;
        if (K <= 15 && r0 + G - 1 < last) {
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            f32x2 wq[2][K];
            lds_issue_row<K, G - 1>(wq[0], abase);
            beta_fast_steps<K, G, 0>(bv, wq, abase, st, voff, lane, r0);
        } else {
            for (int ii = 0; ii < G; ++ii) {
                const int i = G - 1 - ii;
                const int n = r0 + i;
                if (n > last) continue;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Up);
                beta_step<K>(bv, wc, st.dlt);
                if (((n & (kRebase - 1)) == kRebase - 1) || n == last) rebase_lane<K, true>(bv, st, n / kRebase);
                store_diag<K, false>(st.row, voff, lane, bv);
                st.row -= Up;
            }
        }
        if (ck > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_post(lk.consumed, i_ring + 1);
        }
    }
    if (lane == 0) st_f64_wt(p.ll + 2 * b + 1, timed_out ? (double)NAN : (double)st.off + (double)bv[0]);
    if (timed_out && lane == 0) st_f32_wt(p.costs + b, NAN);  // the alpha side may have finished normally
}

template <int K, int G, int NB>
__global__ __launch_bounds__(128) void sweep_ld_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int chunkf = G * 2 * 64 * K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = p.b0 + (int)(blockIdx.x >> 1);
    const bool beta = (blockIdx.x & 1) != 0;
    int *ctr = (int *)(lds + NB * chunkf);
    if (tid < 2) ctr[tid] = 0;
    __syncthreads();
    LdLink lk;
    lk.landed = (uint32_t)(uintptr_t)((lds_void *)ctr);
    lk.consumed = lk.landed + 4u;
    if (wave == 1) {
        if (beta)
            sweep_loader<K, G, NB, true>(p, lds, lk, b, lane);
        else
            sweep_loader<K, G, NB, false>(p, lds, lk, b, lane);
    } else {
        if (beta)
            beta_sweep_ld<K, G, NB>(p, lds, lk, b, lane);
        else
            alpha_sweep_ld<K, G, NB>(p, lds, lk, b, lane);
    }
}

template <int K, int G>
static hipError_t launch_sweep_ld(const LossParams &p, hipStream_t s) {
    constexpr int NB = ((size_t)4 * G * 2 * 64 * K * sizeof(float) + 16 <= 128 * 1024) ? 4 : 3;
    constexpr size_t shm = (size_t)NB * G * 2 * 64 * K * sizeof(float) + 16;
    static_assert(shm <= 160 * 1024, "chunk ring exceeds the LDS");
    if (shm > 64 * 1024) {  // per device and cheap: set on every launch (a process may drive several GPUs)
        hipError_t e = hipFuncSetAttribute((const void *)sweep_ld_kernel<K, G, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((sweep_ld_kernel<K, G, NB>), dim3(2 * p.nb), dim3(128), shm, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Wide sweep (1024 < maxU <= 8192; upstream has no such limit, label sequences this long are rare): one workgroup of 1024
// threads per (utterance, direction), the previous diagonal in LDS, one barrier per diagonal -- the plain formulation, an
// order of magnitude slower per diagonal than the register-resident sweeps, with the SAME arithmetic (log2 domain, finite
// log zero, integer re-basing against the straight-line ridge cell every kRebase diagonals; one offset per block, copied to
// every 64-column group of the offset tables) and the same outputs, so the gradient pass does not know which sweep ran.
// ---------------------------------------------------------------------------------------------
template <bool BETA>
__global__ __launch_bounds__(1024) void sweep_wide_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = p.b0 + (int)blockIdx.x;
    const int Up = p.Up;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1, last = Nb - 1;
    const RidgeLine ridge = make_ridge(Ub, Nb);
    const float2 *Wb = (const float2 *)p.W + (size_t)b * p.Nr * Up;
    float *out = (BETA ? p.Bt : p.A) + (size_t)b * p.Nr * Up;
    float *offs = (BETA ? p.offB : p.offA) + (size_t)b * p.NC * p.NG;
    float *cur = lds, *nxt = lds + Up;
    float off = 0.f;  // cumulative integer offset (every thread tracks the same value)
    const bool bad = lengths_invalid(p, b);
    auto record = [&](const int kc) {
        for (int g = tid; g < p.NG; g += 1024) offs[(size_t)kc * p.NG + g] = off;
    };
    auto rebase_row = [&](float *row, const int n) {  // row is complete and visible; afterwards so is the re-based row
        const float mi = rintf(row[ridge.u_at(n)]);
        __syncthreads();
        for (int u = tid; u < Up; u += 1024) row[u] -= mi;
        off += mi;
        __syncthreads();
    };
    if (!BETA) {
        for (int u = tid; u < Up; u += 1024) {
            cur[u] = (u == 0) ? 0.f : kNeg;
            out[u] = cur[u];
        }
        record(0);
        __syncthreads();
        for (int n = 1; n <= last; ++n) {  // diagonal n from diagonal n - 1 and the edge weights leaving diagonal n - 1
            const float2 *wrow = Wb + (size_t)(n - 1) * Up;
            for (int u = tid; u < Up; u += 1024) {
                const float stay = cur[u] + wrow[u].x;                             // blank: (t-1, u) -> (t, u)
                const float emit = (u > 0) ? cur[u - 1] + wrow[u - 1].y : kNeg;    // label: (t, u-1) -> (t, u)
                nxt[u] = lse2(stay, emit);
            }
            __syncthreads();
            if ((n & (kRebase - 1)) == 0) {
                rebase_row(nxt, n);
                record(n / kRebase);
            }
            for (int u = tid; u < Up; u += 1024) out[(size_t)n * Up + u] = nxt[u];
            float *t = cur;
            cur = nxt, nxt = t;
        }
        if (tid == 0) {
            const double ll2 = bad ? (double)NAN : (double)off + (double)cur[Ub - 1] + (double)Wb[(size_t)last * Up + Ub - 1].x;
            p.ll[2 * b] = ll2;
            p.costs[b] = (float)(-ll2 * 0.6931471805599453);
        }
    } else {
        for (int u = tid; u < Up; u += 1024) cur[u] = (u == Ub - 1) ? 0.f : kNeg;  // the virtual terminal node (T_b, U_b - 1)
        __syncthreads();
        for (int n = last; n >= 0; --n) {  // diagonal n from diagonal n + 1 and the edge weights leaving diagonal n
            const float2 *wrow = Wb + (size_t)n * Up;
            for (int u = tid; u < Up; u += 1024) {
                const float stay = cur[u] + wrow[u].x;
                const float emit = ((u + 1 < Up) ? cur[u + 1] : kNeg) + wrow[u].y;
                nxt[u] = lse2(stay, emit);
            }
            __syncthreads();
            if ((n & (kRebase - 1)) == kRebase - 1 || n == last) {
                rebase_row(nxt, n);
                record(n / kRebase);
            }
            for (int u = tid; u < Up; u += 1024) out[(size_t)n * Up + u] = nxt[u];
            float *t = cur;
            cur = nxt, nxt = t;
        }
        if (tid == 0) {
            p.ll[2 * b + 1] = bad ? (double)NAN : (double)off + (double)cur[0];
            if (bad) p.costs[b] = NAN;
        }
    }
}

static hipError_t launch_sweep_wide(const LossParams &p, hipStream_t s) {
    const size_t shm = (size_t)2 * p.Up * sizeof(float);
    hipError_t e;
    if (shm > 64 * 1024) {
        if ((e = hipFuncSetAttribute((const void *)sweep_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void *)sweep_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL((sweep_wide_kernel<false>), dim3(p.nb), dim3(1024), shm, s, p);
    hipLaunchKernelGGL((sweep_wide_kernel<true>), dim3(p.nb), dim3(1024), shm, s, p);  // (the gradient pass needs both; the
    // alpha launch is not a dependency of the beta launch, but one stream keeps the caller's ordering contract simple)
    return hipGetLastError();
}

// Chunk length G (diagonals per LDS-DMA batch): the longest whose ring fits the LDS (measured at C2: G = 16 beats 8 by
// 3 % of the step, 4 loses 5 %).
hipError_t launch_sweeps(const LossParams &p, hipStream_t s) {
    switch (sweep_K(p.U)) {
        case 1: return launch_sweep_ld<1, 16>(p, s);
        case 2: return launch_sweep_ld<2, 16>(p, s);
        case 3: return launch_sweep_ld<3, 16>(p, s);
        case 4: return launch_sweep_ld<4, 16>(p, s);
        case 6: return launch_sweep_ld<6, 8>(p, s);
        case 8: return launch_sweep_ld<8, 8>(p, s);
        case 12: return launch_sweep_ld<12, 4>(p, s);
        case 16: return launch_sweep_ld<16, 4>(p, s);
        default: return (p.U <= kMaxU) ? launch_sweep_wide(p, s) : hipErrorInvalidValue;  // 1024 < maxU <= 8192
    }
}

}  // namespace rnnt
