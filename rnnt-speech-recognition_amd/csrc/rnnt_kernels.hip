// rnnt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the transducer loss.
//
// Replaces the three GPU stages of the reference's native op:
//   a-6  log-softmax denominator            -> cell_*_kernel<GRAD=false>   ("lsm" pass)
//   a-7/a-8 alpha / beta lattice recurrences -> sweep_kernel
//   a-9  fused-softmax gradient              -> cell_*_kernel<GRAD=true>    ("grad" pass)
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
//    LDS-DMA in double-buffered chunks of G diagonals.  alpha~/beta~ are re-based every 4
//    diagonals (f64 offsets kept aside) so f32 log-space values stay O(100) instead of O(T+U).
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

// Everything one lane does for its lattice cell once the cell's V logits sit in LDS at `xs`:
// GRAD=false: softmax denominator + the two lattice edge weights;  GRAD=true: the V gradients
// (written back into `xs`, zeros for padded cells).
template <int VP, bool V4, bool GRAD, bool OVL = false>
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
            ((float2 *)p.W)[wi] = make_float2(ob, ol);  // plain (write-back) store even in overlap mode, see below
        } else {
            const CellGrad g = cell_grad_setup<OVL>(p, cl, c);
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
// Small-vocabulary path (V <= VP <= 64): one lattice cell per LANE.
// A 256-thread workgroup owns 256 consecutive cells = one contiguous 1024*V-byte span of acts.
// ---------------------------------------------------------------------------------------------
template <int VP, bool V4, bool GRAD>
__global__ __launch_bounds__(256) void cell_small_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int V = p.V;
    const uint32_t c0 = blockIdx.x * 256u;
    const uint32_t c = c0 + (uint32_t)tid;
    const Cell cl = decode(p, c);

    unsigned long long *bm = (unsigned long long *)(lds + 256 * V);
    const unsigned long long msk = __ballot(cl.valid);
    if ((tid & 63) == 0) bm[tid >> 6] = msk;
    __syncthreads();
    const bool any = (bm[0] | bm[1] | bm[2] | bm[3]) != 0ull;

    const uint32_t nchunk = 64u * (uint32_t)V;  // 16-byte chunks in this block's span
    const size_t fbase = (size_t)c0 * V;
    const size_t total = (size_t)p.cells * V;

    if (!any) {
        if (GRAD) {  // an all-padding span: exact zeros, no reads
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (uint32_t k = tid; k < nchunk; k += 256)
                if (fbase + (size_t)k * 4 < total) *(float4 *)(p.grads + fbase + (size_t)k * 4) = z;
        }
        return;
    }

    // ---- stage this span's logits HBM -> LDS with 16-byte LDS-DMA (lane-linear destination) ----
    const float *gsrc = p.acts + fbase;
    for (uint32_t k0 = 0; k0 < nchunk; k0 += 256) {
        const uint32_t k = k0 + (uint32_t)tid;
        if (k < nchunk) {
            const uint32_t f0 = k * 4u;
            const uint32_t ca = fdiv(f0, p.divV);
            const uint32_t cb = V4 ? ca : min(fdiv(f0 + 3u, p.divV), 255u);
            const bool need = (((bm[ca >> 6] >> (ca & 63)) | (bm[cb >> 6] >> (cb & 63))) & 1ull) != 0ull;
            if (need)
                __builtin_amdgcn_global_load_lds((glb_void *)(gsrc + f0),
                                                 (lds_void *)(lds + (k0 + ((uint32_t)tid & ~63u)) * 4u), 16, 0, 0);
        }
    }
    wait_vm0();
    __syncthreads();

    cell_body<VP, V4, GRAD>(p, cl, c, lds + tid * V);

    if (GRAD) {
        __syncthreads();
        float *gdst = p.grads + fbase;
        for (uint32_t k = tid; k < nchunk; k += 256)
            if (fbase + (size_t)k * 4 < total) *(float4 *)(gdst + (size_t)k * 4) = ((const float4 *)lds)[k];
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
template <int VP, bool GRAD, bool OVL, bool AL = true>
__global__ __launch_bounds__(256) void cell_tile_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V;
    const TileGeom &tg = p.tile;
    // XCD-aware remap: hand each XCD (blockIdx % 8) a contiguous range of patches (bijective form)
    uint32_t bid;
    {
        const uint32_t nwg = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
        const uint32_t q = nwg >> 3, r = nwg & 7u;
        bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
        // The gradient pass walks each XCD's range backwards: the logits the lsm pass read LAST are the ones most
        // likely still in the 256 MiB Infinity Cache.
        if (GRAD && p.rev_grad) bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + ((xcd < r ? q + 1u : q) - 1u - idx);
    }
    const uint32_t q1 = fdiv(bid, tg.div_tu);
    const uint32_t tu = bid - q1 * (uint32_t)tg.tiles_u;
    const uint32_t bb = fdiv(q1, tg.div_tt);
    const uint32_t tt = q1 - bb * (uint32_t)tg.tiles_t;
    const int b = p.b0 + (int)bb;
    const int t0 = (int)tt * tg.TT, u0 = (int)tu * tg.UU;
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;

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
                __builtin_nontemporal_store(v, (v4f *)(p.grads + s0 + q * 4));
            }
        } else {
            const int a = (int)(s0 & 3), len = cols_in * V;
            float *base = p.grads + (s0 - a);
            for (int q = lane; q * 4 < a + len; q += 64) {
                const int e0 = q * 4 - a;  // element of the row segment held by the chunk's first float
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = src ? src[q * 4 + k] : 0.f;
                if (e0 >= 0 && e0 + 3 < len) {
                    *(float4 *)(base + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
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
                if (p.tune & (GRAD ? 2 : 4))
                    __builtin_amdgcn_global_load_lds((glb_void *)(src + q * 4), (lds_void *)(dst + q0 * 4), 16, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((glb_void *)(src + q * 4), (lds_void *)(dst + q0 * 4), 16, 0, 0);
            }
        }
    }
    if (OVL && GRAD) {
        // This kernel starts only after the lsm kernel has completed (stream order): tell slow-path sweep waves.
        if (blockIdx.x == 0 && tid == 0)
            __hip_atomic_store(p.flags + flag_lsm_kernel_done(p.B), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the logits are on their way; only now wait until BOTH sweeps of this utterance have published
        if (tid == 0) spin_until_ge(p.flags + flag_sweep(p.B, b), 2, p.flags + flag_err(p.B));
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
        if (GRAD || cl.valid) cell_body<VP, true, GRAD, OVL>(p, cl, c, lds + tid * V);
    } else if ((int)r < tg.TT) {
        const int a = (int)((patch0 + r * row_f) & 3);
        if (GRAD || cl.valid) cell_body<VP, false, GRAD, OVL>(p, cl, c, lds + r * row_lds + a + cu * V);
    }
    if (OVL && !GRAD) {
        // Publish this patch.  Its W words were stored write-back: after vmcnt(0) they sit in THIS XCD's L2,
        // which every CU of this XCD reads coherently.  The counter is kept per XCD so that the consuming
        // sweep wave can prove that all patches of its utterance ran on its own XCD (fast path); if the
        // dispatcher placed them elsewhere it waits for the kernel-end write-back instead (slow, still right).
        wait_vm0();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_add(p.flags + flag_lsm(b, my_xcd()), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (GRAD && !AL) {
        __syncthreads();
        for (int rr = wave; rr < rows_in; rr += 4) store_row(rr, lds + rr * row_lds);
    } else if (GRAD) {
        __syncthreads();
        for (int rr = wave; rr < rows_in; rr += 4) {
            const float4 *srcl = (const float4 *)(lds + rr * row_lds);
            float *dstg = p.grads + patch0 + rr * row_f;
            if (!(p.tune & 1))  // gradients are written once and not re-read by this op: keep them out of L2 / Infinity Cache
                for (int q = lane; q < q_in; q += 64) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(((const v4f *)srcl)[q], (v4f *)(dstg + q * 4));
                }
            else
                for (int q = lane; q < q_in; q += 64) *(float4 *)(dstg + q * 4) = srcl[q];
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

template <bool V4, bool GRAD>
__global__ __launch_bounds__(256) void cell_wave_kernel(const LossParams p) {
    const int lane = threadIdx.x & 63;
    const uint32_t w0 = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * 4u;
    const int V = p.V;
    for (uint32_t c = w0; c < p.cells; c += nw) {
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
#ifdef SWEEP_EXP_NODMA  // timing experiment (results wrong): the chunk buffers are never filled
    return;
#endif
    for (int i0 = 0; i0 < n16; i0 += 64) {
        const int k = i0 + lane;
        if (k < n16) __builtin_amdgcn_global_load_lds((glb_void *)(g + (size_t)k * 4), (lds_void *)(l + i0 * 4), 16, 0, 0);
    }
}

// Precision control.  alpha~/beta~ are stored relative to a per-block offset (kept in f64 on the
// side) so that the f32 values that carry probability mass stay O(10) instead of O(T+U).  The
// reference value is the lattice cell on the straight line (0,0)->(T_b-1,U_b-1) -- NOT the wave
// maximum: for near-uniform posteriors the alpha-maximum of a diagonal sits at the binomial centre,
// ~e^(0.19 n) above the cells that matter, which would leave those at magnitude ~100.
struct RidgeLine {
    uint32_t slope_fx;  // (U_b-1)/(N_b-1) in 16.16 fixed point
    __device__ __forceinline__ int u_at(int n) const { return (int)(((uint32_t)n * slope_fx + 32768u) >> 16); }
};
__device__ __forceinline__ RidgeLine make_ridge(int Ub, int Nb) {
    RidgeLine r;
    r.slope_fx = (Nb > 1) ? (((uint32_t)(Ub - 1) << 16) / (uint32_t)(Nb - 1)) : 0u;
    return r;
}

// Returns the INTEGER amount subtracted (exact in f32, so offsets accumulate exactly in f32 too).  The ridge cell
// (n - u_ref, u_ref) is a lattice node for every diagonal of a well-formed utterance, hence never log zero.
template <int K>
__device__ __forceinline__ float rebase(float (&v)[K], const int u_ref) {
    const int src_lane = u_ref / K, src_j = u_ref - src_lane * K;  // wave-uniform
    float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[0]), src_lane));
#pragma unroll
    for (int j = 1; j < K; ++j) {
        const float mj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[j]), src_lane));
        m = (src_j == j) ? mj : m;
    }
    const float mi = rintf(m);
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] -= mi;  // log zeros stay log zeros: |mi| << 1e30
    return mi;
}

// Log of the offsets, one per block of kRebase diagonals.  Lane (kc & 63) of `hist` holds the offset of block kc;
// every 64 blocks (and at the end) the register is flushed to the table with one coalesced store per column group.
struct OffsetLog {
    float *table;  // this utterance's [NC][NG] floats
    int ng;        // NG (row length of the table)
    int g0, gn;    // column groups this wave writes: [g0, g0 + gn)
    float hist;
    int lo, hi;    // block range recorded since the last flush (lo > hi: empty)
    __device__ __forceinline__ void init(float *t, int ng_) { init(t, ng_, 0, ng_); }
    __device__ __forceinline__ void init(float *t, int ng_, int g0_, int gn_) {
        table = t, ng = ng_, g0 = g0_, gn = gn_, hist = 0.f, lo = 1 << 30, hi = -1;
    }
    __device__ __forceinline__ void flush(const int lane) {
        if (lo > hi) return;
        const int kc = (lo & ~63) + lane;
        if (kc >= lo && kc <= hi)
            for (int g = g0; g < g0 + gn; ++g) st_f32_wt(table + (size_t)kc * ng + g, hist);
        lo = 1 << 30, hi = -1;
    }
    __device__ __forceinline__ void record(const int kc, const float off, const int lane) {
        if (lo <= hi && (kc >> 6) != (lo >> 6)) flush(lane);
        hist = (lane == (kc & 63)) ? off : hist;
        lo = min(lo, kc), hi = max(hi, kc);
    }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Overlap mode (p.flags != nullptr): a sweep wave starts before the lsm pass has finished.  It waits until
// every live patch of ITS utterance has published (one relaxed poll loop), then one agent-scope acquire
// drops any stale L1 lines; the W rows themselves were stored write-through by the lsm pass.
// Overlap mode job assignment.  The dispatcher hands workgroups to XCDs round-robin from a start that differs
// per launch, so which XCD produces an utterance's W rows is only known at run time (the lsm patches count
// themselves per XCD).  A sweep workgroup therefore CLAIMS its utterance: preferably an unclaimed one whose
// patches are being produced on the workgroup's own XCD (then W can be read through the shared L2 with no
// flush); once the lsm kernel is over, anything that is left.  Returns b, or -1 when all are taken.
// Called by ONE wave of the workgroup.
__device__ __forceinline__ int claim_utterance(const LossParams &p, const int lane) {
    const int B = p.B, me = my_xcd();
    int *flags = p.flags;
    for (int it = 0; it < (1 << 22); ++it) {
        if (__hip_atomic_load(flags + flag_nclaimed(B), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= B) return -1;
        const bool any_xcd = __hip_atomic_load(flags + flag_lsm_kernel_done(B), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            bool cand = false;
            if (b < B) {
                const int c0 = __hip_atomic_load(flags + flag_claim(B, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int here = any_xcd ? 1 : __hip_atomic_load(flags + flag_lsm(b, me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cand = (here > 0) && (c0 == 0);
            }
            const unsigned long long m = __ballot(cand);
            if (m == 0ull) continue;
            const int bb = b0 + __builtin_ctzll(m);
            int got = -1;
            if (lane == 0) {
                int expect = 0;
                if (__hip_atomic_compare_exchange_strong(flags + flag_claim(B, bb), &expect, 1, __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    got = bb;
                    __hip_atomic_fetch_add(flags + flag_nclaimed(B), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            got = __shfl(got, 0);
            if (got >= 0) return got;
            break;  // lost the race: rescan
        }
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(flags + flag_err(B), 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
}

__device__ __forceinline__ void sweep_wait_for_lsm(const LossParams &p, const int b, const int Tb, const int Ub,
                                                   const int dir) {
    if (p.flags == nullptr) return;
    const int need = ((Tb + p.tile.TT - 1) / p.tile.TT) * ((Ub + p.tile.UU - 1) / p.tile.UU);
    const int lane = threadIdx.x & 63;
    const int me = my_xcd();
    int *err = p.flags + flag_err(p.B);
    bool all_here = false;
    for (int it = 0; it < (1 << 22); ++it) {
        // lanes 0..7 each read one per-XCD counter of this utterance
        const int c = (lane < 8) ? __hip_atomic_load(p.flags + flag_lsm(b, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int total = c;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) total += __shfl_xor(total, o);
        total = __shfl(total, 0);
        if (total >= need) {
            all_here = (__shfl(c, me) >= need);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (it == (1 << 22) - 1) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) p.flags[flag_diag(p.B, b, dir)] = me | ((int)all_here << 8);
    if (!all_here)  // some patch ran on another XCD: its W words become visible only at the end of the lsm kernel
        spin_until_ge(p.flags + flag_lsm_kernel_done(p.B), 1, err);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale L1 lines; W itself is read from L2 / memory
}
// ... and when it is done: drain its write-through stores, then bump the utterance's "sweeps done" counter
// (alpha and beta each add one; the gradient pass waits for 2).
__device__ __forceinline__ void sweep_publish(const LossParams &p, const int b) {
    if (p.flags == nullptr) return;
    wait_vm0();
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(p.flags + flag_sweep(p.B, b), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void lsm_done_marker_kernel(int *flag) {
    __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
    // kernel is still alive (overlap mode), and nothing on this XCD re-reads them anyway.
    if (!COUNTED) {
        float *dst = row + lane * K + OFF / 4;
#pragma unroll
        for (int j = 0; j < K; ++j) st_f32_wt(dst + j, v[j]);
    } else {
        static_assert(OFF + 4 * K <= 4096 && OFF >= -4096, "store offset outside the immediate range");
#ifdef SWEEP_EXP_NOSTORE
        return;  // timing experiment (results wrong, counted waits over-wait harmlessly)
#endif
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
#ifdef SWEEP_EXP_NOLDS  // timing experiment (results wrong): the weight registers keep whatever they hold
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(q[j]) : "v"(addr));
    return;
#endif
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
__device__ __forceinline__ void alpha_step(float (&a)[K], const f32x2 (&w)[K]) {
    f32x2 de[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const f32x2 aa = {a[j], a[j]};
        de[j] = aa + w[j];  // .x blank: (t-1,u) -> (t,u);  .y label: (t,u) -> (t,u+1)
    }
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
#ifdef SWEEP_EXP_NOTRANS  // timing experiment (results wrong): no transcendentals
#pragma unroll
    for (int j = K - 1; j >= 0; --j) out[j] = vmax(u[j], l[j]) + (1.0f - fabsf(d[j]) * 1e-9f);
    return;
#endif
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
__device__ __forceinline__ void alpha_step_c(float (&a)[K], const f32x2 (&w)[K], float &edge) {
    f32x2 de[K];
    SWEEP_FENCE();
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {  // last column first: the DPP move waits for it
        const f32x2 aa = {a[j], a[j]};
        de[j] = aa + w[j];
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
__device__ __forceinline__ void beta_step_c(float (&bv)[K], const f32x2 (&w)[K], float &edge) {
    SWEEP_FENCE();
    edge = dpp_from_upper_lane(bv[0], edge);
    f32x2 s2[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {  // last column last: it waits for the DPP move
        const f32x2 br = {bv[j], (j == K - 1) ? edge : bv[j + 1]};
        s2[j] = br + w[j];
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
__device__ __forceinline__ void beta_step(float (&bv)[K], const f32x2 (&w)[K]) {
    const float from_right = dpp_from_upper_lane(bv[0], kNeg);
    float nv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const f32x2 br = {bv[j], (j == K - 1) ? from_right : bv[j + 1]};
        const f32x2 s2 = br + w[j];
        nv[j] = lse2(s2[0], s2[1]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = nv[j];
}

// State a sweep carries from step to step besides the diagonal itself.
struct SweepState {
    float off;       // cumulative (integer-valued) offset: true value = stored value + off
    OffsetLog log;
    float *row;      // wave-uniform base of the output row of the NEXT diagonal to be stored
    float edge;      // what DPP shifted in last (edge lane: log zero, see alpha_step_c)
};

// Fully unrolled, explicitly pipelined steps of one chunk (compile-time recursion over the step index
// II so that every LDS offset is an immediate and the two weight register sets ping-pong by name).
template <int K, int G, int II>
__device__ __forceinline__ void alpha_fast_steps(float (&a)[K], f32x2 (&wq)[2][K], const uint32_t abase, SweepState &st,
                                                 const int voff, const int lane, const int r0, const RidgeLine &ridge) {
    if constexpr (II < G) {
        constexpr int cur = II & 1, nxt = cur ^ 1;
        if constexpr (II + 1 < G) {
            lds_issue_row<K, II + 1>(wq[nxt], abase);
            lds_wait<K>();  // row II has landed, row II+1 stays in flight
        } else {
            lds_wait<0>();
        }
        const int n = r0 + II + 1;
        alpha_step_c<K>(a, wq[cur], st.edge);
        if ((n & (kRebase - 1)) == 0) {
            st.off += rebase<K>(a, ridge.u_at(n));
            st.log.record(n / kRebase, st.off, lane);
        }
        constexpr int R = rows_per_base(K);
        store_diag<K, true, (II % R) * 64 * K * 4>(st.row, voff, lane, a);
        if constexpr (II % R == R - 1 || II == G - 1) st.row += (II % R + 1) * 64 * K;
        alpha_fast_steps<K, G, II + 1>(a, wq, abase, st, voff, lane, r0, ridge);
    }
}

template <int K, int G, int II>
__device__ __forceinline__ void beta_fast_steps(float (&bv)[K], f32x2 (&wq)[2][K], const uint32_t abase, SweepState &st,
                                                const int voff, const int lane, const int r0, const RidgeLine &ridge) {
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
        beta_step_c<K>(bv, wq[cur], st.edge);
        if ((n & (kRebase - 1)) == kRebase - 1) {
            st.off += rebase<K>(bv, ridge.u_at(n));
            st.log.record(n / kRebase, st.off, lane);
        }
        constexpr int R = rows_per_base(K);
        store_diag<K, true, -(II % R) * 64 * K * 4>(st.row, voff, lane, bv);
        if constexpr (II % R == R - 1 || II == G - 1) st.row -= (II % R + 1) * 64 * K;
        beta_fast_steps<K, G, II + 1>(bv, wq, abase, st, voff, lane, r0, ridge);
    }
}

template <int K, int G, bool COUNTED>
__device__ void alpha_sweep(const LossParams &p, float *lds, const int b, const int lane) {
    constexpr int Up = 64 * K;
    constexpr int chunkf = G * 2 * Up, n16 = chunkf / 4;
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
    const int Nb = Tb + Ub - 1;
    sweep_wait_for_lsm(p, b, Tb, Ub, 0);
    const RidgeLine ridge = make_ridge(Ub, Nb);
    const float *Wb = p.W + (size_t)b * p.Nr * 2 * Up;
    float *out = p.A + (size_t)b * p.Nr * Up;  // wave-uniform; the lane offset is added at the store
    const int voff = lane * K * 4;
    const int u0 = lane * K;
    float *buf0 = lds, *buf1 = lds + chunkf;

    float a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (u0 + j == 0) ? 0.f : kNeg;
    store_diag<K, false>(out, voff, lane, a);
    SweepState st;
    st.off = 0.f;
    st.edge = kNeg;
    st.log.init(p.offA + (size_t)b * p.NC * p.NG, p.NG);
    st.log.record(0, 0.f, lane);
    st.row = out + Up;  // diagonal 1
    const int last_row = Nb - 1;  // rows 0..Nb-2 feed the steps, row Nb-1 the final likelihood
    const int nchunks = last_row / G + 1;

    dma_rows(Wb, buf0, n16, lane);
    bool prev_full = false;
    for (int ck = 0; ck < nchunks; ++ck) {
        if (COUNTED && prev_full)
            wait_vm_counted<G * store_pieces(K)>();
        else
            wait_vm0();
        const float *cur = ((ck & 1) ? buf1 : buf0) + 2 * u0;
        if (ck + 1 < nchunks) dma_rows(Wb + (size_t)(ck + 1) * chunkf, (ck & 1) ? buf0 : buf1, n16, lane);
        const int r0 = ck * G;
        if (COUNTED && K <= 15 && r0 + G <= last_row) {
            // every row of this chunk feeds a step: straight-line code, explicit software pipeline
            // (row i+1's LDS reads are in flight under step i)
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            f32x2 wq[2][K];
            lds_issue_row<K, 0>(wq[0], abase);
            alpha_fast_steps<K, G, 0>(a, wq, abase, st, voff, lane, r0, ridge);
            prev_full = true;
        } else {
            for (int i = 0; i < G; ++i) {
                const int n = r0 + i + 1;
                if (n > last_row) break;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Up);
                alpha_step<K>(a, wc);
                if ((n & (kRebase - 1)) == 0) {
                    st.off += rebase<K>(a, ridge.u_at(n));
                    st.log.record(n / kRebase, st.off, lane);
                }
                store_diag<K, false>(st.row, voff, lane, a);
                st.row += Up;
            }
            prev_full = false;
        }
    }
    st.log.flush(lane);
    {
        const float *wrow = (((nchunks - 1) & 1) ? buf1 : buf0) + (last_row % G) * 2 * Up + 2 * u0;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (u0 + j == Ub - 1) {
                const double ll2 = (double)st.off + (double)a[j] + (double)wrow[2 * j];
                st_f64_wt(p.ll + 2 * b, ll2);
                st_f32_wt(p.costs + b, (float)(-ll2 * 0.6931471805599453));
            }
    }
    sweep_publish(p, b);
}

template <int K, int G, bool COUNTED>
__device__ void beta_sweep(const LossParams &p, float *lds, const int b, const int lane) {
    constexpr int Up = 64 * K;
    constexpr int chunkf = G * 2 * Up, n16 = chunkf / 4;
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
    const int Nb = Tb + Ub - 1;
    sweep_wait_for_lsm(p, b, Tb, Ub, 1);
    const RidgeLine ridge = make_ridge(Ub, Nb);
    const float *Wb = p.W + (size_t)b * p.Nr * 2 * Up;
    float *out = p.Bt + (size_t)b * p.Nr * Up;  // wave-uniform; the lane offset is added at the store
    const int voff = lane * K * 4;
    const int u0 = lane * K;
    float *buf0 = lds, *buf1 = lds + chunkf;

    float bv[K];  // beta on the diagonal below; starts as the virtual terminal node (Tb, Ub-1) = 0
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = (u0 + j == Ub - 1) ? 0.f : kNeg;
    const int last = Nb - 1;
    const int ckl = last / G;
    SweepState st;
    st.off = 0.f;
    st.edge = kNeg;
    st.log.init(p.offB + (size_t)b * p.NC * p.NG, p.NG);
    st.row = out + (size_t)last * Up;

    dma_rows(Wb + (size_t)ckl * chunkf, (ckl & 1) ? buf1 : buf0, n16, lane);
    bool prev_full = false;
    for (int ck = ckl; ck >= 0; --ck) {
        if (COUNTED && prev_full)
            wait_vm_counted<G * store_pieces(K)>();
        else
            wait_vm0();
        const float *cur = ((ck & 1) ? buf1 : buf0) + 2 * u0;
        if (ck > 0) dma_rows(Wb + (size_t)(ck - 1) * chunkf, (ck & 1) ? buf0 : buf1, n16, lane);
        const int r0 = ck * G;
        if (COUNTED && K <= 15 && r0 + G - 1 < last) {  // whole chunk strictly below the first (terminal) diagonal
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            f32x2 wq[2][K];
            lds_issue_row<K, G - 1>(wq[0], abase);
            beta_fast_steps<K, G, 0>(bv, wq, abase, st, voff, lane, r0, ridge);
            prev_full = true;
        } else {
            for (int ii = 0; ii < G; ++ii) {
                const int i = G - 1 - ii;
                const int n = r0 + i;
                if (n > last) continue;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Up);
                beta_step<K>(bv, wc);
                if (((n & (kRebase - 1)) == kRebase - 1) || n == last) {
                    st.off += rebase<K>(bv, ridge.u_at(n));
                    st.log.record(n / kRebase, st.off, lane);
                }
                store_diag<K, false>(st.row, voff, lane, bv);
                st.row -= Up;
            }
            prev_full = false;
        }
    }
    st.log.flush(lane);
    if (lane == 0) st_f64_wt(p.ll + 2 * b + 1, (double)st.off + (double)bv[0]);
    sweep_publish(p, b);
}

template <int K, int G, bool COUNTED>
__global__ __launch_bounds__(64) void sweep_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int b = p.b0 + (int)(blockIdx.x >> 1);
    if (blockIdx.x & 1)
        beta_sweep<K, G, COUNTED>(p, lds, b, lane);
    else
        alpha_sweep<K, G, COUNTED>(p, lds, b, lane);
}

// Overlap-mode form: one workgroup = the alpha wave and the beta wave of one claimed utterance.  It is launched
// with kPairLdsBytes of dynamic LDS -- far more than it uses -- so that no cell-pass workgroup fits beside it:
// the sweep waves are the critical path and a lone wave that has to share its SIMD's issue port with
// transcendental-heavy lsm/grad waves runs ~1.5x slower (measured), priority or not.
constexpr size_t kPairLdsBytes = 140 * 1024;

template <int K, int G>
__global__ __launch_bounds__(128) void sweep_pair_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (no static __shared__ object: it would shift the dynamic base off its 16-byte alignment)
    int *s_job = (int *)(lds + kPairLdsBytes / sizeof(float) - 4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *my_lds = lds + wave * (2 * G * 2 * 64 * K);  // each direction has its own chunk buffers
    for (;;) {
        if (wave == 0) {
            const int j = claim_utterance(p, lane);
            if (lane == 0) *s_job = j;
        }
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(*s_job);
        __syncthreads();
        if (b < 0) return;
        if (wave)
            beta_sweep<K, G, true>(p, my_lds, b, lane);
        else
            alpha_sweep<K, G, true>(p, my_lds, b, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static bool small_path_ok(const LossParams &p, bool grad) {
    if (p.V > 60) return false;  // 256*V*4 B of LDS must stay under the 64 KiB dynamic limit
    const size_t total = (size_t)p.cells * p.V;
    if (total % 4 != 0 || total >= (1ull << 31)) return false;
    if (((uintptr_t)p.acts & 15) != 0) return false;
    if (grad && ((uintptr_t)p.grads & 15) != 0) return false;
    return true;
}

bool tile_path_ok(const LossParams &p, bool grad) {
    if (p.V > 60) return false;
    // rows that start off a 16-byte boundary (V % 4 != 0) are staged from the enclosing aligned span: the whole tensor
    // must end on a 16-byte boundary for that span never to leave it
    if ((p.V % 4) != 0 && (((size_t)p.cells * (size_t)p.V) % 4) != 0) return false;
    if (((uintptr_t)p.acts & 15) != 0) return false;
    if (grad && ((uintptr_t)p.grads & 15) != 0) return false;
    const char *e = getenv("RNNT_CELL_PATH");  // "flat" forces the 256-consecutive-cells kernels
    if (e && e[0] == 'f') return false;
    return true;
}

template <bool GRAD>
static hipError_t launch_cell(const LossParams &p, hipStream_t s, bool overlap) {
    if (overlap) {  // caller checked overlap_path_ok(): patch kernels, with the hand-off protocol compiled in
        const unsigned blocks = (unsigned)p.nb * p.tile.tiles_t * p.tile.tiles_u;
        const size_t shm = (size_t)256 * p.V * sizeof(float) + 64;
        if (p.V <= 32)
            hipLaunchKernelGGL((cell_tile_kernel<32, GRAD, true>), dim3(blocks), dim3(256), shm, s, p);
        else
            hipLaunchKernelGGL((cell_tile_kernel<64, GRAD, true>), dim3(blocks), dim3(256), shm, s, p);
        return hipGetLastError();
    }
    if (tile_path_ok(p, GRAD)) {
        const unsigned blocks = (unsigned)p.nb * p.tile.tiles_t * p.tile.tiles_u;
        const size_t shm = (size_t)256 * p.V * sizeof(float) + 64;
        if ((p.V % 4) != 0) {
            const size_t pitch = (size_t)((p.tile.UU * p.V + 3 + 3) & ~3);
            size_t shmu = (size_t)p.tile.TT * pitch * sizeof(float);
            if (shmu < shm) shmu = shm;
            if (p.V <= 32)
                hipLaunchKernelGGL((cell_tile_kernel<32, GRAD, false, false>), dim3(blocks), dim3(256), shmu, s, p);
            else
                hipLaunchKernelGGL((cell_tile_kernel<64, GRAD, false, false>), dim3(blocks), dim3(256), shmu, s, p);
        } else if (p.V <= 32)
            hipLaunchKernelGGL((cell_tile_kernel<32, GRAD, false>), dim3(blocks), dim3(256), shm, s, p);
        else
            hipLaunchKernelGGL((cell_tile_kernel<64, GRAD, false>), dim3(blocks), dim3(256), shm, s, p);
    } else if (small_path_ok(p, GRAD)) {
        const unsigned blocks = (p.cells + 255u) / 256u;
        const size_t shm = (size_t)256 * p.V * sizeof(float) + 64;
        const bool v4 = (p.V % 4) == 0;
        if (p.V <= 32) {
            if (v4)
                hipLaunchKernelGGL((cell_small_kernel<32, true, GRAD>), dim3(blocks), dim3(256), shm, s, p);
            else
                hipLaunchKernelGGL((cell_small_kernel<32, false, GRAD>), dim3(blocks), dim3(256), shm, s, p);
        } else {
            if (v4)
                hipLaunchKernelGGL((cell_small_kernel<64, true, GRAD>), dim3(blocks), dim3(256), shm, s, p);
            else
                hipLaunchKernelGGL((cell_small_kernel<64, false, GRAD>), dim3(blocks), dim3(256), shm, s, p);
        }
    } else {
        const bool v4 = (p.V % 4) == 0 && ((uintptr_t)p.acts & 15) == 0 && (!GRAD || ((uintptr_t)p.grads & 15) == 0);
        unsigned blocks = (p.cells + 3u) / 4u;
        if (blocks > 256u * 16u) blocks = 256u * 16u;
        if (v4)
            hipLaunchKernelGGL((cell_wave_kernel<true, GRAD>), dim3(blocks), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((cell_wave_kernel<false, GRAD>), dim3(blocks), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_lsm(const LossParams &p, hipStream_t s, bool overlap) { return launch_cell<false>(p, s, overlap); }
hipError_t launch_grad(const LossParams &p, hipStream_t s, bool overlap) { return launch_cell<true>(p, s, overlap); }

// RNNT_SWEEP_MODE: 1 (default) = sweeping wave + loader wave per (utterance, direction) (sweep_ld_kernel)
//                  5 = one wave that also issues its own LDS-DMA (sweep_kernel; also what the overlap mode runs)
//                  0 = as 5 with compiler-scheduled LDS reads and vmcnt(0) at chunk boundaries
//                  4 = two sweeping waves per direction sharing the columns (sweep_split_kernel)
// Two older multi-wave forms with one lattice column per lane (a barrier per diagonal: 164-175 us at C2; LDS progress
// counters instead of the barrier: 195 us) were removed; profiles/r01_notes.md keeps their measurements.
static int sweep_mode() {
    const char *e = getenv("RNNT_SWEEP_MODE");
    if (e && e[0] == '0') return 0;
    if (e && e[0] == '4') return 4;
    if (e && e[0] == '5') return 5;
    return 1;
}

template <int K, int G>
static hipError_t launch_sweep_pair(const LossParams &p, hipStream_t s) {
    {
        hipError_t e = hipFuncSetAttribute((const void *)sweep_pair_kernel<K, G>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairLdsBytes);
        if (e != hipSuccess) return e;
    }
    const int nwg = p.B < 48 ? p.B : 48;  // at most 48 CUs are taken away from the bandwidth passes
    hipLaunchKernelGGL((sweep_pair_kernel<K, G>), dim3(nwg), dim3(128), kPairLdsBytes, s, p);
    return hipGetLastError();
}

template <int K, int G>
static hipError_t launch_sweep_kg(const LossParams &p, hipStream_t s) {
    if (p.flags != nullptr) return launch_sweep_pair<K, G>(p, s);
    const size_t shm = (size_t)2 * G * 2 * 64 * K * sizeof(float);
    if (sweep_mode() != 0)
        hipLaunchKernelGGL((sweep_kernel<K, G, true>), dim3(2 * p.nb), dim3(64), shm, s, p);
    else
        hipLaunchKernelGGL((sweep_kernel<K, G, false>), dim3(2 * p.nb), dim3(64), shm, s, p);
    return hipGetLastError();
}

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

hipError_t launch_lsm_done_marker(const LossParams &p, hipStream_t s) {
    hipLaunchKernelGGL(lsm_done_marker_kernel, dim3(1), dim3(1), 0, s, p.flags + flag_lsm_kernel_done(p.B));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Split sweep (RNNT_SWEEP_MODE=4): the columns of an utterance are shared by TWO waves per direction -- the low wave
// owns columns [0, 64 KA) with KA columns per lane, the high wave [64 KA, 64 (KA + KB)) with KB -- so each wave issues
// about half the instructions per diagonal of the single-wave sweep (which is issue-bound: 6 transcendentals + ~30 other
// instructions per diagonal at K = 3).  Each wave is the single-wave sweep on its own columns (own LDS-DMA double buffer,
// counted waits, fully unrolled chunks of G diagonals, own integer re-basing against its column nearest the ridge); the
// only coupling is the value that crosses the column boundary once per diagonal:
//   alpha: the low wave's label edge out of column 64 KA - 1 feeds the high wave's column 64 KA;
//   beta : the high wave's beta of column 64 KA feeds the low wave's label edge of column 64 KA - 1.
// The producer writes {value, its offset} into an LDS array indexed by the diagonal (never reused: no back-pressure, so
// the producer never waits and there is no cycle to deadlock on) and publishes a counter after every chunk; the consumer
// runs one chunk behind, checks the counter once per chunk (bounded poll) and reads one entry per diagonal together
// with its weight rows.  Both offsets are integers, so  value + (producer offset - consumer offset)  is exact.
// A high wave whose columns lie beyond U_b has nothing to do and exits; the low wave then reads {log zero, 0}.
// ---------------------------------------------------------------------------------------------
template <int K, int G, int UPT>
__device__ __forceinline__ void dma_rows_part(const float *g, float *l, const int lane) {
    constexpr int per_row = 32 * K, total = G * per_row;  // 16-byte units: one row segment, the whole chunk
    static_assert(total % 64 == 0, "chunk segment must be whole wave-instructions");
#pragma unroll
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int k = i0 + lane;
        const int row = k / per_row, c = k - row * per_row;
        __builtin_amdgcn_global_load_lds((glb_void *)(g + (size_t)row * (2 * UPT) + c * 4), (lds_void *)(l + i0 * 4), 16, 0, 0);
    }
}
constexpr int dma_part_pieces(int K, int G) { return G * 32 * K / 64; }

// Re-base against the wave's own column `u_local` (0 .. 64 K - 1); skipped while that cell is still log zero.
template <int K>
__device__ __forceinline__ float rebase_local(float (&v)[K], const int u_local) {
    const int src_lane = u_local / K, src_j = u_local - src_lane * K;  // wave-uniform
    float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[0]), src_lane));
#pragma unroll
    for (int j = 1; j < K; ++j) {
        const float mj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[j]), src_lane));
        m = (src_j == j) ? mj : m;
    }
    if (!(m > kNegTest)) return 0.f;
    const float mi = rintf(m);
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] -= mi;
    return mi;
}

// alpha step with an explicit value entering column 0 of lane 0 (`fill`); returns nothing, `e_last` = label edge out of
// the lane's last column (lane 63's is what crosses to the next wave).
template <int K>
__device__ __forceinline__ void alpha_step_x(float (&a)[K], const f32x2 (&w)[K], const float fill, float &e_last) {
    f32x2 de[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const f32x2 aa = {a[j], a[j]};
        de[j] = aa + w[j];
    }
    e_last = de[K - 1][1];
    const float from_left = dpp_from_lower_lane(e_last, fill);
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = lse2(de[j][0], (j == 0) ? from_left : de[j - 1][1]);
}
template <int K>
__device__ __forceinline__ void beta_step_x(float (&bv)[K], const f32x2 (&w)[K], const float fill) {
    const float from_right = dpp_from_upper_lane(bv[0], fill);
    float nv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const f32x2 br = {bv[j], (j == K - 1) ? from_right : bv[j + 1]};
        const f32x2 s2 = br + w[j];
        nv[j] = lse2(s2[0], s2[1]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = nv[j];
}

__device__ __forceinline__ int ridge_local(const RidgeLine &ridge, const int n, const int cb, const int width) {
    return min(max(ridge.u_at(n) - cb, 0), width - 1);
}

template <int K, int G, int UPT, int II, bool PROD>
__device__ __forceinline__ void alpha_split_steps(float (&a)[K], f32x2 (&wq)[2][K], f32x2 (&xq)[2], const uint32_t abase,
                                                  const uint32_t xaddr, SweepState &st, const int voff, const int lane,
                                                  const int r0, const RidgeLine &ridge, const int cb) {
    if constexpr (II < G) {
        constexpr int cur = II & 1, nxt = cur ^ 1;
        if constexpr (II + 1 < G) {
            lds_issue_row<K, II + 1>(wq[nxt], abase);
            if constexpr (!PROD) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xq[nxt]) : "v"(xaddr), "n"((II + 1) * 8));
            lds_wait<K + (PROD ? 0 : 1)>();  // row II (and the boundary entry II) landed; the newest reads stay in flight
        } else {
            lds_wait<0>();
        }
        const int n = r0 + II + 1;
        float fill = kNeg, e_last;
        if constexpr (!PROD) fill = xq[cur][0] + (xq[cur][1] - st.off);
        alpha_step_x<K>(a, wq[cur], fill, e_last);
        if constexpr (PROD) {
            const f32x2 o = {e_last, st.off};  // relative to the offset the diagonal-r values carry (before this step's re-base)
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(xaddr), "v"(o), "n"(II * 8) : "memory");
        }
        if ((n & (kRebase - 1)) == 0) {
            st.off += rebase_local<K>(a, ridge_local(ridge, n, cb, 64 * K));
            st.log.record(n / kRebase, st.off, lane);
        }
        store_diag<K, true>(st.row, voff, lane, a);
        st.row += UPT;
        alpha_split_steps<K, G, UPT, II + 1, PROD>(a, wq, xq, abase, xaddr, st, voff, lane, r0, ridge, cb);
    }
}

template <int K, int G, int UPT, int II, bool PROD>
__device__ __forceinline__ void beta_split_steps(float (&bv)[K], f32x2 (&wq)[2][K], f32x2 (&xq)[2], const uint32_t abase,
                                                 const uint32_t xaddr, SweepState &st, const int voff, const int lane,
                                                 const int r0, const RidgeLine &ridge, const int cb) {
    if constexpr (II < G) {
        constexpr int cur = II & 1, nxt = cur ^ 1;
        constexpr int i = G - 1 - II;
        if constexpr (i > 0) {
            lds_issue_row<K, i - 1>(wq[nxt], abase);
            if constexpr (!PROD) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xq[nxt]) : "v"(xaddr), "n"((i - 1) * 8));
            lds_wait<K + (PROD ? 0 : 1)>();
        } else {
            lds_wait<0>();
        }
        const int n = r0 + i;
        float fill = kNeg;
        if constexpr (!PROD) fill = xq[cur][0] + (xq[cur][1] - st.off);
        if constexpr (PROD) {
            const f32x2 o = {bv[0], st.off};  // beta of the diagonal below, before this step
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(xaddr), "v"(o), "n"(i * 8) : "memory");
        }
        beta_step_x<K>(bv, wq[cur], fill);
        if ((n & (kRebase - 1)) == kRebase - 1) {
            st.off += rebase_local<K>(bv, ridge_local(ridge, n, cb, 64 * K));
            st.log.record(n / kRebase, st.off, lane);
        }
        store_diag<K, true>(st.row, voff, lane, bv);
        st.row -= UPT;
        beta_split_steps<K, G, UPT, II + 1, PROD>(bv, wq, xq, abase, xaddr, st, voff, lane, r0, ridge, cb);
    }
}

#ifdef SPLIT_TRACE
__device__ long long *g_split_trace = nullptr;  // [4 waves][256] s_memtime at every chunk start of workgroup 0 (dev tool)
#define SPLIT_STAMP(w, i) \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (i) < 256) g_split_trace[(w) * 256 + (i)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define SPLIT_STAMP(w, i)
#endif
// LDS-side plumbing of one split-sweep wave
struct SplitLink {
    float2 *ring;      // [Nr] boundary entries {value, producer offset}, indexed by the diagonal
    float2 *trash;     // [64 + G] dump for the producer lanes that are not the boundary lane
    uint32_t prog;     // LDS byte address of the producer's progress counter
    bool peer;         // the other wave of this direction is alive
};

template <int K, int G, int UPT, bool PROD>
__device__ void alpha_split_sweep(const LossParams &p, float *buf0, float *buf1, const SplitLink &lk, const int b,
                                  const int lane, const int cb) {
    constexpr int Wd = 64 * K, chunkf = G * 2 * Wd;
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
    const int Nb = Tb + Ub - 1;
    const RidgeLine ridge = make_ridge(Ub, Nb);
    const float *Wb = p.W + (size_t)b * p.Nr * 2 * UPT + 2 * cb;
    float *out = p.A + (size_t)b * p.Nr * UPT + cb;
    const int voff = lane * K * 4;
    const int u0 = cb + lane * K;
    const uint32_t ring_a = (uint32_t)(uintptr_t)((lds_void *)lk.ring);
    const uint32_t trash_a = (uint32_t)(uintptr_t)((lds_void *)lk.trash);

    float a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (u0 + j == 0) ? 0.f : kNeg;
    store_diag<K, false>(out, voff, lane, a);
    SweepState st;
    st.off = 0.f;
    st.edge = kNeg;
    st.log.init(p.offA + (size_t)b * p.NC * p.NG, p.NG, cb / 64, K);
    st.log.record(0, 0.f, lane);
    st.row = out + UPT;
    const int last_row = Nb - 1;
    const int nchunks = last_row / G + 1;

    dma_rows_part<K, G, UPT>(Wb, buf0, lane);
    bool prev_full = false;
    for (int ck = 0; ck < nchunks; ++ck) {
        if (prev_full)
            wait_vm_counted<G * store_pieces(K)>();
        else
            wait_vm0();
        const float *cur = ((ck & 1) ? buf1 : buf0) + 2 * lane * K;
        if (ck + 1 < nchunks) dma_rows_part<K, G, UPT>(Wb + (size_t)(ck + 1) * G * 2 * UPT, (ck & 1) ? buf0 : buf1, lane);
        const int r0 = ck * G;
        const int hi = min(r0 + G, last_row);  // this chunk's steps consume the boundary entries r0 .. hi-1
        SPLIT_STAMP(PROD ? 0 : 1, 2 * ck);
        if (!PROD && lk.peer) lds_wait_ge(lk.prog, hi);
        SPLIT_STAMP(PROD ? 0 : 1, 2 * ck + 1);
        if (r0 + G <= last_row) {
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            const uint32_t xaddr = PROD ? ((lane == 63) ? ring_a + (uint32_t)r0 * 8u : trash_a + (uint32_t)lane * 8u)
                                        : ring_a + (uint32_t)r0 * 8u;
            f32x2 wq[2][K], xq[2];
            lds_issue_row<K, 0>(wq[0], abase);
            if (!PROD) asm volatile("ds_read_b64 %0, %1" : "=v"(xq[0]) : "v"(xaddr));
            alpha_split_steps<K, G, UPT, 0, PROD>(a, wq, xq, abase, xaddr, st, voff, lane, r0, ridge, cb);
            prev_full = true;
        } else {
            for (int i = 0; i < G; ++i) {
                const int r = r0 + i, n = r + 1;
                if (n > last_row) break;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Wd);
                float fill = kNeg, e_last;
                if (!PROD) {
                    const float2 x = lk.ring[r];
                    fill = x.x + (x.y - st.off);
                }
                alpha_step_x<K>(a, wc, fill, e_last);
                if (PROD && lane == 63) lk.ring[r] = make_float2(e_last, st.off);
                if ((n & (kRebase - 1)) == 0) {
                    st.off += rebase_local<K>(a, ridge_local(ridge, n, cb, Wd));
                    st.log.record(n / kRebase, st.off, lane);
                }
                store_diag<K, false>(st.row, voff, lane, a);
                st.row += UPT;
            }
            prev_full = false;
        }
        if (PROD) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this chunk's boundary entries are in LDS
            if (lane == 0) lds_post(lk.prog, hi);
        }
    }
    st.log.flush(lane);
    {
        const float *wrow = (((nchunks - 1) & 1) ? buf1 : buf0) + (last_row % G) * 2 * Wd + 2 * lane * K;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (u0 + j == Ub - 1) {
                const double ll2 = (double)st.off + (double)a[j] + (double)wrow[2 * j];
                st_f64_wt(p.ll + 2 * b, ll2);
                st_f32_wt(p.costs + b, (float)(-ll2 * 0.6931471805599453));
            }
    }
}

template <int K, int G, int UPT, bool PROD>
__device__ void beta_split_sweep(const LossParams &p, float *buf0, float *buf1, const SplitLink &lk, const int b,
                                 const int lane, const int cb) {
    constexpr int Wd = 64 * K, chunkf = G * 2 * Wd;
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
    const int Nb = Tb + Ub - 1;
    const RidgeLine ridge = make_ridge(Ub, Nb);
    const float *Wb = p.W + (size_t)b * p.Nr * 2 * UPT + 2 * cb;
    float *out = p.Bt + (size_t)b * p.Nr * UPT + cb;
    const int voff = lane * K * 4;
    const int u0 = cb + lane * K;
    const uint32_t ring_a = (uint32_t)(uintptr_t)((lds_void *)lk.ring);
    const uint32_t trash_a = (uint32_t)(uintptr_t)((lds_void *)lk.trash);

    float bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = (u0 + j == Ub - 1) ? 0.f : kNeg;
    const int last = Nb - 1;
    const int ckl = last / G;
    SweepState st;
    st.off = 0.f;
    st.edge = kNeg;
    st.log.init(p.offB + (size_t)b * p.NC * p.NG, p.NG, cb / 64, K);
    st.row = out + (size_t)last * UPT;

    dma_rows_part<K, G, UPT>(Wb + (size_t)ckl * G * 2 * UPT, (ckl & 1) ? buf1 : buf0, lane);
    bool prev_full = false;
    for (int ck = ckl; ck >= 0; --ck) {
        if (prev_full)
            wait_vm_counted<G * store_pieces(K)>();
        else
            wait_vm0();
        const float *cur = ((ck & 1) ? buf1 : buf0) + 2 * lane * K;
        if (ck > 0) dma_rows_part<K, G, UPT>(Wb + (size_t)(ck - 1) * G * 2 * UPT, (ck & 1) ? buf0 : buf1, lane);
        const int r0 = ck * G;
        const int done = last - r0 + 1;  // steps finished once this chunk is (entries last .. r0 written)
        SPLIT_STAMP(PROD ? 3 : 2, 2 * (ckl - ck));
        if (!PROD && lk.peer) lds_wait_ge(lk.prog, done);
        SPLIT_STAMP(PROD ? 3 : 2, 2 * (ckl - ck) + 1);
        if (r0 + G - 1 < last) {
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            const uint32_t xaddr = PROD ? ((lane == 0) ? ring_a + (uint32_t)r0 * 8u : trash_a + (uint32_t)lane * 8u)
                                        : ring_a + (uint32_t)r0 * 8u;
            f32x2 wq[2][K], xq[2];
            lds_issue_row<K, G - 1>(wq[0], abase);
            if (!PROD) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xq[0]) : "v"(xaddr), "n"((G - 1) * 8));
            beta_split_steps<K, G, UPT, 0, PROD>(bv, wq, xq, abase, xaddr, st, voff, lane, r0, ridge, cb);
            prev_full = true;
        } else {
            for (int ii = 0; ii < G; ++ii) {
                const int i = G - 1 - ii;
                const int n = r0 + i;
                if (n > last) continue;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Wd);
                float fill = kNeg;
                if (!PROD) {
                    const float2 x = lk.ring[n];
                    fill = x.x + (x.y - st.off);
                }
                if (PROD && lane == 0) lk.ring[n] = make_float2(bv[0], st.off);
                beta_step_x<K>(bv, wc, fill);
                if (((n & (kRebase - 1)) == kRebase - 1) || n == last) {
                    st.off += rebase_local<K>(bv, ridge_local(ridge, n, cb, Wd));
                    st.log.record(n / kRebase, st.off, lane);
                }
                store_diag<K, false>(st.row, voff, lane, bv);
                st.row -= UPT;
            }
            prev_full = false;
        }
        if (PROD) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_post(lk.prog, done);
        }
    }
    st.log.flush(lane);
    if (lane == 0 && cb == 0) st_f64_wt(p.ll + 2 * b + 1, (double)st.off + (double)bv[0]);
}

template <int KA, int KB, int G>
constexpr size_t split_lds_floats() { return (size_t)2 * (2 * G * 2 * 64 * KA) + (size_t)2 * (2 * G * 2 * 64 * KB); }

template <int KA, int KB, int G>
__global__ __launch_bounds__(256) void sweep_split_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int UPT = 64 * (KA + KB);
    constexpr int fA = 2 * G * 2 * 64 * KA, fB = 2 * G * 2 * 64 * KB;  // floats per wave (two chunk buffers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = p.b0 + (int)blockIdx.x;
    const int Ub = p.label_lengths[b] + 1;
    const bool peer = 64 * KA < Ub;  // the high waves own live columns
    float *bA0 = lds, *bA1 = bA0 + fA, *bB0 = bA1 + fB, *bB1 = bB0 + fA;  // alpha low, alpha high, beta low, beta high
    float2 *ringA = (float2 *)(bB1 + fB), *ringB = ringA + p.Nr;
    float2 *trash = ringB + p.Nr;  // two dumps of 64 + G entries
    int *prog = (int *)(trash + 2 * (64 + G));
    if (tid < 2) prog[tid] = 0;
    if (!peer)
        for (int i = tid; i < 2 * p.Nr; i += 256) ringA[i] = make_float2(kNeg, 0.f);
    __syncthreads();
    SplitLink lk;
    lk.peer = peer;
    if (wave < 2) {
        lk.ring = ringA, lk.trash = trash, lk.prog = (uint32_t)(uintptr_t)((lds_void *)prog);
        if (wave == 0)
            alpha_split_sweep<KA, G, UPT, true>(p, bA0, bA0 + fA / 2, lk, b, lane, 0);
        else if (peer)
            alpha_split_sweep<KB, G, UPT, false>(p, bA1, bA1 + fB / 2, lk, b, lane, 64 * KA);
    } else {
        lk.ring = ringB, lk.trash = trash + 64 + G, lk.prog = (uint32_t)(uintptr_t)((lds_void *)(prog + 1));
        if (wave == 2)
            beta_split_sweep<KA, G, UPT, false>(p, bB0, bB0 + fA / 2, lk, b, lane, 0);
        else if (peer)
            beta_split_sweep<KB, G, UPT, true>(p, bB1, bB1 + fB / 2, lk, b, lane, 64 * KA);
    }
}

template <int KA, int KB, int G>
static hipError_t launch_sweep_split(const LossParams &p, hipStream_t s, bool *done) {
    const size_t shm = split_lds_floats<KA, KB, G>() * sizeof(float) + (size_t)2 * p.Nr * sizeof(float2) +
                       (size_t)2 * (64 + G) * sizeof(float2) + 16;
    *done = false;
    if (shm > 160 * 1024) return hipSuccess;  // boundary arrays do not fit: the caller falls back to the single wave
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sweep_split_kernel<KA, KB, G>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
    }
#ifdef SPLIT_TRACE
    static long long *trace_dev = nullptr;
    if (!trace_dev) {
        hipMalloc(&trace_dev, 4 * 256 * sizeof(long long));
        hipMemcpyToSymbol(HIP_SYMBOL(g_split_trace), &trace_dev, sizeof(trace_dev));
    }
    hipMemsetAsync(trace_dev, 0, 4 * 256 * sizeof(long long), s);
#endif
    hipLaunchKernelGGL((sweep_split_kernel<KA, KB, G>), dim3(p.nb), dim3(256), shm, s, p);
#ifdef SPLIT_TRACE
    {
        hipStreamSynchronize(s);
        long long h[4 * 256];
        hipMemcpy(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost);
        const char *path = getenv("SPLIT_TRACE_FILE");
        if (FILE *f = fopen(path ? path : "/tmp/split_trace.bin", "wb")) {
            fwrite(h, 1, sizeof(h), f);
            fclose(f);
        }
    }
#endif
    *done = true;
    return hipGetLastError();
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
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
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
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
    const int Nb = Tb + Ub - 1;
    const RidgeLine ridge = make_ridge(Ub, Nb);
    float *out = p.A + (size_t)b * p.Nr * Up;
    const int voff = lane * K * 4;
    const int u0 = lane * K;

    float a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (u0 + j == 0) ? 0.f : kNeg;
    store_diag<K, false>(out, voff, lane, a);
    SweepState st;
    st.off = 0.f;
    st.edge = kNeg;
    st.log.init(p.offA + (size_t)b * p.NC * p.NG, p.NG);
    st.log.record(0, 0.f, lane);
    st.row = out + Up;
    const int last_row = Nb - 1;
    const int nchunks = last_row / G + 1;

    int have = 0;  // chunks known to have landed (the loader runs up to NB - 1 ahead: most chunks need no look at the counter)
    bool timed_out = false;  // a bounded poll gave up: the result must not look valid
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
            alpha_fast_steps<K, G, 0>(a, wq, abase, st, voff, lane, r0, ridge);
        } else {
            for (int i = 0; i < G; ++i) {
                const int n = r0 + i + 1;
                if (n > last_row) break;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Up);
                alpha_step<K>(a, wc);
                if ((n & (kRebase - 1)) == 0) {
                    st.off += rebase<K>(a, ridge.u_at(n));
                    st.log.record(n / kRebase, st.off, lane);
                }
                store_diag<K, false>(st.row, voff, lane, a);
                st.row += Up;
            }
        }
        if (ck + 1 < nchunks) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of this chunk's buffer has returned
            if (lane == 0) lds_post(lk.consumed, ck + 1);
        }
    }
    st.log.flush(lane);
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
    const int Tb = p.input_lengths[b], Ub = p.label_lengths[b] + 1;
    const int Nb = Tb + Ub - 1;
    const RidgeLine ridge = make_ridge(Ub, Nb);
    float *out = p.Bt + (size_t)b * p.Nr * Up;
    const int voff = lane * K * 4;
    const int u0 = lane * K;

    float bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = (u0 + j == Ub - 1) ? 0.f : kNeg;
    const int last = Nb - 1;
    const int ckl = last / G;
    SweepState st;
    st.off = 0.f;
    st.edge = kNeg;
    st.log.init(p.offB + (size_t)b * p.NC * p.NG, p.NG);
    st.row = out + (size_t)last * Up;

    int have = 0;
    bool timed_out = false;
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
            beta_fast_steps<K, G, 0>(bv, wq, abase, st, voff, lane, r0, ridge);
        } else {
            for (int ii = 0; ii < G; ++ii) {
                const int i = G - 1 - ii;
                const int n = r0 + i;
                if (n > last) continue;
                f32x2 wc[K];
                load_w<K>(wc, cur + i * 2 * Up);
                beta_step<K>(bv, wc);
                if (((n & (kRebase - 1)) == kRebase - 1) || n == last) {
                    st.off += rebase<K>(bv, ridge.u_at(n));
                    st.log.record(n / kRebase, st.off, lane);
                }
                store_diag<K, false>(st.row, voff, lane, bv);
                st.row -= Up;
            }
        }
        if (ck > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_post(lk.consumed, i_ring + 1);
        }
    }
    st.log.flush(lane);
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

bool overlap_path_ok(const LossParams &p, bool grad) {
    // patch kernels on both sides and the single-wave sweep (the hand-off hooks live there)
    return (p.V % 4) == 0 && tile_path_ok(p, false) && (!grad || tile_path_ok(p, true)) && sweep_mode() == 1 &&
           sweep_K(p.U) != 0;
}

hipError_t launch_sweeps(const LossParams &p0, hipStream_t s, bool overlap) {
    LossParams p = p0;
    if (!overlap) p.flags = nullptr;  // the sweep kernels key the hand-off protocol on this pointer
    if (sweep_mode() == 4 && !overlap) {
        bool done = false;
        hipError_t e = hipSuccess;
        switch (sweep_K(p.U)) {  // two waves per direction (see sweep_split_kernel); G as large as the LDS allows
            case 2: e = launch_sweep_split<1, 1, 16>(p, s, &done); break;
            case 3: e = launch_sweep_split<2, 1, 16>(p, s, &done); break;
            case 4: e = launch_sweep_split<2, 2, 8>(p, s, &done); break;
            case 6: e = launch_sweep_split<3, 3, 8>(p, s, &done); break;
            case 8: e = launch_sweep_split<4, 4, 4>(p, s, &done); break;
            default: break;
        }
        if (e != hipSuccess || done) return e;
    }
    if (sweep_mode() == 1 && !overlap) {
        switch (sweep_K(p.U)) {  // sweeping wave + loader wave (see sweep_ld_kernel); same chunk lengths as below
            case 1: return launch_sweep_ld<1, 16>(p, s);
            case 2: return launch_sweep_ld<2, 16>(p, s);
#ifndef SWEEP_LD_G3
#define SWEEP_LD_G3 16
#endif
            case 3: return launch_sweep_ld<3, SWEEP_LD_G3>(p, s);
            case 4: return launch_sweep_ld<4, 16>(p, s);
            case 6: return launch_sweep_ld<6, 8>(p, s);
            case 8: return launch_sweep_ld<8, 8>(p, s);
            case 12: return launch_sweep_ld<12, 4>(p, s);
            case 16: return launch_sweep_ld<16, 4>(p, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (sweep_K(p.U)) {  // RNNT_SWEEP_MODE=5 (the sweeping wave issues its own LDS-DMA), 0, and the overlap mode
        case 1: return launch_sweep_kg<1, 16>(p, s);
        case 2: return launch_sweep_kg<2, 16>(p, s);
        // chunk length G (diagonals per LDS-DMA batch / wait): the longest whose two buffers fit 64 KB (measured at C2:
        // G = 16 beats 8 by 3 % of the step, 4 loses 5 %)
        case 3: return launch_sweep_kg<3, 16>(p, s);
        case 4: return launch_sweep_kg<4, 16>(p, s);
        case 6: return launch_sweep_kg<6, 8>(p, s);
        case 8: return launch_sweep_kg<8, 8>(p, s);
        case 12: return launch_sweep_kg<12, 4>(p, s);
        case 16: return launch_sweep_kg<16, 4>(p, s);
        default: return hipErrorInvalidValue;  // maxU > 1024 is outside the register-resident sweep
    }
}

}  // namespace rnnt
