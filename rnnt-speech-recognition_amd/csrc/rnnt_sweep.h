// rnnt_sweep.h -- device code of the register-resident alpha / beta sweeps (one wave64 per utterance and direction; the
// live anti-diagonal in VGPRs, one DPP wave-shift per step, edge weights streamed HBM -> LDS by a loader wave).
// Shared by rnnt_kernels.hip (the log-domain sweeps: fused joints, large vocabularies, and every lattice the linear-domain
// path hands back) and rnnt_lin_kernels.hip (the linear-domain sweeps of the small-vocabulary loss, and its log-domain redo).
// Replaces warp-transducer's compute_alphas_kernel / compute_betas_kernel (call site
// utils/loss.py:34-35).
#pragma once
#include "rnnt_common.h"
#include "rnnt_cell.h"

#include <float.h>
#include <math.h>

namespace rnnt {
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
__device__ __forceinline__ void dma_rows(const float *g, float *l, int n16, int lane, int first = 0, int step = 1) {
    for (int i0 = 64 * first; i0 < n16; i0 += 64 * step) {  // pieces first, first + step, ... (several loader waves share a chunk)
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

template <int K, int G, int NB, bool BETA, int NL = 1>
__device__ void sweep_loader(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane, const int w = 0) {
    // NL loader waves share EVERY chunk: loader w issues the LDS-DMA pieces w, w + NL, ... of it and counts the chunks whose
    // pieces of its own have landed in its own `landed` word (lk.landed of the LdLink it is handed); a chunk is complete when
    // every loader has counted it.  An LDS-DMA piece costs the issuing wave 60-100 clocks: at 24 pieces per 16 diagonals one
    // loader keeps pace with either sweeping wave (two that split every chunk: 41.9 against 40.6 us -- the sweeper is not waiting
    // for data; a loader that also zeroes the positions no lattice cell owns, instead of fill workgroups in the lsm launch:
    // 103 us -- measured in round 4 and removed).
    constexpr int Up = 64 * K, chunkf = G * 2 * Up, n16 = chunkf / 4, pieces = n16 / 64;
    static_assert(n16 % 64 == 0 && pieces <= 63 && pieces % NL == 0, "chunk must be whole wave-instructions within the vmcnt range");
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int nchunks = (Tb + Ub - 2) / G + 1;
    const float *Wb = p.W + (size_t)b * p.Nr * 2 * Up;
    for (int i = 0; i < nchunks; ++i) {
        const int ck = BETA ? nchunks - 1 - i : i;
        if (i >= NB) lds_wait_ge(lk.consumed, i - NB + 1);  // ring slot i % NB is free again
        dma_rows(Wb + (size_t)ck * chunkf, bufs + (i % NB) * chunkf, n16, lane, w, NL);
        if (i > 0) {
            wait_vm_counted<pieces / NL>();  // loads return in order: everything but the pieces just issued has landed
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
                if (p.costs) st_f32_wt(p.costs + b, (float)(-ll2 * 0.6931471805599453));
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
    if (timed_out && lane == 0 && p.costs) st_f32_wt(p.costs + b, NAN);  // the alpha side may have finished normally
}

// ---------------------------------------------------------------------------------------------
// PRECISE sweeps: the same recurrence carried in float64, for the inputs the float32 one is not good enough for.
// The float32 sweeps round every log-add at the magnitude of its residue (up to hundreds of bits below the lane's reference
// at 4 ... 8 x N(0,1) logits): ~1e-5 bits per step, a random walk of ~3e-4 bits over the ~1,000 steps of a path through a wide
// lattice -- that, not the re-basing granularity, is where 2 ... 5e-4 of gradient error on such inputs came from
// (tests/tools/emulate_sweep.py: per-column offsets and per-diagonal re-basing do not remove it, a float64 recurrence on the
// same float32 edge weights does: 5e-6).  Here: alpha / beta are TRUE log2 values in float64 registers (no frames in the
// recurrence, hence no offset differences at lane crossings either), the transcendental part of a log-add -- log2(1 + 2^-|d|),
// in (0, 1] -- stays on the float32 units (absolute error ~1e-7, not accumulated at the residue's magnitude), and the outputs
// keep their format: float32 residues against the lane's integer offset of the block (one rounding at the store, not carried
// forward), offsets in the tables, ll in float64.  ~3x the instructions of the float32 step; used by the linear lattice's
// hand-back kernel (utterances whose range certificate failed: peaked posteriors) and for lattices of 8 and more columns per lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp64_from_lower_lane(const double x, const double fill) {
    const long long xi = __double_as_longlong(x), fi = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)fi, (int)xi, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fi >> 32), (int)(xi >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double dpp64_from_upper_lane(const double x, const double fill) {
    const long long xi = __double_as_longlong(x), fi = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)fi, (int)xi, 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fi >> 32), (int)(xi >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lse2_pr(const double u, const double l) {
    const float d = (float)(u - l);  // (0 for two log zeros: the result stays a log zero)
    return fmax(u, l) + (double)lg2(1.0f + ex2(-fabsf(d)));
}
// K independent log-adds, stage by stage (a lone wave stalls on every instruction that consumes the result of the one just
// before it: every consumer is K instructions behind its producer, as in lse2_staged)
template <int K>
__device__ __forceinline__ void lse2_pr_staged(double (&out)[K], const double (&u)[K], const double (&l)[K]) {
    float d[K];
    double m[K];
#pragma unroll
    for (int j = 0; j < K; ++j) d[j] = (float)(u[j] - l[j]);
    SWEEP_FENCE();
#pragma unroll
    for (int j = 0; j < K; ++j) d[j] = ex2(-fabsf(d[j]));
    SWEEP_FENCE();
#pragma unroll
    for (int j = 0; j < K; ++j) m[j] = fmax(u[j], l[j]);
    SWEEP_FENCE();
#pragma unroll
    for (int j = 0; j < K; ++j) d[j] = lg2(1.0f + d[j]);
    SWEEP_FENCE();
#pragma unroll
    for (int j = 0; j < K; ++j) out[j] = m[j] + (double)d[j];
}
// The lane's integer offset of a block = its largest value, rounded; a lane without mass copies the lane the mass will come
// from (as rebase_lane does).  Values are NOT re-based in registers.
template <int K, bool BETA>
__device__ __forceinline__ void offset_lane_pr(const double (&v)[K], SweepState &st, const int kc) {
    double m = v[0];
#pragma unroll
    for (int j = 1; j < K; ++j) m = fmax(m, v[j]);
    const bool fin = m > (double)kNegTest;
    float off = fin ? rintf((float)m) : st.off;
    constexpr int R = (kRebase + K - 1) / K;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float nb = BETA ? dpp_from_upper_lane(off, off) : dpp_from_lower_lane(off, off);
        off = fin ? off : nb;
    }
    st.off = off;
    st_f32_wt(st.tab + (size_t)kc * 64, off);
}
template <int K>
__device__ __forceinline__ void store_diag_pr(float *row, const int lane, const double (&v)[K], const float off) {
    float *dst = row + lane * K;
#pragma unroll
    for (int j = 0; j < K; ++j) st_f32_wt(dst + j, (float)(v[j] - (double)off));  // (a log zero, <= -1e30, stays one: readers test against kNegTest)
}

template <int K, int G, int NB>
__device__ void alpha_sweep_pr(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1;
    float *out = p.A + (size_t)b * p.Nr * Up;
    const int u0 = lane * K;
    double a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (u0 + j == 0) ? 0.0 : (double)kNeg;
    SweepState st;
    st.off = 0.f, st.dlt = 0.f, st.edge = kNeg;
    st.tab = p.offA + (size_t)b * p.NC * p.NG + lane;
    st_f32_wt(st.tab, 0.f);  // block 0
    store_diag_pr<K>(out, lane, a, 0.f);
    st.row = out + Up;
    const int last_row = Nb - 1;
    const int nchunks = last_row / G + 1;
    int have = 0;
    bool timed_out = lengths_invalid(p, b);
    for (int ck = 0; ck < nchunks; ++ck) {
        if (have < ck + 1) {
            have = lds_wait_ge(lk.landed, ck + 1);
            timed_out |= have < ck + 1;
        }
        const float *cur = bufs + (ck % NB) * chunkf + 2 * u0;
        const int r0 = ck * G;
#pragma unroll 4
        for (int i = 0; i < G; ++i) {
            const int n = r0 + i + 1;
            if (n > last_row) break;
            f32x2 w[K];
            load_w<K>(w, cur + i * 2 * Up);
            double emit[K], stay[K], left[K];
#pragma unroll
            for (int j = K - 1; j >= 0; --j) emit[j] = a[j] + (double)w[j][1];  // (t, u) -> (t, u + 1); last column first: the DPP moves wait for it
            SWEEP_FENCE();
            const double from_left = dpp64_from_lower_lane(emit[K - 1], (double)kNeg);
#pragma unroll
            for (int j = 0; j < K; ++j) stay[j] = a[j] + (double)w[j][0], left[j] = (j == 0) ? from_left : emit[j - 1];
            SWEEP_FENCE();
            lse2_pr_staged<K>(a, stay, left);
            if ((n & (kRebase - 1)) == 0) offset_lane_pr<K, false>(a, st, n / kRebase);
            store_diag_pr<K>(st.row, lane, a, st.off);
            st.row += Up;
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
                const double ll2 = timed_out ? (double)NAN : a[j] + (double)wrow[2 * j];
                st_f64_wt(p.ll + 2 * b, ll2);
                if (p.costs) st_f32_wt(p.costs + b, (float)(-ll2 * 0.6931471805599453));
            }
    }
}

template <int K, int G, int NB>
__device__ void beta_sweep_pr(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1;
    float *out = p.Bt + (size_t)b * p.Nr * Up;
    const int u0 = lane * K;
    double bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = (u0 + j == Ub - 1) ? 0.0 : (double)kNeg;
    const int last = Nb - 1;
    const int ckl = last / G;
    SweepState st;
    st.off = 0.f, st.dlt = 0.f, st.edge = kNeg;
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
#pragma unroll 4
        for (int ii = 0; ii < G; ++ii) {
            const int i = G - 1 - ii;
            const int n = r0 + i;
            if (n > last) continue;
            f32x2 w[K];
            load_w<K>(w, cur + i * 2 * Up);
            const double from_right = dpp64_from_upper_lane(bv[0], (double)kNeg);
            double stay[K], emit[K];
#pragma unroll
            for (int j = 0; j < K; ++j)
                stay[j] = bv[j] + (double)w[j][0], emit[j] = ((j == K - 1) ? from_right : bv[j + 1]) + (double)w[j][1];
            SWEEP_FENCE();
            lse2_pr_staged<K>(bv, stay, emit);
            if (((n & (kRebase - 1)) == kRebase - 1) || n == last) offset_lane_pr<K, true>(bv, st, n / kRebase);
            store_diag_pr<K>(st.row, lane, bv, st.off);
            st.row -= Up;
        }
        if (ck > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_post(lk.consumed, i_ring + 1);
        }
    }
    if (lane == 0) st_f64_wt(p.ll + 2 * b + 1, timed_out ? (double)NAN : bv[0]);
    if (timed_out && lane == 0 && p.costs) st_f32_wt(p.costs + b, NAN);  // the alpha side may have finished normally
}

}  // namespace rnnt
