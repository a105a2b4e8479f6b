// rnnt_lin_kernels.hip -- LINEAR-domain alpha / beta sweeps of the small-vocabulary transducer loss, and the log-domain redo
// of the utterances that path hands back (rnnt_lin.h has the scheme and the per-cell code of the two cell passes).
//
// Replaces warp-transducer's compute_alphas_kernel / compute_betas_kernel (reached from
// utils/loss.py:34-35).  Same structure as sweep_ld_kernel (rnnt_sweep.h): one workgroup of two waves per (utterance,
// direction) -- a sweeping wave with the live anti-diagonal in VGPRs (K lattice columns per lane, ONE DPP wave-shift per
// diagonal, no barrier, no masks: probability zero is carried by the data) and a loader wave that streams the edge
// probabilities HBM -> LDS through a ring of chunks.  What changed is the arithmetic on the serial chain: a step is
//     alpha(t,u) = alpha(t-1,u) p_blank(t-1,u) + alpha(t,u-1) p_label(t,u-1)
// in float32 -- multiplies and adds, no exp2 / log2 -- on mantissas relative to one integer FRAME per lane; every kLinR = 4
// diagonals a lane renormalises against its own maximum by an exact power of two (v_frexp_exp / v_ldexp), and frames are
// dragged up so that whatever can arrive from the neighbouring lanes within a block fits.  What crosses a lane boundary is
// rescaled by the frame difference (one v_ldexp on the value handed over).  Frames go to [block][64 lanes] tables.
#include "rnnt_common.h"
#include "rnnt_cell.h"
#include "rnnt_sweep.h"
#include "rnnt_cellwave.h"
#include "rnnt_lin.h"
#include "rnnt_cellbody.h"
#include "rnnt_redo.h"

namespace rnnt {

__device__ __forceinline__ float dpp_lower_zero(float x) {  // lane i <- lane i-1, lane 0 <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_upper_zero(float x) {  // lane i <- lane i+1, lane 63 <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ int dpp_lower_i(int x, int fill) {
    return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_upper_i(int x, int fill) {
    return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false);
}

struct LinState {
    int E;       // this lane's frame: true value = mantissa x 2^E
    int d;       // alpha: E - E[lane + 1] (applied by the SENDER to what it hands to lane + 1);  beta: E[lane + 1] - E (applied by
                 // the RECEIVER to what arrives from lane + 1) -- either way the product is formed in the receiver's frame
    int *tab;    // this utterance's frame table [NCl][64], advanced by `lane`
    float *row;  // wave-uniform base of the output row of the next diagonal to be stored
};

// Renormalise a lane's K mantissas against their maximum, choose the frames of the next block, record them.
template <int K, bool BETA, int SH>
__device__ __forceinline__ void lin_renorm(float (&v)[K], LinState &st, const int kc) {
    float m = v[0];
#pragma unroll
    for (int j = 1; j < K; ++j) m = fmaxf(m, v[j]);
    const int own = (m > 0.f) ? st.E + frexp_e(m) : kFrameNone;  // where this lane's mass is (no mass: no claim)
    // Within one block mass travels at most 2^SH columns = LOOK lanes: the frame must leave room for what those lanes hold.
    constexpr int LOOK = ((1 << SH) + K - 1) / K;
    int nb = own, reach = kFrameNone;
#pragma unroll
    for (int r = 0; r < LOOK; ++r) {
        nb = BETA ? dpp_upper_i(nb, kFrameNone) : dpp_lower_i(nb, kFrameNone);
        reach = max(reach, nb);
    }
    const int En = max(own, reach - kLinDrag);
    const int sh = st.E - En;  // <= 0 for a lane with mass; a lane without mass holds zeros whatever its frame
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] = ldexp_f(v[j], sh);
    st.E = En;
    const int up = dpp_upper_i(En, En);  // lane 63 sees itself (nothing crosses)
    st.d = BETA ? up - En : En - up;
    st_i32_wt(st.tab + (size_t)kc * 64, En);
}

// The edge probabilities of one diagonal for this lane: K pairs {blank, label} = 8 K contiguous bytes of LDS at a lane stride of
// 8 K bytes.  Even K: 16-byte reads (a lone wave pays a full issue slot per instruction whatever its width).  Odd K: 8-byte
// reads -- at K = 3 the 16-byte form would be only 8-byte aligned (lane stride 24); gfx950 serves such reads (unaligned DS
// access; scripts/probes/probe_lat.hip) but slower: 44.3 against 40.6 us for the sweeps at B32 T600 U150.
template <int K>
struct LinRow {
    static constexpr bool WIDE = (K % 2) == 0;
    static constexpr int NR = WIDE ? K / 2 : K;  // LDS instructions per row
    f32x4 q4[WIDE ? K / 2 : 1];
    f32x2 q2[WIDE ? 1 : K];
    template <int J>
    __device__ __forceinline__ float b() const {  // p(blank) of column J
        if constexpr (WIDE) return q4[J / 2][2 * (J & 1)];
        else return q2[J][0];
    }
    template <int J>
    __device__ __forceinline__ float l() const {  // p(label) of column J
        if constexpr (WIDE) return q4[J / 2][2 * (J & 1) + 1];
        else return q2[J][1];
    }
};
template <int K, int J = 0>
__device__ __forceinline__ void lin_row_unpack(const LinRow<K> &w, float (&wb)[K], float (&wl)[K]) {
    if constexpr (J < K) {
        wb[J] = w.template b<J>(), wl[J] = w.template l<J>();
        lin_row_unpack<K, J + 1>(w, wb, wl);
    }
}
// Issue the LDS reads of row ROW of the chunk buffer; completion is NOT tracked by the compiler: the caller waits with
// lds_wait<N>() (LDS returns in order) and nothing may touch the row's registers before that.
template <int K, int ROW>
__device__ __forceinline__ void lin_issue_row(LinRow<K> &w, const uint32_t addr) {
    if constexpr (LinRow<K>::WIDE) {
#pragma unroll
        for (int i = 0; i < K / 2; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w.q4[i]) : "v"(addr), "n"(ROW * 2 * 64 * K * 4 + i * 16));
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w.q2[j]) : "v"(addr), "n"(ROW * 2 * 64 * K * 4 + j * 8));
    }
}
// (the tail of a sweep reads its rows with ordinary, compiler-tracked loads)
template <int K>
__device__ __forceinline__ void lin_load_row(LinRow<K> &w, const float *wrow) {
    if constexpr (LinRow<K>::WIDE) {
#pragma unroll
        for (int i = 0; i < K / 2; ++i) {
            const f32x2 a = ((const f32x2 *)wrow)[2 * i], c = ((const f32x2 *)wrow)[2 * i + 1];
            w.q4[i] = (f32x4){a[0], a[1], c[0], c[1]};
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) w.q2[j] = ((const f32x2 *)wrow)[j];
    }
}

// One step of the recurrence.  A lone wave issues ONE instruction of any kind (VALU, LDS, VMEM, s_nop) per ~5 clocks, ~8.5 when
// a VALU instruction consumes the result of the one just before it (scripts/probes/probe_lat.hip; profiles/r04_notes.md): the
// time of a step is its instruction count.  Two VALU instructions per lattice column plus the hand-over (v_ldexp into the
// receiver's frame, one DPP wave shift), in an order in which no instruction reads what its predecessor wrote and the DPP move
// finds its source two instructions old.  Plain scalar C++ with scheduling fences: this file is compiled with
// -fno-slp-vectorize (build.py), because the packed-f32 forms the vectoriser picks cost more in register moves than they save,
// and inline-asm statements get padded with s_nop 0 by the compiler (a full issue slot each).
#define LIN_FENCE() __builtin_amdgcn_sched_barrier(0)
// alpha: diagonal r -> r+1 with the outgoing edge probabilities w[j] = {blank, label} of diagonal r:
//   a[j] <- a[j] p_blank[j] + a[j-1] p_label[j-1];  what leaves column K-1 is scaled into lane + 1's frame BEFORE the product
template <int K>
__device__ __forceinline__ void lin_alpha_step(float (&a)[K], const LinRow<K> &w, const int d) {
    // ONE asm block: the compiler puts the hand-over product right in front of the DPP move (s_nop 1).  Inputs and outputs are
    // separate registers, so that the new diagonal can land in the register triple the store instruction wants.
    if constexpr (K > 4) {  // wide lattices: left to the compiler (their step is long enough to hide its own latencies)
        float pr[K], wb[K], wl[K];
        lin_row_unpack<K>(w, wb, wl);
        const float hand = ldexp_f(a[K - 1], d) * wl[K - 1];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) pr[j] = a[j] * wl[j];
        const float left = dpp_lower_zero(hand);
#pragma unroll
        for (int j = K - 1; j >= 1; --j) a[j] = fmaf(a[j], wb[j], pr[j - 1]);
        a[0] = fmaf(a[0], wb[0], left);
    } else {
    float n[K], as, q, pr[K > 1 ? K - 1 : 1];
    if constexpr (K == 1) {
        asm volatile("v_ldexp_f32 %[as], %[a0], %[d]\n\tv_mul_f32 %[as], %[as], %[l0]\n\ts_nop 1\n\tv_mov_b32_dpp %[q], %[as] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f32 %[n0], %[a0], %[b0], %[q]"
                     : [n0] "=&v"(n[0]), [as] "=&v"(as), [q] "=&v"(q)
                     : [a0] "v"(a[0]), [b0] "v"(w.template b<0>()), [l0] "v"(w.template l<0>()), [d] "v"(d));
    } else if constexpr (K == 2) {
        asm volatile("v_ldexp_f32 %[as], %[a1], %[d]\n\tv_mul_f32 %[p0], %[a0], %[l0]\n\tv_mul_f32 %[as], %[as], %[l1]\n\tv_fma_f32 %[n1], %[a1], %[b1], %[p0]\n\ts_nop 0\n\tv_mov_b32_dpp %[q], %[as] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f32 %[n0], %[a0], %[b0], %[q]"
                     : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [as] "=&v"(as), [q] "=&v"(q), [p0] "=&v"(pr[0])
                     : [a0] "v"(a[0]), [a1] "v"(a[1]), [b0] "v"(w.template b<0>()), [b1] "v"(w.template b<1>()), [l0] "v"(w.template l<0>()), [l1] "v"(w.template l<1>()), [d] "v"(d));
    } else if constexpr (K == 3) {
        asm volatile("v_ldexp_f32 %[as], %[a2], %[d]\n\tv_mul_f32 %[p0], %[a0], %[l0]\n\tv_mul_f32 %[p1], %[a1], %[l1]\n\tv_mul_f32 %[as], %[as], %[l2]\n\tv_fma_f32 %[n2], %[a2], %[b2], %[p1]\n\tv_fma_f32 %[n1], %[a1], %[b1], %[p0]\n\tv_mov_b32_dpp %[q], %[as] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f32 %[n0], %[a0], %[b0], %[q]"
                     : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [as] "=&v"(as), [q] "=&v"(q), [p0] "=&v"(pr[0]), [p1] "=&v"(pr[1])
                     : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [b0] "v"(w.template b<0>()), [b1] "v"(w.template b<1>()), [b2] "v"(w.template b<2>()), [l0] "v"(w.template l<0>()), [l1] "v"(w.template l<1>()), [l2] "v"(w.template l<2>()), [d] "v"(d));
    } else {
        asm volatile("v_ldexp_f32 %[as], %[a3], %[d]\n\tv_mul_f32 %[p0], %[a0], %[l0]\n\tv_mul_f32 %[p1], %[a1], %[l1]\n\tv_mul_f32 %[p2], %[a2], %[l2]\n\tv_mul_f32 %[as], %[as], %[l3]\n\tv_fma_f32 %[n3], %[a3], %[b3], %[p2]\n\tv_fma_f32 %[n2], %[a2], %[b2], %[p1]\n\tv_fma_f32 %[n1], %[a1], %[b1], %[p0]\n\tv_mov_b32_dpp %[q], %[as] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f32 %[n0], %[a0], %[b0], %[q]"
                     : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [as] "=&v"(as), [q] "=&v"(q), [p0] "=&v"(pr[0]), [p1] "=&v"(pr[1]), [p2] "=&v"(pr[2])
                     : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(w.template b<0>()), [b1] "v"(w.template b<1>()), [b2] "v"(w.template b<2>()), [b3] "v"(w.template b<3>()), [l0] "v"(w.template l<0>()), [l1] "v"(w.template l<1>()), [l2] "v"(w.template l<2>()), [l3] "v"(w.template l<3>()), [d] "v"(d));
    }
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = n[j];
    }
}
// beta: diagonal n+1 -> n with the outgoing edge probabilities of diagonal n:
//   b[j] <- b[j] p_blank[j] + b[j+1] p_label[j];  what arrives from lane + 1 is scaled into this lane's frame BEFORE the product
template <int K>
__device__ __forceinline__ void lin_beta_step(float (&bv)[K], const LinRow<K> &w, const int d) {
    float st[K], wb[K], wl[K];
    lin_row_unpack<K>(w, wb, wl);  // (register names only: no instruction)
    float right = dpp_upper_zero(bv[0]);
    LIN_FENCE();
#pragma unroll
    for (int j = 0; j < K; ++j) st[j] = bv[j] * wb[j];
    LIN_FENCE();
    right = ldexp_f(right, d);
    LIN_FENCE();
#pragma unroll
    for (int j = 0; j < K - 1; ++j) bv[j] = fmaf(bv[j + 1], wl[j], st[j]);
    LIN_FENCE();
    bv[K - 1] = fmaf(right, wl[K - 1], st[K - 1]);
    LIN_FENCE();
}

// Fully unrolled steps of one chunk.  Edge probabilities are read from LDS TWO rows ahead (three register sets): the reads of
// row II + 2 are issued behind the arithmetic of step II (LDS latency ~64 clocks against a step of ~65), and they separate the
// step's asm block from the diagonal's store (the compiler would otherwise pad with an s_nop).
template <int K, int G, int SH, int II>
__device__ __forceinline__ void lin_alpha_fast_steps(float (&a)[K], LinRow<K> (&wq)[3], const uint32_t abase, LinState &st,
                                                     const int voff, const int lane, const int r0) {
    if constexpr (II < G) {
        constexpr int R = 1 << SH;
        if constexpr (II + 1 < G)
            lds_wait<LinRow<K>::NR>();  // row II has landed, row II+1 stays in flight
        else
            lds_wait<0>();
        lin_alpha_step<K>(a, wq[II % 3], st.d);
        if constexpr (II + 2 < G) lin_issue_row<K, II + 2>(wq[(II + 2) % 3], abase);
        if constexpr (((II + 1) % R) == 0) lin_renorm<K, false, SH>(a, st, (r0 + II + 1) >> SH);  // (r0 is a multiple of G)
        constexpr int RB = rows_per_base(K);
        store_diag<K, true, (II % RB) * 64 * K * 4>(st.row, voff, lane, a);
        if constexpr (II % RB == RB - 1 || II == G - 1) st.row += (II % RB + 1) * 64 * K;
        lin_alpha_fast_steps<K, G, SH, II + 1>(a, wq, abase, st, voff, lane, r0);
    }
}
template <int K, int G, int SH, int II>
__device__ __forceinline__ void lin_beta_fast_steps(float (&bv)[K], LinRow<K> (&wq)[3], const uint32_t abase, LinState &st,
                                                    const int voff, const int lane, const int r0) {
    if constexpr (II < G) {
        constexpr int R = 1 << SH;
        constexpr int i = G - 1 - II;  // row inside the chunk (descending)
        if constexpr (i > 0)
            lds_wait<LinRow<K>::NR>();
        else
            lds_wait<0>();
        lin_beta_step<K>(bv, wq[II % 3], st.d);
        if constexpr (i >= 2) lin_issue_row<K, (i >= 2 ? i - 2 : 0)>(wq[(II + 2) % 3], abase);
        if constexpr ((i % R) == R - 1) lin_renorm<K, true, SH>(bv, st, (r0 + i) >> SH);
        constexpr int RB = rows_per_base(K);
        store_diag<K, true, -(II % RB) * 64 * K * 4>(st.row, voff, lane, bv);
        if constexpr (II % RB == RB - 1 || II == G - 1) st.row -= (II % RB + 1) * 64 * K;
        lin_beta_fast_steps<K, G, SH, II + 1>(bv, wq, abase, st, voff, lane, r0);
    }
}

// The likelihood one side arrived at: {mantissa in [0.5, 1), frame}, its log2 in float64, the hand-back flag.
__device__ __forceinline__ void lin_record(const LossParams &p, const int b, const int side, const float L, const int E, bool bad) {
    bad = bad || !(L > 0.f) || !(L <= FLT_MAX);  // zero (everything flushed), NaN (an edge the lsm pass refused), inf
    const float mL = frexp_m(L);
    const int EL = E + frexp_e(L);
    st_f32_wt(p.lik + 4 * b + 2 * side, bad ? NAN : mL);
    st_i32_wt((int *)p.lik + 4 * b + 2 * side + 1, EL);
    const double ll2 = bad ? (double)NAN : log2((double)mL) + (double)EL;
    st_f64_wt(p.ll + 2 * b + side, ll2);
    st_i32_wt(p.flags + 4 * b + (side ? kFlagB : kFlagA), bad ? 1 : 0);
    if (side == 0 && p.costs) st_f32_wt(p.costs + b, (float)(-ll2 * 0.6931471805599453));
}

constexpr int kLinLoaders = 1;  // loader waves per sweep workgroup (rnnt_sweep.h sweep_loader); measured: two loaders that split every chunk give the same sweep time (41.9 vs 40.6 us at B32 T600 U150) -- the sweeping wave is not waiting for data

// Chunk i of the loading order has landed when EVERY loader has counted i + 1 chunks (each brings its share of the pieces).
struct LandedView {
    uint32_t addr0;  // LDS byte address of loader 0's counter; loader w's is 8 w bytes further
    int have[kLinLoaders];
    __device__ __forceinline__ bool wait(const int i) {  // false: a bounded poll gave up
        bool ok = true;
#pragma unroll
        for (int w = 0; w < kLinLoaders; ++w) {
            if (have[w] < i + 1) have[w] = lds_wait_ge(addr0 + 8u * (uint32_t)w, i + 1);
            ok &= have[w] >= i + 1;
        }
        return ok;
    }
};

template <int K, int G, int NB, int SH>
__device__ void lin_alpha_sweep(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up;
    static_assert(G % (1 << SH) == 0, "frame blocks must not straddle chunks");
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1;
    float *out = p.A + (size_t)b * p.Nr * Up;
    const int voff = lane * K * 4;
    const int u0 = lane * K;
    if (lane == 0) {  // this forward call owns the utterance's hand-back state from here on
        st_i32_wt(p.flags + 4 * b + kFlagG, 0);
        st_i32_wt(p.flags + 4 * b + kFlagState, 0);
        st_i32_wt(p.lshift + b, SH);  // the block length this utterance's frame tables are indexed with
        // the ticket / completion counters of the utterance's hand-back team (rnnt_redo.h; read by a LATER launch: plain stores)
        ((int4 *)p.bar)[2 * b] = make_int4(0, 0, 0, 0), ((int4 *)p.bar)[2 * b + 1] = make_int4(0, 0, 0, 0);
    }
    float a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (u0 + j == 0) ? 1.f : 0.f;
    LinState st;
    st.E = 0, st.d = 0;
    st.tab = p.EA + (size_t)b * p.NCl * 64 + lane;
    lin_renorm<K, false, SH>(a, st, 0);
    store_diag<K, false>(out, voff, lane, a);
    st.row = out + Up;
    const int last_row = Nb - 1;
    const int nchunks = last_row / G + 1;

    LandedView lv;
    lv.addr0 = lk.landed;
#pragma unroll
    for (int w = 0; w < kLinLoaders; ++w) lv.have[w] = 0;
    bool timed_out = lengths_invalid(p, b);
    for (int ck = 0; ck < nchunks; ++ck) {
        timed_out |= !lv.wait(ck);
        const float *cur = bufs + (ck % NB) * chunkf + 2 * u0;
        const int r0 = ck * G;
        if (r0 + G <= last_row) {
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            LinRow<K> wq[3];
            lin_issue_row<K, 0>(wq[0], abase);
            lin_issue_row<K, 1>(wq[1], abase);
            lin_alpha_fast_steps<K, G, SH, 0>(a, wq, abase, st, voff, lane, r0);
        } else {
            for (int i = 0; i < G; ++i) {
                const int n = r0 + i + 1;
                if (n > last_row) break;
                LinRow<K> wc;
                lin_load_row<K>(wc, cur + i * 2 * Up);
                lin_alpha_step<K>(a, wc, st.d);
                if ((n & ((1 << SH) - 1)) == 0) lin_renorm<K, false, SH>(a, st, n >> SH);
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
            if (u0 + j == Ub - 1) lin_record(p, b, 0, a[j] * wrow[2 * j], st.E, timed_out);  // x p(blank | T_b-1, U_b-1)
    }
}

template <int K, int G, int NB, int SH>
__device__ void lin_beta_sweep(const LossParams &p, float *bufs, const LdLink lk, const int b, const int lane) {
    constexpr int Up = 64 * K, chunkf = G * 2 * Up;
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1;
    float *out = p.Bt + (size_t)b * p.Nr * Up;
    const int voff = lane * K * 4;
    const int u0 = lane * K;

    float bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bv[j] = (u0 + j == Ub - 1) ? 1.f : 0.f;  // the virtual terminal node (T_b, U_b - 1)
    const int last = Nb - 1;
    const int ckl = last / G;
    LinState st;
    st.E = 0, st.d = 0;
    st.tab = p.EB + (size_t)b * p.NCl * 64 + lane;
    st.row = out + (size_t)last * Up;

    LandedView lv;
    lv.addr0 = lk.landed;
#pragma unroll
    for (int w = 0; w < kLinLoaders; ++w) lv.have[w] = 0;
    bool timed_out = lengths_invalid(p, b);
    for (int ck = ckl; ck >= 0; --ck) {
        const int i_ring = ckl - ck;  // the loader's chunk index
        timed_out |= !lv.wait(i_ring);
        const float *cur = bufs + (i_ring % NB) * chunkf + 2 * u0;
        const int r0 = ck * G;
        if (r0 + G - 1 < last) {
            const uint32_t abase = (uint32_t)(uintptr_t)((lds_void *)cur);
            LinRow<K> wq[3];
            lin_issue_row<K, G - 1>(wq[0], abase);
            lin_issue_row<K, G - 2>(wq[1], abase);
            lin_beta_fast_steps<K, G, SH, 0>(bv, wq, abase, st, voff, lane, r0);
        } else {
            for (int ii = 0; ii < G; ++ii) {
                const int i = G - 1 - ii;
                const int n = r0 + i;
                if (n > last) continue;
                LinRow<K> wc;
                lin_load_row<K>(wc, cur + i * 2 * Up);
                lin_beta_step<K>(bv, wc, st.d);
                if ((n & ((1 << SH) - 1)) == (1 << SH) - 1 || n == last) lin_renorm<K, true, SH>(bv, st, n >> SH);
                store_diag<K, false>(st.row, voff, lane, bv);
                st.row -= Up;
            }
        }
        if (ck > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_post(lk.consumed, i_ring + 1);
        }
    }
    if (lane == 0) lin_record(p, b, 1, bv[0], st.E, timed_out);
}

template <int K, int G, int NB>
__global__ __launch_bounds__(64 * (1 + kLinLoaders)) void lin_sweep_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int chunkf = G * 2 * 64 * K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = p.b0 + (int)(blockIdx.x >> 1);
    const bool beta = (blockIdx.x & 1) != 0;
    // counters: [0] landed by loader 0, [1] consumed, [2] landed by loader 1, ...
    int *ctr = (int *)(lds + NB * chunkf);
    if (tid < 2 * kLinLoaders) ctr[tid] = 0;
    __syncthreads();
    LdLink lk;
    lk.landed = (uint32_t)(uintptr_t)((lds_void *)ctr);
    lk.consumed = lk.landed + 4u;
    if (wave >= 1) {
        LdLink mine = lk;
        mine.landed = lk.landed + 8u * (uint32_t)(wave - 1);
        if (beta)
            sweep_loader<K, G, NB, true, kLinLoaders>(p, lds, mine, b, lane, wave - 1);
        else
            sweep_loader<K, G, NB, false, kLinLoaders>(p, lds, mine, b, lane, wave - 1);
    } else {
        // Block length of this utterance (rnnt_lin.h): the mean decay statistic of its cells, from the per-(patch, wave) sums the
        // lsm launch left (a fixed-order sum: both directions arrive at the same choice; the loads overlap with the loader
        // wave's first chunk, which this wave has to wait for anyway).
        constexpr int SHmax = lin_shift_max(K);
        int sh = SHmax;
        if constexpr (SHmax > 2) {
            // (a sample is enough: wave 0's slot of every patch = the first quarter of the patch's lanes; eight loads in flight
            // per lane -- read one after the other the 1,500 slots of a 600 x 150 lattice cost the sweep 9 us)
            float sum = 0.f, cnt = 0.f;
            const float2 *ps = p.pstat + (size_t)b * p.nPstat;
            const int stride = p.pstatStride, npatch = p.nPstat / stride;
            for (int i0 = 0; i0 < npatch; i0 += 512) {
                float2 q[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    q[k] = (i < npatch) ? ps[(size_t)stride * i] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += q[k].x, cnt += q[k].y;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off), cnt += __shfl_xor(cnt, off);
            sh = (sum > kLinDecayBits * cnt) ? 2 : SHmax;  // (NaN statistics: the comparison is false, the certificate decides)
            sh = __builtin_amdgcn_readfirstlane(sh);
        }
        if constexpr (SHmax > 2) {
            if (sh == 2) {
                if (beta)
                    lin_beta_sweep<K, G, NB, 2>(p, lds, lk, b, lane);
                else
                    lin_alpha_sweep<K, G, NB, 2>(p, lds, lk, b, lane);
                return;
            }
        }
        if (beta)
            lin_beta_sweep<K, G, NB, SHmax>(p, lds, lk, b, lane);
        else
            lin_alpha_sweep<K, G, NB, SHmax>(p, lds, lk, b, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// The hand-back: ONE small launch at the end of every call of the linear path (grid = utterances x team; a workgroup whose
// utterance is fine reads two 16-byte words and returns).  An utterance is redone when a sweep flagged it (kFlagA / kFlagB),
// when the two likelihoods disagree, when the gradient pass's certificate failed for one of its cells (kFlagG), or when
// `force` is set (a gradient buffer the patch kernels cannot write): its team of workgroups (rnnt_redo.h) rebuilds the
// utterance's edge weights in the log2 domain from the logits (one lattice cell per thread, staged through LDS), runs the
// log-domain sweeps with the float64 recurrence (any range; alpha and beta side by side on two members) and, if gradients are
// wanted, writes all of the utterance's gradients (again a cell per thread, split over the team).  Afterwards the utterance's
// state word says that its lattice is in the log format, so that a later backward-only call goes straight to the gradient
// stage here.
// ---------------------------------------------------------------------------------------------
template <int K, int G, int NB>
__global__ __launch_bounds__(kRedoThreads) void lin_redo_kernel(const LossParams p, const int force, const int team) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int chunkf = G * 2 * 64 * K;
    const int tid = threadIdx.x;
    const int ub = (int)blockIdx.x / team;
    const int b = p.b0 + ub;
    RedoTeam tm;
    tm.k = (int)blockIdx.x - ub * team, tm.n = team, tm.bar = p.bar + kRedoCtr * b, tm.ok = true;
    int *fl = p.flags + 4 * b;
    // one round trip: the four flag words and the two likelihoods (both written write-through by the sweeps / the gradient pass)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const i32x4 fw = __builtin_nontemporal_load((const i32x4 *)fl);
    const f64x2 lw = __builtin_nontemporal_load((const f64x2 *)(p.ll + 2 * b));
    const int state = fw[kFlagState];
    bool redo = false;
    if (state != 2) {
        // |cost_alpha - cost_beta| in nats against what two float32 sweeps of the same lattice differ by (<= 1e-6 measured)
        const bool agree = fabs(lw[0] - lw[1]) * 0.6931471805599453 <= 2e-5 + 1e-8 * fabs(lw[0]);
        redo = force || (fw[kFlagA] | fw[kFlagB] | fw[kFlagG]) != 0 || !agree;
        if (!redo) return;
    } else if (!p.grads) {
        return;
    }
    if (redo) redo_lattice<K, G, NB>(p, b, tm, lds, tid);
    uint32_t lo, hi;
    redo_cell_range(p, b, tm, lo, hi);
    if (p.grads) redo_cells<true>(p, lo, hi, tid, lds, NB * chunkf);
#ifdef RNNT_REDO_TRACE
    REDO_STAMP(tm, 7);
    if (redo && tid == 0 && blockIdx.x < 8)
        printf("redo trace wg %d t0 %lld (clocks): fill %lld | sync %lld | lsm %lld | sync %lld | sweep %lld | sync %lld | grad %lld\n", (int)blockIdx.x, tm.ts[0] % 100000000ll,
               tm.ts[1] - tm.ts[0], tm.ts[2] - tm.ts[1], tm.ts[3] - tm.ts[2], tm.ts[4] - tm.ts[3], tm.ts[5] - tm.ts[4], tm.ts[6] - tm.ts[5],
               tm.ts[7] - tm.ts[6]);
#endif
    if (!tm.ok) {  // a team member never arrived (bounded spin): the utterance's results must not look valid
        if (p.costs && tm.k == 0 && tid == 0) st_f32_wt(p.costs + b, NAN);
        if (p.grads)
            for (size_t i = (size_t)lo * p.V + tid; i < (size_t)hi * p.V; i += kRedoThreads) p.grads[i] = NAN;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The linear path covers what the patch kernels cover (V <= 60, 16-byte-aligned logits) on every lattice the register-resident
// sweeps cover (up to 1024 columns).  One frame per lane spans up to 16 columns there: lattices on which that is too coarse
// (more label columns than frames, unstructured logits) fail the certificate and are redone in the log domain.
// (V >= 2: the lsm pass parks a cell's two edge probabilities in the cell's own LDS slot of V floats)
static int lin_cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return n;
}

bool lin_path_ok(const LossParams &p) { return p.V >= 2 && sweep_K(p.U) >= 1 && tile_path_ok(p, false); }

template <int K, int G>
static hipError_t launch_lin_sweep(const LossParams &p, hipStream_t s) {
    constexpr int NB = ((size_t)4 * G * 2 * 64 * K * sizeof(float) + 16 <= 128 * 1024) ? 4 : 3;
    constexpr size_t shm = (size_t)NB * G * 2 * 64 * K * sizeof(float) + 16;
    static_assert(shm <= 160 * 1024 && 2 * kLinLoaders * sizeof(int) <= 16, "chunk ring exceeds the LDS");
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)lin_sweep_kernel<K, G, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((lin_sweep_kernel<K, G, NB>), dim3(2 * p.nb), dim3(64 * (1 + kLinLoaders)), shm, s, p);
    return hipGetLastError();
}
template <int K, int G>
static hipError_t launch_lin_redo(const LossParams &p, const bool force, hipStream_t s) {
    constexpr int NB = ((size_t)4 * G * 2 * 64 * K * sizeof(float) + 16 <= 128 * 1024) ? 4 : 3;
    constexpr size_t shm = (size_t)NB * G * 2 * 64 * K * sizeof(float) + 16;
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)lin_redo_kernel<K, G, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    const int team = redo_team_size(p.nb, p.T, p.U, lin_cu_count());
    hipLaunchKernelGGL((lin_redo_kernel<K, G, NB>), dim3(p.nb * team), dim3(kRedoThreads), shm, s, p, force ? 1 : 0, team);
    return hipGetLastError();
}

hipError_t launch_sweeps_lin(const LossParams &p, hipStream_t s) {
    switch (sweep_K(p.U)) {
        case 1: return launch_lin_sweep<1, 16>(p, s);
        case 2: return launch_lin_sweep<2, 16>(p, s);
        case 3: return launch_lin_sweep<3, 16>(p, s);
        case 4: return launch_lin_sweep<4, 16>(p, s);
        case 6: return launch_lin_sweep<6, 8>(p, s);
        case 8: return launch_lin_sweep<8, 8>(p, s);
        case 12: return launch_lin_sweep<12, 4>(p, s);
        case 16: return launch_lin_sweep<16, 4>(p, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_redo_lin(const LossParams &p, const bool force, hipStream_t s) {
    switch (sweep_K(p.U)) {
        case 1: return launch_lin_redo<1, 16>(p, force, s);
        case 2: return launch_lin_redo<2, 16>(p, force, s);
        case 3: return launch_lin_redo<3, 16>(p, force, s);
        case 4: return launch_lin_redo<4, 16>(p, force, s);
        case 6: return launch_lin_redo<6, 8>(p, force, s);
        case 8: return launch_lin_redo<8, 8>(p, force, s);
        case 12: return launch_lin_redo<12, 4>(p, force, s);
        case 16: return launch_lin_redo<16, 4>(p, force, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rnnt
