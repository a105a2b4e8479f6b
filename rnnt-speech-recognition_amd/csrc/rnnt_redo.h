// rnnt_redo.h -- the hand-back of the linear-domain lattice (rnnt_lin.h): device code shared by the loss op's hand-back launch
// (rnnt_lin_kernels.hip lin_redo_kernel) and the fused joint's (joint_kernels.hip joint_redo_kernel).
//
// An utterance the linear lattice cannot represent (a sweep flagged it, the two likelihoods disagree, the gradient pass's range
// certificate failed) is redone in the LOG domain from its logits: edge weights (a lattice cell per thread), the log2 recurrence
// in float64 registers (rnnt_sweep.h alpha_sweep_pr / beta_sweep_pr), then whatever the caller forms per cell from that lattice.
// Replaces the same stages of warp-transducer's GPU path as the kernels it falls back to (SURVEY.md 2.1, 8a-6 ... a-9; call
// site utils/loss.py:34-35), which has no such fast path and hence no hand-back.
//
// Round 5: a TEAM of workgroups per utterance.  One workgroup -- one CU's worth of exponentials -- took ~2 ms for a 600 x 150
// lattice, a 10x cliff for a step that holds one such utterance.  The launch carries `team` workgroups per utterance (all of
// them read the utterance's flag words and return at once when it is fine); a flagged utterance's cell phases are cut into `team`
// parts, the alpha and beta sweeps are two parts of their own.
// Round 6: no workgroup is ASSIGNED a part.  Every phase has a ticket counter and a completion counter in the workspace; a
// workgroup of the team draws tickets of the current phase and works them off until none is left, then waits for the phase's
// completion count before it moves on.  Whoever holds a ticket is running and waits for nothing while it works, so a phase
// completes however few of the team's workgroups are resident -- a lone one draws every ticket and does the utterance by itself.
// (Round 5 gave member k part k and let everybody wait for `team` arrivals: correct only while all members become resident
// together, i.e. while nothing else -- another stream's kernels, a CU mask -- holds the CUs; a late member meant NaN results.)
// The spin on the completion count stays bounded (minutes, not milliseconds) and poisons the results if it ever gives up.
#pragma once
#include "rnnt_sweep.h"
#include "rnnt_lin.h"
#include "rnnt_cellbody.h"

namespace rnnt {

constexpr int kRedoThreads = 1024;

constexpr int kRedoCtr = 8;  // counter words per utterance: tickets of phases 0..2, completions of phases 0..2 (LossParams::bar)
struct RedoTeam {
    int k, n;   // this workgroup's index in the team (only the LAST phase -- the caller's, which nobody waits for -- is dealt by it), team size
    int *bar;   // the utterance's counters (zeroed by the forward sweeps)
    bool ok;    // false once a bounded spin gave up: the caller poisons its outputs
#ifdef RNNT_REDO_TRACE  // dev builds: s_memtime stamps of the phases (printed by member 0 of the launch's first utterance)
    long long ts[8];
#endif
};
#ifdef RNNT_REDO_TRACE
#define REDO_STAMP(tm, i) ((tm).ts[i] = (long long)__builtin_amdgcn_s_memtime())
#else
#define REDO_STAMP(tm, i) ((void)0)
#endif

// Workgroups per flagged utterance: the whole chip for a small batch, at least ~1024 lattice cells per part.
inline int redo_team_size(int nb, int T, int U, int cus) {
#ifdef RNNT_REDO_TEAM  // dev builds: fixed team size (timing experiments)
    (void)nb, (void)T, (void)U, (void)cus;
    return RNNT_REDO_TEAM;
#else
    int t = (cus > 0 ? cus : 256) / (nb > 0 ? nb : 1);
    const long long cells = (long long)T * U;
    if ((long long)t * 1024 > cells) t = (int)(cells / 1024);
    return t < 1 ? 1 : (t > 16 ? 16 : t);
#endif
}

// A ticket of phase `ph`: the index of a part nobody else works on (>= the phase's part count: none left).  Workgroup-uniform.
__device__ __forceinline__ int team_take(RedoTeam &tm, const int ph, const int tid) {
    __shared__ int team_ticket;
    __syncthreads();  // (the previous ticket has been read by everybody)
    if (tid == 0) team_ticket = __hip_atomic_fetch_add(tm.bar + ph, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return team_ticket;
}
// The part this workgroup held is complete.  Everything a phase writes for OTHER workgroups leaves through write-through (sc1)
// stores -- the log-zero fill, the cells' edge weights and lse, the sweeps' lattice, offsets and likelihoods -- so the release
// side is "my stores have been acknowledged" (s_waitcnt vmcnt(0)), not an L2 write-back: with a __threadfence() here every
// workgroup's write-back scanned its XCD's whole L2, 32 of them queueing per XCD and boundary (round 5 trace: 140 us of 1.0 ms).
__device__ __forceinline__ void team_done(RedoTeam &tm, const int ph, const int tid) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this thread's (write-through) stores are out
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(tm.bar + 3 + ph, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wait until all `nparts` parts of phase `ph` are complete (every one of them is held by a running workgroup by now: the caller
// has seen the tickets run out).  The acquire side is one cache invalidate after the wait (a reader's L2 may still hold what the
// linear sweeps read).
__device__ __forceinline__ void team_wait(RedoTeam &tm, const int ph, const int nparts, const int tid) {
    __shared__ int team_ok;
    if (tid == 0) {
        int ok = 0;
        // RELAXED polls (an L2-coherent load each, nothing else): an acquire load per poll carries a cache invalidate
        for (int spin = 0; spin < (1 << 24); ++spin) {
            if (__hip_atomic_load(tm.bar + 3 + ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nparts) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
        team_ok = ok;
    }
    __syncthreads();
    if (!team_ok) tm.ok = false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // nothing stale is served from this CU's vector L1 / this XCD's L2
}

// log2-domain edge weights + lse of one valid cell from its V logits at `xs`, without a register image of the row (any V): what
// cell_body<VP, ., false> computes, for vocabularies beyond its 64 registers.
__device__ __forceinline__ void cell_lsm_loop(const LossParams &p, const Cell &cl, const uint32_t c, const float *xs) {
    const int V = p.V;
    float m = xs[0];
    for (int i = 1; i < V; ++i) m = fmaxf(m, xs[i]);
    const float nml = -m * kLog2e;
    float s = 0.f;
    for (int i = 0; i < V; ++i) s += ex2(fmaf(xs[i], kLog2e, nml));
    const float lg2s = lg2(s);
    const bool blank_stays = (cl.t < cl.Tb - 1) || (cl.u == cl.Ub - 1);
    const float ob = blank_stays ? fmaf(xs[p.blank] - m, kLog2e, -lg2s) : kNeg;
    float ol = kNeg;
    if (cl.u < cl.Ub - 1) {
        const int lab = clamp_label(p.labels[(size_t)cl.b * (p.U - 1) + cl.u], V);
        ol = fmaf(xs[lab] - m, kLog2e, -lg2s);
    }
    const size_t wi = ((size_t)cl.b * p.Nr + (cl.t + cl.u)) * p.Up + cl.u;
    st_f32_wt(p.lse + c, m + kLn2 * lg2s);  // (write-through: read by other workgroups of the team within this launch)
    st_f32_wt(p.W + 2 * wi, ob), st_f32_wt(p.W + 2 * wi + 1, ol);
}

// The cells [c0, c1) of one utterance, one per lane.  With 16-byte-aligned rows the logits are staged through LDS (the workgroup's
// chunk ring is idle during the cell phases) and the gradients leave the same way: every global access is a coalesced 16-byte
// piece.  (A lane per cell straight from global memory made every load / store instruction of a wave touch 56 cache lines.)
// Round 5: every WAVE stages its own cells in its own slice of the ring -- all of a pass's loads in flight at once, no workgroup
// barrier anywhere -- so that the sixteen waves of the one workgroup a CU holds cover each other's memory latency (the
// workgroup-wide "load a chunk, barrier, compute, barrier, store, barrier" of round 4 left the CU waiting on every step: a batch
// in which every utterance is handed back took 1.0 ms in this kernel).  Otherwise (V % 4 != 0 or unaligned tensors): straight
// from / to global memory.
template <bool GRAD>
__device__ __forceinline__ void redo_cells(const LossParams &p, const uint32_t c0, const uint32_t c1, const int tid, float *lds,
                                           const int lds_floats) {
    const bool v4 = (p.V % 4) == 0 && (((uintptr_t)p.acts | (uintptr_t)p.grads) & 15) == 0;
    const int V = p.V;
    if (v4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr int kWaves = kRedoThreads / 64, kPieces = 6;  // 16-byte pieces a lane holds in flight per pass (a 96 KB ring: 6 KB per wave)
        const int lane = tid & 63, wave = tid >> 6;
        const int per = min(lds_floats / kWaves, kPieces * 64 * 4) & ~3;  // floats of this wave's slice
        const uint32_t CW = (uint32_t)min(64, per / V);                   // cells per wave and pass (>= 1: V <= 128)
        float *my = lds + (size_t)wave * (lds_floats / kWaves);
        for (uint32_t cs = c0 + (uint32_t)wave * CW; cs < c1; cs += (uint32_t)kWaves * CW) {
            const uint32_t n = min(CW, c1 - cs), nq = n * (uint32_t)V / 4u;
            const v4f *src = (const v4f *)(p.acts + (size_t)cs * V);
            v4f r[kPieces];
#pragma unroll
            for (int k = 0; k < kPieces; ++k)
                if ((uint32_t)(lane + 64 * k) < nq) r[k] = src[lane + 64 * k];
#pragma unroll
            for (int k = 0; k < kPieces; ++k)
                if ((uint32_t)(lane + 64 * k) < nq) ((v4f *)my)[lane + 64 * k] = r[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (a wave's LDS operations complete in order: no wait needed,
            __builtin_amdgcn_wave_barrier();                        //  only the compiler must keep the order)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((uint32_t)lane < n) {
                const uint32_t c = cs + (uint32_t)lane;
                const Cell cl = decode(p, c);
                float *xs = my + (size_t)lane * V;
                if (GRAD || cl.valid) {
                    if (V <= 32)
                        cell_body<32, true, GRAD, false, true>(p, cl, c, xs, xs);
                    else if (V <= 64)
                        cell_body<64, true, GRAD, false, true>(p, cl, c, xs, xs);
                    else if (!GRAD)
                        cell_lsm_loop(p, cl, c, xs);  // (the fused joint's parked logits at 65 ... 128 symbols: edge weights only)
                    else  // no gradient body beyond 64 symbols (the loss op's linear path stops at 60, the joint passes no grads):
                        for (int i = 0; i < V; ++i) xs[i] = NAN;  // never the staged LOGITS as gradients -- NaN if a caller ever gets here
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (GRAD) {
                v4f *dst = (v4f *)(p.grads + (size_t)cs * V);
#pragma unroll
                for (int k = 0; k < kPieces; ++k)
                    if ((uint32_t)(lane + 64 * k) < nq) __builtin_nontemporal_store(((const v4f *)my)[lane + 64 * k], dst + lane + 64 * k);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        return;
    }
    for (uint32_t c = c0 + (uint32_t)tid; c < c1; c += kRedoThreads) {
        const Cell cl = decode(p, c);
        const float *xs = p.acts + (size_t)c * p.V;
        float *out = GRAD ? p.grads + (size_t)c * p.V : nullptr;
        if (!GRAD && !cl.valid) continue;
        if (p.V <= 32)
            cell_body<32, false, GRAD, false, true>(p, cl, c, xs, out);
        else if (p.V <= 64)
            cell_body<64, false, GRAD, false, true>(p, cl, c, xs, out);
        else if (!GRAD)
            cell_lsm_loop(p, cl, c, xs);
        else  // (unreachable today, see above: loud, not wrong)
            for (int i = 0; i < p.V; ++i) out[i] = NAN;
    }
}

// Part `k` of `n` of the utterance's cells (the same cut in every phase)
__device__ __forceinline__ void redo_cell_part(const LossParams &p, const int b, const int k, const int n, uint32_t &lo, uint32_t &hi) {
    const uint32_t c0 = (uint32_t)b * (uint32_t)p.T * (uint32_t)p.U, nc = (uint32_t)p.T * (uint32_t)p.U;
    lo = c0 + (uint32_t)((unsigned long long)nc * (unsigned)k / (unsigned)n);
    hi = c0 + (uint32_t)((unsigned long long)nc * (unsigned)(k + 1) / (unsigned)n);
}
// ... the part the LAST phase deals to this workgroup by its index (nobody waits for that phase: no ticket needed)
__device__ __forceinline__ void redo_cell_range(const LossParams &p, const int b, const RedoTeam &tm, uint32_t &lo, uint32_t &hi) {
    redo_cell_part(p, b, tm.k, tm.n, lo, hi);
}

// The log-domain lattice of utterance b from the logits at p.acts ([cells][p.V]): log2 edge weights (W), lse, alpha~ / beta~
// with their offset tables, ll (and the cost when p.costs is set).  Called by every workgroup of the utterance's team;
// `lds` = the sweep workgroup's chunk ring (NB chunks + the two counters).  Returns once all three phases are complete.
template <int K, int G, int NB>
__device__ __forceinline__ void redo_lattice(const LossParams &p, const int b, RedoTeam &tm, float *lds, const int tid) {
    constexpr int chunkf = G * 2 * 64 * K;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- phase 0: log zero everywhere ----
    uint32_t *Wb = (uint32_t *)(p.W + (size_t)b * p.Nr * 2 * p.Up);
    const uint32_t lz = (uint32_t)kFillByte * 0x01010101u;
    const size_t nW = (size_t)p.Nr * 2 * p.Up;
    REDO_STAMP(tm, 0);
    for (int part = team_take(tm, 0, tid); part < tm.n; part = team_take(tm, 0, tid)) {
        for (size_t i = (size_t)part * kRedoThreads + tid; i < nW; i += (size_t)tm.n * kRedoThreads) st_i32_wt((int *)Wb + i, (int)lz);
        team_done(tm, 0, tid);
    }
    REDO_STAMP(tm, 1);
    team_wait(tm, 0, tm.n, tid);
    REDO_STAMP(tm, 2);
    // ---- phase 1: the cells' edge weights ----
    for (int part = team_take(tm, 1, tid); part < tm.n; part = team_take(tm, 1, tid)) {
        uint32_t lo, hi;
        redo_cell_part(p, b, part, tm.n, lo, hi);
        redo_cells<false>(p, lo, hi, tid, lds, NB * chunkf);
        team_done(tm, 1, tid);
    }
    REDO_STAMP(tm, 3);
    team_wait(tm, 1, tm.n, tid);
    REDO_STAMP(tm, 4);
    // ---- phase 2: the log-domain sweeps (float64 recurrence: whatever failed the certificate is a hard input), two parts:
    //      alpha and beta (side by side on two workgroups, or one after the other); waves 0 (sweeping) and 1 (loading) ----
    int *ctr = (int *)(lds + NB * chunkf);
    LdLink lk;
    lk.landed = (uint32_t)(uintptr_t)((lds_void *)ctr);
    lk.consumed = lk.landed + 4u;
    for (int part = team_take(tm, 2, tid); part < 2; part = team_take(tm, 2, tid)) {
        if (tid < 2) ctr[tid] = 0;
        __syncthreads();
        if (part == 0) {
            if (wave == 1)
                sweep_loader<K, G, NB, false>(p, lds, lk, b, lane);
            else if (wave == 0)
                alpha_sweep_pr<K, G, NB>(p, lds, lk, b, lane);
        } else {
            if (wave == 1)
                sweep_loader<K, G, NB, true>(p, lds, lk, b, lane);
            else if (wave == 0)
                beta_sweep_pr<K, G, NB>(p, lds, lk, b, lane);
        }
        team_done(tm, 2, tid);
    }
    REDO_STAMP(tm, 5);
    team_wait(tm, 2, 2, tid);
    REDO_STAMP(tm, 6);
    if (tm.k == 0 && tid == 0) st_i32_wt(p.flags + 4 * b + kFlagState, 2);  // "log-domain lattice ready": later calls honour it
}

}  // namespace rnnt
