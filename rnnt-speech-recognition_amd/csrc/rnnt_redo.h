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
// lattice, a 10x cliff for a step that holds one such utterance.  The launch now carries `team` workgroups per utterance (all of
// them read the utterance's flag words and return at once when it is fine); a flagged utterance's cell phases are split over
// the team, the alpha and beta sweeps run side by side on two of its members, and the phases are separated by a counter in the
// workspace (one agent-scope add per workgroup and phase, bounded spin, NaN results on a timeout -- never a hang).  Members of
// a team have consecutive block indices; workgroups are dispatched in block order, so the members of the oldest unfinished team
// are always resident or next in line and a waiting team cannot starve the one it waits for.
#pragma once
#include "rnnt_sweep.h"
#include "rnnt_lin.h"
#include "rnnt_cellbody.h"

namespace rnnt {

constexpr int kRedoThreads = 1024;

struct RedoTeam {
    int k, n;   // this workgroup's index in the team, team size
    int *bar;   // the utterance's phase counter (LossParams::bar; zeroed by the forward sweeps)
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

// Workgroups per flagged utterance: the whole chip for a small batch, at least ~1024 lattice cells per member.
inline int redo_team_size(int nb, int T, int U) {
#ifdef RNNT_REDO_TEAM  // dev builds: fixed team size (timing experiments)
    (void)nb, (void)T, (void)U;
    return RNNT_REDO_TEAM;
#else
    int t = 256 / (nb > 0 ? nb : 1);
    const long long cells = (long long)T * U;
    if ((long long)t * 1024 > cells) t = (int)(cells / 1024);
    return t < 1 ? 1 : (t > 16 ? 16 : t);
#endif
}

// Phase boundary of a team: everything this workgroup wrote is visible device-wide, then wait until `phase * n` arrivals.
// Everything a phase writes for OTHER workgroups leaves through write-through (sc1) stores -- the log-zero fill, the cells' edge
// weights and lse, the sweeps' lattice, offsets and likelihoods -- so the release side of the boundary is "my stores have been
// acknowledged" (s_waitcnt vmcnt(0)), not an L2 write-back: with a __threadfence() here every workgroup's write-back scanned
// its XCD's whole L2, 32 of them queueing per XCD and boundary -- 140 us of the 1.0 ms a fully handed-back batch took (round 5
// trace).  The acquire side is one cache invalidate after the wait (a reader's L2 may still hold what the linear sweeps read).
__device__ __forceinline__ void team_sync(RedoTeam &tm, const int phase, const int tid) {
    __shared__ int team_ok;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this thread's (write-through) stores are out
    __syncthreads();
    if (tm.n > 1) {
        if (tid == 0) {
            __hip_atomic_fetch_add(tm.bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int need = phase * tm.n;
            int ok = 0;
            // RELAXED polls (an L2-coherent load each, nothing else): an acquire load per poll carries a cache invalidate
            for (int spin = 0; spin < (1 << 18); ++spin) {
                if (__hip_atomic_load(tm.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            team_ok = ok;
        }
        __syncthreads();
        if (!team_ok) tm.ok = false;
    }
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
        else
            cell_body<64, false, GRAD, false, true>(p, cl, c, xs, out);
    }
}

// This member's share of the utterance's cells: the same split in every phase (a member re-reads what it wrote itself)
__device__ __forceinline__ void redo_cell_range(const LossParams &p, const int b, const RedoTeam &tm, uint32_t &lo, uint32_t &hi) {
    const uint32_t c0 = (uint32_t)b * (uint32_t)p.T * (uint32_t)p.U, n = (uint32_t)p.T * (uint32_t)p.U;
    lo = c0 + (uint32_t)((unsigned long long)n * (unsigned)tm.k / (unsigned)tm.n);
    hi = c0 + (uint32_t)((unsigned long long)n * (unsigned)(tm.k + 1) / (unsigned)tm.n);
}

// The log-domain lattice of utterance b from the logits at p.acts ([cells][p.V]): log2 edge weights (W), lse, alpha~ / beta~
// with their offset tables, ll (and the cost when p.costs is set).  Called by every workgroup of the utterance's team;
// `lds` = the sweep workgroup's chunk ring (NB chunks + the two counters).  Ends with the team in step (a phase boundary).
template <int K, int G, int NB>
__device__ __forceinline__ void redo_lattice(const LossParams &p, const int b, RedoTeam &tm, float *lds, const int tid) {
    constexpr int chunkf = G * 2 * 64 * K;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- log zero everywhere, then the cells ----
    uint32_t *Wb = (uint32_t *)(p.W + (size_t)b * p.Nr * 2 * p.Up);
    const uint32_t lz = (uint32_t)kFillByte * 0x01010101u;
    const size_t nW = (size_t)p.Nr * 2 * p.Up;
    REDO_STAMP(tm, 0);
    for (size_t i = (size_t)tm.k * kRedoThreads + tid; i < nW; i += (size_t)tm.n * kRedoThreads) st_i32_wt((int *)Wb + i, (int)lz);
    REDO_STAMP(tm, 1);
    team_sync(tm, 1, tid);
    REDO_STAMP(tm, 2);
    uint32_t lo, hi;
    redo_cell_range(p, b, tm, lo, hi);
    redo_cells<false>(p, lo, hi, tid, lds, NB * chunkf);
    REDO_STAMP(tm, 3);
    team_sync(tm, 2, tid);
    REDO_STAMP(tm, 4);
    // ---- the log-domain sweeps (float64 recurrence: whatever failed the certificate is a hard input): alpha by member 0,
    //      beta by member 1 (a team of one: one after the other); waves 0 (sweeping) and 1 (loading) ----
    int *ctr = (int *)(lds + NB * chunkf);
    LdLink lk;
    lk.landed = (uint32_t)(uintptr_t)((lds_void *)ctr);
    lk.consumed = lk.landed + 4u;
    const int kbeta = tm.n > 1 ? 1 : 0;
    if (tm.k == 0) {
        if (tid < 2) ctr[tid] = 0;
        __syncthreads();
        if (wave == 1)
            sweep_loader<K, G, NB, false>(p, lds, lk, b, lane);
        else if (wave == 0)
            alpha_sweep_pr<K, G, NB>(p, lds, lk, b, lane);
        __syncthreads();
    }
    if (tm.k == kbeta) {
        if (tid < 2) ctr[tid] = 0;
        __syncthreads();
        if (wave == 1)
            sweep_loader<K, G, NB, true>(p, lds, lk, b, lane);
        else if (wave == 0)
            beta_sweep_pr<K, G, NB>(p, lds, lk, b, lane);
    }
    REDO_STAMP(tm, 5);
    team_sync(tm, 3, tid);
    REDO_STAMP(tm, 6);
    if (tm.k == 0 && tid == 0) st_i32_wt(p.flags + 4 * b + kFlagState, 2);  // "log-domain lattice ready": later calls honour it
}

}  // namespace rnnt
