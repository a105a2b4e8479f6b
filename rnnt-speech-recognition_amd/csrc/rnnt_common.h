// rnnt_common.h -- shared device/host definitions for the gfx950 transducer-loss kernels.
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   acts / grads   f32 [B][T][U][V]            caller-owned, row-major (batch_first)
//   lse            f32 [B*T*U]                 natural-log softmax denominator per lattice cell
//   W              f32 [B][Nr][Up][2]          edge weights in LOG2 domain, DIAGONAL-MAJOR (skewed):
//                                              W[b][n][u][0] = log2 p(blank | t=n-u, u)
//                                              W[b][n][u][1] = log2 p(y_{u+1} | t=n-u, u)
//   A, Bt          f32 [B][Nr][Up]             alpha~/beta~ lattices, log2 domain, skewed,
//                                              stored relative to a per-block offset (kRebase diagonals)
//   offA, offB     f32 [B][NC][NG]             the offsets (integer-valued; per block of kRebase diagonals and group of OG
//                                              lattice columns: OG = K, the columns of one sweep lane, NG = 64 for the
//                                              register-resident sweeps; OG = 64, NG = Up/64 for the wide sweep)
//   ll             f64 [B][2]                  log2-likelihood from the alpha side / beta side
// with N = T+U-1 diagonals, Nr = N rounded up to 16 (a multiple of every chunk length), Up = 64*K (K = lattice columns per sweep lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnnt {

constexpr float kNeg = -1.0e30f;      // "log zero" that survives additions without inf-inf
constexpr float kNegTest = -1.0e29f;  // anything below this is "log zero"
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kRebase = 8;  // diagonals per precision block (alpha~/beta~ are re-based each block)

// Unsigned division by a launch-constant for n < 2^31 (Granlund-Montgomery, add-shift form).
struct FastDiv {
    uint32_t d, m, s;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)((((1ull << s) - d) << 32) / d) + 1u;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv f) {
    return (__umulhi(n, f.m) + n) >> f.s;
}

// Patch geometry of the small-vocabulary tile kernels (TT*UU <= 256 lattice cells per workgroup).
struct TileGeom {
    int TT, UU, tiles_t, tiles_u, cpr;  // cpr = 16-byte chunks per patch row = UU*V/4
    FastDiv div_tu, div_tt, divUU, div_cpr, divTT;
};

struct LossParams {
    const float *acts;
    float *grads;
    const int *labels;
    const int *label_lengths;
    const int *input_lengths;
    const float *cost_scale;  // nullable
    float *costs;
    float *lse;
    float *W;
    float *A;
    float *Bt;
    float *offA;  // integer-valued offsets, exact in f32
    float *offB;
    double *ll;
    // linear-domain lattice (rnnt_lin.h): integer frames per (block of kLinR diagonals, sweep lane), the likelihoods as
    // {mantissa, frame} pairs, and the per-utterance hand-back flags
    int *EA;
    int *EB;
    float *lik;  // [B][4]: alpha side {mantissa (float), frame (int)}, beta side {mantissa, frame}
    int *flags;  // [B][4]: kFlagA, kFlagB, kFlagG, kFlagState
    int NCl;     // frame blocks per utterance (tables are [NCl][64])
    float2 *pstat;  // [B][nPstat]: per (patch, wave) of the lsm launch {sum of -log2 max(p_blank, p_label), cells}: how fast mass decays
    int *lshift;    // [B]: log2 of the diagonals per frame block the sweeps chose for the utterance (rnnt_lin.h)
    int *bar;       // [B][8]: ticket / completion counters of the utterance's hand-back team (rnnt_redo.h), zeroed by the forward sweeps
    int nPstat;
    int pstatStride;  // the sweeps sample slot pstatStride * i, i < nPstat / pstatStride (4: wave 0 of every patch of the lsm launch)
    int B, T, U, V, blank;
    int b0, nb;  // this launch covers utterances [b0, b0+nb)
    int visit_all;  // 1: no occupancy floor -- the gradient kernels visit every cell / lattice row (RNNT_VISIT_ALL, include/rnnt.h)
    int precise;  // 1: the log-domain sweeps carry their recurrence in float64 (rnnt_sweep.h alpha_sweep_pr); lattices of 8+ columns per lane always do
    int N, Nr, Up, NC, NG;  // NG = Up/OG offset groups (offset tables are [NC][NG])
    uint32_t cells;  // B*T*U
    FastDiv divU, divT, divV, divOG;  // divOG: lattice column -> offset group
    TileGeom tile;
};

struct WsLayout {
    size_t lse, W, A, Bt, offA, offB, ll, EA, EB, lik, flags, pstat, lshift, bar, total;
    int NCl, nPstat;
    int N, Nr, Up, NC, NG, OG;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kMaxU = 8192;  // the wide sweep keeps two diagonals in LDS
// Lattice columns per lane in the register-resident sweeps (one of the instantiated widths); 0 = U > 1024: wide sweep.
inline int sweep_K(int U) {
    const int k = (U + 63) / 64;
    const int avail[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int a : avail)
        if (k <= a) return a;
    return 0;
}
// Byte every W word is pre-filled with: 0xF1F1F1F1 = -2.39e30f, a finite "log zero".
constexpr int kFillByte = 0xF1;

inline WsLayout make_layout(int T, int U, int B) {
    WsLayout w;
    w.N = T + U - 1;
    w.Nr = (int)align_up((size_t)w.N, 16);  // multiple of every sweep chunk length G
    // row stride of the skewed arrays = 64 lanes x K columns (U <= 1024), else U rounded up to 64
    w.Up = sweep_K(U) ? 64 * sweep_K(U) : (int)align_up((size_t)U, 64);
    w.NC = w.Nr / kRebase + 1;
    w.OG = sweep_K(U) ? sweep_K(U) : 64;  // columns per offset group: one sweep lane, or 64 columns of the wide sweep
    w.NG = w.Up / w.OG;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    w.lse = take((size_t)B * T * U * sizeof(float));
    w.W = take((size_t)B * w.Nr * 2 * w.Up * sizeof(float));
    w.A = take((size_t)B * w.Nr * w.Up * sizeof(float));
    w.Bt = take((size_t)B * w.Nr * w.Up * sizeof(float));
    w.offA = take((size_t)B * w.NC * w.NG * sizeof(float));
    w.offB = take((size_t)B * w.NC * w.NG * sizeof(float));
    w.ll = take((size_t)B * 2 * sizeof(double));
    w.NCl = w.Nr / 4 + 1;  // frame blocks of >= 4 diagonals (rnnt_lin.h)
    w.EA = take((size_t)B * w.NCl * 64 * sizeof(int));
    w.EB = take((size_t)B * w.NCl * 64 * sizeof(int));
    w.lik = take((size_t)B * 4 * sizeof(float));
    w.flags = take((size_t)B * 4 * sizeof(int));
    {
        const int nu = (U + 31) / 32, UU = (U + nu - 1) / nu, TT = (256 / UU) > T ? T : (256 / UU);  // make_tile's patch geometry
        w.nPstat = ((T + TT - 1) / TT) * ((U + UU - 1) / UU) * 4;  // four waves per patch workgroup
    }
    w.pstat = take((size_t)B * w.nPstat * 2 * sizeof(float));
    w.lshift = take((size_t)B * sizeof(int));
    w.bar = take((size_t)B * 8 * sizeof(int));  // rnnt_redo.h kRedoCtr words per utterance
    w.total = off;
    return w;
}

inline TileGeom make_tile(int T, int U, int V) {
    TileGeom g;
    const int nu = (U + 31) / 32;
    g.UU = (U + nu - 1) / nu;  // <= 32, splits U evenly (measured at U = 150: 30-wide patches beat 15-, 38- and 50-wide ones)
    g.TT = 256 / g.UU;
    if (g.TT > T) g.TT = T;
    g.tiles_u = (U + g.UU - 1) / g.UU;
    g.tiles_t = (T + g.TT - 1) / g.TT;
    g.cpr = g.UU * V / 4;
    g.div_tu = make_fastdiv((uint32_t)g.tiles_u);
    g.div_tt = make_fastdiv((uint32_t)g.tiles_t);
    g.divUU = make_fastdiv((uint32_t)g.UU);
    g.divTT = make_fastdiv((uint32_t)(g.TT > 0 ? g.TT : 1));
    g.div_cpr = make_fastdiv((uint32_t)(g.cpr > 0 ? g.cpr : 1));
    return g;
}

// The fused joints' tanh tables (joint_kernels.hip, joint_f16_kernels.hip; also written by the dense layer's GEMM epilogue,
// dense_kernels.hip): tab(x) = e^{2x} = 2^(x * kExpTab2Log2e), valid while |x| <= kExpTabLimit (beyond that the table-range flag
// sends the kernels to the exact tanh on the raw projections).  One definition for the three files that must agree.
constexpr float kExpTabLimit = 43.0f;
constexpr float kExpTab2Log2e = 2.8853900817779268f;
#ifdef __HIPCC__
__device__ __forceinline__ float exp_tab(float x) { return __builtin_amdgcn_exp2f(x * kExpTab2Log2e); }
__device__ __forceinline__ bool exp_tab_out_of_range(float x) { return !(fabsf(x) <= kExpTabLimit); }  // NaN too
#endif

// Hooks of the fused joint for a caller that also owns the joint's first Dense layer (rnnt_entrypoint.hip joint_net_call,
// dense_kernels.hip): work the dense kernels can do on the way, so that the joint need not redo it.
constexpr int kHookBlocks = 1024;  // grid of the two reductions that produce d enc_proj / d pred_proj
struct JointHooks {
    int prep_mode;        // 0: full prep kernel.  1: the e^{2x} tables of the projections and tflag[0] (zeroed, then raised by the
                          //    table writer) are already in the workspace: the prep kernel only does the W2 images.  2: the
                          //    workspace still holds the state of the forward call with the same inputs: no prep at all.
    int prefilled;        // 1: the caller has already run joint_prefill (edge array + flag words) in front of its own launches
    unsigned *dmax_enc;   // nullable: [kHookBlocks] per-block abs-max bit patterns of d enc_proj, written by its reduction
    unsigned *dmax_pred;  // nullable: the same for d pred_proj
};

// LDS-DMA (global -> LDS, 16 bytes per lane, lane l lands at lds + 16 l; `lds` wave-uniform).
// The compiler models the builtin as a FLAT access that may touch LDS: from then on every wait for a plain ds_read is
// `s_waitcnt lgkmcnt(0)` ("pending flat"), so an MFMA loop that prefetches its LDS fragments exposes one LDS latency per DMA
// instruction.  Issued as inline assembly instead the waits become lgkmcnt(N) again -- measured at config 5 (round 3): K1's chunk
// period drops from 4830 to 4210 shader clocks and its run time does not move (the kernels are power-limited: the clock drops
// instead), and K3 + K4 lose 3 ms (the asm statement is a barrier for the compiler's own LDS scheduling): the builtin stays.
#ifdef __HIPCC__
__device__ __forceinline__ void lds_dma16(const void *global, void *lds) {
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)global, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
#endif

// Every fill the library enqueues goes through its own kernel, not hipMemsetAsync: a memset NODE recorded by stream capture
// replayed a 16-byte garbage pattern instead of the 0xF1 bytes on this stack (ROCm 7.2, found by the HIP-graph tests) -- the
// fused paths survived it only because positions no lattice cell owns are never a source of probability mass.
// `bytes` must be a multiple of 4, `dst` 4-byte aligned; `byte` is replicated into every byte.
hipError_t launch_fill(void *dst, int byte, size_t bytes, hipStream_t s);
// the same for two regions in ONE launch
hipError_t launch_fill2(void *dst0, int byte0, size_t bytes0, void *dst1, int byte1, size_t bytes1, hipStream_t s);

// kernel launchers (rnnt_kernels.hip); return hipError_t from the launch
bool tile_path_ok(const LossParams &p, bool grad);
hipError_t launch_lsm(const LossParams &p, hipStream_t s);
hipError_t launch_sweeps(const LossParams &p, hipStream_t s);
hipError_t launch_grad(const LossParams &p, hipStream_t s);
// the linear-domain path of the small-vocabulary loss (rnnt_lin.h, rnnt_lin_kernels.hip)
bool lin_path_ok(const LossParams &p);
hipError_t launch_lsm_lin(const LossParams &p, hipStream_t s);
hipError_t launch_sweeps_lin(const LossParams &p, hipStream_t s);
hipError_t launch_grad_lin(const LossParams &p, hipStream_t s);
hipError_t launch_redo_lin(const LossParams &p, bool force, hipStream_t s);

}  // namespace rnnt
