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
#include "rnnt_sweep.h"
#include "rnnt_cellwave.h"
#include "rnnt_lin.h"
#include "rnnt_cellbody.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

namespace rnnt {


// The gradient pass's 16-byte stores: non-temporal (written once, never re-read by this op).  Measured alternatives (round 3):
// plain, sc1, sc0 sc1, sc1 nt, sc0 sc1 nt stores and non-temporal logit LOADS were all slower or equal.
typedef float gs_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void grad_store16(gs_v4f *dst, const gs_v4f v) { __builtin_nontemporal_store(v, dst); }

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
constexpr int kFillRows = 16;  // W diagonals per fill workgroup of the lsm launch

template <int VP, bool GRAD, bool AL = true, bool LIN = false>
__global__ __launch_bounds__(256) void cell_tile_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V;
    const TileGeom &tg = p.tile;
    // The lsm launch carries extra workgroups that write "log zero" into the W positions no lattice cell writes (the sweeps
    // read whole rows of the skewed array): kFillRows diagonals of one utterance each.  This replaces a 37 MB memset in front of
    // the launch (7.5 us at C2) by ~14 MB of stores.  They are INTERLEAVED with the patch workgroups -- every P-th group of eight
    // consecutive block indices (one per XCD) is a fill group -- so that their stores run under the patches' reads; appended behind
    // the patches (rounds 2-3) they formed a write-only tail of the launch (~4.7 us of 68 at B32 T600 U150).
    const uint32_t n_patch_wg = (uint32_t)p.nb * (uint32_t)tg.tiles_t * (uint32_t)tg.tiles_u;
    uint32_t vblock = blockIdx.x;  // index among the patch workgroups
    if (!GRAD) {
        const uint32_t per = (uint32_t)p.Nr / kFillRows, n_fill = (uint32_t)p.nb * per;
        const uint32_t pgs = (n_patch_wg + 7u) >> 3, fgs = (n_fill + 7u) >> 3, P = (pgs + fgs) / fgs;
        const uint32_t g = blockIdx.x >> 3, l8 = blockIdx.x & 7u;
        const uint32_t k = g / P;
        const bool is_fill = g - k * P == P - 1u && k < fgs;  // fill groups sit at positions P - 1, 2 P - 1, ..., fgs P - 1
        if (is_fill) {
            const uint32_t f = k * 8u + l8;
            if (f >= n_fill) return;
            const int fb = p.b0 + (int)(f / per), chunk = (int)(f % per);
            const int Tf = length_T(p, fb), Uf = length_U(p, fb);
            const int Nf = Tf + Uf - 1;
            float2 *Wb = (float2 *)p.W + (size_t)fb * p.Nr * p.Up;
            const float2 z = LIN ? make_float2(0.f, 0.f) : make_float2(kNeg, kNeg);  // linear lattice: probability zero
            for (int idx = tid; idx < kFillRows * p.Up; idx += 256) {
                const int rr = idx / p.Up, u = idx - rr * p.Up;
                const int n = chunk * kFillRows + rr;
                // lattice cells of diagonal n: u in [lo, hi]; diagonals past the lattice are never read by the sweeps
                if (n < Nf && (u < max(0, n - Tf + 1) || u > min(n, Uf - 1))) Wb[(size_t)n * p.Up + u] = z;
            }
            return;
        }
        vblock = (g - min(k, fgs)) * 8u + l8;  // fill groups in front of group g: min(g / P, fgs)
        if (vblock >= n_patch_wg) return;
    }
    // XCD-aware remap: hand each XCD (blockIdx % 8) a contiguous range of patches (bijective form)
    uint32_t bid;
    {
        const uint32_t nwg = n_patch_wg, xcd = vblock & 7u, idx = vblock >> 3;
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
    if (GRAD && LIN && lin_skip(p, b)) return;  // handed back to the log-domain path: lin_redo_kernel writes this utterance's gradients

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

    // LIN lsm: every wave of every patch leaves {sum of the cells' decay statistic, cells} for the sweeps (rnnt_lin.h)
    float2 *const pslot = p.pstat + (size_t)b * p.nPstat + ((size_t)tt * tg.tiles_u + tu) * 4 + wave;
    if (rows_valid == 0 || cols_valid == 0) {
        if (GRAD) {  // an all-padding patch: exact zeros, no reads
            for (int r = wave; r < rows_in; r += 4) store_row(r, nullptr);
        }
        if (!GRAD && LIN && lane == 0) *pslot = make_float2(0.f, 0.f);
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
                __builtin_amdgcn_global_load_lds((glb_void *)(src + q * 4), (lds_void *)(dst + q0 * 4), 16, 0, 0);
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
    float stat = 0.f;
    if (AL) {
        if ((GRAD && tid < tg.TT * tg.UU) || cl.valid) stat = cell_body<VP, true, GRAD, LIN>(p, cl, c, lds + tid * V, lds + tid * V);  // (lanes beyond the patch own no LDS)
    } else if ((int)r < tg.TT) {
        const int a = (int)((patch0 + r * row_f) & 3);
        if (GRAD || cl.valid) stat = cell_body<VP, false, GRAD, LIN>(p, cl, c, lds + r * row_lds + a + cu * V, lds + r * row_lds + a + cu * V);
    }
    if (!GRAD && LIN) {  // wave sums by butterfly (no LDS: the patch image is still being read by other waves)
        float cnt = cl.valid ? 1.f : 0.f;
        stat = cl.valid ? stat : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) stat += __shfl_xor(stat, off), cnt += __shfl_xor(cnt, off);
        if (lane == 0) *pslot = make_float2(stat, cnt);
        // The edge probabilities, from the cells' LDS slots to the skewed array DIAGONAL by diagonal: thread j takes row j % TT of
        // patch diagonal j / TT, so TT consecutive lanes store TT consecutive (descending) positions of one row of W -- 64-byte
        // runs at TT = 8 -- where a lane per cell scattered every wave-store over 64 rows (2.9 M 8-byte requests per step at
        // B32 T600 U150: ~10 us of the lsm pass).
        __syncthreads();
        float2 *const Wp = (float2 *)p.W + ((size_t)b * p.Nr + t0 + u0) * p.Up + u0;  // cell (r, c) of the patch: Wp[(r + c) Up + c]
        const uint32_t total = (uint32_t)(tg.TT + tg.UU - 1) * (uint32_t)tg.TT;
        for (uint32_t j = tid; j < total; j += 256) {
            const uint32_t d = fdiv(j, tg.divTT);
            const int rr = (int)(j - d * (uint32_t)tg.TT), cc = (int)d - rr;
            if (rr < rows_valid && cc >= 0 && cc < cols_valid) {
                float2 v;
                if (AL) {
                    v = *(const float2 *)(lds + (rr * tg.UU + cc) * V);
                } else {
                    const float *src = lds + rr * row_lds + (int)((patch0 + rr * row_f) & 3) + cc * V;
                    v = make_float2(src[0], src[1]);
                }
                Wp[(size_t)(rr + cc) * p.Up + cc] = v;
            }
        }
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
// General path (any V, any alignment): one lattice cell per WAVE, lanes stride over V (rnnt_cellwave.h).
// ---------------------------------------------------------------------------------------------
template <bool V4, bool GRAD>
__global__ __launch_bounds__(256) void cell_wave_kernel(const LossParams p) {
    // No grid stride: a workgroup owns ONE contiguous span of cells (about 16 KB of logits: 4 waves x wave_cells(V) cells), as the
    // streaming kernels that reach 6 TB/s on this part do (scripts/probes/probe_hbm.hip; grid-stride loops: 4.7-5.0 TB/s).
    const uint32_t cpw = (uint32_t)wave_cells(p.V);
    const uint32_t c_lo = (blockIdx.x * 4u + (threadIdx.x >> 6)) * cpw;
    cell_wave_range<V4, GRAD>(p, c_lo, min(c_lo + cpw, p.cells), threadIdx.x & 63);
}

// ---------------------------------------------------------------------------------------------
// fills (see launch_fill in rnnt_common.h)
// ---------------------------------------------------------------------------------------------
// One contiguous 16 KB span per workgroup (four 16-byte stores per thread), no grid stride.
__device__ __forceinline__ void fill_span(uint32_t *dst, const uint32_t word, const size_t nwords, const unsigned blk) {
    if ((((uintptr_t)dst) & 15) == 0) {
        const size_t n4 = nwords >> 2;
        const uint4 q = make_uint4(word, word, word, word);
        const size_t base = (size_t)blk * 1024 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + 256 * k < n4) ((uint4 *)dst)[base + 256 * k] = q;
        if (blk == 0 && threadIdx.x < (nwords & 3)) dst[n4 * 4 + threadIdx.x] = word;
    } else {
        const size_t base = (size_t)blk * 4096 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (base + 256 * k < nwords) dst[base + 256 * k] = word;
    }
}
__global__ __launch_bounds__(256) void fill_kernel(uint32_t *dst, const uint32_t word, const size_t nwords) {
    fill_span(dst, word, nwords, blockIdx.x);
}
// two regions in one launch (the fused joints zero their flag words with the same launch that pre-fills the edge array)
__global__ __launch_bounds__(256) void fill2_kernel(uint32_t *dst0, const uint32_t word0, const size_t nwords0, const unsigned grid0,
                                                    uint32_t *dst1, const uint32_t word1, const size_t nwords1) {
    if (blockIdx.x < grid0)
        fill_span(dst0, word0, nwords0, blockIdx.x);
    else
        fill_span(dst1, word1, nwords1, blockIdx.x - grid0);
}

static inline uint32_t fill_word(int byte) {
    const uint32_t b = (uint32_t)(byte & 0xff);
    return b | (b << 8) | (b << 16) | (b << 24);
}

hipError_t launch_fill(void *dst, int byte, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) != 0 || (((uintptr_t)dst) & 3) != 0) return hipErrorInvalidValue;
    const size_t nwords = bytes >> 2;
    const size_t grid = (nwords + 4095) / 4096;  // 16 KB per workgroup
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)grid), dim3(256), 0, s, (uint32_t *)dst, fill_word(byte), nwords);
    return hipGetLastError();
}

hipError_t launch_fill2(void *dst0, int byte0, size_t bytes0, void *dst1, int byte1, size_t bytes1, hipStream_t s) {
    if (bytes0 == 0) return launch_fill(dst1, byte1, bytes1, s);
    if (bytes1 == 0) return launch_fill(dst0, byte0, bytes0, s);
    if (((bytes0 | bytes1) & 3) != 0 || ((((uintptr_t)dst0) | ((uintptr_t)dst1)) & 3) != 0) return hipErrorInvalidValue;
    const size_t n0 = bytes0 >> 2, n1 = bytes1 >> 2;
    const size_t g0 = (n0 + 4095) / 4096, g1 = (n1 + 4095) / 4096;
    if (g0 + g1 > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fill2_kernel, dim3((unsigned)(g0 + g1)), dim3(256), 0, s, (uint32_t *)dst0, fill_word(byte0), n0, (unsigned)g0,
                       (uint32_t *)dst1, fill_word(byte1), n1);
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
template <bool GRAD, bool LIN = false>
static hipError_t launch_cell(const LossParams &p, hipStream_t s) {
    if (LIN && !tile_path_ok(p, GRAD)) return hipErrorInvalidValue;
    if (tile_path_ok(p, GRAD)) {
        // (the lsm launch fills the log-zero part of W itself: see cell_tile_kernel)
        const unsigned n_patch = (unsigned)p.nb * p.tile.tiles_t * p.tile.tiles_u;
        unsigned blocks = n_patch;
        if (!GRAD)  // + the fill workgroups, in groups of eight among the patch groups (cell_tile_kernel)
            blocks = (((n_patch + 7u) >> 3) + (((unsigned)p.nb * (unsigned)(p.Nr / kFillRows) + 7u) >> 3)) * 8u;
        // the patch image: TT x UU cells (<= 256) of V floats, to the byte.  Sized by the patch, not by the 256 lanes, and without
        // padding: at V = 28 that is 26,880 B -- six workgroups per CU instead of five (gradient pass 115 -> 110.5 us); at V = 32
        // 30,720 B instead of 32,832 B -- five instead of four (146 -> 140 us)
        const size_t shm = (size_t)p.tile.TT * p.tile.UU * p.V * sizeof(float);
        if ((p.V % 4) != 0) {
            const size_t pitch = (size_t)((p.tile.UU * p.V + 3 + 3) & ~3);
            size_t shmu = (size_t)p.tile.TT * pitch * sizeof(float);
            if (shmu < shm) shmu = shm;
            if (p.V <= 32)
                hipLaunchKernelGGL((cell_tile_kernel<32, GRAD, false, LIN>), dim3(blocks), dim3(256), shmu, s, p);
            else
                hipLaunchKernelGGL((cell_tile_kernel<64, GRAD, false, LIN>), dim3(blocks), dim3(256), shmu, s, p);
        } else if (p.V <= 32)
            hipLaunchKernelGGL((cell_tile_kernel<32, GRAD, true, LIN>), dim3(blocks), dim3(256), shm, s, p);
        else
            hipLaunchKernelGGL((cell_tile_kernel<64, GRAD, true, LIN>), dim3(blocks), dim3(256), shm, s, p);
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
hipError_t launch_lsm_lin(const LossParams &p, hipStream_t s) { return launch_cell<false, true>(p, s); }
hipError_t launch_grad_lin(const LossParams &p, hipStream_t s) { return launch_cell<true, true>(p, s); }


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
    } else if (K >= 8 || p.precise) {  // the float64 recurrence (rnnt_sweep.h): the loss op, and every lattice of 8+ columns per lane (wider than 384 columns)
        if (beta)
            beta_sweep_pr<K, G, NB>(p, lds, lk, b, lane);
        else
            alpha_sweep_pr<K, G, NB>(p, lds, lk, b, lane);
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
// order of magnitude slower per diagonal than the register-resident sweeps.  Same arithmetic as the precise sweeps of
// rnnt_sweep.h (alpha / beta as TRUE log2 values in float64, the log2(1 + 2^-|d|) term on the float32 units) and the same
// outputs, so the gradient pass does not know which sweep ran: float32 residues against one integer offset per block of
// kRebase diagonals and group of 64 columns (the group's largest value, rounded; a group without mass copies the neighbour
// the mass will come from), offsets in the tables.  (Round 3 carried float32 residues against ONE offset per block -- the
// straight-line ridge cell -- for all columns: 1.4e-4 of relative cost error on a trained-like 1200 x 1100 lattice.)
// ---------------------------------------------------------------------------------------------
template <bool BETA>
__global__ __launch_bounds__(1024) void sweep_wide_kernel(const LossParams p) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = p.b0 + (int)blockIdx.x;
    const int Up = p.Up;  // a multiple of 64: every wave iteration covers one whole offset group
    const int Tb = length_T(p, b), Ub = length_U(p, b);
    const int Nb = Tb + Ub - 1, last = Nb - 1;
    const float2 *Wb = (const float2 *)p.W + (size_t)b * p.Nr * Up;
    float *out = (BETA ? p.Bt : p.A) + (size_t)b * p.Nr * Up;
    float *offs = (BETA ? p.offB : p.offA) + (size_t)b * p.NC * p.NG;
    double *cur = ldsd, *nxt = ldsd + Up;
    float *gm = (float *)(ldsd + 2 * Up);  // [NG] the groups' rounded maxima of the row being re-based (NaN: no mass)
    constexpr int kIter = kMaxU / 1024;    // column iterations per thread at most
    float goff[kIter];                     // offset in force for this thread's column of iteration i (its 64-column group)
#pragma unroll
    for (int i = 0; i < kIter; ++i) goff[i] = 0.f;
    const bool bad = lengths_invalid(p, b);
    const double negd = (double)kNeg;
    // new offsets of the complete, visible row `row` for block kc; ends with a barrier
    auto reoffset = [&](const double *row, const int kc) {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int u = tid + 1024 * i;
            if (u < Up) {
                float m = (float)row[u];
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
                if (lane == 0) gm[u >> 6] = (m > kNegTest) ? rintf(m) : NAN;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int u = tid + 1024 * i;
            if (u < Up) {
                const int g = u >> 6, gn = BETA ? min(g + 1, p.NG - 1) : max(g - 1, 0);
                const float own = gm[g], nb = gm[gn];
                goff[i] = (own == own) ? own : ((nb == nb) ? nb : goff[i]);
                if (lane == 0) offs[(size_t)kc * p.NG + g] = goff[i];
            }
        }
        __syncthreads();
    };
    auto store_row = [&](const double *row, const int n) {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int u = tid + 1024 * i;
            if (u < Up) out[(size_t)n * Up + u] = (row[u] > (double)kNegTest) ? (float)(row[u] - (double)goff[i]) : kNeg;
        }
    };
    if (!BETA) {
        for (int u = tid; u < Up; u += 1024) cur[u] = (u == 0) ? 0.0 : negd;
        __syncthreads();
        reoffset(cur, 0);
        store_row(cur, 0);
        for (int n = 1; n <= last; ++n) {  // diagonal n from diagonal n - 1 and the edge weights leaving diagonal n - 1
            const float2 *wrow = Wb + (size_t)(n - 1) * Up;
            for (int u = tid; u < Up; u += 1024) {
                const double stay = cur[u] + (double)wrow[u].x;                           // blank: (t-1, u) -> (t, u)
                const double emit = (u > 0) ? cur[u - 1] + (double)wrow[u - 1].y : negd;  // label: (t, u-1) -> (t, u)
                nxt[u] = lse2_pr(stay, emit);
            }
            __syncthreads();
            if ((n & (kRebase - 1)) == 0) reoffset(nxt, n / kRebase);
            store_row(nxt, n);
            double *t = cur;
            cur = nxt, nxt = t;
        }
        if (tid == 0) {
            const double ll2 = bad ? (double)NAN : cur[Ub - 1] + (double)Wb[(size_t)last * Up + Ub - 1].x;
            p.ll[2 * b] = ll2;
            p.costs[b] = (float)(-ll2 * 0.6931471805599453);
        }
    } else {
        for (int u = tid; u < Up; u += 1024) cur[u] = (u == Ub - 1) ? 0.0 : negd;  // the virtual terminal node (T_b, U_b - 1)
        __syncthreads();
        for (int n = last; n >= 0; --n) {  // diagonal n from diagonal n + 1 and the edge weights leaving diagonal n
            const float2 *wrow = Wb + (size_t)n * Up;
            for (int u = tid; u < Up; u += 1024) {
                const double stay = cur[u] + (double)wrow[u].x;
                const double emit = ((u + 1 < Up) ? cur[u + 1] : negd) + (double)wrow[u].y;
                nxt[u] = lse2_pr(stay, emit);
            }
            __syncthreads();
            if ((n & (kRebase - 1)) == kRebase - 1 || n == last) reoffset(nxt, n / kRebase);
            store_row(nxt, n);
            double *t = cur;
            cur = nxt, nxt = t;
        }
        if (tid == 0) {
            p.ll[2 * b + 1] = bad ? (double)NAN : cur[0];
            if (bad) p.costs[b] = NAN;
        }
    }
}

static hipError_t launch_sweep_wide(const LossParams &p, hipStream_t s) {
    const size_t shm = (size_t)2 * p.Up * sizeof(double) + (size_t)p.NG * sizeof(float);  // two diagonals in float64 + the groups' maxima
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
