/*
 * include/rnnt.h -- C ABI of libwarprnnt.so, the MI355X-native (gfx950) transducer-loss engine.
 *
 * This is the drop-in boundary for the reference's native op.  The reference
 * (noahchalifour/rnnt-speech-recognition) reaches its loss through
 *     utils/loss.py:6        from warprnnt_tensorflow import rnnt_loss
 *     utils/loss.py:34-35    rnnt_loss(y_pred, y_true, spec_lengths, label_lengths)
 * and builds the library behind it with scripts/build_rnnt.sh:1-13, which compiles the
 * warp-transducer submodule into `libwarprnnt.so` (cmake/warp-rnnt-cmakelist.txt:99,119) and
 * installs this header's namesake (cmake/warp-rnnt-cmakelist.txt:137: include/rnnt.h).  The
 * submodule source is absent from the reference tree, so the entry points below restate the
 * library's published C interface; each one names the reference-side
 * interface it replaces.
 *
 * Ownership: the caller owns every buffer.  The library allocates no device memory and never
 * synchronises the host; all work is enqueued on the caller's HIP stream.
 * Location: this library is device-only.  `loc == RNNT_CPU` is rejected with
 * RNNT_STATUS_INVALID_VALUE -- there is deliberately no CPU fallback inside the product.
 */
#ifndef MI355X_RNNT_H
#define MI355X_RNNT_H

#include <stdbool.h>
#include <stddef.h>

/* The library is built with -fvisibility=hidden: only the entry points declared here are exported. */
#if defined(__GNUC__)
#define RNNT_API __attribute__((visibility("default")))
#else
#define RNNT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque stand-in for hipStream_t so that C callers need no HIP headers.
 * Replaces the `CUstream stream` member of upstream's rnntOptions. */
typedef struct ihipStream_t *rnntStream_t;

/* Replaces upstream rnntStatus_t (include/rnnt.h of warp-transducer; installed by
 * cmake/warp-rnnt-cmakelist.txt:137).  Values keep upstream's order. */
typedef enum {
    RNNT_STATUS_SUCCESS = 0,
    RNNT_STATUS_MEMOPS_FAILED = 1,
    RNNT_STATUS_INVALID_VALUE = 2,
    RNNT_STATUS_EXECUTION_FAILED = 3,
    RNNT_STATUS_UNKNOWN_ERROR = 4
} rnntStatus_t;

/* Replaces upstream rnntComputeLocation. */
typedef enum { RNNT_CPU = 0, RNNT_GPU = 1 } rnntComputeLocation;

/* Replaces upstream rnntOptions (passed by value). */
typedef struct rnntOptions {
    rnntComputeLocation loc; /* must be RNNT_GPU */
    union {
        unsigned int num_threads; /* ignored (CPU location is not provided) */
        rnntStream_t stream;      /* HIP stream all kernels are enqueued on */
    };
    int blank_label; /* reference never passes it -> op default 0 (utils/vocabulary.py:3-6) */
    int maxT;        /* acts.shape[1] */
    int maxU;        /* acts.shape[2] = L_max + 1 (utils/preprocessing.py:177-183); at most 8192, see below */
    bool batch_first; /* must be true: acts is [B, maxT, maxU, V] row-major (1 byte, as upstream) */
} rnntOptions;

/* Replaces upstream get_warprnnt_version(). */
RNNT_API int get_warprnnt_version(void);

/* Replaces upstream rnntGetStatusString(). */
RNNT_API const char *rnntGetStatusString(rnntStatus_t status);

/* Replaces upstream get_workspace_size(maxT, maxU, minibatch, gpu, &size_bytes).
 * `gpu` must be true.  The size depends on (maxT, maxU, minibatch) only. */
RNNT_API rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t *size_bytes);

/* Deliberate limits of this library (upstream has none of them; all are reported as RNNT_STATUS_INVALID_VALUE at
 * enqueue time, never as wrong numbers):
 *   maxU <= 8192       up to 1024 the alpha/beta sweeps keep a whole anti-diagonal in the registers of ONE wavefront
 *                      (64 lanes x up to 16 lattice columns); longer label sequences take a plain multi-wave sweep with
 *                      the previous diagonal in LDS and a barrier per diagonal (same results, much slower per diagonal);
 *                      the fused joint entry points stop at maxU = 1024;
 *   B*maxT*maxU < 2^31 cell indices are 32-bit;
 *   workspace 256-byte aligned.
 * Numerics.  Small vocabularies (alphabet_size <= 60, 16-byte-aligned acts) on lattices of up to 1024 columns run on a
 * LINEAR-domain lattice: edge probabilities, alpha / beta as float32 mantissas times 2^(integer frame per sweep lane and block of
 * 4 or 8 diagonals), exact power-of-two renormalisation -- every rounding is RELATIVE, so the gradients come out 1e-7 ... 4e-6
 * from a float64 evaluation of the same logits (costs 1e-9 ... 5e-6 relative).  Mass that falls more than 126 bits below its
 * frame is flushed; whether that mattered is decided per lattice cell by the gradient pass (what a flush can have cost, times
 * the other side's mass, over the likelihood, must stay below 2^-40) and per utterance by the sweeps (likelihood zero /
 * non-finite, alpha-side vs beta-side likelihood, an edge probability below 2^-100).  An utterance that fails is redone by the
 * LOG-domain kernels, exact for any range, so results never depend on the shortcut: a TEAM of workgroups per utterance (round 5:
 * cell phases split over up to 16 CUs, alpha and beta side by side; one flagged utterance in a B=32 T=600 U=150 batch costs
 * +0.3 ms on a 0.23 ms step -- one workgroup per utterance took +2 ms -- and a batch in which EVERY utterance is handed back
 * 1.2 ms instead of 2.3).
 * N(0,1) logits and trained-like posteriors (one dominant symbol per cell along any monotone alignment) stay on the linear
 * lattice.  Unstructured logits of 4 x N(0,1) sit on the edge at that size (the sweeps shorten their frame blocks from 8 to 4
 * diagonals where the lsm pass saw the mass decay fast; some draws pass the certificate, others -- about a third of the
 * utterances of a batch -- are handed back); from about 5 x N(0,1) on every utterance is.  Larger vocabularies, unaligned
 * tensors, more than 1024 columns, the wide (640 < joint_size <= 704) and the f16 fused joints run on the log-domain kernels
 * throughout; the f32-grade fused joint (joint_dtype 0, joint_size <= 640) runs on the linear lattice since round 5, with the
 * same certificate and the same hand-back (from its parked logits).
 * The log-domain kernels keep alpha / beta as log2 values.  Wherever THIS op uses them (the hand-back, vocabularies above 60
 * symbols, unaligned tensors, more than 1024 columns) the recurrence is carried in float64 registers -- the log2(1 + 2^-|d|) term
 * of a log-add, in (0, 1], on the float32 units -- and only the stored lattice is float32 (residues against an integer offset per
 * block of 8 diagonals and sweep lane / group of 64 columns): a float32 recurrence rounds every log-add at the magnitude of its
 * residue, a random walk that reached 1e-4 ... 5e-4 of gradient error over the ~1,000-step paths of peaked or wide lattices
 * (rounds 1-3; tests/tools/emulate_sweep.py).  The wide f32-grade fused joint (joint_dtype 0, 640 < joint_size <= 704) and the
 * hand-back of the ordinary one use the float64 recurrence as well; the f16 joint (joint_dtype 1: binary16 roundings set its error)
 * keeps the float32 one up to 6 columns per lane (maxU <= 384).
 * Bars, against a float64 evaluation of the same logits, all tested with FIXED bars (tests/test_lin_gpu.py,
 * tests/test_peaky_gpu.py, tests/test_peaky_wide_gpu.py, tests/test_loss_gpu.py; measured values in profiles/r04_accuracy*.json):
 *   costs      within 1e-4 max(1, |cost|) everywhere (measured <= 6e-6);
 *   gradients  within 1e-4 absolute on every input: N(0,1), 4 x N(0,1), 8 x N(0,1) logits and trained-like posteriors, lattices
 *              of up to 8192 columns, every vocabulary size (measured: linear lattice 1e-7 ... 4e-6; log-domain paths 4e-6 at
 *              4 x N(0,1) and 1.1e-5 at 8 x N(0,1) at B=32 T=600 U=150 V=28, up to 4.4e-5 at 8 x N(0,1) on 1000-column lattices,
 *              5.5e-5 at T=1500 U=300 V=1024; 5,000 random shapes up to 1000 columns, half of them at 4 x N(0,1): 3.0e-5).
 * Vocabularies of more than 60 symbols (one lattice cell per wavefront): a cell whose occupancy alpha.beta/L is below 2^-50 gets
 * exactly zero gradients -- every gradient of a cell is bounded by 2 |cost_scale| x its occupancy -- and its logits are not read
 * (the reference's kernel leaves values around 1e-15 there).  The headline path (alphabet_size <= 60) visits every cell.
 * Out-of-range per-utterance lengths (T_b < 1, T_b > maxT, L_b < 0, L_b > maxU-1) are device data and cannot be
 * checked at enqueue time: the kernels clamp them into the tensor (no out-of-bounds access) and report that
 * utterance with a NaN cost and NaN gradients.  Labels outside [0, alphabet_size) are clamped into range. */

/* Replaces upstream compute_rnnt_loss(...): the body of the WarpRNNT TensorFlow op that
 * utils/loss.py:34-35 calls.
 *
 *   acts           device f32 [minibatch, maxT, maxU, alphabet_size]  RAW LOGITS (the log-softmax
 *                  is fused, as in upstream's GPU path; utils/loss.py:29-30 skips it on CUDA builds)
 *   grads          device f32, same shape, or NULL for score-only.  Receives d cost_b / d acts;
 *                  padded cells (t >= T_b or u > L_b) are written as exact zeros, so the caller
 *                  need not pre-zero it.
 *   flat_labels    device i32 [minibatch, maxU-1] (padded rows, as run_rnnt.py:262-263 passes them)
 *   label_lengths  device i32 [minibatch]   L_b
 *   input_lengths  device i32 [minibatch]   T_b (already divided by the time-reduction factor,
 *                  utils/loss.py:31-33)
 *   costs          device f32 [minibatch]   -ln P(y_b | x_b)
 *   workspace      device, >= get_workspace_size() bytes, 256-byte aligned
 *
 * Returns immediately after enqueueing; errors detected at enqueue time are returned, device
 * faults surface on the caller's next stream synchronisation. */
RNNT_API rnntStatus_t compute_rnnt_loss(const float *acts, float *grads, const int *flat_labels,
                               const int *label_lengths, const int *input_lengths,
                               int alphabet_size, int minibatch, float *costs, void *workspace,
                               rnntOptions options);

/* Build-only split of compute_rnnt_loss (no upstream counterpart).  The reference multiplies the
 * op's gradient by the upstream gradient afterwards (the TF binding's registered gradient;
 * run_rnnt.py:278 makes that factor 1/global_batch).  Splitting the call lets an autograd host
 * run the gradient pass only when backward is requested and fold that factor in for free:
 *   compute_rnnt_loss_fwd  = costs + lattice state in `workspace` (reads acts once)
 *   compute_rnnt_loss_bwd  = grads[b] = cost_scale[b] * d cost_b / d acts  (cost_scale NULL = 1),
 *                            from the SAME acts and the workspace left by _fwd.
 * compute_rnnt_loss(acts, grads, ...) == _fwd followed by _bwd(cost_scale = NULL).
 * _bwd may be called more than once per _fwd (an utterance the linear lattice handed back keeps its log-domain state for it).
 *   compute_rnnt_loss_ex   = compute_rnnt_loss with cost_scale folded into grads, in ONE call
 *                            (_fwd followed by _bwd on the caller's stream; nothing else differs). */
RNNT_API rnntStatus_t compute_rnnt_loss_fwd(const float *acts, const int *flat_labels,
                                   const int *label_lengths, const int *input_lengths,
                                   int alphabet_size, int minibatch, float *costs, void *workspace,
                                   rnntOptions options);

RNNT_API rnntStatus_t compute_rnnt_loss_bwd(const float *acts, float *grads, const int *flat_labels,
                                   const int *label_lengths, const int *input_lengths,
                                   const float *cost_scale, int alphabet_size, int minibatch,
                                   void *workspace, rnntOptions options);

RNNT_API rnntStatus_t compute_rnnt_loss_ex(const float *acts, float *grads, const int *flat_labels,
                                  const int *label_lengths, const int *input_lengths,
                                  const float *cost_scale, int alphabet_size, int minibatch,
                                  float *costs, void *workspace, rnntOptions options);

/* Build-only flag (no upstream counterpart): RNNT_VISIT_ALL switches the occupancy floor OFF -- the gradient kernels then visit
 * every lattice cell / row, as the reference's op and TensorFlow's autodiff do (run_rnnt.py:284), whatever the data.  Where the
 * floor applies (vocabularies above 60 symbols here; the backward of the fused joints below) a cell, or a lattice row of a
 * 32-column tile, whose occupancy alpha.beta/L is at most 2^-50 (the op) / 2^-40 (the fused joints: see get_rnnt_joint_backward_rows)
 * gets exact zeros without its logits being read: results differ from the all-visited ones by less than 2^-44 |cost_scale| per
 * element (the op; the fused joints skip only what is an exact zero already) and run times follow the width of the alignment band.
 * The floor hides no NaN: the forward pass reads every cell, a NaN logit makes the utterance's lattice, cost and occupancies NaN,
 * and a NaN occupancy counts as occupied (tests/test_loss_gpu.py::test_occupancy_floor_and_its_opt_out).  The flag is there for
 * parity debugging and for timing that does not depend on the data (bench.py reports both).
 *   compute_rnnt_loss_flags = compute_rnnt_loss_ex with `flags` (0 or RNNT_VISIT_ALL);  costs == NULL: the gradient pass alone
 *   (compute_rnnt_loss_bwd), grads == NULL: the forward alone.
 *   The fused-joint entry points take the same bit OR-ed into joint_dtype (joint_dtype | RNNT_VISIT_ALL). */
#define RNNT_VISIT_ALL 0x100
RNNT_API rnntStatus_t compute_rnnt_loss_flags(const float *acts, float *grads, const int *flat_labels,
                                     const int *label_lengths, const int *input_lengths,
                                     const float *cost_scale, int alphabet_size, int minibatch,
                                     float *costs, void *workspace, rnntOptions options, unsigned int flags);

/* ------------------------------------------------------------------------------------------
 * Build-only extension (no upstream counterpart): the joint network fused with the loss, so the
 * [B,T,U,J] and [B,T,U,V] tensors of model.py:158-166 are never materialised.
 *
 * The first Dense layer is factored exactly:  W1^T(e_t + p_u) + b1 = (W1^T e_t + b1) + W1^T p_u,
 * so the caller passes the two small projections
 *   enc_proj   device f32 [B, maxT, J]   = enc  @ W1 + b1   (model.py:162-163, bias folded here)
 *   pred_proj  device f32 [B, maxU, J]   = pred @ W1
 * and the kernels evaluate  logits[b,t,u,:] = tanh(enc_proj[b,t,:] + pred_proj[b,u,:]) @ W2 + b2
 * (model.py:162-166) tile by tile on the matrix cores.
 *
 *   W2 [J, V], b2 [V]            device f32 (model.py:165-166)
 *   cost_scale                   device f32 [B] or NULL (=1): upstream gradient of each cost, e.g.
 *                                1/global_batch (run_rnnt.py:278)
 *   d_enc_proj [B,maxT,J], d_pred_proj [B,maxU,J], dW2 [J,V], db2 [V]
 *                                device f32 outputs: gradients of sum_b cost_scale[b]*cost_b.
 *                                Fully overwritten.  May all be NULL for score-only.
 *   joint_dtype                  arithmetic of the J x V products.
 *                                0 = f32-grade products (operands split into binary16 hi + lo parts, three f16 MFMAs per product, f32
 *                                    accumulation).  W2 enters the products scaled by the power of two that puts max |W2| into
 *                                    [2^13, 2^14): any finite magnitude is taken (round 5; before, weights beyond binary16's
 *                                    65504 switched the call to plain f32 MFMA kernels), weights within 2^-13 of the largest
 *                                    keep 22 significand bits, smaller ones an absolute error of 2^-38 max |W2|.  Small
 *                                    vocabularies: alphabet_size <= 128 for joint_size a multiple of 64 up to 640 (vocabulary
 *                                    tiles of 32 symbols, one pass of the kernels per tile; <= 32 is the reference's character
 *                                    set), alphabet_size <= 32 for joint_size 704.
 *                                    The backward (joint_size <= 640) does not visit lattice rows, in tiles of 32 columns, whose
 *                                    cells all have an occupancy alpha.beta/L below 2^-40: those cells get exactly zero where the
 *                                    reference leaves ~1e-12 (get_rnnt_joint_backward_rows below: the bound, and how many rows a
 *                                    call visited); its run time follows the width of the alignment band.
 *                                1 = f16 MFMA, larger vocabularies: alphabet_size a multiple of 128 (128 ... 8192),
 *                                    joint_size a multiple of 128 (128 ... 640).  h = tanh(.) and W2 are rounded to binary16
 *                                    (round-to-nearest-even) before the products, accumulation is f32; the loss gradient
 *                                    w.r.t. the logits is scaled by 2^(14 - ceil(log2 max|cost_scale|)) and rounded to
 *                                    binary16 before dh = dl.W2^T and dW2 = h^T.dl (the backward skips lattice rows without mass,
 *                                    as joint_dtype 0 does -- groups of four rows of a 32-column tile here, below an occupancy of
 *                                    2^-40: every binary16 dlogits value of such a row is an exact zero already, the scaled
 *                                    gradient being below 2^-26 there; RNNT_VISIT_ALL switches that off).  The lattice (log-softmax, alpha,
 *                                    beta, costs) stays f32.  A forward pass that knows a backward pass follows (the
 *                                    one-call entry with gradients, or _fwd) PARKS the softmax numerators in the workspace:
 *                                    per (cell, 32-symbol chunk) 2^(x log2 e - R) rounded to binary16, R = the integer at or
 *                                    above the chunk's largest x log2 e (|R| <= 30000); the backward pass multiplies them back
 *                                    with one f32 factor per chunk, in place (one more binary16 rounding of occupancy x softmax;
 *                                    the blank and label columns come from the unrounded f32 edge logits).  A SECOND _bwd
 *                                    call on the same workspace finds the parked values consumed and recomputes the
 *                                    logits instead (one rounding less; results differ by binary16 rounding noise).
 *                                    Counterpart of the reference's `mixed_float16` policy (run_rnnt.py:96-99); oracle:
 *                                    oracle/rnnt_oracle.py joint_loss_and_grads_f16 (parked=True / False).
 *                                Either value may carry RNNT_VISIT_ALL (above): joint_dtype = 0 | RNNT_VISIT_ALL makes the backward
 *                                visit every lattice row.
 *                                Any other (joint_dtype, shape) combination returns RNNT_STATUS_INVALID_VALUE -- checked before
 *                                anything is enqueued, in every entry point that takes joint_dtype.
 *                                get_joint_workspace_size needs no dtype: where both arithmetic types take the shape (alphabet_size 128)
 *                                it returns the larger of the two layouts.
 * Both: maxU <= 1024; enc_proj, pred_proj (and b2 for joint_dtype 1) 16-byte aligned.
 * compute_rnnt_joint_loss      = costs and all four gradients in one call
 * compute_rnnt_joint_loss_fwd  = costs (+ the state a _bwd call needs, kept in `workspace`); for costs ONLY (evaluation) call
 *                                compute_rnnt_joint_loss with NULL gradient pointers -- it leaves nothing for a backward pass
 * compute_rnnt_joint_loss_bwd  = the gradients, from the same inputs and that workspace (autograd split,
 *                                as compute_rnnt_loss_fwd/_bwd); may be called more than once per _fwd
 */
RNNT_API rnntStatus_t get_joint_workspace_size(int maxT, int maxU, int minibatch, int joint_size,
                                      int alphabet_size, size_t *size_bytes);

RNNT_API rnntStatus_t compute_rnnt_joint_loss(const float *enc_proj, const float *pred_proj,
                                     const float *W2, const float *b2, const int *flat_labels,
                                     const int *label_lengths, const int *input_lengths,
                                     const float *cost_scale, int joint_size, int alphabet_size,
                                     int minibatch, float *costs, float *d_enc_proj,
                                     float *d_pred_proj, float *dW2, float *db2, int joint_dtype,
                                     void *workspace, rnntOptions options);

RNNT_API rnntStatus_t compute_rnnt_joint_loss_fwd(const float *enc_proj, const float *pred_proj,
                                         const float *W2, const float *b2, const int *flat_labels,
                                         const int *label_lengths, const int *input_lengths,
                                         int joint_size, int alphabet_size, int minibatch,
                                         float *costs, int joint_dtype, void *workspace,
                                         rnntOptions options);

RNNT_API rnntStatus_t compute_rnnt_joint_loss_bwd(const float *enc_proj, const float *pred_proj,
                                         const float *W2, const float *b2, const int *flat_labels,
                                         const int *label_lengths, const int *input_lengths,
                                         const float *cost_scale, int joint_size, int alphabet_size,
                                         int minibatch, float *d_enc_proj, float *d_pred_proj,
                                         float *dW2, float *db2, int joint_dtype, void *workspace,
                                         rnntOptions options);

/* Build-only extension: the WHOLE joint network of model.py:158-166 fused with the loss -- the first Dense layer
 * (model.py:162-163) and its backward run inside the library too, on the matrix cores, f32-grade:
 *   enc   device f32 [B, maxT, H]  encoder output        pred  device f32 [B, maxU, H]  prediction-network output
 *   W1 [H, J], b1 [J]   (Keras Dense kernel / bias, model.py:162-163)          W2 [J, V], b2 [V]  (model.py:165-166)
 *   logits[b,t,u,:] = tanh((enc[b,t,:] + pred[b,u,:]) @ W1 + b1) @ W2 + b2, never materialised; the layer is factored exactly
 *   as (enc @ W1 + b1) + pred @ W1 (three GEMMs over B (maxT + maxU) rows with their operands split into binary16 hi + lo
 *   parts, as for joint_dtype 0: csrc/dense_kernels.hip) and the rest is compute_rnnt_joint_loss on those projections.
 *   d_enc [B,maxT,H], d_pred [B,maxU,H], dW1 [H,J], db1 [J], dW2 [J,V], db2 [V]: gradients of sum_b cost_scale[b] * cost_b
 *   (all six or none; fully overwritten; padded frames / label positions get exact zeros in d_enc / d_pred).
 * hidden_size a multiple of 32, joint_size a multiple of 64 (both <= 4096), then the (joint_size, alphabet_size, joint_dtype)
 * rules of compute_rnnt_joint_loss; enc, pred, W1, b1 and the four first-layer gradients (d_enc, d_pred, dW1, db1) 16-byte aligned.  workspace: get_joint_net_workspace_size() bytes, 256-byte
 * aligned; _bwd needs the workspace left by _fwd of the same inputs (projections and operand images live there). */
RNNT_API rnntStatus_t get_joint_net_workspace_size(int maxT, int maxU, int minibatch, int hidden_size,
                                                   int joint_size, int alphabet_size, size_t *size_bytes);

RNNT_API rnntStatus_t compute_rnnt_joint_net_loss(const float *enc, const float *pred, const float *W1,
                                                  const float *b1, const float *W2, const float *b2,
                                                  const int *flat_labels, const int *label_lengths,
                                                  const int *input_lengths, const float *cost_scale,
                                                  int hidden_size, int joint_size, int alphabet_size,
                                                  int minibatch, float *costs, float *d_enc, float *d_pred,
                                                  float *dW1, float *db1, float *dW2, float *db2,
                                                  int joint_dtype, void *workspace, rnntOptions options);

RNNT_API rnntStatus_t compute_rnnt_joint_net_loss_fwd(const float *enc, const float *pred, const float *W1,
                                                      const float *b1, const float *W2, const float *b2,
                                                      const int *flat_labels, const int *label_lengths,
                                                      const int *input_lengths, int hidden_size,
                                                      int joint_size, int alphabet_size, int minibatch,
                                                      float *costs, int joint_dtype, void *workspace,
                                                      rnntOptions options);

RNNT_API rnntStatus_t compute_rnnt_joint_net_loss_bwd(const float *enc, const float *pred, const float *W1,
                                                      const float *b1, const float *W2, const float *b2,
                                                      const int *flat_labels, const int *label_lengths,
                                                      const int *input_lengths, const float *cost_scale,
                                                      int hidden_size, int joint_size, int alphabet_size,
                                                      int minibatch, float *d_enc, float *d_pred, float *dW1,
                                                      float *db1, float *dW2, float *db2, int joint_dtype,
                                                      void *workspace, rnntOptions options);

/* Build-only extension: the joint network alone, for decoding.  Replaces `joint(model, f, g)` of the reference's greedy
 * decoder (utils/decoding.py:6-18: dense_1 (tanh) and dense_2 on f + g for one lattice cell per step; called at :63-69):
 *   logits[b,t,u,:] = tanh(enc_proj[b,t,:] + pred_proj[b,u,:]) @ W2 + b2        device f32 [minibatch, maxT, maxU, alphabet_size]
 * with enc_proj / pred_proj as for compute_rnnt_joint_loss (the first Dense layer factored, bias folded into enc_proj).
 * It runs the forward kernels of compute_rnnt_joint_loss with the same joint_dtype on a lattice whose every cell is live, so a
 * decoder sees the logits the loss was trained on:
 *   joint_dtype 0  f32-grade split-precision products (the same power-of-two scale of W2): alphabet_size <= 32,
 *                  joint_size a multiple of 64 (<= 704);
 *   joint_dtype 1  operands rounded to binary16, f32 accumulation (the reference's default vocabulary of 4096 word pieces,
 *                  hparams.py:4): alphabet_size a multiple of 128 (128 ... 8192), joint_size a multiple of 128 (128 ... 640); logits 16-byte aligned.
 * maxU <= 1024; a greedy decoder calls it with maxT = maxU = 1 and minibatch = the number of hypotheses.
 * workspace: get_joint_workspace_size(maxT, maxU, minibatch, joint_size, alphabet_size) bytes, 256-byte aligned. */
RNNT_API rnntStatus_t compute_rnnt_joint_logits(const float *enc_proj, const float *pred_proj,
                                                const float *W2, const float *b2, int joint_size,
                                                int alphabet_size, int minibatch, float *logits,
                                                int joint_dtype, void *workspace, rnntOptions options);

/* Diagnostics of the fused joints' backward (joint_dtype 0 at joint_size <= 640; joint_dtype 1 since round 6): how many lattice rows x 32-column tiles the
 * LAST backward on this workspace visited (rows[0]) out of those inside the utterances (rows[1]).  The backward skips a row of a
 * tile when none of its 32 cells has an occupancy alpha.beta/L above 2^-40: every dlogits value of a cell is bounded by
 * 2 |cost_scale| x that occupancy, and both joints hand dlogits to their products as binary16 parts of (power of two <= 2^13 / |cost_scale|) x
 * dlogits -- below 2^-26 in such a row, i.e. exact zeros already: the row adds nothing, its cells get exactly zero where
 * the reference leaves 1e-12's.  How many rows that is depends on the data (unstructured N(0,1) logits on a 600 x 150 lattice: about
 * half; a trained model: most).  Synchronises options.stream.  rows = {-1, -1} where nothing is skipped (the wide joint, 640 < joint_size)
 * and where the workspace does not hold the counts of a backward of THIS shape (fresh, or used by another shape since: the row plan stamps them).
 * The work of a backward call is divided among the workgroups by these counts, deterministically. */
RNNT_API rnntStatus_t get_rnnt_joint_backward_rows(void *workspace, int joint_size, int alphabet_size, int minibatch,
                                                   rnntOptions options, int rows[2]);

/* The same from the encoder / prediction-network outputs: the first Dense layer (model.py:162-163, utils/decoding.py:8,15) runs
 * in the library too (the split-precision GEMMs of compute_rnnt_joint_net_loss), then the joint as above.
 *   enc [minibatch, maxT, hidden_size], pred [minibatch, maxU, hidden_size], W1 [hidden_size, joint_size], b1 [joint_size]
 *   (16-byte aligned; hidden_size a multiple of 32).  workspace: get_joint_net_workspace_size(...) bytes. */
RNNT_API rnntStatus_t compute_rnnt_joint_net_logits(const float *enc, const float *pred, const float *W1, const float *b1,
                                                    const float *W2, const float *b2, int hidden_size, int joint_size,
                                                    int alphabet_size, int minibatch, float *logits, int joint_dtype,
                                                    void *workspace, rnntOptions options);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_RNNT_H */
