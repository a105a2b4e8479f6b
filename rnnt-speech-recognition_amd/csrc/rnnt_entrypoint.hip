// rnnt_entrypoint.hip -- the extern "C" boundary of libwarprnnt.so (declared in include/rnnt.h).
//
// Replaces the reference's native entry points (warp-transducer `src/rnnt_entrypoint.cu`, named at
// cmake/warp-rnnt-cmakelist.txt:99; reached from utils/loss.py:34-35).  Argument validation follows
// the published contract: null pointers / non-positive sizes ->
// RNNT_STATUS_INVALID_VALUE; nothing is allocated; everything is enqueued on the caller's stream.
#include "../../include/rnnt.h"
#include "rnnt_common.h"
#include "rnnt_lin.h"

using namespace rnnt;

namespace rnnt {
// joint_kernels.hip
hipError_t joint_workspace_bytes(int T, int U, int B, int J, int V, int joint_dtype, size_t *bytes);
hipError_t joint_f16_backward_rows(void *workspace, int T, int U, int B, int J, int V, int rows[2], hipStream_t s);
hipError_t joint_backward_rows(void *workspace, int T, int U, int B, int J, int V, int rows[2], hipStream_t s);
hipError_t launch_joint_loss(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                             const int *labels, const int *label_lengths, const int *input_lengths,
                             const float *cost_scale, int J, int V, int B, int T, int U, int blank, float *costs,
                             float *d_enc_proj, float *d_pred_proj, float *dW2, float *db2, int joint_dtype,
                             int phases, void *workspace, hipStream_t s, const JointHooks *hooks);
hipError_t joint_aux_pointers(void *workspace, int T, int U, int B, int J, int V, float **expE, float **expP, float **tflag);
hipError_t launch_joint_prefill(void *workspace, int T, int U, int B, int J, int V, hipStream_t s);
bool joint_dtype_supported(int joint_dtype, int J, int V);
// dense_kernels.hip (the joint's first Dense layer)
bool dense_supported(int H, int J);
hipError_t dense_workspace_bytes(int B, int T, int U, int H, int J, size_t base, size_t *bytes);
void dense_proj_pointers(void *workspace, int B, int T, int U, int H, int J, size_t base, float **enc_proj, float **pred_proj,
                         float **d_enc_proj, float **d_pred_proj);
void dense_hook_pointers(void *workspace, int B, int T, int U, int H, int J, size_t base, unsigned **dmax_enc, unsigned **dmax_pred);
hipError_t launch_dense_fwd(const float *enc, const float *pred, const float *W1, const float *b1, int B, int T, int U, int H, int J,
                            void *workspace, size_t base, float *expE, float *expP, float *tflag, hipStream_t s);
hipError_t launch_dense_bwd(int B, int T, int U, int H, int J, float *d_enc, float *d_pred, float *dW1, float *db1, void *workspace,
                            size_t base, hipStream_t s);
hipError_t launch_joint_logits(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2, int J, int V,
                               int B, int T, int U, float *logits, void *workspace, hipStream_t s);
// joint_f16_kernels.hip
hipError_t joint_f16_workspace_bytes(int T, int U, int B, int J, int V, size_t *bytes);
hipError_t launch_joint_logits_f16(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2, int J, int V,
                                   int B, int T, int U, float *logits, void *workspace, hipStream_t s);
}  // namespace rnnt

static rnntStatus_t check_options(const rnntOptions &o) {
    if (o.loc != RNNT_GPU) return RNNT_STATUS_INVALID_VALUE;  // device-only library: no CPU fallback
    if (!o.batch_first) return RNNT_STATUS_INVALID_VALUE;
    if (o.maxT <= 0 || o.maxU <= 0 || o.blank_label < 0) return RNNT_STATUS_INVALID_VALUE;
    if (o.maxU > kMaxU) return RNNT_STATUS_INVALID_VALUE;  // the wide sweep keeps two diagonals in LDS (include/rnnt.h)
    return RNNT_STATUS_SUCCESS;
}

static bool fill_params(LossParams &p, const float *acts, float *grads, const int *labels,
                        const int *label_lengths, const int *input_lengths, const float *cost_scale, int V,
                        int B, float *costs, void *workspace, const rnntOptions &o) {
    const long long cells = (long long)B * o.maxT * o.maxU;
    if (cells <= 0 || cells >= (1ll << 31)) return false;
    if (((uintptr_t)workspace & 255) != 0) return false;
    const WsLayout w = make_layout(o.maxT, o.maxU, B);
    char *ws = (char *)workspace;
    p.acts = acts;
    p.grads = grads;
    p.labels = labels;
    p.label_lengths = label_lengths;
    p.input_lengths = input_lengths;
    p.cost_scale = cost_scale;
    p.costs = costs;
    p.lse = (float *)(ws + w.lse);
    p.W = (float *)(ws + w.W);
    p.A = (float *)(ws + w.A);
    p.Bt = (float *)(ws + w.Bt);
    p.offA = (float *)(ws + w.offA);
    p.offB = (float *)(ws + w.offB);
    p.ll = (double *)(ws + w.ll);
    p.EA = (int *)(ws + w.EA);
    p.EB = (int *)(ws + w.EB);
    p.lik = (float *)(ws + w.lik);
    p.flags = (int *)(ws + w.flags);
    p.NCl = w.NCl;
    p.pstat = (float2 *)(ws + w.pstat);
    p.lshift = (int *)(ws + w.lshift);
    p.bar = (int *)(ws + w.bar);
    p.nPstat = w.nPstat;
    p.pstatStride = 4;
    p.B = B, p.T = o.maxT, p.U = o.maxU, p.V = V, p.blank = o.blank_label;
    p.b0 = 0, p.nb = B;
    p.precise = 0;
    p.visit_all = 0;
    p.tile = make_tile(o.maxT, o.maxU, V);
    p.N = w.N, p.Nr = w.Nr, p.Up = w.Up, p.NC = w.NC, p.NG = w.NG;
    p.cells = (uint32_t)cells;
    p.divU = make_fastdiv((uint32_t)o.maxU);
    p.divT = make_fastdiv((uint32_t)o.maxT);
    p.divV = make_fastdiv((uint32_t)V);
    p.divOG = make_fastdiv((uint32_t)w.OG);
    return true;
}

namespace rnnt {
bool fill_loss_params(LossParams &p, const float *acts, float *grads, const int *labels, const int *label_lengths,
                      const int *input_lengths, const float *cost_scale, int V, int B, float *costs, void *workspace,
                      int maxT, int maxU, int blank) {
    rnntOptions o;
    o.loc = RNNT_GPU, o.stream = nullptr, o.blank_label = blank, o.maxT = maxT, o.maxU = maxU, o.batch_first = true;
    return fill_params(p, acts, grads, labels, label_lengths, input_lengths, cost_scale, V, B, costs, workspace, o);
}
}  // namespace rnnt

static rnntStatus_t from_hip(hipError_t e) {
    if (e == hipSuccess) return RNNT_STATUS_SUCCESS;
    if (e == hipErrorInvalidValue) return RNNT_STATUS_INVALID_VALUE;
    return RNNT_STATUS_EXECUTION_FAILED;
}

extern "C" {

int get_warprnnt_version(void) { return 1; }

const char *rnntGetStatusString(rnntStatus_t status) {
    switch (status) {
        case RNNT_STATUS_SUCCESS: return "no error";
        case RNNT_STATUS_MEMOPS_FAILED: return "hip memcpy or memset failed";
        case RNNT_STATUS_INVALID_VALUE: return "invalid value";
        case RNNT_STATUS_EXECUTION_FAILED: return "execution failed";
        case RNNT_STATUS_UNKNOWN_ERROR:
        default: return "unknown error";
    }
}

rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t *size_bytes) {
    if (!size_bytes || maxT <= 0 || maxU <= 0 || minibatch <= 0 || !gpu) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = make_layout(maxT, maxU, minibatch).total;
    return RNNT_STATUS_SUCCESS;
}

// fill + lsm + sweeps on the caller's stream (stream order is the only dependency between the stages)
static rnntStatus_t run_forward(LossParams &p, const WsLayout &w, hipStream_t s) {
    if (lin_path_ok(p)) {  // small vocabulary, <= 256 lattice columns: the linear-domain lattice (rnnt_lin.h)
        hipError_t e = launch_lsm_lin(p, s);
        if (e != hipSuccess) return from_hip(e);
        return from_hip(launch_sweeps_lin(p, s));
    }
    // the patch kernels write the log-zero part of W themselves; the wave-per-cell kernels (large or unaligned vocabularies)
    // rely on a pre-filled W
    if (!tile_path_ok(p, false) && launch_fill(p.W, kFillByte, w.A - w.W, s) != hipSuccess) return RNNT_STATUS_MEMOPS_FAILED;
    p.precise = 1;  // the op's contract is 1e-4 on every input: float64 recurrence (the fused joints keep the float32 one up to 6 columns per lane)
    hipError_t e = launch_lsm(p, s);
    if (e != hipSuccess) return from_hip(e);
    return from_hip(launch_sweeps(p, s));
}

static rnntStatus_t validate(const void *acts, const void *labels, const void *ll, const void *il, const void *ws,
                      int V, int B, const rnntOptions &o) {
    if (!acts || !labels || !ll || !il || !ws) return RNNT_STATUS_INVALID_VALUE;
    if (V <= 0 || B <= 0) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(o);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (o.blank_label >= V) return RNNT_STATUS_INVALID_VALUE;
    return RNNT_STATUS_SUCCESS;
}

// Build-only split of compute_rnnt_loss so that an autograd caller can delay the gradient pass
// until the upstream gradient (run_rnnt.py:278: 1/global_batch) is known, and fold it in for free.
rnntStatus_t compute_rnnt_loss_fwd(const float *acts, const int *flat_labels, const int *label_lengths,
                                   const int *input_lengths, int alphabet_size, int minibatch, float *costs,
                                   void *workspace, rnntOptions options) {
    if (!costs) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = validate(acts, flat_labels, label_lengths, input_lengths, workspace, alphabet_size, minibatch,
                               options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    LossParams p;
    if (!fill_params(p, acts, nullptr, flat_labels, label_lengths, input_lengths, nullptr, alphabet_size,
                     minibatch, costs, workspace, options))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t s = (hipStream_t)options.stream;
    const WsLayout w = make_layout(options.maxT, options.maxU, minibatch);
    st = run_forward(p, w, s);
    if (st != RNNT_STATUS_SUCCESS || !lin_path_ok(p)) return st;
    return from_hip(launch_redo_lin(p, false, s));  // utterances the linear lattice handed back: log-domain sweeps (costs)
}

// The gradient pass.  On the linear path it ends with the hand-back launch (rnnt_lin_kernels.hip); a gradient buffer the
// patch kernels cannot write (not 16-byte aligned) sends every utterance through that launch.
static rnntStatus_t run_backward(LossParams &p, hipStream_t s) {
    if (!lin_path_ok(p)) return from_hip(launch_grad(p, s));
    const bool patch = tile_path_ok(p, true);
    if (patch) {
        hipError_t e = launch_grad_lin(p, s);
        if (e != hipSuccess) return from_hip(e);
    }
    return from_hip(launch_redo_lin(p, !patch, s));
}

rnntStatus_t compute_rnnt_loss_bwd(const float *acts, float *grads, const int *flat_labels,
                                   const int *label_lengths, const int *input_lengths, const float *cost_scale,
                                   int alphabet_size, int minibatch, void *workspace, rnntOptions options) {
    if (!grads) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = validate(acts, flat_labels, label_lengths, input_lengths, workspace, alphabet_size, minibatch,
                               options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    LossParams p;
    if (!fill_params(p, acts, grads, flat_labels, label_lengths, input_lengths, cost_scale, alphabet_size,
                     minibatch, nullptr, workspace, options))
        return RNNT_STATUS_INVALID_VALUE;
    return run_backward(p, (hipStream_t)options.stream);
}

// compute_rnnt_loss with the upstream gradient folded in (cost_scale NULL = 1) and the build-only flags (include/rnnt.h):
// costs == NULL = the gradient pass alone (compute_rnnt_loss_bwd), grads == NULL = the forward alone.
rnntStatus_t compute_rnnt_loss_flags(const float *acts, float *grads, const int *flat_labels,
                                     const int *label_lengths, const int *input_lengths, const float *cost_scale,
                                     int alphabet_size, int minibatch, float *costs, void *workspace,
                                     rnntOptions options, unsigned int flags) {
    if (flags & ~(unsigned)RNNT_VISIT_ALL) return RNNT_STATUS_INVALID_VALUE;
    if (!grads)
        return compute_rnnt_loss_fwd(acts, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                     costs, workspace, options);
    rnntStatus_t st = validate(acts, flat_labels, label_lengths, input_lengths, workspace, alphabet_size, minibatch,
                               options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    LossParams p;
    if (!fill_params(p, acts, grads, flat_labels, label_lengths, input_lengths, cost_scale, alphabet_size,
                     minibatch, costs, workspace, options))
        return RNNT_STATUS_INVALID_VALUE;
    p.visit_all = (flags & RNNT_VISIT_ALL) ? 1 : 0;
    hipStream_t s = (hipStream_t)options.stream;
    if (costs) {
        const WsLayout w = make_layout(options.maxT, options.maxU, minibatch);
        st = run_forward(p, w, s);
        if (st != RNNT_STATUS_SUCCESS) return st;
    }
    return run_backward(p, s);
}

rnntStatus_t compute_rnnt_loss_ex(const float *acts, float *grads, const int *flat_labels,
                                  const int *label_lengths, const int *input_lengths, const float *cost_scale,
                                  int alphabet_size, int minibatch, float *costs, void *workspace,
                                  rnntOptions options) {
    if (grads && !costs) return RNNT_STATUS_INVALID_VALUE;
    return compute_rnnt_loss_flags(acts, grads, flat_labels, label_lengths, input_lengths, cost_scale, alphabet_size, minibatch,
                                   costs, workspace, options, 0u);
}

rnntStatus_t compute_rnnt_loss(const float *acts, float *grads, const int *flat_labels,
                               const int *label_lengths, const int *input_lengths, int alphabet_size,
                               int minibatch, float *costs, void *workspace, rnntOptions options) {
    return compute_rnnt_loss_ex(acts, grads, flat_labels, label_lengths, input_lengths, nullptr, alphabet_size,
                                minibatch, costs, workspace, options);
}

rnntStatus_t get_joint_workspace_size(int maxT, int maxU, int minibatch, int joint_size, int alphabet_size,
                                      size_t *size_bytes) {
    if (!size_bytes || maxT <= 0 || maxU <= 0 || minibatch <= 0 || joint_size <= 0 || alphabet_size <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    return from_hip(joint_workspace_bytes(maxT, maxU, minibatch, joint_size, alphabet_size, -1, size_bytes));
}

static rnntStatus_t joint_call(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                               const int *flat_labels, const int *label_lengths, const int *input_lengths,
                               const float *cost_scale, int joint_size, int alphabet_size, int minibatch, float *costs,
                               float *d_enc_proj, float *d_pred_proj, float *dW2, float *db2, int joint_dtype,
                               int phases, void *workspace, const rnntOptions &options) {
    if (!enc_proj || !pred_proj || !W2 || !b2 || !flat_labels || !label_lengths || !input_lengths || !workspace)
        return RNNT_STATUS_INVALID_VALUE;
    if ((phases & 1) && !costs) return RNNT_STATUS_INVALID_VALUE;
    if (joint_size <= 0 || alphabet_size <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (options.blank_label >= alphabet_size) return RNNT_STATUS_INVALID_VALUE;
    if (options.maxU > 1024) return RNNT_STATUS_INVALID_VALUE;  // the fused joint paths are built on the register-resident sweeps
    if (joint_dtype & ~(0xff | RNNT_VISIT_ALL)) return RNNT_STATUS_INVALID_VALUE;
    if (joint_dtype & RNNT_VISIT_ALL) phases |= 8;  // (launch_joint_loss: no occupancy floor in the backward)
    joint_dtype &= 0xff;
    if (!joint_dtype_supported(joint_dtype, joint_size, alphabet_size)) return RNNT_STATUS_INVALID_VALUE;
    const bool any_grad = d_enc_proj || d_pred_proj || dW2 || db2;
    if (any_grad && !(d_enc_proj && d_pred_proj && dW2 && db2)) return RNNT_STATUS_INVALID_VALUE;
    if ((phases & 2) && !(phases & 1) && !any_grad) return RNNT_STATUS_INVALID_VALUE;
    return from_hip(launch_joint_loss(enc_proj, pred_proj, W2, b2, flat_labels, label_lengths, input_lengths,
                                      cost_scale, joint_size, alphabet_size, minibatch, options.maxT, options.maxU,
                                      options.blank_label, costs, d_enc_proj, d_pred_proj, dW2, db2, joint_dtype,
                                      phases, workspace, (hipStream_t)options.stream, nullptr));
}

rnntStatus_t compute_rnnt_joint_loss(const float *enc_proj, const float *pred_proj, const float *W2,
                                     const float *b2, const int *flat_labels, const int *label_lengths,
                                     const int *input_lengths, const float *cost_scale, int joint_size,
                                     int alphabet_size, int minibatch, float *costs, float *d_enc_proj,
                                     float *d_pred_proj, float *dW2, float *db2, int joint_dtype, void *workspace,
                                     rnntOptions options) {
    return joint_call(enc_proj, pred_proj, W2, b2, flat_labels, label_lengths, input_lengths, cost_scale, joint_size,
                      alphabet_size, minibatch, costs, d_enc_proj, d_pred_proj, dW2, db2, joint_dtype, 3, workspace,
                      options);
}

rnntStatus_t compute_rnnt_joint_loss_fwd(const float *enc_proj, const float *pred_proj, const float *W2,
                                         const float *b2, const int *flat_labels, const int *label_lengths,
                                         const int *input_lengths, int joint_size, int alphabet_size, int minibatch,
                                         float *costs, int joint_dtype, void *workspace, rnntOptions options) {
    return joint_call(enc_proj, pred_proj, W2, b2, flat_labels, label_lengths, input_lengths, nullptr, joint_size,
                      alphabet_size, minibatch, costs, nullptr, nullptr, nullptr, nullptr, joint_dtype, 1 | 4, workspace,
                      options);  // bit 2: a backward-only call follows (the forward leaves what that call needs)
}

rnntStatus_t compute_rnnt_joint_loss_bwd(const float *enc_proj, const float *pred_proj, const float *W2,
                                         const float *b2, const int *flat_labels, const int *label_lengths,
                                         const int *input_lengths, const float *cost_scale, int joint_size,
                                         int alphabet_size, int minibatch, float *d_enc_proj, float *d_pred_proj,
                                         float *dW2, float *db2, int joint_dtype, void *workspace,
                                         rnntOptions options) {
    if (!d_enc_proj) return RNNT_STATUS_INVALID_VALUE;
    return joint_call(enc_proj, pred_proj, W2, b2, flat_labels, label_lengths, input_lengths, cost_scale, joint_size,
                      alphabet_size, minibatch, nullptr, d_enc_proj, d_pred_proj, dW2, db2, joint_dtype, 2, workspace,
                      options);
}

// ---- the whole joint network (first Dense layer included) fused with the loss ----
rnntStatus_t get_joint_net_workspace_size(int maxT, int maxU, int minibatch, int hidden_size, int joint_size, int alphabet_size,
                                          size_t *size_bytes) {
    if (!size_bytes || maxT <= 0 || maxU <= 0 || minibatch <= 0 || hidden_size <= 0 || joint_size <= 0 || alphabet_size <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    size_t base = 0;
    hipError_t e = joint_workspace_bytes(maxT, maxU, minibatch, joint_size, alphabet_size, -1, &base);
    if (e != hipSuccess) return from_hip(e);
    return from_hip(dense_workspace_bytes(minibatch, maxT, maxU, hidden_size, joint_size, base, size_bytes));
}

static rnntStatus_t joint_net_call(const float *enc, const float *pred, const float *W1, const float *b1, const float *W2,
                                   const float *b2, const int *flat_labels, const int *label_lengths, const int *input_lengths,
                                   const float *cost_scale, int hidden_size, int joint_size, int alphabet_size, int minibatch,
                                   float *costs, float *d_enc, float *d_pred, float *dW1, float *db1, float *dW2, float *db2,
                                   int joint_dtype, int phases, void *workspace, const rnntOptions &options) {
    if (!enc || !pred || !W1 || !b1 || !W2 || !b2 || !flat_labels || !label_lengths || !input_lengths || !workspace)
        return RNNT_STATUS_INVALID_VALUE;
    if ((phases & 1) && !costs) return RNNT_STATUS_INVALID_VALUE;
    if (hidden_size <= 0 || joint_size <= 0 || alphabet_size <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (options.blank_label >= alphabet_size || options.maxU > 1024) return RNNT_STATUS_INVALID_VALUE;
    if (((uintptr_t)workspace & 255) != 0 || !dense_supported(hidden_size, joint_size)) return RNNT_STATUS_INVALID_VALUE;
    const bool any_grad = d_enc || d_pred || dW1 || db1 || dW2 || db2;
    if (any_grad && !(d_enc && d_pred && dW1 && db1 && dW2 && db2)) return RNNT_STATUS_INVALID_VALUE;
    if (any_grad && ((((uintptr_t)dW1 | (uintptr_t)db1 | (uintptr_t)d_enc | (uintptr_t)d_pred) & 15) != 0)) return RNNT_STATUS_INVALID_VALUE;  // 16-byte stores
    if ((((uintptr_t)enc | (uintptr_t)pred | (uintptr_t)W1 | (uintptr_t)b1) & 15) != 0) return RNNT_STATUS_INVALID_VALUE;
    if ((phases & 2) && !(phases & 1) && !any_grad) return RNNT_STATUS_INVALID_VALUE;
    if (joint_dtype & ~(0xff | RNNT_VISIT_ALL)) return RNNT_STATUS_INVALID_VALUE;
    if (joint_dtype & RNNT_VISIT_ALL) phases |= 8;  // (launch_joint_loss: no occupancy floor in the backward)
    joint_dtype &= 0xff;
    if (!joint_dtype_supported(joint_dtype, joint_size, alphabet_size)) return RNNT_STATUS_INVALID_VALUE;  // before anything is enqueued
    const int B = minibatch, T = options.maxT, U = options.maxU;
    hipStream_t s = (hipStream_t)options.stream;
    size_t base = 0;
    hipError_t e = joint_workspace_bytes(T, U, B, joint_size, alphabet_size, -1, &base);
    if (e != hipSuccess) return from_hip(e);
    // What the dense layer does on the way for the fused joint (JointHooks): with the f32-grade joint, the forward GEMM's
    // epilogue writes the e^{2x} tables and the table-range flag (the prep kernel then only builds the W2 images), the
    // backward-only call reuses the forward's workspace state; the reductions that produce d enc_proj / d pred_proj leave
    // their per-block abs-max entries for the backward GEMMs' operand scales.
    JointHooks hooks;
    hooks.prep_mode = 0;
    hooks.prefilled = 0;
    dense_hook_pointers(workspace, B, T, U, hidden_size, joint_size, base, &hooks.dmax_enc, &hooks.dmax_pred);
    float *expE = nullptr, *expP = nullptr, *tflag = nullptr;
    if (joint_dtype == 0) {
        if ((e = joint_aux_pointers(workspace, T, U, B, joint_size, alphabet_size, &expE, &expP, &tflag)) != hipSuccess) return from_hip(e);
        hooks.prep_mode = (phases & 1) ? 1 : 2;
    }
    if (phases & 1) {
        if (tflag) {  // the joint's edge-array pre-fill and its flag words (the GEMM epilogue raises one of them) in ONE launch
            if (launch_joint_prefill(workspace, T, U, B, joint_size, alphabet_size, s) != hipSuccess) return RNNT_STATUS_MEMOPS_FAILED;
            hooks.prefilled = 1;
        }
        if ((e = launch_dense_fwd(enc, pred, W1, b1, B, T, U, hidden_size, joint_size, workspace, base, expE, expP, tflag, s)) != hipSuccess)
            return from_hip(e);
    }
    float *ep, *pp, *dep, *dpp;
    dense_proj_pointers(workspace, B, T, U, hidden_size, joint_size, base, &ep, &pp, &dep, &dpp);
    const bool bwd = (phases & 2) && any_grad;
    e = launch_joint_loss(ep, pp, W2, b2, flat_labels, label_lengths, input_lengths, cost_scale, joint_size, alphabet_size, B, T, U,
                          options.blank_label, costs, bwd ? dep : nullptr, bwd ? dpp : nullptr, bwd ? dW2 : nullptr,
                          bwd ? db2 : nullptr, joint_dtype, phases, workspace, s, &hooks);
    if (e != hipSuccess || !bwd) return from_hip(e);
    return from_hip(launch_dense_bwd(B, T, U, hidden_size, joint_size, d_enc, d_pred, dW1, db1, workspace, base, s));
}

rnntStatus_t compute_rnnt_joint_net_loss(const float *enc, const float *pred, const float *W1, const float *b1, const float *W2,
                                         const float *b2, const int *flat_labels, const int *label_lengths,
                                         const int *input_lengths, const float *cost_scale, int hidden_size, int joint_size,
                                         int alphabet_size, int minibatch, float *costs, float *d_enc, float *d_pred, float *dW1,
                                         float *db1, float *dW2, float *db2, int joint_dtype, void *workspace, rnntOptions options) {
    return joint_net_call(enc, pred, W1, b1, W2, b2, flat_labels, label_lengths, input_lengths, cost_scale, hidden_size, joint_size,
                          alphabet_size, minibatch, costs, d_enc, d_pred, dW1, db1, dW2, db2, joint_dtype, 3, workspace, options);
}

rnntStatus_t compute_rnnt_joint_net_loss_fwd(const float *enc, const float *pred, const float *W1, const float *b1, const float *W2,
                                             const float *b2, const int *flat_labels, const int *label_lengths,
                                             const int *input_lengths, int hidden_size, int joint_size, int alphabet_size,
                                             int minibatch, float *costs, int joint_dtype, void *workspace, rnntOptions options) {
    return joint_net_call(enc, pred, W1, b1, W2, b2, flat_labels, label_lengths, input_lengths, nullptr, hidden_size, joint_size,
                          alphabet_size, minibatch, costs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, joint_dtype, 1 | 4,
                          workspace, options);
}

rnntStatus_t compute_rnnt_joint_net_loss_bwd(const float *enc, const float *pred, const float *W1, const float *b1, const float *W2,
                                             const float *b2, const int *flat_labels, const int *label_lengths,
                                             const int *input_lengths, const float *cost_scale, int hidden_size, int joint_size,
                                             int alphabet_size, int minibatch, float *d_enc, float *d_pred, float *dW1, float *db1,
                                             float *dW2, float *db2, int joint_dtype, void *workspace, rnntOptions options) {
    if (!d_enc) return RNNT_STATUS_INVALID_VALUE;
    return joint_net_call(enc, pred, W1, b1, W2, b2, flat_labels, label_lengths, input_lengths, cost_scale, hidden_size, joint_size,
                          alphabet_size, minibatch, nullptr, d_enc, d_pred, dW1, db1, dW2, db2, joint_dtype, 2, workspace, options);
}

// The joint alone, for decoding (utils/decoding.py:6-18 evaluates dense_1 / dense_2 on one lattice cell per step).
rnntStatus_t compute_rnnt_joint_logits(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                                       int joint_size, int alphabet_size, int minibatch, float *logits, int joint_dtype,
                                       void *workspace, rnntOptions options) {
    if (!enc_proj || !pred_proj || !W2 || !b2 || !logits || !workspace) return RNNT_STATUS_INVALID_VALUE;
    if (joint_size <= 0 || alphabet_size <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    if (joint_dtype != 0 && joint_dtype != 1) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (options.maxU > 1024) return RNNT_STATUS_INVALID_VALUE;
    if (((uintptr_t)workspace & 255) != 0) return RNNT_STATUS_INVALID_VALUE;
    if (!joint_dtype_supported(joint_dtype, joint_size, alphabet_size)) return RNNT_STATUS_INVALID_VALUE;
    hipStream_t s = (hipStream_t)options.stream;
    if (joint_dtype == 1)
        return from_hip(launch_joint_logits_f16(enc_proj, pred_proj, W2, b2, joint_size, alphabet_size, minibatch, options.maxT,
                                                options.maxU, logits, workspace, s));
    return from_hip(launch_joint_logits(enc_proj, pred_proj, W2, b2, joint_size, alphabet_size, minibatch, options.maxT,
                                        options.maxU, logits, workspace, s));
}

// Rows (x u-tiles) the last f32-grade backward on this workspace visited / rows inside the utterances (include/rnnt.h).
rnntStatus_t get_rnnt_joint_backward_rows(void *workspace, int joint_size, int alphabet_size, int minibatch, rnntOptions options,
                                          int rows[2]) {
    if (!workspace || !rows || joint_size <= 0 || alphabet_size <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (((uintptr_t)workspace & 255) != 0) return RNNT_STATUS_INVALID_VALUE;
    const bool t32 = joint_dtype_supported(0, joint_size, alphabet_size), t16 = joint_dtype_supported(1, joint_size, alphabet_size);
    if (!t32 && !t16) return RNNT_STATUS_INVALID_VALUE;
    rows[0] = rows[1] = -1;
    rnntStatus_t r = RNNT_STATUS_SUCCESS;
    if (t32)
        r = from_hip(joint_backward_rows(workspace, options.maxT, options.maxU, minibatch, joint_size, alphabet_size, rows,
                                         (hipStream_t)options.stream));
    // (a shape both arithmetic types take, alphabet_size 128: whichever backward stamped the workspace last answers)
    if (r == RNNT_STATUS_SUCCESS && rows[1] < 0 && t16)
        r = from_hip(joint_f16_backward_rows(workspace, options.maxT, options.maxU, minibatch, joint_size, alphabet_size, rows,
                                             (hipStream_t)options.stream));
    return r;
}

// The whole joint network without the loss: first Dense layer (the library's split-precision GEMMs, as in the fused loss) + the
// joint, logits [minibatch, maxT, maxU, alphabet_size] out.  Workspace: get_joint_net_workspace_size().
rnntStatus_t compute_rnnt_joint_net_logits(const float *enc, const float *pred, const float *W1, const float *b1, const float *W2,
                                           const float *b2, int hidden_size, int joint_size, int alphabet_size, int minibatch,
                                           float *logits, int joint_dtype, void *workspace, rnntOptions options) {
    if (!enc || !pred || !W1 || !b1 || !W2 || !b2 || !logits || !workspace) return RNNT_STATUS_INVALID_VALUE;
    if (hidden_size <= 0 || joint_size <= 0 || alphabet_size <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    if (joint_dtype != 0 && joint_dtype != 1) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (options.maxU > 1024) return RNNT_STATUS_INVALID_VALUE;
    if (((uintptr_t)workspace & 255) != 0 || !dense_supported(hidden_size, joint_size)) return RNNT_STATUS_INVALID_VALUE;
    if ((((uintptr_t)enc | (uintptr_t)pred | (uintptr_t)W1 | (uintptr_t)b1) & 15) != 0) return RNNT_STATUS_INVALID_VALUE;
    if (!joint_dtype_supported(joint_dtype, joint_size, alphabet_size)) return RNNT_STATUS_INVALID_VALUE;  // before the GEMM is enqueued
    const int B = minibatch, T = options.maxT, U = options.maxU;
    hipStream_t s = (hipStream_t)options.stream;
    size_t base = 0;
    hipError_t e = joint_workspace_bytes(T, U, B, joint_size, alphabet_size, -1, &base);
    if (e != hipSuccess) return from_hip(e);
    // (the joint's own prep kernel builds the tanh tables here: one cell per call is the common case, nothing to save)
    if ((e = launch_dense_fwd(enc, pred, W1, b1, B, T, U, hidden_size, joint_size, workspace, base, nullptr, nullptr, nullptr, s)) != hipSuccess)
        return from_hip(e);
    float *ep, *pp, *dep, *dpp;
    dense_proj_pointers(workspace, B, T, U, hidden_size, joint_size, base, &ep, &pp, &dep, &dpp);
    if (joint_dtype == 1)
        return from_hip(launch_joint_logits_f16(ep, pp, W2, b2, joint_size, alphabet_size, B, T, U, logits, workspace, s));
    return from_hip(launch_joint_logits(ep, pp, W2, b2, joint_size, alphabet_size, B, T, U, logits, workspace, s));
}

}  // extern "C"
