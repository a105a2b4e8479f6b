// rnnt_entrypoint.hip -- the extern "C" boundary of libwarprnnt.so (declared in include/rnnt.h).
//
// Replaces the reference's native entry points (warp-transducer `src/rnnt_entrypoint.cu`, named at
// cmake/warp-rnnt-cmakelist.txt:99; reached from utils/loss.py:34-35).  Argument validation follows
// the published contract: null pointers / non-positive sizes ->
// RNNT_STATUS_INVALID_VALUE; nothing is allocated; everything is enqueued on the caller's stream.
#include "../../include/rnnt.h"
#include "rnnt_common.h"

#include <stdlib.h>

using namespace rnnt;

namespace rnnt {
// joint_kernels.hip
hipError_t joint_workspace_bytes(int T, int U, int B, int J, int V, size_t *bytes);
hipError_t launch_joint_loss(const float *enc_proj, const float *pred_proj, const float *W2, const float *b2,
                             const int *labels, const int *label_lengths, const int *input_lengths,
                             const float *cost_scale, int J, int V, int B, int T, int U, int blank, float *costs,
                             float *d_enc_proj, float *d_pred_proj, float *dW2, float *db2, int joint_dtype,
                             int phases, void *workspace, hipStream_t s);
}  // namespace rnnt

static rnntStatus_t check_options(const rnntOptions &o) {
    if (o.loc != RNNT_GPU) return RNNT_STATUS_INVALID_VALUE;  // device-only library: no CPU fallback
    if (!o.batch_first) return RNNT_STATUS_INVALID_VALUE;
    if (o.maxT <= 0 || o.maxU <= 0 || o.blank_label < 0) return RNNT_STATUS_INVALID_VALUE;
    if (o.maxU > 1024) return RNNT_STATUS_INVALID_VALUE;  // register-resident sweep limit
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
    p.flags = (int *)(ws + w.flags);
    p.B = B, p.T = o.maxT, p.U = o.maxU, p.V = V, p.blank = o.blank_label;
    p.b0 = 0, p.nb = B;
    {
        static const int rev = [] { const char *e = getenv("RNNT_GRAD_ORDER"); return (e && e[0] == 'f') ? 0 : 1; }();
        static const int tune = [] { const char *e = getenv("RNNT_TUNE"); return e ? atoi(e) : 0; }();
        p.tune = tune;
        p.rev_grad = rev;  // RNNT_GRAD_ORDER=fwd restores the same order as the lsm pass
    }
    p.tile = make_tile(o.maxT, o.maxU, V);
    p.N = w.N, p.Nr = w.Nr, p.Up = w.Up, p.NC = w.NC, p.NG = w.NG;
    p.cells = (uint32_t)cells;
    p.divU = make_fastdiv((uint32_t)o.maxU);
    p.divT = make_fastdiv((uint32_t)o.maxT);
    p.divV = make_fastdiv((uint32_t)V);
    return true;
}

namespace rnnt {
bool fill_loss_params(LossParams &p, const float *acts, float *grads, const int *labels, const int *label_lengths,
                      const int *input_lengths, const float *cost_scale, int V, int B, float *costs, void *workspace,
                      int maxT, int maxU, int blank) {
    rnntOptions o;
    o.loc = RNNT_GPU, o.stream = nullptr, o.blank_label = blank, o.maxT = maxT, o.maxU = maxU, o.batch_first = 1;
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

rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, int gpu, size_t *size_bytes) {
    if (!size_bytes || maxT <= 0 || maxU <= 0 || minibatch <= 0 || !gpu) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = make_layout(maxT, maxU, minibatch).total;
    return RNNT_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// Utterance-group pipelining.  The sweeps are latency-bound (T+U-1 dependent steps, one wave per
// utterance and direction) and use a fraction of the CUs, while the lsm / gradient passes are
// HBM-bound.  Splitting the batch into groups lets group g's sweeps run on a side stream while the
// caller's stream streams the next group's logits:
//     main :  memset  lsm(0) lsm(1) ... lsm(G-1)          [wait s(0)] grad(0) [wait s(1)] grad(1) ...
//     side g:          [wait lsm(g)] sweeps(g)
// Side streams and events are created once per device and owned by the library (no device memory
// is allocated); every fork is joined back into the caller's stream before the call returns, so the
// caller still sees ordinary stream-ordered semantics.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxGroups = 8;
constexpr int kMaxDevices = 16;
struct Pipe {
    bool ready = false;
    hipStream_t side[kMaxGroups];
    hipEvent_t lsm_done[kMaxGroups], sweep_done[kMaxGroups];
};
Pipe g_pipes[kMaxDevices];

Pipe *get_pipe() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    Pipe &pp = g_pipes[dev];
    if (!pp.ready) {
        for (int i = 0; i < kMaxGroups; ++i) {
            if (hipStreamCreateWithFlags(&pp.side[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&pp.lsm_done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&pp.sweep_done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        pp.ready = true;
    }
    return &pp;
}

int choose_groups(const LossParams &p, bool grad) {
    // Measured on MI355X (profiles/r01_notes.md): with 4 HW queues and ~10 us per cross-stream event hop,
    // kernel-level group pipelining LOSES to one group (0.465 ms vs 0.334 ms at C2), so it is opt-in.
    int g = 1;
    if (const char *e = getenv("RNNT_GROUPS")) g = atoi(e);
    if (g > kMaxGroups) g = kMaxGroups;
    if (g > p.B) g = p.B;
    if (g < 1) g = 1;
    if (!tile_path_ok(p, grad)) g = 1;  // only the patch kernels take an utterance range
    return g;
}

// lsm + sweeps for all groups; on return (status success) the caller's stream has either been
// joined with every sweep (join_all) or `ngroups`/`pipe` tell the caller which events to wait for.
rnntStatus_t run_forward(LossParams &p, const WsLayout &w, hipStream_t s, bool grad_follows, int &ngroups,
                         Pipe *&pipe) {
    if (hipMemsetAsync(p.W, kFillByte, w.A - w.W, s) != hipSuccess) return RNNT_STATUS_MEMOPS_FAILED;
    ngroups = choose_groups(p, grad_follows);
    pipe = (ngroups > 1) ? get_pipe() : nullptr;
    if (!pipe) ngroups = 1;
    if (ngroups == 1) {
        hipError_t e = launch_lsm(p, s);
        if (e != hipSuccess) return from_hip(e);
        return from_hip(launch_sweeps(p, s));
    }
    const int B = p.B;
    for (int g = 0; g < ngroups; ++g) {
        LossParams q = p;
        q.b0 = (int)((long long)B * g / ngroups);
        q.nb = (int)((long long)B * (g + 1) / ngroups) - q.b0;
        hipError_t e = launch_lsm(q, s);
        if (e != hipSuccess) return from_hip(e);
        if (hipEventRecord(pipe->lsm_done[g], s) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
        if (hipStreamWaitEvent(pipe->side[g], pipe->lsm_done[g], 0) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
        e = launch_sweeps(q, pipe->side[g]);
        if (e != hipSuccess) return from_hip(e);
        if (hipEventRecord(pipe->sweep_done[g], pipe->side[g]) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    }
    return RNNT_STATUS_SUCCESS;
}

// EXPERIMENTAL, opt-in (RNNT_OVERLAP=1), bit-identical to the serial schedule (tests), currently NOT faster:
// 0.327 ms vs 0.315 ms per step at C2 (profiles/r01_notes.md has the timelines).
// Workgroup-granular overlap of the three stages:
//     side : [after the memsets] sweeps   -- every sweep wave polls its utterance's "lsm patches done" counter
//     main : memsets, lsm (patches publish per utterance), grad (patches stage their logits, then poll the
//            utterance's "sweeps done" counter), join.
// The sweep kernel (2B one-wave workgroups) is enqueued FIRST so it is resident before the gradient patches
// that wait on it; all polls are bounded (flags[2B] reports a timeout instead of hanging the device).
static int overlap_enabled() {
    const char *e = getenv("RNNT_OVERLAP");
    return (e && e[0] == '1') ? 1 : 0;
}

rnntStatus_t run_overlapped(LossParams &p, const WsLayout &w, hipStream_t s, bool with_grad) {
    Pipe *pipe = get_pipe();
    if (!pipe) return RNNT_STATUS_EXECUTION_FAILED;
    if (hipMemsetAsync(p.W, kFillByte, w.A - w.W, s) != hipSuccess) return RNNT_STATUS_MEMOPS_FAILED;
    if (hipMemsetAsync(p.flags, 0, flag_words(p.B) * sizeof(int), s) != hipSuccess)
        return RNNT_STATUS_MEMOPS_FAILED;
    if (hipEventRecord(pipe->lsm_done[0], s) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    if (hipStreamWaitEvent(pipe->side[0], pipe->lsm_done[0], 0) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    hipError_t e = launch_sweeps(p, pipe->side[0], true);
    if (e != hipSuccess) return from_hip(e);
    if (hipEventRecord(pipe->sweep_done[0], pipe->side[0]) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    e = launch_lsm(p, s, true);
    if (e != hipSuccess) return from_hip(e);
    // kernel boundary behind lsm = every XCD's L2 written back: the slow-path signal for the sweep waves.
    // The gradient kernel raises it itself (its first workgroup); score-only calls need a marker kernel.
    if (with_grad)
        e = launch_grad(p, s, true);
    else
        e = launch_lsm_done_marker(p, s);
    if (e != hipSuccess) return from_hip(e);
    if (hipStreamWaitEvent(s, pipe->sweep_done[0], 0) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t validate(const void *acts, const void *labels, const void *ll, const void *il, const void *ws,
                      int V, int B, const rnntOptions &o) {
    if (!acts || !labels || !ll || !il || !ws) return RNNT_STATUS_INVALID_VALUE;
    if (V <= 0 || B <= 0) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = check_options(o);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (o.blank_label >= V) return RNNT_STATUS_INVALID_VALUE;
    return RNNT_STATUS_SUCCESS;
}
}  // namespace

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
    if (overlap_enabled() && overlap_path_ok(p, false)) return run_overlapped(p, w, s, false);
    int ng = 1;
    Pipe *pipe = nullptr;
    st = run_forward(p, w, s, false, ng, pipe);
    if (st != RNNT_STATUS_SUCCESS) return st;
    for (int g = 0; g < ng && pipe; ++g)
        if (hipStreamWaitEvent(s, pipe->sweep_done[g], 0) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    return RNNT_STATUS_SUCCESS;
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
    return from_hip(launch_grad(p, (hipStream_t)options.stream));
}

// compute_rnnt_loss with the upstream gradient folded in (cost_scale NULL = 1): the pipelined form.
rnntStatus_t compute_rnnt_loss_ex(const float *acts, float *grads, const int *flat_labels,
                                  const int *label_lengths, const int *input_lengths, const float *cost_scale,
                                  int alphabet_size, int minibatch, float *costs, void *workspace,
                                  rnntOptions options) {
    if (!grads)
        return compute_rnnt_loss_fwd(acts, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                     costs, workspace, options);
    if (!costs) return RNNT_STATUS_INVALID_VALUE;
    rnntStatus_t st = validate(acts, flat_labels, label_lengths, input_lengths, workspace, alphabet_size, minibatch,
                               options);
    if (st != RNNT_STATUS_SUCCESS) return st;
    LossParams p;
    if (!fill_params(p, acts, grads, flat_labels, label_lengths, input_lengths, cost_scale, alphabet_size,
                     minibatch, costs, workspace, options))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t s = (hipStream_t)options.stream;
    const WsLayout w = make_layout(options.maxT, options.maxU, minibatch);
    if (overlap_enabled() && overlap_path_ok(p, true)) return run_overlapped(p, w, s, true);
    int ng = 1;
    Pipe *pipe = nullptr;
    st = run_forward(p, w, s, true, ng, pipe);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (ng == 1 || !pipe) return from_hip(launch_grad(p, s));
    for (int g = 0; g < ng; ++g) {
        LossParams q = p;
        q.b0 = (int)((long long)p.B * g / ng);
        q.nb = (int)((long long)p.B * (g + 1) / ng) - q.b0;
        if (hipStreamWaitEvent(s, pipe->sweep_done[g], 0) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
        hipError_t e = launch_grad(q, s);
        if (e != hipSuccess) return from_hip(e);
    }
    return RNNT_STATUS_SUCCESS;
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
    return from_hip(joint_workspace_bytes(maxT, maxU, minibatch, joint_size, alphabet_size, size_bytes));
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
    const bool any_grad = d_enc_proj || d_pred_proj || dW2 || db2;
    if (any_grad && !(d_enc_proj && d_pred_proj && dW2 && db2)) return RNNT_STATUS_INVALID_VALUE;
    if ((phases & 2) && !(phases & 1) && !any_grad) return RNNT_STATUS_INVALID_VALUE;
    return from_hip(launch_joint_loss(enc_proj, pred_proj, W2, b2, flat_labels, label_lengths, input_lengths,
                                      cost_scale, joint_size, alphabet_size, minibatch, options.maxT, options.maxU,
                                      options.blank_label, costs, d_enc_proj, d_pred_proj, dW2, db2, joint_dtype,
                                      phases, workspace, (hipStream_t)options.stream));
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
                      alphabet_size, minibatch, costs, nullptr, nullptr, nullptr, nullptr, joint_dtype, 1, workspace,
                      options);
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

}  // extern "C"
