"""Host-side mirror of the reference's loss surface, on PyTorch-ROCm over the C ABI.

Reference interface (same names, argument order and meaning):
  utils/loss.py:12-38   get_loss_fn(reduction_factor) -> _loss_fn(y_true, y_pred, spec_lengths, label_lengths)
  utils/loss.py:6,34-35 warprnnt_tensorflow.rnnt_loss(acts, labels, input_lengths, label_lengths, blank_label=0)
Call sites: run_rnnt.py:493-494 (construction), :272-273 (positional), :405-407 (keywords).

Differences, all deliberate:
  * the native op is libwarprnnt.so for gfx950 (include/rnnt.h); PyTorch only provides device
    memory, the HIP stream and autograd plumbing;
  * there is no silent fallback (utils/loss.py:14-22 returns the logits when the op is missing):
    a missing library or a CPU tensor raises;
  * the gradient pass runs in backward() with the upstream gradient folded in
    (compute_rnnt_loss_bwd), instead of computing unscaled grads in forward and multiplying later.
"""
from __future__ import annotations

import math

import torch

from . import _lib


def _as_i32(x: torch.Tensor, device) -> torch.Tensor:
    return x.to(device=device, dtype=torch.int32).contiguous()


class _RNNTLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, input_lengths, label_lengths, blank_label):
        lib = _lib.load()
        if not acts.is_cuda:
            raise RuntimeError(
                "rnnt_loss: acts must live on an MI355X (cuda/HIP) device; this engine has no CPU path"
            )
        if acts.dim() != 4:
            raise ValueError("rnnt_loss: acts must be [B, T, U, V]")
        if acts.dtype != torch.float32:
            raise TypeError("rnnt_loss: acts must be float32 (the reference op is float32-only)")
        B, T, U, V = acts.shape
        dev = acts.device
        acts_c = acts.detach().contiguous()
        labels = _as_i32(labels, dev)
        input_lengths = _as_i32(input_lengths, dev)
        label_lengths = _as_i32(label_lengths, dev)
        if U > 1 and tuple(labels.shape) != (B, U - 1):
            raise ValueError(f"rnnt_loss: labels must be [B, U-1] = [{B}, {U - 1}], got {tuple(labels.shape)}")
        if input_lengths.numel() != B or label_lengths.numel() != B:
            raise ValueError("rnnt_loss: input_lengths and label_lengths must be [B]")
        if labels.numel() == 0:
            labels = torch.zeros((B, 1), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, int(blank_label), T, U)
            st = lib.compute_rnnt_loss_fwd(
                acts_c.data_ptr(), labels.data_ptr(), label_lengths.data_ptr(), input_lengths.data_ptr(),
                V, B, costs.data_ptr(), ws.data_ptr(), opts)
        _lib.check(st, "compute_rnnt_loss_fwd")
        ctx.save_for_backward(acts_c, labels, input_lengths, label_lengths, ws)
        ctx.blank = int(blank_label)
        return costs

    @staticmethod
    def backward(ctx, grad_costs):
        acts, labels, input_lengths, label_lengths, ws = ctx.saved_tensors
        lib = _lib.load()
        B, T, U, V = acts.shape
        dev = acts.device
        scale = grad_costs.to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            grads = torch.empty_like(acts)
            opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, ctx.blank, T, U)
            st = lib.compute_rnnt_loss_bwd(
                acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), label_lengths.data_ptr(),
                input_lengths.data_ptr(), scale.data_ptr(), V, B, ws.data_ptr(), opts)
        _lib.check(st, "compute_rnnt_loss_bwd")
        return grads, None, None, None, None


def rnnt_loss(acts, labels, input_lengths, label_lengths, blank_label: int = 0):
    """Per-utterance transducer negative log-likelihood, differentiable in `acts`.

    Same contract as warprnnt_tensorflow.rnnt_loss on a CUDA build (utils/loss.py:34-35):
    acts are RAW LOGITS [B, T, U, V] (the log-softmax is fused), labels [B, U-1] int,
    input_lengths / label_lengths [B] int; returns costs [B] float32."""
    return _RNNTLossFunction.apply(acts, labels, input_lengths, label_lengths, blank_label)


def rnnt_loss_and_grad(acts, labels, input_lengths, label_lengths, blank_label: int = 0, visit_all: bool = False):
    """The upstream C entry point as one call: compute_rnnt_loss(acts, grads, ...) ->
    (costs [B], grads [B,T,U,V]) with grads = d cost_b / d acts (unscaled), like the two outputs of
    the reference's WarpRNNT op.  No autograd graph is built.
    visit_all: compute_rnnt_loss_flags(..., RNNT_VISIT_ALL) -- no occupancy floor (vocabularies above 60 symbols otherwise
    write zeros for cells whose occupancy is below 2^-50 without reading their logits)."""
    lib = _lib.load()
    if not acts.is_cuda:
        raise RuntimeError("rnnt_loss_and_grad: acts must live on an MI355X (cuda/HIP) device")
    if acts.dtype != torch.float32 or acts.dim() != 4:
        raise TypeError("rnnt_loss_and_grad: acts must be float32 [B, T, U, V]")
    B, T, U, V = acts.shape
    dev = acts.device
    acts_c = acts.detach().contiguous()
    labels = _as_i32(labels, dev)
    if labels.numel() == 0:
        labels = torch.zeros((B, 1), dtype=torch.int32, device=dev)
    input_lengths = _as_i32(input_lengths, dev)
    label_lengths = _as_i32(label_lengths, dev)
    with torch.cuda.device(dev):
        ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        grads = torch.empty_like(acts_c)
        opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, int(blank_label), T, U)
        if visit_all:
            st = lib.compute_rnnt_loss_flags(
                acts_c.data_ptr(), grads.data_ptr(), labels.data_ptr(), label_lengths.data_ptr(),
                input_lengths.data_ptr(), None, V, B, costs.data_ptr(), ws.data_ptr(), opts, _lib.RNNT_VISIT_ALL)
        else:
            st = lib.compute_rnnt_loss(
                acts_c.data_ptr(), grads.data_ptr(), labels.data_ptr(), label_lengths.data_ptr(),
                input_lengths.data_ptr(), V, B, costs.data_ptr(), ws.data_ptr(), opts)
    _lib.check(st, "compute_rnnt_loss")
    return costs, grads


class RNNTLoss(torch.nn.Module):
    """nn.Module wrapper; reduction 'none' returns the reference's per-utterance costs."""

    def __init__(self, blank_label: int = 0, reduction: str = "none"):
        super().__init__()
        if reduction not in ("none", "sum", "mean"):
            raise ValueError(reduction)
        self.blank_label = blank_label
        self.reduction = reduction

    def forward(self, acts, labels, input_lengths, label_lengths):
        costs = rnnt_loss(acts, labels, input_lengths, label_lengths, self.blank_label)
        if self.reduction == "sum":
            return costs.sum()
        if self.reduction == "mean":
            return costs.mean()
        return costs


def reduced_lengths(spec_lengths: torch.Tensor, reduction_factor) -> torch.Tensor:
    """T_b = ceil(spec_length_b / reduction_factor) as int32 (utils/loss.py:31-33)."""
    return torch.ceil(spec_lengths.to(torch.float64) / float(reduction_factor)).to(torch.int32)


def get_loss_fn(reduction_factor):
    """Mirror of utils/loss.py:12-38.  Returns fn(y_true, y_pred, spec_lengths, label_lengths) -> costs [B].

    y_true: labels [B, U-1]; y_pred: joint logits [B, T', U, V]; spec_lengths: encoder input
    lengths BEFORE time reduction; label_lengths [B].  The reference log-softmaxes first only on
    non-CUDA builds (utils/loss.py:29-30); here the fused-softmax device op is the only path, so a
    CPU tensor raises instead of silently training on garbage."""
    if reduction_factor is None or float(reduction_factor) <= 0 or math.isnan(float(reduction_factor)):
        raise ValueError("reduction_factor must be positive")
    _lib.load()  # fail at construction time (run_rnnt.py:493-494), not at the first step

    def _loss_fn(y_true, y_pred, spec_lengths, label_lengths):
        y_true = y_true.to(torch.int32)
        spec_lengths = reduced_lengths(spec_lengths, reduction_factor)
        return rnnt_loss(y_pred, y_true, spec_lengths, label_lengths)

    return _loss_fn
