"""Joint network fused with the transducer loss (host side).

Reference: model.py:158-166 (broadcast add -> Dense(J, tanh) -> Dense(V)) followed by
utils/loss.py:24-36 and TF autodiff (run_rnnt.py:284).  The first Dense layer is applied to the
encoder and prediction-network outputs separately (exact factorisation) and the whole network runs
behind the C ABI: libwarprnnt.so's compute_rnnt_joint_net_loss_* entry points do the two W1 GEMMs and
their backward (csrc/dense_kernels.hip), tanh, the J x V projection on the MFMA units, log-softmax,
alpha/beta, and the gradient scatter back to enc / pred / W1 / b1 / W2 / b2, without ever materialising
[B,T,U,J] or [B,T,U,V] tensors.  Hidden sizes the dense kernels do not take (not a multiple of 32) go
through torch.matmul + autograd around compute_rnnt_joint_loss_* (first_layer="torch").
"""
from __future__ import annotations

import math

import torch

from . import _lib


class _JointLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_proj, pred_proj, W2, b2, labels, input_lengths, label_lengths, blank_label, joint_dtype):
        lib = _lib.load()
        for name, x in (("enc_proj", enc_proj), ("pred_proj", pred_proj), ("W2", W2), ("b2", b2)):
            if not x.is_cuda:
                raise RuntimeError(f"rnnt_joint_loss: {name} must live on an MI355X (cuda/HIP) device; no CPU path")
            if x.dtype != torch.float32:
                raise TypeError(f"rnnt_joint_loss: {name} must be float32")
        B, T, J = enc_proj.shape
        U = pred_proj.shape[1]
        V = W2.shape[1]
        if pred_proj.shape != (B, U, J) or W2.shape[0] != J or b2.shape != (V,):
            raise ValueError("rnnt_joint_loss: inconsistent shapes")
        dev = enc_proj.device
        ep, pp, w2, bb = (x.detach().contiguous() for x in (enc_proj, pred_proj, W2, b2))
        labels = labels.to(device=dev, dtype=torch.int32).contiguous()
        # the kernels index labels with row stride U-1: any other width would silently read the wrong labels
        if U > 1 and tuple(labels.shape) != (B, U - 1):
            raise ValueError(f"rnnt_joint_loss: labels must be [B, U-1] = [{B}, {U - 1}], got {tuple(labels.shape)}")
        if labels.numel() == 0:
            labels = torch.zeros((B, 1), dtype=torch.int32, device=dev)
        il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
        ll = label_lengths.to(device=dev, dtype=torch.int32).contiguous()
        if il.numel() != B or ll.numel() != B:
            raise ValueError("rnnt_joint_loss: input_lengths and label_lengths must be [B]")
        with torch.cuda.device(dev):
            ws = _new_workspace(_lib.joint_workspace_bytes(T, U, B, J, V), dev)
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, int(blank_label), T, U)
            if any(ctx.needs_input_grad[:4]):
                # _fwd = "a _bwd call on this workspace follows": the f16 joint parks its softmax numerators for it
                st = lib.compute_rnnt_joint_loss_fwd(ep.data_ptr(), pp.data_ptr(), w2.data_ptr(), bb.data_ptr(),
                                                     labels.data_ptr(), ll.data_ptr(), il.data_ptr(), J, V, B,
                                                     costs.data_ptr(), int(joint_dtype), ws.data_ptr(), opts)
            else:  # costs only (evaluation)
                st = lib.compute_rnnt_joint_loss(ep.data_ptr(), pp.data_ptr(), w2.data_ptr(), bb.data_ptr(),
                                                 labels.data_ptr(), ll.data_ptr(), il.data_ptr(), None, J, V, B,
                                                 costs.data_ptr(), None, None, None, None, int(joint_dtype), ws.data_ptr(), opts)
        _lib.check(st, "compute_rnnt_joint_loss_fwd")
        ctx.save_for_backward(ep, pp, w2, bb, labels, il, ll, ws)
        ctx.blank = int(blank_label)
        ctx.joint_dtype = int(joint_dtype)
        return costs

    @staticmethod
    def backward(ctx, grad_costs):
        ep, pp, w2, bb, labels, il, ll, ws = ctx.saved_tensors
        lib = _lib.load()
        B, T, J = ep.shape
        U, V = pp.shape[1], w2.shape[1]
        dev = ep.device
        scale = grad_costs.to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            d_ep, d_pp, d_w2, d_b2 = (torch.empty_like(x) for x in (ep, pp, w2, bb))
            opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, ctx.blank, T, U)
            st = lib.compute_rnnt_joint_loss_bwd(ep.data_ptr(), pp.data_ptr(), w2.data_ptr(), bb.data_ptr(),
                                                 labels.data_ptr(), ll.data_ptr(), il.data_ptr(), scale.data_ptr(),
                                                 J, V, B, d_ep.data_ptr(), d_pp.data_ptr(), d_w2.data_ptr(),
                                                 d_b2.data_ptr(), ctx.joint_dtype, ws.data_ptr(), opts)
            _lib.check(st, "compute_rnnt_joint_loss_bwd")
            _note_backward_rows(ws, T, U, B, J, V, ctx.blank)
        return d_ep, d_pp, d_w2, d_b2, None, None, None, None, None


class _JointNetLossFunction(torch.autograd.Function):
    """The whole joint network + loss behind the C ABI (compute_rnnt_joint_net_loss_fwd / _bwd): the first Dense layer and its
    backward run in the library too (csrc/dense_kernels.hip), not in torch."""

    @staticmethod
    def forward(ctx, enc, pred, W1, b1, W2, b2, labels, input_lengths, label_lengths, blank_label, joint_dtype):
        lib = _lib.load()
        for name, x in (("enc", enc), ("pred", pred), ("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2)):
            if not x.is_cuda:
                raise RuntimeError(f"rnnt_joint_loss: {name} must live on an MI355X (cuda/HIP) device; no CPU path")
            if x.dtype != torch.float32:
                raise TypeError(f"rnnt_joint_loss: {name} must be float32")
        B, T, H = enc.shape
        U = pred.shape[1]
        J, V = W2.shape
        if pred.shape != (B, U, H) or W1.shape != (H, J) or b1.shape != (J,) or b2.shape != (V,):
            raise ValueError("rnnt_joint_loss: inconsistent shapes")
        dev = enc.device
        e, p, w1, bb1, w2, bb2 = (x.detach().contiguous() for x in (enc, pred, W1, b1, W2, b2))
        # the dense kernels move 16 bytes per access: an operand that is contiguous but starts off that grid (a slice of a flat
        # parameter bucket, say) is copied to a fresh allocation instead of being refused
        e, p, w1, bb1 = (x if x.data_ptr() % 16 == 0 else x.clone() for x in (e, p, w1, bb1))
        labels = labels.to(device=dev, dtype=torch.int32).contiguous()
        if U > 1 and tuple(labels.shape) != (B, U - 1):
            raise ValueError(f"rnnt_joint_loss: labels must be [B, U-1] = [{B}, {U - 1}], got {tuple(labels.shape)}")
        if labels.numel() == 0:
            labels = torch.zeros((B, 1), dtype=torch.int32, device=dev)
        il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
        ll = label_lengths.to(device=dev, dtype=torch.int32).contiguous()
        if il.numel() != B or ll.numel() != B:
            raise ValueError("rnnt_joint_loss: input_lengths and label_lengths must be [B]")
        with torch.cuda.device(dev):
            ws = _new_workspace(_lib.joint_net_workspace_bytes(T, U, B, H, J, V), dev)
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, int(blank_label), T, U)
            if any(ctx.needs_input_grad[:6]):
                st = lib.compute_rnnt_joint_net_loss_fwd(e.data_ptr(), p.data_ptr(), w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(),
                                                         bb2.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(), H, J, V, B,
                                                         costs.data_ptr(), int(joint_dtype), ws.data_ptr(), opts)
            else:  # costs only (evaluation)
                st = lib.compute_rnnt_joint_net_loss(e.data_ptr(), p.data_ptr(), w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(),
                                                     bb2.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(), None, H, J, V, B,
                                                     costs.data_ptr(), None, None, None, None, None, None, int(joint_dtype),
                                                     ws.data_ptr(), opts)
        _lib.check(st, "compute_rnnt_joint_net_loss_fwd")
        ctx.save_for_backward(e, p, w1, bb1, w2, bb2, labels, il, ll, ws)
        ctx.blank = int(blank_label)
        ctx.joint_dtype = int(joint_dtype)
        return costs

    @staticmethod
    def backward(ctx, grad_costs):
        e, p, w1, bb1, w2, bb2, labels, il, ll, ws = ctx.saved_tensors
        lib = _lib.load()
        B, T, H = e.shape
        U = p.shape[1]
        J, V = w2.shape
        dev = e.device
        scale = grad_costs.to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            grads = [torch.empty_like(x) for x in (e, p, w1, bb1, w2, bb2)]
            opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, ctx.blank, T, U)
            st = lib.compute_rnnt_joint_net_loss_bwd(e.data_ptr(), p.data_ptr(), w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(),
                                                     bb2.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(),
                                                     scale.data_ptr(), H, J, V, B, *(g.data_ptr() for g in grads),
                                                     ctx.joint_dtype, ws.data_ptr(), opts)
            _lib.check(st, "compute_rnnt_joint_net_loss_bwd")
            _note_backward_rows(ws, T, U, B, J, V, ctx.blank)
        return (*grads, None, None, None, None, None)


JOINT_DTYPES = {"f32": 0, "f16": 1}

# Diagnostics (off by default): with TRACK_BACKWARD_ROWS set, every backward of this module asks the library how many lattice rows
# (x 32-column tiles) it visited -- the data-dependent part of the f32-grade joint's run time (include/rnnt.h
# get_rnnt_joint_backward_rows; the query synchronises the stream) -- and last_backward_rows() returns the answer.
TRACK_BACKWARD_ROWS = False
_LAST_ROWS = None


def _note_backward_rows(ws, T, U, B, J, V, blank):
    global _LAST_ROWS
    if not TRACK_BACKWARD_ROWS:
        return
    import ctypes

    rows = (ctypes.c_int * 2)(-1, -1)
    opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, blank, T, U)
    st = _lib.load().get_rnnt_joint_backward_rows(ws.data_ptr(), J, V, B, opts, rows)
    _LAST_ROWS = (rows[0], rows[1]) if st == 0 else (-1, -1)


def last_backward_rows():
    """(rows x 32-column tiles the last tracked backward visited, rows inside the utterances); (-1, -1) where nothing is skipped
    (the f16 joint, the wide joint); None when nothing was tracked (TRACK_BACKWARD_ROWS)."""
    return _LAST_ROWS


# Test hook: when set to a byte value, every workspace this module allocates is filled with it before the forward call
# (0xFF = a NaN bit pattern in every float: a backward kernel that read a workspace word nobody wrote would show it).
_WORKSPACE_FILL = None


def _new_workspace(nbytes: int, dev) -> torch.Tensor:
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if _WORKSPACE_FILL is not None:
        ws.fill_(int(_WORKSPACE_FILL))
    return ws


def rnnt_joint_loss(enc, pred, W1, b1, W2, b2, labels, input_lengths, label_lengths, blank_label: int = 0,
                    joint_dtype: str = "auto", first_layer: str = "auto", visit_all: bool = False):
    """costs[b] = transducer NLL of  logits = tanh((enc[:,:,None]+pred[:,None]) @ W1 + b1) @ W2 + b2.

    enc [B,T,H] (encoder output), pred [B,U,H] (prediction-network output), W1 [H,J], b1 [J],
    W2 [J,V], b2 [V]  (Keras Dense kernels are stored [in, out], model.py:162-166).

    joint_dtype: arithmetic of the J x V product.  "f32": f32-grade products (binary16 hi + lo operands on the f16 MFMA units, f32 accumulation), small vocabularies (V <= 32, the reference's
    character set; up to 128 symbols at joint sizes up to 640, as up to four vocabulary tiles -- "auto" uses it up to 64).  "f16": operands rounded to binary16, f32 accumulation, for large vocabularies -- the counterpart
    of the reference's `mixed_float16` policy (run_rnnt.py:96-99); the lattice stays f32 either way.  "auto" picks by V.
    Shapes the kernels do not take natively (f16: V a multiple of 128, J a multiple of 128 up to 640; f32: J a multiple of
    64) are padded up exactly (zero units / zero-probability symbols).

    first_layer: where the first Dense layer (enc @ W1 + b1, pred @ W1, and dW1 / db1 / d enc / d pred) runs.  "engine": inside
    libwarprnnt.so (compute_rnnt_joint_net_loss_*: split-precision MFMA GEMMs, csrc/dense_kernels.hip; hidden size a multiple
    of 32).  "torch": torch.matmul + autograd around compute_rnnt_joint_loss_*.  "auto": the engine whenever it takes the shape.

    visit_all: RNNT_VISIT_ALL of include/rnnt.h -- the backward visits every lattice row instead of skipping the rows (x 32-column
    tiles) whose cells all have an occupancy below 2^-40 (their binary16 dlogits parts are exact zeros already: same results up to the
    order of a few f32 sums; timing then does not depend on the data)."""
    dtype_word = lambda name: JOINT_DTYPES[name] | (_lib.RNNT_VISIT_ALL if visit_all else 0)  # noqa: E731
    if joint_dtype == "auto":
        joint_dtype = _auto_joint_dtype(W2.shape[0], W2.shape[1])
    if joint_dtype not in JOINT_DTYPES:
        raise ValueError(f"rnnt_joint_loss: joint_dtype must be one of {sorted(JOINT_DTYPES)} or 'auto'")
    # The kernels take a fixed set of (J, V) shapes; anything else is padded up here, exactly:
    #   joint units  -- extra units get zero W1 columns / b1 entries (projections 0) and zero W2 rows: h = tanh(0) = 0 contributes nothing;
    #   vocabulary   -- extra columns get zero weights and a bias of -1e4: their softmax mass is exp(-1e4) = 0 in f32.
    # Autograd slices the gradients back through the pads.
    J, V = W2.shape
    H = W1.shape[0]
    Jp, Vp = padded_joint_shape(J, V, joint_dtype)
    if Jp != J:
        W1 = torch.nn.functional.pad(W1, (0, Jp - J))
        b1 = torch.nn.functional.pad(b1, (0, Jp - J))
        W2 = torch.nn.functional.pad(W2, (0, 0, 0, Jp - J))
    if Vp != V:
        W2 = torch.nn.functional.pad(W2, (0, Vp - V))
        b2 = torch.nn.functional.pad(b2, (0, Vp - V), value=_PAD_BIAS)
    if first_layer == "auto":
        first_layer = "engine" if (H % 32 == 0 and Jp % 64 == 0 and max(H, Jp) <= 4096) else "torch"
    if first_layer == "engine":
        # the whole joint network behind the C ABI: W1 GEMMs, their backward, tanh, W2, the lattice (include/rnnt.h)
        return _JointNetLossFunction.apply(enc, pred, W1, b1, W2, b2, labels, input_lengths, label_lengths, blank_label,
                                           dtype_word(joint_dtype))
    if first_layer != "torch":
        raise ValueError("rnnt_joint_loss: first_layer must be 'auto', 'engine' or 'torch'")
    # hidden sizes the library's dense kernels do not take (not a multiple of 32): the first layer through torch.matmul
    # (hipBLASLt) and autograd, the rest through compute_rnnt_joint_loss
    enc_proj = torch.matmul(enc, W1) + b1
    pred_proj = torch.matmul(pred, W1)
    return _JointLossFunction.apply(enc_proj, pred_proj, W2, b2, labels, input_lengths, label_lengths, blank_label,
                                    dtype_word(joint_dtype))


_LOGITS_CACHE = {}  # (device, entry, shape) -> (workspace, output): a greedy decoder asks for one cell per emitted symbol


@torch.no_grad()
def joint_logits(enc, pred, W1, b1, W2, b2, joint_dtype: str = "auto", reuse_buffers: bool = False):
    """logits [B, T, U, V] of the joint network through libwarprnnt.so (no autograd): the decoding twin of the joint
    (utils/decoding.py:6-18).  Same factorisation, tables and products as the fused loss of the same joint_dtype ("f32": V <= 32,
    f32-grade; "f16": binary16 operands, up to 8192 symbols -- the reference's default 4096 word pieces; "auto" picks by V), so a
    decoder sees the logits the loss was trained on.  The first Dense layer runs in the library too (compute_rnnt_joint_net_logits)
    when the hidden size is a multiple of 32, else through torch.matmul in front of compute_rnnt_joint_logits.  Joint sizes /
    vocabularies the kernels do not take natively are padded exactly as in rnnt_joint_loss.

    reuse_buffers=True returns a VIEW OF A CACHED BUFFER (one workspace + output pair per device, stream and shape, at most
    eight pairs): the next call with the same shape on the same stream overwrites it.  Only for callers that consume the result
    before they call again (the greedy decoder asks for one lattice cell per emitted symbol); the default allocates."""
    lib = _lib.load()
    for name, x in (("enc", enc), ("pred", pred), ("W1", W1), ("W2", W2)):
        if not x.is_cuda:
            raise RuntimeError(f"joint_logits: {name} must live on an MI355X (cuda/HIP) device; no CPU path")
    B, T, H = enc.shape
    U = pred.shape[1]
    J, V = W2.shape
    if joint_dtype == "auto":
        joint_dtype = _auto_joint_dtype(J, V)
    Jp, Vp = padded_joint_shape(J, V, joint_dtype)
    if Jp != J:
        W1 = torch.nn.functional.pad(W1, (0, Jp - J))
        b1 = torch.nn.functional.pad(b1, (0, Jp - J))
        W2 = torch.nn.functional.pad(W2, (0, 0, 0, Jp - J))
    if Vp != V:
        W2 = torch.nn.functional.pad(W2, (0, Vp - V))
        b2 = torch.nn.functional.pad(b2, (0, Vp - V), value=_PAD_BIAS)
    dev = enc.device
    engine_first_layer = H % 32 == 0 and max(H, Jp) <= 4096
    with torch.cuda.device(dev):
        key = (dev, torch.cuda.current_stream().cuda_stream, engine_first_layer, T, U, B, H, Jp, Vp)
        if reuse_buffers and key in _LOGITS_CACHE:
            ws, out = _LOGITS_CACHE[key]
        else:
            nbytes = (_lib.joint_net_workspace_bytes(T, U, B, H, Jp, Vp) if engine_first_layer
                      else _lib.joint_workspace_bytes(T, U, B, Jp, Vp))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            out = torch.empty(B, T, U, Vp, dtype=torch.float32, device=dev)
            if reuse_buffers:  # (the caller consumes `out` before the next call: decoding.greedy_decode_fn does)
                if len(_LOGITS_CACHE) > 8:
                    _LOGITS_CACHE.clear()
                _LOGITS_CACHE[key] = (ws, out)
        opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, 0, T, U)
        w2, bb = W2.detach().contiguous().float(), b2.detach().contiguous().float()
        if engine_first_layer:
            e, p, w1, bb1 = (x.detach().contiguous().float() for x in (enc, pred, W1, b1))
            e, p, w1, bb1 = (x if x.data_ptr() % 16 == 0 else x.clone() for x in (e, p, w1, bb1))
            st = lib.compute_rnnt_joint_net_logits(e.data_ptr(), p.data_ptr(), w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(),
                                                   bb.data_ptr(), H, Jp, Vp, B, out.data_ptr(), JOINT_DTYPES[joint_dtype],
                                                   ws.data_ptr(), opts)
            _lib.check(st, "compute_rnnt_joint_net_logits")
        else:
            ep = (torch.matmul(enc.float(), W1) + b1).contiguous()
            pp = torch.matmul(pred.float(), W1).contiguous()
            st = lib.compute_rnnt_joint_logits(ep.data_ptr(), pp.data_ptr(), w2.data_ptr(), bb.data_ptr(), Jp, Vp, B,
                                               out.data_ptr(), JOINT_DTYPES[joint_dtype], ws.data_ptr(), opts)
            _lib.check(st, "compute_rnnt_joint_logits")
    return out if Vp == V else out[..., :V]


_PAD_BIAS = -1.0e4
_F16_J = (128, 256, 384, 512, 640)


def _auto_joint_dtype(J: int, V: int) -> str:
    """f32-grade products up to 64 symbols (32 at joint sizes above 640): one or two vocabulary tiles of the split-precision joint,
    as fast as the f16 joint on its smallest (128-column) shape and f32-grade; the f16 MFMA joint beyond.  (joint_dtype="f32"
    takes up to 128 symbols -- four tiles, four passes -- when f32-grade gradients matter more than time.)"""
    return "f32" if (V <= 32 or (V <= 64 and J <= 640)) else "f16"



def padded_joint_shape(J: int, V: int, joint_dtype: str):
    """(J, V) -> the nearest shape the chosen kernels accept (include/rnnt.h), or raises if there is none."""
    if joint_dtype == "f32":
        Jp = (J + 63) // 64 * 64
        if Jp > 704:
            raise ValueError("rnnt_joint_loss: the f32 joint takes joint sizes of at most 704")
        if V > (128 if Jp <= 640 else 32):
            raise ValueError("rnnt_joint_loss: the f32 joint takes vocabularies of at most 128 symbols (32 at joint sizes above 640); "
                             "use joint_dtype='f16'")
        return Jp, V
    Jp = next((j for j in _F16_J if j >= J), None)
    if Jp is None:
        raise ValueError("rnnt_joint_loss: the f16 joint takes joint sizes of at most 640")
    Vp = max(128, (V + 127) // 128 * 128)
    if Vp > 8192:
        raise ValueError("rnnt_joint_loss: the f16 joint takes vocabularies of at most 8192 symbols")
    return Jp, Vp


class JointLoss(torch.nn.Module):
    """The reference's joint network (model.py:158-166) + loss as one module.  Parameters follow Keras'
    Dense defaults: glorot-uniform kernels, zero biases."""

    def __init__(self, hidden: int, joint_size: int, vocab_size: int, blank_label: int = 0, visit_all: bool = False):
        super().__init__()
        self.blank_label = blank_label
        self.visit_all = visit_all  # RNNT_VISIT_ALL: no occupancy floor in the backward (rnnt_joint_loss)
        self.W1 = torch.nn.Parameter(torch.empty(hidden, joint_size))
        self.b1 = torch.nn.Parameter(torch.zeros(joint_size))
        self.W2 = torch.nn.Parameter(torch.empty(joint_size, vocab_size))
        self.b2 = torch.nn.Parameter(torch.zeros(vocab_size))
        for w in (self.W1, self.W2):
            lim = math.sqrt(6.0 / (w.shape[0] + w.shape[1]))
            torch.nn.init.uniform_(w, -lim, lim)

    def forward(self, enc, pred, labels, input_lengths, label_lengths):
        return rnnt_joint_loss(enc, pred, self.W1, self.b1, self.W2, self.b2, labels, input_lengths,
                               label_lengths, self.blank_label, visit_all=self.visit_all)

    def logits(self, enc, pred):
        """Unfused reference form (materialises [B,T,U,J] and [B,T,U,V] in torch); for tests and host-logic checks on CPU."""
        z = enc.unsqueeze(2) + pred.unsqueeze(1)
        return torch.tanh(z @ self.W1 + self.b1) @ self.W2 + self.b2

    def cell_logits(self, enc, pred, reuse_buffers: bool = False):
        """Joint logits [B, T, U, V] for decoding (utils/decoding.py:6-18).  On an MI355X this is the ENGINE at every vocabulary
        size (compute_rnnt_joint_net_logits / compute_rnnt_joint_logits: the fused loss's own forward kernels, first Dense layer
        included); CPU tensors -- the host-logic tests -- and shapes the kernels do not take (joint sizes beyond 704 / 640) use
        the torch composition.  The result is a fresh tensor unless `reuse_buffers` is set (see joint_logits: the buffer is then
        overwritten by the next call of the same shape -- the greedy decoder opts in, a beam search must not)."""
        if enc.is_cuda:
            try:
                padded_joint_shape(self.W2.shape[0], self.W2.shape[1], _auto_joint_dtype(*self.W2.shape))
            except ValueError:
                return self.logits(enc, pred)
            return joint_logits(enc, pred, self.W1, self.b1, self.W2, self.b2, reuse_buffers=reuse_buffers)
        return self.logits(enc, pred)
