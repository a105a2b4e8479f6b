"""Float64 NumPy oracle for the RNN-T joint + transducer-loss hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.

PARITY STATUS: *unpinned by the reference itself*.  The arithmetic of this path
lives in the un-vendored ``warp-transducer`` submodule
(``/root/reference/.gitmodules:1-3``, commit unknowable, directory empty), the
reference ships no tests, and its Python cannot be imported here (TensorFlow is
absent).  This oracle is therefore a restatement of the *published* algorithm
(Graves 2012, "Sequence Transduction with Recurrent Neural Networks",
eqs. 16-20) anchored on

* the reference's own call sites and boundary semantics
  (``utils/loss.py:24-36``, ``run_rnnt.py:262-278``, ``model.py:158-166``,
  ``utils/preprocessing.py:177-183``, ``utils/vocabulary.py:3-6``), and
* one upstream known-answer vector (warp-transducer ``tests/test_cpu.cpp``
  ``small_test``) committed as
  ``tests/golden/kat_small.json``, plus finite differences and the
  alpha-side/beta-side likelihood identity.

Conventions:
  acts    [B, T, U, V]  joint logits (or log-probs), U = L_max + 1
  labels  [B, U-1]      int, padded with anything (reference pads with 0)
  T_b = input_lengths[b], L_b = label_lengths[b], U_b = L_b + 1
  blank   = 0 in the reference (``utils/vocabulary.py:3-6``; the op's default)
  cost_b  = -ln P(y_b | x_b);   padded cells get exactly zero gradient.
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


# --------------------------------------------------------------------------
# log-softmax (reference: utils/loss.py:29-30 applies tf.nn.log_softmax on
# non-CUDA builds; the GPU op fuses it: SURVEY.md a-6)
# --------------------------------------------------------------------------
def log_softmax(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def _lse2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """log(exp(a)+exp(b)) with -inf handling."""
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore"):
        r = m + np.log1p(np.exp(-np.abs(a - b)))
    return np.where(np.isneginf(m), NEG_INF, r)


# --------------------------------------------------------------------------
# single-utterance lattice, evaluated by anti-diagonals
# --------------------------------------------------------------------------
def _gather(lp: np.ndarray, labels: np.ndarray, blank: int):
    """lp [T,U,V] log-probs -> (lpb [T,U], lpl [T,U-1])."""
    T, U, _ = lp.shape
    lpb = lp[:, :, blank]
    if U > 1:
        lpl = np.take_along_axis(
            lp[:, : U - 1, :], np.asarray(labels[: U - 1], dtype=np.int64)[None, :, None], axis=2
        )[:, :, 0]
    else:
        lpl = np.zeros((T, 0), dtype=np.float64)
    return lpb, lpl


def alphas(lpb: np.ndarray, lpl: np.ndarray):
    """alpha(0,0)=0; alpha(t,u)=lse(alpha(t-1,u)+lpb(t-1,u), alpha(t,u-1)+lpl(t,u-1)).
    Returns (alpha [T,U], ll = alpha(T-1,U-1)+lpb(T-1,U-1))."""
    T, U = lpb.shape
    a = np.full((T, U), NEG_INF)
    a[0, 0] = 0.0
    for n in range(1, T + U - 1):
        u = np.arange(max(0, n - T + 1), min(n, U - 1) + 1)
        t = n - u
        up = np.full(u.shape, NEG_INF)
        lf = np.full(u.shape, NEG_INF)
        mt = t >= 1
        up[mt] = a[t[mt] - 1, u[mt]] + lpb[t[mt] - 1, u[mt]]
        mu = u >= 1
        lf[mu] = a[t[mu], u[mu] - 1] + lpl[t[mu], u[mu] - 1]
        a[t, u] = _lse2(up, lf)
    return a, a[T - 1, U - 1] + lpb[T - 1, U - 1]


def betas(lpb: np.ndarray, lpl: np.ndarray):
    """beta(T-1,U-1)=lpb(T-1,U-1); beta(t,u)=lse(beta(t+1,u)+lpb(t,u), beta(t,u+1)+lpl(t,u)).
    Returns (beta [T,U], ll = beta(0,0))."""
    T, U = lpb.shape
    b = np.full((T, U), NEG_INF)
    b[T - 1, U - 1] = lpb[T - 1, U - 1]
    for n in range(T + U - 3, -1, -1):
        u = np.arange(max(0, n - T + 1), min(n, U - 1) + 1)
        t = n - u
        dn = np.full(u.shape, NEG_INF)
        rt = np.full(u.shape, NEG_INF)
        mt = t + 1 < T
        dn[mt] = b[t[mt] + 1, u[mt]] + lpb[t[mt], u[mt]]
        mu = u + 1 < U
        rt[mu] = b[t[mu], u[mu] + 1] + lpl[t[mu], u[mu]]
        b[t, u] = _lse2(dn, rt)
    return b, b[0, 0]


def utterance_cost_and_grad(x, labels, blank=0, fused_softmax=True):
    """One utterance, exact lengths.  x [T,U,V] (logits if fused_softmax else log-probs).

    Returns (cost, grad [T,U,V], alpha, beta).
    fused_softmax=True  -> gradient w.r.t. *logits*  (the GPU-op convention)
    fused_softmax=False -> gradient w.r.t. *log-probs* (the CPU-op convention: only the
                           blank and label entries are non-zero)
    """
    x = np.asarray(x, dtype=np.float64)
    T, U, V = x.shape
    lp = log_softmax(x) if fused_softmax else x
    lpb, lpl = _gather(lp, labels, blank)
    a, ll_f = alphas(lpb, lpl)
    b, ll_b = betas(lpb, lpl)
    ll = ll_f
    g = np.zeros((T, U, V))
    # occupancy-weighted transition posteriors
    with np.errstate(invalid="ignore", over="ignore"):
        # blank transitions (t,u)->(t+1,u), and the terminal blank at (T-1,U-1)
        gb = np.zeros((T, U))
        if T > 1:
            gb[: T - 1, :] = np.exp(a[: T - 1, :] + lpb[: T - 1, :] + b[1:, :] - ll)
        gb[T - 1, U - 1] = np.exp(a[T - 1, U - 1] + lpb[T - 1, U - 1] - ll)
        gl = np.zeros((T, max(U - 1, 0)))
        if U > 1:
            gl = np.exp(a[:, : U - 1] + lpl + b[:, 1:] - ll)
    gb = np.nan_to_num(gb, nan=0.0)
    gl = np.nan_to_num(gl, nan=0.0)
    if fused_softmax:
        with np.errstate(invalid="ignore"):
            occ = np.exp(a + b - ll)  # gamma(t,u)
        occ = np.nan_to_num(occ, nan=0.0)
        g = occ[:, :, None] * np.exp(lp)
    g[:, :, blank] -= gb
    if U > 1:
        idx = np.asarray(labels[: U - 1], dtype=np.int64)
        tt = np.arange(T)[:, None]
        uu = np.arange(U - 1)[None, :]
        np.subtract.at(g, (tt, uu, idx[None, :]), gl)
    return -ll, g, a, b, ll_b


def rnnt_loss_and_grad(acts, labels, input_lengths, label_lengths, blank=0, fused_softmax=True):
    """Batched oracle with ragged lengths.  Returns (costs [B] f64, grads [B,T,U,V] f64).

    Mirrors the op contract at utils/loss.py:34-35 (costs) and the TF op's second
    output (grads, zero in padded cells)."""
    acts = np.asarray(acts)
    B, T, U, V = acts.shape
    costs = np.zeros(B)
    grads = np.zeros((B, T, U, V))
    for i in range(B):
        Tb = int(input_lengths[i])
        Ub = int(label_lengths[i]) + 1
        if not (1 <= Tb <= T and 1 <= Ub <= U):
            raise ValueError("length out of range")
        c, g, _, _, _ = utterance_cost_and_grad(
            acts[i, :Tb, :Ub, :], np.asarray(labels[i])[: Ub - 1], blank, fused_softmax
        )
        costs[i] = c
        grads[i, :Tb, :Ub, :] = g
    return costs, grads


# --------------------------------------------------------------------------
# joint network (reference: model.py:158-166; decode twin utils/decoding.py:6-18)
# --------------------------------------------------------------------------
def joint_forward(enc, pred, W1, b1, W2, b2):
    """enc [B,T,H], pred [B,U,H] -> logits [B,T,U,V], plus h [B,T,U,J] for backward.
    z0 = enc[:,:,None,:] + pred[:,None,:,:]      (model.py:158-160)
    h  = tanh(z0 @ W1 + b1)                       (model.py:162-163)
    y  = h @ W2 + b2                              (model.py:165-166)"""
    enc = np.asarray(enc, np.float64)
    pred = np.asarray(pred, np.float64)
    z0 = enc[:, :, None, :] + pred[:, None, :, :]
    h = np.tanh(z0 @ np.asarray(W1, np.float64) + np.asarray(b1, np.float64))
    y = h @ np.asarray(W2, np.float64) + np.asarray(b2, np.float64)
    return y, h


def joint_backward(dlogits, enc, pred, W1, b1, W2, b2, h=None):
    """Exact backward of joint_forward (what TF autodiff does at run_rnnt.py:284)."""
    enc = np.asarray(enc, np.float64)
    pred = np.asarray(pred, np.float64)
    W1 = np.asarray(W1, np.float64)
    W2 = np.asarray(W2, np.float64)
    if h is None:
        _, h = joint_forward(enc, pred, W1, b1, W2, b2)
    dl = np.asarray(dlogits, np.float64)
    J = W1.shape[1]
    V = W2.shape[1]
    dW2 = h.reshape(-1, J).T @ dl.reshape(-1, V)
    db2 = dl.reshape(-1, V).sum(0)
    dh = dl @ W2.T
    dz = dh * (1.0 - h * h)
    db1 = dz.reshape(-1, J).sum(0)
    d_a = dz.sum(axis=2)  # [B,T,J]  reduce over u
    d_c = dz.sum(axis=1)  # [B,U,J]  reduce over t
    H = W1.shape[0]
    dW1 = enc.reshape(-1, H).T @ d_a.reshape(-1, J) + pred.reshape(-1, H).T @ d_c.reshape(-1, J)
    d_enc = d_a @ W1.T
    d_pred = d_c @ W1.T
    return dict(d_enc=d_enc, d_pred=d_pred, dW1=dW1, db1=db1, dW2=dW2, db2=db2,
                d_a=d_a, d_c=d_c)


def joint_loss_and_grads(enc, pred, W1, b1, W2, b2, labels, input_lengths, label_lengths,
                         blank=0, cost_scale=None):
    """Fused path oracle: joint -> transducer loss -> gradients of sum_b(scale_b*cost_b).
    cost_scale defaults to 1 per utterance (the train step uses 1/global_batch,
    run_rnnt.py:278)."""
    y, h = joint_forward(enc, pred, W1, b1, W2, b2)
    costs, g = rnnt_loss_and_grad(y, labels, input_lengths, label_lengths, blank, True)
    B = y.shape[0]
    s = np.ones(B) if cost_scale is None else np.broadcast_to(np.asarray(cost_scale, np.float64), (B,))
    g = g * s[:, None, None, None]
    out = joint_backward(g, enc, pred, W1, b1, W2, b2, h)
    out["costs"] = costs
    out["dlogits"] = g
    return out


def _rne_half(x):
    """Round to IEEE binary16 (round-to-nearest-even, what v_cvt_f16_f32 does) and widen back to f64."""
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def dl_scale_f16(cost_scale, B):
    """Power-of-two scale the f16 joint applies to dlogits before rounding them to half: 2^(14 - ceil(log2 max|s|))."""
    s = np.ones(B) if cost_scale is None else np.broadcast_to(np.asarray(cost_scale, np.float64), (B,))
    m = float(np.abs(s).max())
    e = int(np.ceil(np.log2(m))) if m > 0 else 0
    return 2.0 ** (14 - e)


_LOG2E = 1.4426950408889634


def park_factor_f16(y, labels, blank, has_label=None):
    """What parking the softmax numerators in binary16 does to the loss gradient w.r.t. the logits, as a factor per logit.

    The forward kernel of the f16 joint keeps, per (cell, chunk of 32 symbols), 2^(y log2 e - R) rounded to binary16, R = the
    integer at or above the chunk's largest y log2 e; the backward pass multiplies them back (one f32 factor per chunk).
    occupancy * softmax therefore carries the relative rounding error of its parked numerator: factor = rne(p) / p.  The
    blank column of every cell and the label column of the cells that have a label edge are formed from the unrounded
    edge logits instead (factor 1).  y [..., U, V] natural-log logits, labels [..., U-1] broadcastable to y's leading axes,
    has_label (same shape, optional): which of them exist (u < L_b)."""
    y2 = np.asarray(y, np.float64) * _LOG2E
    V = y2.shape[-1]
    Vp = (V + 31) // 32 * 32  # (the kernels only take whole chunks; a ragged last chunk is a chunk of its own here)
    ch = np.full(y2.shape[:-1] + (Vp,), -np.inf)
    ch[..., :V] = y2
    ch = ch.reshape(y2.shape[:-1] + (Vp // 32, 32))
    R = np.ceil(ch.max(axis=-1, keepdims=True))
    pe = np.exp2(ch - R)
    with np.errstate(invalid="ignore", divide="ignore"):
        f = np.where(pe > 0, _rne_half(pe) / pe, 1.0).reshape(y2.shape[:-1] + (Vp,))[..., :V].copy()
    f[..., blank] = 1.0
    U = y2.shape[-2]
    if U > 1:
        shape = y2.shape[:-2] + (U - 1,)
        lab = np.broadcast_to(np.asarray(labels, np.int64)[..., : U - 1], shape)
        sub = f[..., : U - 1, :]
        cur = np.take_along_axis(sub, lab[..., None], axis=-1)
        if has_label is not None:
            new = np.where(np.broadcast_to(np.asarray(has_label, bool)[..., : U - 1], shape)[..., None], 1.0, cur)
        else:
            new = np.ones_like(cur)
        np.put_along_axis(sub, lab[..., None], new, axis=-1)
        f[..., blank] = 1.0
    return f


def joint_loss_and_grads_f16(enc, pred, W1, b1, W2, b2, labels, input_lengths, label_lengths,
                             blank=0, cost_scale=None, parked=True):
    """Oracle of the f16-MFMA joint (BASELINE config 5: "fp16 joint MFMA", fp32 lattice).

    Same mathematics as joint_loss_and_grads (model.py:158-166 + utils/loss.py:24-36 + autodiff), with the operand
    roundings the MFMA path makes stated explicitly: h = tanh(.) and W2 are rounded to binary16 before the J x V
    product (exact products, wide accumulation), and the loss gradient w.r.t. the logits is scaled by a power of two
    and rounded to binary16 before the two backward products (dh = dl . W2^T, dW2 = h^T . dl, db2 = sum dl).
    tanh' = 1 - h^2 uses the unrounded h (straight-through for the rounding).  parked=True (what a forward + backward
    pair does): the softmax numerators went through binary16 on their way from the forward to the backward pass
    (park_factor_f16); parked=False: the backward pass recomputed the logits (a second backward over one forward)."""
    enc = np.asarray(enc, np.float64)
    pred = np.asarray(pred, np.float64)
    W1 = np.asarray(W1, np.float64)
    z0 = enc[:, :, None, :] + pred[:, None, :, :]
    h = np.tanh(z0 @ W1 + np.asarray(b1, np.float64))
    hq = _rne_half(h)
    W2q = _rne_half(W2)
    y = hq @ W2q + np.asarray(b2, np.float64)
    costs, g = rnnt_loss_and_grad(y, labels, input_lengths, label_lengths, blank, True)
    B = y.shape[0]
    s = np.ones(B) if cost_scale is None else np.broadcast_to(np.asarray(cost_scale, np.float64), (B,))
    g = g * s[:, None, None, None]
    if parked:
        lab = np.asarray(labels, np.int64).reshape(B, -1)
        U = y.shape[2]
        lab = lab[:, : U - 1] if lab.shape[1] >= U - 1 else np.zeros((B, U - 1), np.int64)
        has = np.arange(U - 1)[None, :] < np.asarray(label_lengths, np.int64).reshape(B, 1)
        g = g * park_factor_f16(y, np.clip(lab, 0, y.shape[-1] - 1)[:, None, :], blank, has[:, None, :])
    S = dl_scale_f16(cost_scale, B)
    gq = _rne_half(g * S) / S
    J, V = W2q.shape
    dW2 = hq.reshape(-1, J).T @ gq.reshape(-1, V)
    db2 = gq.reshape(-1, V).sum(0)
    dh = gq @ W2q.T
    dz = dh * (1.0 - h * h)
    db1 = dz.reshape(-1, J).sum(0)
    d_a = dz.sum(axis=2)
    d_c = dz.sum(axis=1)
    H = W1.shape[0]
    dW1 = enc.reshape(-1, H).T @ d_a.reshape(-1, J) + pred.reshape(-1, H).T @ d_c.reshape(-1, J)
    return dict(costs=costs, d_enc=d_a @ W1.T, d_pred=d_c @ W1.T, dW1=dW1, db1=db1, dW2=dW2, db2=db2,
                d_a=d_a, d_c=d_c, dlogits=gq)


# --------------------------------------------------------------------------
# One utterance of the fused path at ANY size, in row chunks (memory-light): used by the parity tests
# that run at BASELINE.json's full configurations (C2 fused: T600 U150 J640 V28; C5: T1500 U300 J640 V1024),
# where the [T, U, J] and [T, U, V] tensors of one utterance do not fit comfortably in float64.
# Same mathematics as joint_loss_and_grads / joint_loss_and_grads_f16 (model.py:162-166 after the exact
# factorisation of the first Dense layer into enc_proj + pred_proj, utils/loss.py:24-36, autodiff of both),
# evaluated in two passes over the lattice rows: (1) log-probs of the blank/label edges -> alpha, beta;
# (2) dlogits chunk by chunk -> dW2, db2, d enc_proj, d pred_proj.
# --------------------------------------------------------------------------
def joint_utterance_streamed(enc_proj, pred_proj, W2, b2, labels, blank=0, cost_scale=1.0, f16=False,
                             dl_scale=None, rows_per_chunk=None, want_grads=True, parked=True):
    """enc_proj [T, J] (= enc @ W1 + b1), pred_proj [U, J] (= pred @ W1), W2 [J, V], b2 [V], labels [U-1], exact lengths.

    f16=True states the operand roundings of the f16-MFMA joint (h, W2, the parked softmax numerators and the scaled
    dlogits to binary16, see joint_loss_and_grads_f16); dl_scale is that path's power-of-two dlogits scale (dl_scale_f16 of the WHOLE batch's
    cost_scale).  Returns dict(cost, d_enc_proj [T,J], d_pred_proj [U,J], dW2, db2)."""
    A = np.asarray(enc_proj, np.float64)
    C = np.asarray(pred_proj, np.float64)
    W = np.asarray(W2, np.float64)
    bias = np.asarray(b2, np.float64)
    T, J = A.shape
    U = C.shape[0]
    V = W.shape[1]
    lab = np.asarray(labels, np.int64)[: U - 1]
    Wq = _rne_half(W) if f16 else W
    if rows_per_chunk is None:
        rows_per_chunk = max(1, int(2.5e7 // (U * max(J, V))))  # ~200 MB per [rows, U, max(J, V)] float64 array

    def rows(t0, t1):
        h = np.tanh(A[t0:t1, None, :] + C[None, :, :])           # [r, U, J]
        hq = _rne_half(h) if f16 else h
        y = hq @ Wq + bias                                        # [r, U, V]
        return h, hq, log_softmax(y), (park_factor_f16(y, lab, blank) if (f16 and parked) else None)

    lpb = np.empty((T, U))
    lpl = np.empty((T, max(U - 1, 0)))
    for t0 in range(0, T, rows_per_chunk):
        t1 = min(T, t0 + rows_per_chunk)
        _, _, lp, _ = rows(t0, t1)
        lpb[t0:t1] = lp[:, :, blank]
        if U > 1:
            lpl[t0:t1] = np.take_along_axis(lp[:, : U - 1, :], lab[None, :, None], axis=2)[:, :, 0]
    a, ll = alphas(lpb, lpl)
    if not want_grads:
        return dict(cost=-ll)
    b, _ = betas(lpb, lpl)
    s = float(cost_scale)
    S = float(dl_scale) if (f16 and dl_scale is not None) else 1.0
    dW2 = np.zeros((J, V))
    db2 = np.zeros(V)
    d_a = np.zeros((T, J))
    d_c = np.zeros((U, J))
    uu = np.arange(max(U - 1, 0))
    for t0 in range(0, T, rows_per_chunk):
        t1 = min(T, t0 + rows_per_chunk)
        h, hq, lp, pf = rows(t0, t1)
        r = t1 - t0
        with np.errstate(invalid="ignore", over="ignore"):
            occ = np.nan_to_num(np.exp(a[t0:t1] + b[t0:t1] - ll), nan=0.0)
            g = occ[:, :, None] * np.exp(lp)
            gb = np.zeros((r, U))
            tt = np.arange(t0, t1)
            inner = tt < T - 1
            if inner.any():
                gb[inner] = np.exp(a[tt[inner]] + lpb[tt[inner]] + b[tt[inner] + 1] - ll)
            if t1 == T:
                gb[r - 1, U - 1] = np.exp(a[T - 1, U - 1] + lpb[T - 1, U - 1] - ll)
            gb = np.nan_to_num(gb, nan=0.0)
            g[:, :, blank] -= gb
            if U > 1:
                gl = np.nan_to_num(np.exp(a[t0:t1, : U - 1] + lpl[t0:t1] + b[t0:t1, 1:] - ll), nan=0.0)
                np.subtract.at(g, (np.arange(r)[:, None], uu[None, :], lab[None, :]), gl)
        g *= s
        if pf is not None:
            g *= pf
        if f16:
            g = _rne_half(g * S) / S
        g2 = g.reshape(-1, V)
        dW2 += hq.reshape(-1, J).T @ g2
        db2 += g2.sum(0)
        dz = (g @ Wq.T) * (1.0 - h * h)
        d_a[t0:t1] = dz.sum(axis=1)
        d_c += dz.sum(axis=0)
    return dict(cost=-ll, d_enc_proj=d_a, d_pred_proj=d_c, dW2=dW2, db2=db2)
