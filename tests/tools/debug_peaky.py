"""Dev tool (GPU box): where does the gradient error of a peaked-logit utterance come from?  Runs the op through the C ABI
with a caller-owned workspace, reads the lattice state back (alpha~, beta~, offset tables, ll) and compares alpha, beta, the
per-cell exponents and the gradient with the float64 oracle at the worst cell."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rnnt_speech_recognition_amd as pkg  # noqa: E402
from oracle import rnnt_oracle as orc  # noqa: E402
from rnnt_speech_recognition_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
pkg.build()
lib = _lib.load()


def layout(T, U, B):
    N = T + U - 1
    Nr = (N + 15) // 16 * 16
    k = (U + 63) // 64
    K = next(a for a in (1, 2, 3, 4, 6, 8, 12, 16) if k <= a)
    Up = 64 * K
    NC = Nr // 8 + 1
    NG = 64 if os.environ.get("DEBUG_OFFSETS", "lane") == "lane" else Up // 64
    off = 0
    out = {}
    for name, n in (("lse", B * T * U * 4), ("W", B * Nr * 2 * Up * 4), ("A", B * Nr * Up * 4), ("Bt", B * Nr * Up * 4),
                    ("offA", B * NC * NG * 4), ("offB", B * NC * NG * 4), ("ll", B * 2 * 8)):
        out[name] = (off, n)
        off = (off + n + 255) // 256 * 256
    return out, dict(N=N, Nr=Nr, K=K, Up=Up, NC=NC, NG=NG)


def alpha_from_weights(W, T, U):
    """float64 alpha (log2 domain) from the GPU's own f32 edge weights W[n][u][2] (diagonal-major): isolates the sweep's
    rounding from the rounding of the weights."""
    a = np.full((T, U), -np.inf)
    a[0, 0] = 0.0
    Wd = W.astype(np.float64)
    Wd[Wd < -1e29] = -np.inf
    for n in range(1, T + U - 1):
        u = np.arange(max(0, n - T + 1), min(n, U - 1) + 1)
        t = n - u
        up = np.full(u.shape, -np.inf)
        lf = np.full(u.shape, -np.inf)
        mt = t >= 1
        up[mt] = a[t[mt] - 1, u[mt]] + Wd[n - 1, u[mt], 0]
        mu = u >= 1
        lf[mu] = a[t[mu], u[mu] - 1] + Wd[n - 1, u[mu] - 1, 1]
        m = np.maximum(up, lf)
        with np.errstate(invalid="ignore"):
            a[t, u] = np.where(np.isneginf(m), -np.inf, m + np.log2(np.exp2(up - m) + np.exp2(lf - m)))
    return a


def run(sigma, seed, T=600, U=150, V=28, batch=None):
    rng = np.random.default_rng(seed)
    if batch is None:
        x = (sigma * rng.normal(size=(1, T, U, V))).astype(np.float32)
        labels = rng.integers(1, V, size=(1, U - 1)).astype(np.int32)
    else:  # the data of tests/test_peaky_gpu.py::make_logits (B = 32), utterance `batch`
        labels = rng.integers(1, V, size=(32, U - 1)).astype(np.int32)
        x = rng.normal(size=(32, T, U, V)).astype(np.float32) * np.float32(sigma)
        x, labels = x[batch:batch + 1], labels[batch:batch + 1]
    B = 1
    acts = torch.tensor(x, device=dev)
    grads = torch.empty_like(acts)
    lab, il, ll = (torch.tensor(a, device=dev) for a in (labels, np.array([T], np.int32), np.array([U - 1], np.int32)))
    costs = torch.empty(B, device=dev)
    ws = torch.zeros(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
    opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, 0, T, U)
    _lib.check(lib.compute_rnnt_loss(acts.data_ptr(), grads.data_ptr(), lab.data_ptr(), ll.data_ptr(), il.data_ptr(), V, B,
                                     costs.data_ptr(), ws.data_ptr(), opts), "loss")
    torch.cuda.synchronize()
    lay, d = layout(T, U, B)
    raw = ws.cpu().numpy()
    get = lambda name, dt: raw[lay[name][0]: lay[name][0] + lay[name][1]].view(dt)
    A = get("A", np.float32).reshape(d["Nr"], d["Up"])
    Bt = get("Bt", np.float32).reshape(d["Nr"], d["Up"])
    offA = get("offA", np.float32).reshape(d["NC"], d["NG"])
    offB = get("offB", np.float32).reshape(d["NC"], d["NG"])
    Wg = get("W", np.float32).reshape(d["Nr"], d["Up"], 2)
    OG = d["K"] if d["NG"] == 64 else 64
    ll2 = get("ll", np.float64)
    print(f"sigma {sigma} seed {seed}: K={d['K']}  offA block 40 lanes 0..9: {offA[40, :10]}  distinct offsets in that block: "
          f"{len(np.unique(offA[40]))}")
    c_ref, g_ref, al, be, lp = orc.utterance_cost_and_grad(x[0], labels[0])
    g = grads.cpu().numpy()[0]
    err = np.abs(g - g_ref)
    t, u, v = np.unravel_index(err.argmax(), err.shape)
    n, K = t + u, OG
    a_gpu = (float(A[n, u]) + float(offA[n // 8, u // K])) * np.log(2)
    b_gpu = (float(Bt[n, u]) + float(offB[n // 8, u // K])) * np.log(2)
    print(f"  cost gpu {float(costs[0]):.4f} ref {c_ref:.4f}; max|dgrad| {err.max():.3e} at (t,u,v)=({t},{u},{v}) grad_ref {g_ref[t,u,v]:.5f}")
    print(f"  alpha: gpu {a_gpu:.6f} ref {al[t,u]:.6f} d {a_gpu-al[t,u]:.2e} | beta: gpu {b_gpu:.6f} ref {be[t,u]:.6f} d {b_gpu-be[t,u]:.2e} "
          f"| ll gpu {ll2[0]*np.log(2):.6f} / {ll2[1]*np.log(2):.6f} ref {-c_ref:.6f}")
    print(f"  residues there: alpha~ {A[n,u]:.3f} beta~ {Bt[n,u]:.3f}; occupancy log (ref) {al[t,u]+be[t,u]+c_ref:.4f}")
    # error of alpha / beta over the whole lattice (true = residue + offset)
    tt, uu = np.meshgrid(np.arange(T), np.arange(U), indexing="ij")
    nn = tt + uu
    At = (A[nn, uu].astype(np.float64) + offA[nn // 8, uu // K]) * np.log(2)
    Btt = (Bt[nn, uu].astype(np.float64) + offB[nn // 8, uu // K]) * np.log(2)
    occ = al + be + c_ref
    live = occ > -12
    aw = alpha_from_weights(Wg, T, U) * np.log(2)
    print(f"  alpha: sweep error (gpu - f64 sweep over the gpu's f32 weights) max {np.abs(At - aw)[live].max():.2e}; "
          f"weight-rounding error (f64 sweep over f32 weights - oracle) max {np.abs(aw - al)[live].max():.2e}")
    print(f"  over cells with occupancy > e^-12: max|d alpha| {np.abs(At-al)[live].max():.2e}  max|d beta| {np.abs(Btt-be)[live].max():.2e} "
          f" max|d(alpha+beta)| {np.abs(At+Btt-al-be)[live].max():.2e}; max|residue| alpha {np.abs(A[nn,uu])[live].max():.1f} beta {np.abs(Bt[nn,uu])[live].max():.1f}")


for sigma, seed in ((8.0, 81), (8.0, 3), (1.0, 5)):
    run(sigma, seed)
for b in (4, 12):
    run(8.0, 81, batch=b)
