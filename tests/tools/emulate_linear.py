"""Dev tool (no GPU needed): NumPy float32 emulation of the LINEAR-domain alpha/beta sweeps (round 4) -- the lattice as
mantissas x 2^(integer frame per sweep lane and block of R diagonals), the gradient set-up from mantissas and frames, and the
per-cell range certificate that decides whether an utterance stays on this path or is redone in the log domain.

  python tests/tools/emulate_linear.py [T,U,V] [R]     (R = diagonals per frame block; omitted: chosen as the sweeps choose it)

Prints, per input family: cost error, max|dgrad| against the float64 oracle, and the certificate margin (worst of the three
per-cell terms, in bits: <= CERT_BITS passes).  Follows csrc/rnnt_lin_kernels.hip closely enough for error LEVELS (not
bit-exact: the hardware's exp2 / rcp differ in the last ulp; NumPy keeps float32 denormals like the kernels do).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rnnt_oracle as orc  # noqa: E402
from tests.tools.emulate_sweep import make_inputs  # noqa: E402

F = np.float32
LOG2E = F(1.4426950408889634)
BIG = -1.0e30       # frame of a lane that holds no mass and has no neighbour to copy from
DRAG = 118.0        # a lane's frame is at most this far below the frames of the lanes mass can arrive from within a block
CERT_BITS = -40.0   # (csrc/rnnt_lin.h kCertBits) per-cell bound on (lost mass x other side) / likelihood, in bits
TINY_EDGE = -100.0  # an edge the lattice owns with log2 p below this is not representable safely: log-domain path


def edge_probs(x, labels, blank=0):
    """f32 edge probabilities as the lsm pass forms them: e_i / sum, e_i = 2^((x_i - m) log2e)."""
    x = x.astype(F)
    T, U, V = x.shape
    m = x.max(-1)
    e = np.exp2(((x - m[..., None]) * LOG2E).astype(F)).astype(F)
    s = e.sum(-1, dtype=F)
    inv = (F(1.0) / s).astype(F)
    pb = (e[:, :, blank] * inv).astype(F)
    pl = np.zeros((T, U), F)
    if U > 1:
        el = np.take_along_axis(e[:, :U - 1], labels[None, :U - 1, None].astype(np.int64), axis=2)[:, :, 0]
        pl[:, :U - 1] = (el * inv[:, :U - 1]).astype(F)
    pb2 = pb.copy()
    pb2[T - 1, :U - 1] = 0  # a blank from the last frame leaves the lattice unless terminal
    tiny = bool(((pb2 < F(2.0 ** TINY_EDGE)) & (np.arange(T)[:, None] < T - 1)).any() or
                (U > 1 and (pl[:, :U - 1] < F(2.0 ** TINY_EDGE)).any()))
    return pb2, pl, e, s, tiny


def _frexp_e(m):
    """exponent e with m = f 2^e, f in [0.5, 1); 0 for m == 0 (v_frexp_exp_i32_f32)."""
    _, e = np.frexp(m)
    return e.astype(np.float64)


def _renorm(a, E, R, K, beta):
    """a [L,K] mantissas, E [L] frames.  Every lane normalises against its own maximum; a lane's frame is dragged up so
    that whatever can arrive from the lanes mass comes from within one block fits; empty lanes copy (through the drag)."""
    L = a.shape[0]
    m = a.max(axis=1)
    nonempty = m > 0
    E_own = np.where(nonempty, E + _frexp_e(m), BIG)
    E_new = E_own.copy()
    look = (R + K - 1) // K
    for i in range(1, look + 1):
        sh = np.full(L, BIG)
        if beta:
            sh[:-i] = E_own[i:]
        else:
            sh[i:] = E_own[:-i]
        E_new = np.maximum(E_new, sh - DRAG)
    shift = np.where(nonempty, E - E_new, 0.0)
    a = np.ldexp(a, shift.astype(np.int64)[:, None]).astype(F)
    return a, E_new


def _cross(E, beta):
    """scale of what crosses a lane boundary: alpha lane l -> l+1: 2^(E[l]-E[l+1]); beta lane l <- l+1: 2^(E[l+1]-E[l])."""
    L = E.shape[0]
    d = np.zeros(L)
    d[:-1] = (E[1:] - E[:-1]) if beta else (E[:-1] - E[1:])
    d = np.where(np.abs(d) > 1e20, np.sign(d) * DRAG, d)
    return np.clip(d, -DRAG, DRAG)


def sweep_lin(pb, pl, K=3, R=8):
    """Returns mantissas mA, mB [T,U] f32, frames EA, EB [NC, L] (f64 integers), likelihood (mL, EL)."""
    T, U = pb.shape
    N = T + U - 1
    L = (U + K - 1) // K
    Up = L * K
    NC = N // R + 1
    ar = np.arange(Up)

    def diag_w(n):
        t = n - ar
        ok = (t >= 0) & (t < T) & (ar < U)
        tc, uc = np.clip(t, 0, T - 1), np.clip(ar, 0, U - 1)
        return (np.where(ok, pb[tc, uc], 0).astype(F).reshape(L, K), np.where(ok, pl[tc, uc], 0).astype(F).reshape(L, K), ok)

    # ---- alpha ----
    mA = np.zeros((T, U), F)
    EA = np.full((NC, L), BIG)
    a = np.zeros((L, K), F)
    a[0, 0] = 1
    E = np.zeros(L)
    a, E = _renorm(a, E, R, K, False)
    EA[0] = E
    dl = _cross(E, False)
    mA[0, 0] = a[0, 0]
    for n in range(1, N):
        wb, wl, _ = diag_w(n - 1)
        wl = wl.copy()
        wl[:, K - 1] = np.ldexp(wl[:, K - 1], dl.astype(np.int64)).astype(F)  # into the next lane's frame
        p = (a * wl).astype(F)
        left = np.zeros((L, K), F)
        left[:, 1:] = p[:, :-1]
        left[1:, 0] = p[:-1, K - 1]
        a = (a * wb + left).astype(F)  # fma: one rounding
        if n % R == 0:
            a, E = _renorm(a, E, R, K, False)
            EA[n // R] = E
            dl = _cross(E, False)
        t = n - ar
        ok = (t >= 0) & (t < T) & (ar < U)
        mA[t[ok], ar[ok]] = a.reshape(-1)[ok]
    lu, lj = (U - 1) // K, (U - 1) % K
    Lval = F(a[lu, lj]) * pb[T - 1, U - 1]
    mL, eL = np.frexp(F(Lval))
    EL = float(E[lu]) + float(eL)
    # ---- beta ----
    mB = np.zeros((T, U), F)
    EB = np.full((NC, L), BIG)
    b = np.zeros((L, K), F)
    b[lu, lj] = 1  # the virtual terminal node
    E = np.zeros(L)
    dl = np.zeros(L)
    first = True
    for n in range(N - 1, -1, -1):
        wb, wl, _ = diag_w(n)
        right = np.zeros((L, K), F)
        right[:, :-1] = b[:, 1:]
        right[:-1, K - 1] = b[1:, 0]
        wl = wl.copy()
        wl[:, K - 1] = np.ldexp(wl[:, K - 1], dl.astype(np.int64)).astype(F)
        b = (b * wb + (right * wl).astype(F)).astype(F)
        if n % R == R - 1 or first:
            b, E = _renorm(b, E, R, K, True)
            EB[n // R] = E
            dl = _cross(E, True)
            first = False
        t = n - ar
        ok = (t >= 0) & (t < T) & (ar < U)
        mB[t[ok], ar[ok]] = b.reshape(-1)[ok]
    mLb, eLb = np.frexp(F(b[0, 0]))
    return mA, mB, EA, EB, (float(mL), EL), (float(mLb), float(E[0]) + float(eLb))


def grad_lin(x, labels, e, s, mA, mB, EA, EB, lik, K, R, blank=0):
    """f32 gradient set-up from mantissas + frames; returns (g, worst certificate term in bits)."""
    T, U, V = x.shape
    mL, EL = lik
    tt, uu = np.meshgrid(np.arange(T), np.arange(U), indexing="ij")
    n = tt + uu
    kc, kc1 = n // R, np.minimum(n + 1, T + U - 2) // R
    l0, l1 = uu // K, np.minimum(uu + 1, U - 1) // K
    ea, eb = EA[kc, l0], EB[kc, l0]
    fa, xa_ = np.frexp(mA)  # mantissas may sit anywhere in the f32 range (a dragged frame): split before multiplying
    qa = (fa.astype(F) * (F(1.0) / F(mL))).astype(F)
    xa_ = xa_.astype(np.float64)

    def prod(mb, X):  # (mA / mL) * mb * 2^X without intermediate under- / overflow
        fb, xb_ = np.frexp(mb)
        X = np.clip(X + xa_ + xb_, -400, 400).astype(np.int64)  # (frames of empty lanes are -1e30)
        return np.ldexp((qa * fb.astype(F)).astype(F), X).astype(F)

    inv_s = (F(1.0) / s).astype(F)
    occ = prod(mB, ea + eb - EL)
    g = (e * (occ * inv_s)[..., None]).astype(F)
    mB_t1 = np.vstack([mB[1:], np.zeros((1, U), F)])
    cb = prod(mB_t1, ea + EB[kc1, l0] - EL)
    cb[T - 1, :] = 0
    cb[T - 1, U - 1] = np.ldexp(qa[T - 1, U - 1], int(np.clip(ea[T - 1, U - 1] + xa_[T - 1, U - 1] - EL, -400, 400)))
    g[:, :, blank] -= (cb * (e[:, :, blank] * inv_s)).astype(F)
    if U > 1:
        mB_u1 = np.hstack([mB[:, 1:], np.zeros((T, 1), F)])
        cl = prod(mB_u1, ea + EB[kc1, l1] - EL)[:, :U - 1]
        el = np.take_along_axis(e[:, :U - 1], labels[None, :U - 1, None].astype(np.int64), axis=2)[:, :, 0]
        corr = (cl * (el * inv_s[:, :U - 1])).astype(F)
        np.subtract.at(g, (tt[:, :U - 1], uu[:, :U - 1], np.broadcast_to(labels[None, :U - 1], (T, U - 1))), corr)
    # certificate (bits): what a flush below 2^-126 of the cell's frame can cost, times the other side, over the likelihood
    ELf = EL + float(np.frexp(F(mL))[1])
    with np.errstate(divide="ignore"):
        xa = np.where(mA > 0, ea + _frexp_e(mA), -np.inf)
        xb = np.where(mB > 0, eb + _frexp_e(mB), -np.inf)
    t1 = ea - 126 + xb - ELf
    t2 = eb - 126 + xa - ELf
    t3 = ea + eb - 252 - ELf
    worst = float(np.max(np.maximum(np.maximum(t1, t2), t3)))
    return g, worst


DECAY_BITS = 6.2  # (csrc/rnnt_lin.h kLinDecayBits) mean -log2 max(p_blank, p_label) beyond which the sweeps take blocks of four


def decay_statistic(pb, pl):
    """What the lsm pass leaves for the sweeps: mean over the cells of -log2 of the better edge's probability."""
    with np.errstate(divide="ignore"):
        return float((-np.log2(np.maximum(np.maximum(pb, pl), F(1e-37)))).mean())


def run(kind, T, U, V, seed, K, R):
    rng = np.random.default_rng(seed)
    x, labels = make_inputs(kind, T, U, V, rng)
    c_ref, g_ref, _, _, _ = orc.utterance_cost_and_grad(x, labels)
    pb, pl, e, s, tiny = edge_probs(x, labels)
    stat = decay_statistic(pb, pl)
    if R == 0:  # the kernels' own choice
        R = 4 if (K == 1 or K >= 12 or stat > DECAY_BITS) else 8
    mA, mB, EA, EB, lik, likb = sweep_lin(pb, pl, K, R)
    g, worst = grad_lin(x.astype(F), labels, e, s, mA, mB, EA, EB, lik, K, R)
    cost = -(np.log2(lik[0]) + lik[1]) * np.log(2.0)
    costb = -(np.log2(likb[0]) + likb[1]) * np.log(2.0)
    return dict(cost=c_ref, dcost=abs(cost - c_ref) / max(1.0, abs(c_ref)), dab=abs(cost - costb) / max(1.0, abs(c_ref)),
                dgrad=float(np.abs(g - g_ref).max()), cert=worst, tiny=tiny, stat=stat, R=R)


if __name__ == "__main__":
    T, U, V = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "600,150,28").split(","))
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: chosen per utterance from the decay statistic, as the sweeps do
    k = (U + 63) // 64
    K = next(a for a in (1, 2, 3, 4, 6, 8, 12, 16) if k <= a)
    kinds = ["sigma1", "sigma4", "sigma8", "trained10", "trained10late", "trained20late"]
    print(f"T={T} U={U} V={V} K={K} R={R or 'auto'}: linear-domain lattice vs the float64 oracle")
    for kind in kinds:
        for seed in (1, 2):
            r = run(kind, T, U, V, seed, K, R)
            print(f"{kind:14s} seed {seed} decay {r['stat']:5.2f} bits -> blocks of {r['R']}  cost {r['cost']:10.2f} dcost {r['dcost']:.1e} |a-b| {r['dab']:.1e} "
                  f"dgrad {r['dgrad']:.2e} cert {r['cert']:7.1f} bits {'PASS' if r['cert'] <= CERT_BITS and not r['tiny'] else 'LOG-DOMAIN'}"
                  f"{' (tiny edge)' if r['tiny'] else ''}")
