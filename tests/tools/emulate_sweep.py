"""Dev tool (no GPU needed): NumPy float32 emulation of the shipped alpha/beta sweeps + gradient set-up, to compare
re-basing references before spending GPU time.  Follows csrc/rnnt_kernels.hip (log2 domain, finite log zero, integer
re-basing every kRebase diagonals, offsets on the side, gradient from f32 residues + f64 offset differences) closely
enough to reproduce the error LEVELS the GPU shows (not bit-exact: hardware exp2/log2 differ in the last ulp).

  python tests/tools/emulate_sweep.py            # table: max|dgrad| per input family and reference rule
Rules: "ridge" / "follow" (one offset per block of diagonals), "laneK" (one integer offset per K columns, float32 recurrence),
"laneKf64" (the float64 recurrence the loss op uses wherever it falls back to the log domain since round 4).
"""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rnnt_oracle as orc  # noqa: E402

F = np.float32
NEG = F(-1.0e30)
LOG2E = F(1.4426950408889634)
KREB = 8


def lse2_f32(a, b):
    d = (a - b).astype(F)
    e = np.exp2(-np.abs(d)).astype(F)
    return (np.maximum(a, b) + np.log2((F(1.0) + e).astype(F)).astype(F)).astype(F)


def edge_weights(x, labels, blank=0):
    """f32 W as the lsm pass forms it: (x - m) * log2e - log2(sum)."""
    x = x.astype(F)
    T, U, V = x.shape
    m = x.max(-1)
    s = np.exp2(((x - m[..., None]) * LOG2E).astype(F)).astype(F).sum(-1, dtype=F)
    lg2s = np.log2(s).astype(F)
    lse = (m + F(0.6931471805599453) * lg2s).astype(F)
    wb = ((x[:, :, blank] - m) * LOG2E - lg2s).astype(F)
    wl = np.full((T, U), NEG, F)
    if U > 1:
        xl = np.take_along_axis(x[:, :U - 1], labels[None, :U - 1, None].astype(np.int64), axis=2)[:, :, 0]
        wl[:, :U - 1] = ((xl - m[:, :U - 1]) * LOG2E - lg2s[:, :U - 1]).astype(F)
    wb2 = wb.copy()
    wb2[T - 1, :U - 1] = NEG  # blank from the last frame leaves the lattice unless terminal
    return wb2, wl, lse


def pick_ref(v, us, n, rule, slope, ridge_u, sign):
    fin = v > F(-1e29)
    if rule == "ridge":
        return ridge_u
    if rule == "max":
        key = np.where(fin, v, -np.inf)
    else:  # follow: alpha - s*u (alpha side), beta + s*u (beta side)
        key = np.where(fin, v - sign * F(slope) * us.astype(F), -np.inf)
    return int(us[int(np.argmax(key))])


def sweep(wb, wl, rule, kreb=KREB):
    """Returns alpha~, beta~ [T,U] residues (f32), offsets per diagonal (f64) for both, ll2 (alpha side, f64)."""
    T, U = wb.shape
    N = T + U - 1
    slope = np.log2((T - 1) / (U - 1)) if (T > 1 and U > 1) else 0.0
    ridge = lambda n: int(round(n * (U - 1) / (N - 1))) if N > 1 else 0
    A = np.full((T, U), NEG, F)
    offA = np.zeros(N)
    A[0, 0] = 0
    off = 0.0
    cur = np.full(U, NEG, F)
    cur[0] = 0
    for n in range(1, N):
        # diagonal n from n-1: cur[u] is alpha~(n-1-u, u)
        tprev = (n - 1) - np.arange(U)
        valid_prev = (tprev >= 0) & (tprev < T)
        wbp = np.where(valid_prev, wb[np.clip(tprev, 0, T - 1), np.arange(U)], NEG).astype(F)
        wlp = np.where(valid_prev, wl[np.clip(tprev, 0, T - 1), np.arange(U)], NEG).astype(F)
        stay = (cur + wbp).astype(F)
        emit = np.concatenate([[NEG], (cur + wlp).astype(F)[:-1]]).astype(F)
        nxt = lse2_f32(stay, emit)
        t = n - np.arange(U)
        ok = (t >= 0) & (t < T)
        nxt = np.where(ok, nxt, NEG).astype(F)
        if n % kreb == 0:
            us = np.arange(U)[ok]
            r = pick_ref(nxt[ok], us, n, rule, slope, ridge(n), +1)
            mi = np.rint(nxt[r])
            nxt = np.where(nxt > F(-1e29), (nxt - F(mi)).astype(F), nxt)
            off += float(mi)
        offA[n] = off
        A[t[ok], np.arange(U)[ok]] = nxt[ok]
        cur = nxt
    ll2 = off + float(cur[U - 1]) + float(wb[T - 1, U - 1])
    # beta
    Bt = np.full((T, U), NEG, F)
    offB = np.zeros(N)
    off = 0.0
    cur = np.full(U + 1, NEG, F)  # diagonal n+1 (virtual terminal for n = N-1)
    cur[U - 1] = 0
    for n in range(N - 1, -1, -1):
        t = n - np.arange(U)
        ok = (t >= 0) & (t < T)
        wbn = np.where(ok, wb[np.clip(t, 0, T - 1), np.arange(U)], NEG).astype(F)
        wln = np.where(ok, wl[np.clip(t, 0, T - 1), np.arange(U)], NEG).astype(F)
        stay = (cur[:U] + wbn).astype(F)
        emit = (cur[1:U + 1] + wln).astype(F)
        nxt = lse2_f32(stay, emit)
        nxt = np.where(ok, nxt, NEG).astype(F)
        if n % kreb == kreb - 1 or n == N - 1:
            us = np.arange(U)[ok]
            r = pick_ref(nxt[ok], us, n, rule, slope, ridge(n), -1)
            mi = np.rint(nxt[r])
            nxt = np.where(nxt > F(-1e29), (nxt - F(mi)).astype(F), nxt)
            off += float(mi)
        offB[n] = off
        Bt[t[ok], np.arange(U)[ok]] = nxt[ok]
        cur = np.concatenate([nxt, [NEG]]).astype(F)
    # offsets are per BLOCK of kreb diagonals in the kernels: offA[n] as recorded here changes only at block starts
    return A, Bt, offA, offB, ll2


def sweep_lane_f64(wb, wl, K=16, kreb=KREB, store_bits=17):
    """The float64 recurrence of csrc/rnnt_sweep.h alpha_sweep_pr / beta_sweep_pr: the SAME float32 edge weights, alpha / beta
    carried as true log2 values in float64 (here: sweep_lane with float64 registers), stored as float32 residues against the
    lane's integer offset -- emulated by rounding the stored values to 2^-store_bits (the ulp of a residue of ~64).  The
    log2(1 + 2^-|d|) term is float64 here and float32 on the GPU (absolute error ~1e-7 per step, not accumulated at the
    residue's magnitude)."""
    global F, NEG
    F_old, NEG_old = F, NEG
    F, NEG = np.float64, np.float64(-1.0e30)
    try:
        A, Bt, ll2 = sweep_lane(wb.astype(np.float64), wl.astype(np.float64), K, kreb)
    finally:
        F, NEG = F_old, NEG_old
    q = 2.0 ** -store_bits
    A = np.where(np.isfinite(A), np.round(A / q) * q, A)
    Bt = np.where(np.isfinite(Bt), np.round(Bt / q) * q, Bt)
    return A, Bt, ll2


def sweep_lane(wb, wl, K=3, kreb=KREB):
    """Per-LANE integer offsets (a lane owns K consecutive columns): every lane re-bases against its own maximum; a value
    that crosses a lane boundary is shifted by the (integer) offset difference.  Returns TRUE alpha / beta as f64 arrays
    (residue + offset) plus ll2, and the residues for inspection."""
    T, U = wb.shape
    N = T + U - 1
    L = (U + K - 1) // K
    lane = np.arange(U) // K
    ar = np.arange(U)

    def rebase(v, off, ok):
        for l in range(L):
            sel = (lane == l) & ok & (v > F(-1e29))
            if sel.any():
                mi = np.rint(v[sel].max())
                v[(lane == l) & (v > F(-1e29))] -= F(mi)
                off[l] += float(mi)

    A = np.full((T, U), -np.inf)
    Ares = np.full((T, U), NEG, F)
    A[0, 0] = 0.0
    Ares[0, 0] = 0
    off = np.zeros(L)
    cur = np.full(U, NEG, F)
    cur[0] = 0
    for n in range(1, N):
        tprev = (n - 1) - ar
        vp = (tprev >= 0) & (tprev < T)
        wbp = np.where(vp, wb[np.clip(tprev, 0, T - 1), ar], NEG).astype(F)
        wlp = np.where(vp, wl[np.clip(tprev, 0, T - 1), ar], NEG).astype(F)
        stay = (cur + wbp).astype(F)
        em = (cur + wlp).astype(F)
        # shift into the receiving lane's frame where the edge crosses a lane boundary
        delta = np.zeros(U, F)
        delta[1:] = (off[lane[:-1]] - off[lane[1:]]).astype(F)
        emit = np.concatenate([[NEG], em[:-1]]).astype(F)
        emit = np.where(emit > F(-1e29), (emit + delta).astype(F), emit)
        nxt = lse2_f32(stay, emit)
        t = n - ar
        ok = (t >= 0) & (t < T)
        nxt = np.where(ok, nxt, NEG).astype(F)
        if n % kreb == 0:
            rebase(nxt, off, ok)
        fin = ok & (nxt > F(-1e29))
        A[t[fin], ar[fin]] = nxt[fin].astype(np.float64) + off[lane[fin]]
        Ares[t[fin], ar[fin]] = nxt[fin]
        cur = nxt
    ll2 = off[lane[U - 1]] + float(cur[U - 1]) + float(wb[T - 1, U - 1])
    Bt = np.full((T, U), -np.inf)
    off = np.zeros(L)
    cur = np.full(U + 1, NEG, F)
    cur[U - 1] = 0
    for n in range(N - 1, -1, -1):
        t = n - ar
        ok = (t >= 0) & (t < T)
        wbn = np.where(ok, wb[np.clip(t, 0, T - 1), ar], NEG).astype(F)
        wln = np.where(ok, wl[np.clip(t, 0, T - 1), ar], NEG).astype(F)
        stay = (cur[:U] + wbn).astype(F)
        right = cur[1:U + 1].copy()
        delta = np.zeros(U, F)
        delta[:-1] = (off[lane[1:]] - off[lane[:-1]]).astype(F)
        right = np.where(right > F(-1e29), (right + delta).astype(F), right)
        emit = (right + wln).astype(F)
        nxt = lse2_f32(stay, emit)
        nxt = np.where(ok, nxt, NEG).astype(F)
        if n % kreb == kreb - 1 or n == N - 1:
            rebase(nxt, off, ok)
        fin = ok & (nxt > F(-1e29))
        Bt[t[fin], ar[fin]] = nxt[fin].astype(np.float64) + off[lane[fin]]
        cur = np.concatenate([nxt, [NEG]]).astype(F)
    return A, Bt, ll2


def grad_from_true(x, labels, A, Bt, ll2, lse, blank=0):
    """gradient from alpha, beta given as f64 (residue + offset): the set-up forms every exponent in f64, rounds once."""
    x = x.astype(F)
    T, U, V = x.shape
    D = np.float64
    tt, uu = np.meshgrid(np.arange(T), np.arange(U), indexing="ij")
    nl64 = -(lse.astype(D)) * D(LOG2E)
    with np.errstate(invalid="ignore"):
        c0 = np.maximum(A + Bt - ll2 + nl64, -1e30).astype(F)
        g = np.exp2((x * LOG2E + c0[..., None]).astype(F)).astype(F)
        Bt_t1 = np.vstack([Bt[1:], np.full((1, U), -np.inf)])
        cb = np.maximum(A + Bt_t1 - ll2 + nl64, -1e30)
        cb[T - 1, :] = -1e30
        cb[T - 1, U - 1] = A[T - 1, U - 1] - ll2 + nl64[T - 1, U - 1]
        g[:, :, blank] -= np.exp2((x[:, :, blank] * LOG2E + cb.astype(F)).astype(F))
        if U > 1:
            Bt_u1 = np.hstack([Bt[:, 1:], np.full((T, 1), -np.inf)])
            cl = np.maximum(A + Bt_u1 - ll2 + nl64, -1e30)[:, :U - 1]
            xl = np.take_along_axis(x[:, :U - 1], labels[None, :U - 1, None].astype(np.int64), axis=2)[:, :, 0]
            corr = np.exp2((xl * LOG2E + cl.astype(F)).astype(F))
            np.subtract.at(g, (tt[:, :U - 1], uu[:, :U - 1], np.broadcast_to(labels[None, :U - 1], (T, U - 1))), corr)
    return g


def grad_f32(x, labels, A, Bt, offA, offB, ll2, lse, blank=0, setup64=False):
    """setup64: form alpha + beta + offsets - ll - lse in float64 and round ONCE (instead of four f32 additions)."""
    if setup64:
        return grad_setup64(x, labels, A, Bt, offA, offB, ll2, lse, blank)
    x = x.astype(F)
    T, U, V = x.shape
    g = np.zeros((T, U, V), F)
    tt, uu = np.meshgrid(np.arange(T), np.arange(U), indexing="ij")
    n = tt + uu
    nl = (-lse * LOG2E).astype(F)
    E0 = (offA[n] + offB[n] - ll2).astype(F)
    c0 = (((A + Bt).astype(F) + E0).astype(F) + nl).astype(F)
    g[:] = np.exp2((x * LOG2E + c0[..., None]).astype(F))
    # blank correction
    n1 = np.minimum(n + 1, T + U - 2)
    E1 = (offA[n] + offB[n1] - ll2).astype(F)
    Bt_t1 = np.vstack([Bt[1:], np.full((1, U), NEG, F)])
    cb = ((A + Bt_t1).astype(F) + E1).astype(F)
    cb[T - 1, :] = NEG
    cb[T - 1, U - 1] = (A[T - 1, U - 1] + F(offA[T + U - 2] - ll2)).astype(F)
    g[:, :, blank] -= np.exp2(((x[:, :, blank] * LOG2E + nl).astype(F) + cb).astype(F))
    if U > 1:
        Bt_u1 = np.hstack([Bt[:, 1:], np.full((T, 1), NEG, F)])
        cl = ((A + Bt_u1).astype(F) + E1).astype(F)[:, :U - 1]
        xl = np.take_along_axis(x[:, :U - 1], labels[None, :U - 1, None].astype(np.int64), axis=2)[:, :, 0]
        corr = np.exp2(((xl * LOG2E + nl[:, :U - 1]).astype(F) + cl).astype(F))
        np.subtract.at(g, (tt[:, :U - 1], uu[:, :U - 1], np.broadcast_to(labels[None, :U - 1], (T, U - 1))), corr)
    return g


def grad_setup64(x, labels, A, Bt, offA, offB, ll2, lse, blank=0):
    x = x.astype(F)
    T, U, V = x.shape
    D = np.float64
    tt, uu = np.meshgrid(np.arange(T), np.arange(U), indexing="ij")
    n = tt + uu
    nl64 = -(lse.astype(D)) * D(LOG2E)
    c0 = (A.astype(D) + Bt.astype(D) + (offA[n] + offB[n] - ll2) + nl64).astype(F)
    g = np.exp2((x * LOG2E + c0[..., None]).astype(F)).astype(F)
    n1 = np.minimum(n + 1, T + U - 2)
    Bt_t1 = np.vstack([Bt[1:], np.full((1, U), NEG, F)]).astype(D)
    cb = A.astype(D) + Bt_t1 + (offA[n] + offB[n1] - ll2) + nl64
    cb[T - 1, :] = -1e30
    cb[T - 1, U - 1] = A[T - 1, U - 1].astype(D) + (offA[T + U - 2] - ll2) + nl64[T - 1, U - 1]
    g[:, :, blank] -= np.exp2((x[:, :, blank] * LOG2E + cb.astype(F)).astype(F))
    if U > 1:
        Bt_u1 = np.hstack([Bt[:, 1:], np.full((T, 1), NEG, F)]).astype(D)
        cl = (A.astype(D) + Bt_u1 + (offA[n] + offB[n1] - ll2) + nl64)[:, :U - 1]
        xl = np.take_along_axis(x[:, :U - 1], labels[None, :U - 1, None].astype(np.int64), axis=2)[:, :, 0]
        corr = np.exp2((xl * LOG2E + cl.astype(F)).astype(F))
        np.subtract.at(g, (tt[:, :U - 1], uu[:, :U - 1], np.broadcast_to(labels[None, :U - 1], (T, U - 1))), corr)
    return g


def make_inputs(kind, T, U, V, rng):
    labels = rng.integers(1, V, size=U - 1).astype(np.int32)
    x = rng.normal(size=(T, U, V))
    if kind.startswith("sigma"):
        return (x * float(kind[5:])).astype(np.float32), labels
    if kind.startswith("trained"):  # one dominant symbol per cell along a monotone alignment
        bonus = float(re.sub(r"[a-z]", "", kind[7:]) or 10)
        emit = np.sort(rng.integers(0, T, size=U - 1))  # frame at which label u+1 is emitted
        if "late" in kind:
            emit = np.sort(rng.integers(int(0.6 * T), T, size=U - 1))
        for u in range(U):
            te = emit[u] if u < U - 1 else T
            x[:te, u, 0] += bonus
            if u < U - 1:
                x[te:, u, labels[u]] += bonus
        return x.astype(np.float32), labels
    raise ValueError(kind)


def run(kind, T, U, V, seed, rules=("ridge+s64", "follow+s64", "lane3", "lane1"), kreb=KREB):
    rng = np.random.default_rng(seed)
    x, labels = make_inputs(kind, T, U, V, rng)
    c_ref, g_ref, _, _, _ = orc.utterance_cost_and_grad(x, labels)
    wb, wl, lse = edge_weights(x, labels)
    out = {}
    for rule in rules:
        if rule.startswith("lane"):
            if rule.endswith("f64"):  # e.g. "lane16f64": the float64 recurrence, 16 columns per lane
                A, Bt, ll2 = sweep_lane_f64(wb, wl, int(rule[4:-3] or 16), kreb)
            else:
                A, Bt, ll2 = sweep_lane(wb, wl, int(rule[4:] or 3), kreb)
            g = grad_from_true(x, labels, A, Bt, ll2, lse)
            cost = -ll2 * np.log(2.0)
            out[rule] = (abs(cost - c_ref) / max(1.0, abs(c_ref)), float(np.abs(g - g_ref).max()))
            continue
        A, Bt, offA, offB, ll2 = sweep(wb, wl, rule.split("+")[0], kreb)
        g = grad_f32(x, labels, A, Bt, offA, offB, ll2, lse, setup64=rule.endswith("+s64"))
        cost = -ll2 * np.log(2.0)
        out[rule] = (abs(cost - c_ref) / max(1.0, abs(c_ref)), float(np.abs(g - g_ref).max()))
    return c_ref, out


if __name__ == "__main__":
    T, U, V = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "600,150,28").split(","))
    kinds = ["sigma1", "sigma4", "sigma8", "trained10", "trained10late", "trained20late"]
    print(f"T={T} U={U} V={V}: max|dgrad| (rel. cost error) per re-basing reference")
    for kind in kinds:
        for seed in (1, 2):
            c_ref, out = run(kind, T, U, V, seed)
            print(f"{kind:14s} seed {seed} cost {c_ref:10.2f}  " +
                  "  ".join(f"{r}: {g:.2e} ({c:.1e})" for r, (c, g) in out.items()))
