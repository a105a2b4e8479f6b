"""CPU tests of the oracles themselves (no GPU): the float64 NumPy oracle is pinned by the upstream
known-answer vector, finite differences and the alpha/beta likelihood identity; the C restatement of
the reference CPU path is checked against it; committed goldens are regression-checked."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import cpu_oracle, rnnt_oracle as orc


def _kat(golden_dir):
    with open(os.path.join(golden_dir, "kat_small.json")) as f:
        return json.load(f)


def test_kat_cost_and_grad(golden_dir):
    k = _kat(golden_dir)
    costs, grads = orc.rnnt_loss_and_grad(
        np.array(k["logits"]), np.array(k["labels"]), k["input_lengths"], k["label_lengths"], blank=k["blank"])
    assert abs(costs[0] - k["cost_f64"]) < 1e-12
    assert abs(costs[0] - k["cost"]) < 1e-6
    np.testing.assert_allclose(grads, np.array(k["grads_wrt_logits"]), atol=2e-7, rtol=0)


def test_kat_c_restatement(golden_dir):
    """The C restatement takes log-probs and returns log-prob gradients (CPU-op convention);
    chaining the log-softmax backward must reproduce the KAT's logits-gradient."""
    k = _kat(golden_dir)
    x = np.array(k["logits"], dtype=np.float64)
    lp = orc.log_softmax(x)
    costs, glp = cpu_oracle.rnnt_cpu(lp, k["labels"], k["input_lengths"], k["label_lengths"], blank=k["blank"])
    assert abs(costs[0] - k["cost"]) < 2e-6
    g = glp - np.exp(lp) * glp.sum(-1, keepdims=True)
    np.testing.assert_allclose(g, np.array(k["grads_wrt_logits"]), atol=1e-6, rtol=0)


def _kat_b2(golden_dir):
    with open(os.path.join(golden_dir, "kat_b2.json")) as f:
        k = json.load(f)
    x = np.array(k["logits_flat"], np.float64).reshape(k["B"], k["T"], k["U"], k["V"])
    g = np.array(k["grads_wrt_logits_flat_published"]).reshape(x.shape)
    return k, x, g


def test_kat_b2_upstream_two_utterance_vector(golden_dir):
    """Second external anchor: upstream's B=2 T=4 U=3 V=3 vector (published costs and gradients, 6-7 decimals)."""
    k, x, g_pub = _kat_b2(golden_dir)
    costs, grads = orc.rnnt_loss_and_grad(x, np.array(k["labels"]), k["input_lengths"], k["label_lengths"], blank=k["blank"])
    np.testing.assert_allclose(costs, k["costs_published"], atol=5e-7, rtol=0)
    np.testing.assert_allclose(grads, g_pub, atol=1e-6, rtol=0)


def test_kat_b2_c_restatement(golden_dir):
    k, x, g_pub = _kat_b2(golden_dir)
    lp = orc.log_softmax(x)
    costs, glp = cpu_oracle.rnnt_cpu(lp, k["labels"], k["input_lengths"], k["label_lengths"], blank=k["blank"])
    np.testing.assert_allclose(costs, k["costs_published"], atol=3e-6, rtol=0)
    g = glp - np.exp(lp) * glp.sum(-1, keepdims=True)
    np.testing.assert_allclose(g, g_pub, atol=2e-6, rtol=0)


@pytest.mark.parametrize("fused", [True, False])
def test_finite_differences(fused):
    rng = np.random.default_rng(7)
    B, T, U, V = 2, 5, 4, 6
    x = rng.normal(size=(B, T, U, V))
    if not fused:
        x = orc.log_softmax(x)  # treat entries as free variables anyway
    lab = rng.integers(1, V, size=(B, U - 1))
    il, ll = np.array([5, 3]), np.array([3, 2])
    _, g = orc.rnnt_loss_and_grad(x, lab, il, ll, fused_softmax=fused)
    eps = 1e-6
    fd = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        cp, _ = orc.rnnt_loss_and_grad(xp, lab, il, ll, fused_softmax=fused)
        cm, _ = orc.rnnt_loss_and_grad(xm, lab, il, ll, fused_softmax=fused)
        fd[idx] = (cp.sum() - cm.sum()) / (2 * eps)
    np.testing.assert_allclose(g, fd, atol=1e-7, rtol=0)
    # padded cells carry exactly zero gradient
    assert np.all(g[1, 3:, :, :] == 0) and np.all(g[1, :, 3:, :] == 0)


def test_alpha_beta_likelihood_identity_and_cell_sums():
    rng = np.random.default_rng(3)
    T, U, V = 17, 6, 9
    x = rng.normal(size=(T, U, V))
    lab = rng.integers(1, V, size=U - 1)
    cost, g, a, b, ll_b = orc.utterance_cost_and_grad(x, lab)
    assert abs(-cost - ll_b) < 1e-10
    # occupancy sums to one on every anti-diagonal
    occ = np.exp(a + b + cost)
    for n in range(T + U - 1):
        s = sum(occ[n - u, u] for u in range(U) if 0 <= n - u < T)
        assert abs(s - 1) < 1e-10
    # fused-softmax gradient of every cell sums to zero over the vocabulary
    assert np.abs(g.sum(-1)).max() < 1e-12


@pytest.mark.parametrize("T,U", [(1, 1), (1, 4), (6, 1), (2, 2)])
def test_edge_shapes(T, U):
    rng = np.random.default_rng(T * 10 + U)
    V = 5
    x = rng.normal(size=(1, T, U, V))
    lab = rng.integers(1, V, size=(1, max(U - 1, 0)))
    costs, g = orc.rnnt_loss_and_grad(x, lab, [T], [U - 1])
    lp = orc.log_softmax(x[0])
    if U == 1:  # only blanks
        assert abs(costs[0] + lp[:, 0, 0].sum()) < 1e-12
    if T == 1:  # all labels in the single frame, then the terminal blank
        ref = -(sum(lp[0, u, lab[0, u]] for u in range(U - 1)) + lp[0, U - 1, 0])
        assert abs(costs[0] - ref) < 1e-12
    assert np.isfinite(g).all()


def test_c_restatement_matches_numpy_oracle_ragged():
    rng = np.random.default_rng(11)
    B, T, U, V = 6, 40, 13, 28
    x = rng.normal(size=(B, T, U, V)).astype(np.float32)
    lab = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    il = np.array([40, 31, 1, 25, 40, 7])
    ll = np.array([12, 0, 5, 12, 3, 9])
    lp = orc.log_softmax(x)
    c64, g64 = orc.rnnt_loss_and_grad(lp, lab, il, ll, fused_softmax=False)
    for nt in (1, 0):
        c32, g32 = cpu_oracle.rnnt_cpu(lp, lab, il, ll, num_threads=nt)
        np.testing.assert_allclose(c32, c64, rtol=2e-6, atol=1e-5)
        np.testing.assert_allclose(g32, g64, atol=5e-5, rtol=0)
    # score-only mode
    c_only, none = cpu_oracle.rnnt_cpu(lp, lab, il, ll, want_grad=False)
    assert none is None
    np.testing.assert_allclose(c_only, c64, rtol=2e-6, atol=1e-5)


def test_c_restatement_rejects_bad_lengths():
    lp = np.zeros((1, 3, 2, 4), np.float32)
    with pytest.raises(RuntimeError):
        cpu_oracle.rnnt_cpu(lp, [[1]], [4], [1])
    with pytest.raises(RuntimeError):
        cpu_oracle.rnnt_cpu(lp, [[1]], [3], [2])


def test_goldens_regression(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "*.npz")))
    assert len(files) >= 5
    for f in files:
        d = np.load(f)
        costs, grads = orc.rnnt_loss_and_grad(
            d["acts"], d["labels"], d["input_lengths"], d["label_lengths"], blank=int(d["blank"]))
        np.testing.assert_allclose(costs, d["costs"], rtol=1e-12)
        np.testing.assert_allclose(grads, d["grads"], atol=1e-7)


def test_joint_oracle_backward_fd():
    rng = np.random.default_rng(5)
    B, T, U, H, J, V = 2, 4, 3, 5, 6, 7
    enc, pred = rng.normal(size=(B, T, H)), rng.normal(size=(B, U, H))
    W1, b1 = rng.normal(size=(H, J)) * 0.5, rng.normal(size=J) * 0.1
    W2, b2 = rng.normal(size=(J, V)) * 0.5, rng.normal(size=V) * 0.1
    lab = rng.integers(1, V, size=(B, U - 1))
    il, ll = np.array([4, 3]), np.array([2, 1])
    scale = np.array([0.5, 0.25])

    def total(**kw):
        args = dict(enc=enc, pred=pred, W1=W1, b1=b1, W2=W2, b2=b2)
        args.update(kw)
        out = orc.joint_loss_and_grads(args["enc"], args["pred"], args["W1"], args["b1"], args["W2"], args["b2"],
                                       lab, il, ll, cost_scale=scale)
        return (out["costs"] * scale).sum(), out

    _, out = total()
    eps = 1e-6
    for name, arr, key in [("enc", enc, "d_enc"), ("pred", pred, "d_pred"), ("W1", W1, "dW1"), ("b1", b1, "db1"),
                           ("W2", W2, "dW2"), ("b2", b2, "db2")]:
        fd = np.zeros_like(arr)
        for idx in np.ndindex(*arr.shape):
            ap, am = arr.copy(), arr.copy()
            ap[idx] += eps
            am[idx] -= eps
            fd[idx] = (total(**{name: ap})[0] - total(**{name: am})[0]) / (2 * eps)
        np.testing.assert_allclose(out[key], fd, atol=2e-7, err_msg=name)


def test_joint_goldens_regression(golden_dir):
    """The committed fused-joint fixtures are what today's oracle produces (f32 joint and f16-MFMA joint)."""
    files = sorted(glob.glob(os.path.join(golden_dir, "joint", "*.npz")))
    assert len(files) >= 2
    for f in files:
        z = np.load(f)
        fn = orc.joint_loss_and_grads if str(z["joint_dtype"]) == "f32" else orc.joint_loss_and_grads_f16
        r = fn(z["enc"], z["pred"], z["W1"], z["b1"], z["W2"], z["b2"], z["labels"], z["input_lengths"],
               z["label_lengths"], cost_scale=z["cost_scale"])
        np.testing.assert_allclose(r["costs"], z["costs"], rtol=1e-12)
        for k in ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2"):
            np.testing.assert_allclose(r[k], z[k], rtol=0, atol=1e-6 * max(1.0, np.abs(z[k]).max()))


def test_f16_joint_oracle_stays_close_to_the_exact_joint():
    """The stated binary16 roundings cost ~1e-4 relative on the costs and ~1e-3 on the gradients, no more."""
    rng = np.random.default_rng(3)
    B, T, U, H, J, V = 2, 7, 5, 6, 32, 64
    enc, pred = rng.normal(size=(B, T, H)), rng.normal(size=(B, U, H))
    W1, b1 = 0.3 * rng.normal(size=(H, J)), 0.1 * rng.normal(size=J)
    W2, b2 = 0.3 * rng.normal(size=(J, V)), 0.1 * rng.normal(size=V)
    labels = rng.integers(1, V, size=(B, U - 1))
    il, ll = np.array([T, T - 2]), np.array([U - 1, U - 3])
    a = orc.joint_loss_and_grads(enc, pred, W1, b1, W2, b2, labels, il, ll)
    h = orc.joint_loss_and_grads_f16(enc, pred, W1, b1, W2, b2, labels, il, ll)
    np.testing.assert_allclose(h["costs"], a["costs"], rtol=5e-4)
    for k in ("d_enc", "d_pred", "dW2", "db2"):
        assert np.abs(h[k] - a[k]).max() <= 5e-3 * max(1.0, np.abs(a[k]).max()), k
    assert orc.dl_scale_f16(None, 3) == 2.0 ** 14
    assert orc.dl_scale_f16([0.25, -1.0], 2) == 2.0 ** 14 and orc.dl_scale_f16([3.0], 1) == 2.0 ** 12
    assert orc.dl_scale_f16([1.0 / 512], 1) == 2.0 ** 23


@pytest.mark.parametrize("f16", [False, True])
def test_streamed_utterance_oracle_equals_the_dense_oracles(f16):
    """joint_utterance_streamed (used by the BASELINE-size GPU tests) against joint_loss_and_grads[_f16]."""
    rng = np.random.default_rng(0)
    B, T, U, H, J, V = 2, 13, 7, 5, 16, 11
    enc, pred = rng.normal(size=(B, T, H)), rng.normal(size=(B, U, H))
    W1, b1 = rng.normal(size=(H, J)) * 0.5, rng.normal(size=J) * 0.1
    W2, b2 = rng.normal(size=(J, V)), rng.normal(size=V) * 0.1
    labels = rng.integers(1, V, size=(B, U - 1))
    il, ll, sc = np.array([13, 9]), np.array([6, 4]), np.array([0.7, 1.3])
    ref = (orc.joint_loss_and_grads_f16 if f16 else orc.joint_loss_and_grads)(enc, pred, W1, b1, W2, b2, labels, il, ll,
                                                                             cost_scale=sc)
    S = orc.dl_scale_f16(sc, B)
    dW2, db2 = 0.0, 0.0
    for b in range(B):
        Tb, Ub = il[b], ll[b] + 1
        o = orc.joint_utterance_streamed(enc[b, :Tb] @ W1 + b1, pred[b, :Ub] @ W1, W2, b2, labels[b, : Ub - 1],
                                         cost_scale=sc[b], f16=f16, dl_scale=S, rows_per_chunk=4)
        assert abs(o["cost"] - ref["costs"][b]) < 1e-10
        assert np.abs(o["d_enc_proj"] - ref["d_a"][b, :Tb]).max() < 1e-10
        assert np.abs(o["d_pred_proj"] - ref["d_c"][b, :Ub]).max() < 1e-10
        dW2, db2 = dW2 + o["dW2"], db2 + o["db2"]
    assert np.abs(dW2 - ref["dW2"]).max() < 1e-10 and np.abs(db2 - ref["db2"]).max() < 1e-10


def test_park_factor_of_the_f16_joint_oracle():
    """park_factor_f16 (the binary16 parking of the softmax numerators between the f16 joint's forward and backward pass):
    a relative perturbation of at most half a binary16 step for everything within 2^14 of its chunk's maximum, exactly 1 at the
    blank column and at the label columns of cells that have a label edge, and independent of the other chunks' values."""
    rng = np.random.default_rng(5)
    T, U, V = 3, 4, 96
    y = rng.normal(size=(T, U, V))  # (nothing more than 2^14 below its chunk's maximum)
    labels = np.array([5, 40, 70])
    f = orc.park_factor_f16(y, labels, blank=0)
    assert f.shape == y.shape
    assert np.all(np.abs(f - 1.0) <= 2.0 ** -11 * (1 + 1e-12))
    assert np.all(f[..., 0] == 1.0)
    for u, lab in enumerate(labels):
        assert np.all(f[:, u, lab] == 1.0)
    assert np.any(f[:, U - 1, :] != 1.0)  # the last column has no label edge, only the blank column is exact there
    y2 = y.copy()
    y2[..., 32:64] += 40.0  # another chunk's values do not move this chunk's factors
    f2 = orc.park_factor_f16(y2, labels, blank=0)
    assert np.array_equal(f[..., :32], f2[..., :32]) and np.array_equal(f[..., 64:], f2[..., 64:])
    # has_label: a cell without a label edge keeps the parked value in what would be its label column
    f3 = orc.park_factor_f16(y, labels, blank=0, has_label=np.array([True, False, True]))
    assert np.array_equal(f3[:, 0], f[:, 0]) and np.array_equal(f3[:, 2], f[:, 2])
    assert not np.all(f3[:, 1, labels[1]] == 1.0) or np.all(orc._rne_half(np.exp2(y[:, 1, labels[1]] * orc._LOG2E)) == 1)
    # far below the chunk maximum the numerator sinks into binary16's subnormals and finally to zero: factor 0, not NaN
    y4 = np.zeros((1, 2, 32))
    y4[..., 1] = -30.0
    f4 = orc.park_factor_f16(y4, np.array([3]), blank=0)
    assert f4[0, 0, 1] == 0.0 and np.isfinite(f4).all()
