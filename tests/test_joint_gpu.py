"""GPU parity tests of the fused joint + loss path (compute_rnnt_joint_loss_* through the C ABI)
against the float64 oracle (oracle/rnnt_oracle.py: joint_forward / joint_backward, which restate
model.py:158-166 and its autodiff).

Tolerances: costs relative 1e-4; every gradient tensor max|d| <= 1e-4 * max(1, max|ref|)."""
import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc

pytestmark = pytest.mark.gpu


def make(B, T, U, H, J, V, ragged, seed):
    rng = np.random.default_rng(seed)
    enc = rng.normal(size=(B, T, H)).astype(np.float32)
    pred = rng.normal(size=(B, U, H)).astype(np.float32)
    lim1, lim2 = np.sqrt(6.0 / (H + J)), np.sqrt(6.0 / (J + V))
    W1 = rng.uniform(-lim1, lim1, size=(H, J)).astype(np.float32)
    b1 = (0.1 * rng.normal(size=J)).astype(np.float32)
    W2 = rng.uniform(-lim2, lim2, size=(J, V)).astype(np.float32) * 3.0  # livelier logits than glorot alone
    b2 = (0.1 * rng.normal(size=V)).astype(np.float32)
    labels = rng.integers(1, V, size=(B, max(U - 1, 1))).astype(np.int32)[:, : max(U - 1, 0)]
    if ragged:
        il = rng.integers((T + 1) // 2, T + 1, size=B)
        ll = rng.integers(U // 2, U, size=B)
        il[0], ll[0] = T, U - 1
    else:
        il, ll = np.full(B, T), np.full(B, U - 1)
    return enc, pred, W1, b1, W2, b2, labels, il.astype(np.int32), ll.astype(np.int32)


def run(case, scale):
    enc, pred, W1, b1, W2, b2, labels, il, ll = case
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    params = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
    costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll))
    (costs * t(scale.astype(np.float32))).sum().backward()
    torch.cuda.synchronize()
    return costs.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in params]


SHAPES = [
    # B, T, U, H, J, V
    (2, 9, 5, 16, 64, 12),      # single u-tile, single J slab
    (3, 21, 40, 24, 128, 28),   # two u-tiles, two slabs, char-sized vocabulary
    (2, 50, 33, 32, 192, 31),   # u-tile boundary at 32/33, V = 31 (reference vocabulary)
    (1, 7, 70, 8, 64, 32),      # three u-tiles, V = 32 exactly
    (2, 300, 20, 16, 64, 5),    # row splits (T >= 256)
]


@pytest.mark.parametrize("B,T,U,H,J,V", SHAPES)
@pytest.mark.parametrize("ragged", [False, True])
def test_joint_matches_oracle(B, T, U, H, J, V, ragged):
    case = make(B, T, U, H, J, V, ragged, seed=B * 100 + T + U + J + V)
    scale = np.linspace(0.5, 1.5, B)
    costs, grads = run(case, scale)
    ref = orc.joint_loss_and_grads(*case, cost_scale=scale)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        tol = 1e-4 * max(1.0, np.abs(ref[key]).max())
        assert np.abs(g - ref[key]).max() <= tol, key
    # padded frames / label positions receive exactly zero gradient
    enc_g, pred_g = grads[0], grads[1]
    il, ll = case[7], case[8]
    for b in range(B):
        assert not enc_g[b, il[b]:].any() and not pred_g[b, ll[b] + 1:].any()


def test_joint_equals_unfused_composition():
    """Fused path == materialised logits + rnnt_loss (the reference's own composition)."""
    case = make(2, 30, 12, 16, 64, 28, True, seed=7)
    enc, pred, W1, b1, W2, b2, labels, il, ll = case
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    m = pkg.JointLoss(16, 64, 28).to(dev)
    with torch.no_grad():
        m.W1.copy_(t(W1)), m.b1.copy_(t(b1)), m.W2.copy_(t(W2)), m.b2.copy_(t(b2))
    fused = m(t(enc), t(pred), t(labels), t(il), t(ll))
    unfused = pkg.rnnt_loss(m.logits(t(enc), t(pred)), t(labels), t(il), t(ll))
    np.testing.assert_allclose(fused.detach().cpu().numpy(), unfused.detach().cpu().numpy(), rtol=2e-5)


def test_joint_large_preactivations_take_the_exact_tanh_path():
    """|enc_proj| > 43: the tabulated e^{2x} factors would overflow; the kernels must switch to tanh(a + c)."""
    case = list(make(2, 13, 9, 16, 64, 28, True, seed=5))
    case[0] = case[0] * 60.0
    case[1] = case[1] * 60.0 + 2.0
    scale = np.ones(2)
    costs, grads = run(tuple(case), scale)
    ref = orc.joint_loss_and_grads(*case, cost_scale=scale)
    assert np.isfinite(costs).all() and all(np.isfinite(g).all() for g in grads)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        assert np.abs(g - ref[key]).max() <= 1e-4 * max(1.0, np.abs(ref[key]).max()), key


def test_joint_is_deterministic():
    case = make(2, 40, 40, 16, 128, 28, True, seed=11)
    scale = np.ones(2)
    c1, g1 = run(case, scale)
    c2, g2 = run(case, scale)
    assert np.array_equal(c1, c2)
    for a, b in zip(g1, g2):
        assert np.array_equal(a, b)


def test_joint_limits_are_reported():
    dev = torch.device("cuda:0")
    enc, pred = torch.zeros(1, 4, 8, device=dev), torch.zeros(1, 3, 8, device=dev)
    W1, b1 = torch.zeros(8, 832, device=dev), torch.zeros(832, device=dev)  # J = 832 > 768: beyond the f32 joint
    W2, b2 = torch.zeros(832, 5, device=dev), torch.zeros(5, device=dev)
    with pytest.raises(ValueError, match="at most 768"):
        pkg.rnnt_joint_loss(enc, pred, W1, b1, W2, b2, torch.ones(1, 2, dtype=torch.int32, device=dev),
                            torch.tensor([4], device=dev), torch.tensor([2], device=dev))
    from rnnt_speech_recognition_amd import _lib
    with pytest.raises(RuntimeError, match="invalid value"):  # the raw C ABI rejects J = 48 (not a multiple of 64)
        _lib.joint_workspace_bytes(4, 3, 1, 48, 5)


def test_joint_odd_joint_width_is_padded_exactly():
    """J = 100: padded to 128 with zero units by the host layer; results equal the oracle on the unpadded problem."""
    case = make(2, 15, 8, 12, 100, 28, True, seed=33)
    scale = np.ones(2)
    costs, grads = run(case, scale)
    ref = orc.joint_loss_and_grads(*case, cost_scale=scale)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        assert g.shape == ref[key].shape and np.abs(g - ref[key]).max() <= 1e-4 * max(1.0, np.abs(ref[key]).max()), key
