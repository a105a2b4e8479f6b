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


def run(case, scale, joint_dtype="auto"):
    enc, pred, W1, b1, W2, b2, labels, il, ll = case
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    params = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
    costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll), joint_dtype=joint_dtype)
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
    (2, 40, 37, 16, 704, 28),   # 640 < J <= 704: W2 streams through the LDS (joint_phase1s_kernel), two-kernel backward
    (1, 33, 70, 8, 704, 31),    # widest joint of the f32 path, three u-tiles, V = 31
    (2, 23, 37, 16, 128, 40),   # round 5: TWO vocabulary tiles (32 < V <= 64), f32-grade; labels and blank in either tile
    (3, 40, 20, 24, 640, 64),   # both tiles full, the widest single-kernel joint (two J groups)
    (2, 70, 66, 16, 64, 33),    # one symbol in the second tile, three u-tiles
]
SHAPES_F32_WIDE_VOCAB = [(2, 30, 21, 16, 192, 100), (1, 45, 40, 24, 640, 128)]  # three / four tiles: joint_dtype="f32" on request


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


@pytest.mark.parametrize("B,T,U,H,J,V", SHAPES_F32_WIDE_VOCAB)
def test_f32_grade_joint_up_to_128_symbols(B, T, U, H, J, V):
    """joint_dtype="f32" beyond 64 symbols: three / four vocabulary tiles of the split-precision joint (the f16 joint is "auto"'s
    choice there); f32-grade bars, peaked logits included (the hand-back works from the parked 128-column tile)."""
    for gain in (1.0, 12.0):
        case = list(make(B, T, U, H, J, V, True, seed=V + J))
        case[4] = case[4] * np.float32(gain)
        scale = np.linspace(0.5, 1.5, B)
        costs, grads = run(tuple(case), scale, joint_dtype="f32")
        ref = orc.joint_loss_and_grads(*case, cost_scale=scale)
        np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
        for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
            assert np.abs(g - ref[key]).max() <= 1e-4 * max(1.0, np.abs(ref[key]).max()), (key, gain)


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
    W1, b1 = torch.zeros(8, 832, device=dev), torch.zeros(832, device=dev)  # J = 832 > 704: beyond the f32 joint
    W2, b2 = torch.zeros(832, 5, device=dev), torch.zeros(5, device=dev)
    with pytest.raises(ValueError, match="at most 704"):
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


def test_fused_joint_calls_can_be_captured_in_a_hip_graph():
    """Forward + backward of the fused joint (persistent kernels with LDS hand-offs, memsets, device-side path flags)
    record into a HIP graph and replay on new inputs: nothing in the library allocates or synchronises."""
    from rnnt_speech_recognition_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    B, T, U, J, V = 2, 40, 37, 64, 28
    ep, pp = torch.zeros(B, T, J, device=dev), torch.zeros(B, U, J, device=dev)
    W2 = torch.tensor(rng.normal(size=(J, V)) * 0.2, dtype=torch.float32, device=dev)
    b2 = torch.tensor(rng.normal(size=V) * 0.1, dtype=torch.float32, device=dev)
    labels = torch.tensor(rng.integers(1, V, size=(B, U - 1)), dtype=torch.int32, device=dev)
    il = torch.tensor([T, T - 9], dtype=torch.int32, device=dev)
    ll = torch.tensor([U - 1, U - 6], dtype=torch.int32, device=dev)
    scale = torch.ones(B, device=dev)
    costs = torch.empty(B, device=dev)
    d_ep, d_pp, d_w2, d_b2 = (torch.empty_like(x) for x in (ep, pp, W2, b2))
    ws = torch.empty(_lib.joint_workspace_bytes(T, U, B, J, V), dtype=torch.uint8, device=dev)

    def call(stream):
        opts = _lib.make_options(stream.cuda_stream, 0, T, U)
        _lib.check(lib.compute_rnnt_joint_loss_fwd(ep.data_ptr(), pp.data_ptr(), W2.data_ptr(), b2.data_ptr(), labels.data_ptr(),
                                                   ll.data_ptr(), il.data_ptr(), J, V, B, costs.data_ptr(), 0, ws.data_ptr(), opts), "fwd")
        _lib.check(lib.compute_rnnt_joint_loss_bwd(ep.data_ptr(), pp.data_ptr(), W2.data_ptr(), b2.data_ptr(), labels.data_ptr(),
                                                   ll.data_ptr(), il.data_ptr(), scale.data_ptr(), J, V, B, d_ep.data_ptr(),
                                                   d_pp.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(), 0, ws.data_ptr(), opts), "bwd")

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        call(side)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream())
    for seed in (3, 4):
        r = np.random.default_rng(seed)
        e_np, p_np = r.normal(size=(B, T, J)).astype(np.float32), r.normal(size=(B, U, J)).astype(np.float32)
        ep.copy_(torch.from_numpy(e_np)), pp.copy_(torch.from_numpy(p_np))
        graph.replay()
        torch.cuda.synchronize()
        eye = np.eye(J)
        ref = orc.joint_loss_and_grads(e_np.astype(np.float64), p_np.astype(np.float64), eye, np.zeros(J), W2.cpu().numpy().astype(np.float64),
                                       b2.cpu().numpy().astype(np.float64), labels.cpu().numpy(), il.cpu().numpy(), ll.cpu().numpy())
        assert np.abs(costs.cpu().numpy() - ref["costs"]).max() <= 1e-4 * max(1.0, np.abs(ref["costs"]).max())
        assert np.abs(d_w2.cpu().numpy() - ref["dW2"]).max() <= 1e-4 * max(1.0, np.abs(ref["dW2"]).max())
        assert np.abs(d_ep.cpu().numpy() - ref["d_a"]).max() <= 1e-4 * max(1.0, np.abs(ref["d_a"]).max())
