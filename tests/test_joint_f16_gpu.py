"""GPU parity tests of the large-vocabulary fused joint + loss path on the f16 MFMA units
(compute_rnnt_joint_loss_* with joint_dtype = 1, through the C ABI) against the float64 oracle that states the
same operand roundings (oracle/rnnt_oracle.py: joint_loss_and_grads_f16, restating model.py:158-166 under the
reference's mixed_float16 policy, run_rnnt.py:96-99).

Tolerances: costs relative 1e-4; every gradient tensor max|d| <= 1e-3 * max(1, max|ref|).  The gradient bound is
wider than the f32 paths' 1e-4 for a structural reason: dlogits are rounded to binary16 (relative step 4.9e-4) before
the two backward products, and an f32 kernel value that differs from the f64 oracle value by ~3e-6 relative lands on
the other side of a rounding boundary for ~1 % of the elements; each such element then differs by a whole binary16
step.  Measured: typically 5e-5 .. 4e-4, worst 6.1e-4 over 400 random cases (tests/tools/fuzz_parity.py); perturbing the
oracle's own pre-rounding values by 3e-6 relative moves its gradients by 1.5e-4 (same mechanism).
The distance to the UNROUNDED joint is bounded too (costs 1e-4 relative, the north-star bar; measured <= 3e-5): that is
the price of binary16 operands (the reference's mixed_float16 policy pays the same), not a kernel error."""
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
    W2 = rng.uniform(-lim2, lim2, size=(J, V)).astype(np.float32) * 3.0
    b2 = (0.1 * rng.normal(size=V)).astype(np.float32)
    labels = rng.integers(1, V, size=(B, max(U - 1, 1))).astype(np.int32)[:, : max(U - 1, 0)]
    if ragged:
        il = rng.integers((T + 1) // 2, T + 1, size=B)
        ll = rng.integers(U // 2, U, size=B)
        il[0], ll[0] = T, U - 1
    else:
        il, ll = np.full(B, T), np.full(B, U - 1)
    return enc, pred, W1, b1, W2, b2, labels, il.astype(np.int32), ll.astype(np.int32)


def run(case, scale, joint_dtype="f16", blank=0):
    enc, pred, W1, b1, W2, b2, labels, il, ll = case
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    params = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
    costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll), blank_label=blank, joint_dtype=joint_dtype)
    (costs * t(scale.astype(np.float32))).sum().backward()
    torch.cuda.synchronize()
    return costs.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in params]


SHAPES = [
    # B, T, U, H, J, V
    (2, 11, 7, 16, 128, 512),     # one tile of everything
    (2, 20, 40, 24, 256, 1024),   # two u-tiles, two J tiles, two V tiles, three row tiles
    (1, 140, 33, 16, 128, 512),   # row splits of the dh kernel, several dW2 work units, u-tile boundary at 32/33
    (1, 9, 5, 32, 640, 512),      # the BASELINE joint width
    (3, 17, 12, 8, 512, 1536),    # three V tiles
    (4, 64, 64, 16, 128, 1024),   # 16.8 M logits, eight row tiles x two column tiles per utterance
    # mid-sized vocabularies and joint widths (round 4): V any multiple of 128, J any multiple of 128 up to 640
    (2, 19, 9, 16, 128, 128),     # one 128-column group: three of the dW2 kernel's four column waves have no tile
    (2, 40, 37, 24, 384, 256),    # J = 384 (three 128-unit tiles), half a dW2 tile, bias table = exactly one 1 KB piece
    (3, 13, 33, 16, 640, 384),    # the BASELINE joint width on 384 symbols: bias table = one and a half pieces
    (2, 21, 11, 8, 256, 640),     # one full 512-column dW2 tile + a 128-column one
    (1, 70, 35, 16, 384, 1152),   # two full tiles + 128 columns, row splits
]


def _append_accuracy(entry):
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "r05_accuracy_f16.json")
    try:
        rows = json.load(open(path))
    except (OSError, ValueError):
        rows = []
    rows = [r for r in rows if (r["shape"], r["ragged"]) != (entry["shape"], entry["ragged"])] + [entry]
    json.dump(rows, open(path, "w"), indent=1)


@pytest.mark.parametrize("B,T,U,H,J,V", SHAPES)
@pytest.mark.parametrize("ragged", [False, True])
def test_joint_f16_matches_oracle(B, T, U, H, J, V, ragged):
    case = make(B, T, U, H, J, V, ragged, seed=B + T + U + H + J + V)
    scale = np.linspace(0.5, 1.5, B)
    costs, grads = run(case, scale)
    ref = orc.joint_loss_and_grads_f16(*case, cost_scale=scale)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        tol = 1e-3 * max(1.0, np.abs(ref[key]).max())
        assert np.abs(g - ref[key]).max() <= tol, key
    exact = orc.joint_loss_and_grads(*case, cost_scale=scale)
    np.testing.assert_allclose(costs, exact["costs"], rtol=1e-4)
    # ... and the gradients against the UNROUNDED float64 joint as well (round 5; before: costs only): binary16 operands (h, W2,
    # dlogits) put them 0.4e-4 ... 5.7e-4 of the largest entry away from it -- the bar is 1e-3, the measured ratios are appended to
    # gpurun_out/r05_accuracy_f16.json
    ratios = {}
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        ratios[key] = float(np.abs(g - exact[key]).max() / max(1.0, np.abs(exact[key]).max()))
        assert ratios[key] <= 1e-3, (key, ratios[key])
    _append_accuracy({"shape": [B, T, U, H, J, V], "ragged": bool(ragged), "max_abs_err_over_max_ref_vs_unrounded_f64": ratios})
    enc_g, pred_g = grads[0], grads[1]
    il, ll = case[7], case[8]
    for b in range(B):
        assert not enc_g[b, il[b]:].any() and not pred_g[b, ll[b] + 1:].any()


def test_joint_f16_large_preactivations_take_the_exact_tanh_path():
    """|enc_proj| > 43: the tabulated e^{2x} factors would overflow, the kernels must switch to tanh(a + c)."""
    case = list(make(2, 13, 9, 16, 128, 512, True, seed=5))
    case[0] = case[0] * 60.0   # encoder outputs -> pre-activations of magnitude ~100
    case[1] = case[1] * 60.0 + 2.0
    scale = np.ones(2)
    costs, grads = run(tuple(case), scale)
    ref = orc.joint_loss_and_grads_f16(*case, cost_scale=scale)
    assert np.isfinite(costs).all() and all(np.isfinite(g).all() for g in grads)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        assert np.abs(g - ref[key]).max() <= 1e-3 * max(1.0, np.abs(ref[key]).max()), key


EDGE = [
    # B, T, U, H, J, V, blank, scale
    (2, 1, 6, 8, 128, 512, 0, [1.0, 1.0]),         # a single frame
    (2, 9, 1, 8, 128, 512, 0, [1.0, 0.5]),         # no labels at all (U = 1): blank-only lattice
    (1, 1, 1, 8, 128, 512, 0, [1.0]),              # one cell
    (2, 10, 7, 8, 128, 512, 5, [1.0, 1.0]),        # blank in the middle of a chunk
    (2, 10, 7, 8, 128, 1024, 1023, [1.0, 1.0]),    # blank = last column of the last chunk
    (3, 12, 5, 8, 128, 512, 0, [0.0, -0.5, 3.0]),  # zero / negative / large upstream gradients (dlogits scale 2^12)
]


@pytest.mark.parametrize("B,T,U,H,J,V,blank,scale", EDGE)
def test_joint_f16_edge_cases(B, T, U, H, J, V, blank, scale):
    enc, pred, W1, b1, W2, b2, labels, il, ll = make(B, T, U, H, J, V, B > 1, seed=T * 7 + U + blank)
    if blank:  # labels must avoid the blank id
        labels = np.where(labels == blank, (blank + 1) % V, labels).astype(np.int32)
    case = (enc, pred, W1, b1, W2, b2, labels, il, ll)
    scale = np.asarray(scale, np.float64)
    costs, grads = run(case, scale, blank=blank)
    ref = orc.joint_loss_and_grads_f16(*case, blank=blank, cost_scale=scale)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        assert np.isfinite(g).all(), key
        assert np.abs(g - ref[key]).max() <= 1e-3 * max(1.0, np.abs(ref[key]).max()), key


def test_joint_f16_wide_logit_range_moves_the_softmax_reference():
    """Logits spread over +-60 nats: the lazy softmax reference of K1 must be moved (its rare path) and nothing may
    overflow; the gradients are then essentially one-hot differences."""
    case = list(make(2, 10, 6, 16, 128, 512, True, seed=9))
    case[4] = (case[4] * 25.0).astype(np.float32)  # W2
    scale = np.ones(2)
    costs, grads = run(tuple(case), scale)
    ref = orc.joint_loss_and_grads_f16(*case, cost_scale=scale)
    assert np.isfinite(costs).all() and all(np.isfinite(g).all() for g in grads)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        assert np.abs(g - ref[key]).max() <= 1e-3 * max(1.0, np.abs(ref[key]).max()), key


def test_joint_goldens(golden_dir):
    """HIP fused joint (both dtypes) against the committed fixtures tests/golden/joint/*.npz."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(golden_dir, "joint", "*.npz")))
    assert len(files) >= 2
    for f in files:
        z = np.load(f)
        case = tuple(z[k] for k in ("enc", "pred", "W1", "b1", "W2", "b2", "labels", "input_lengths", "label_lengths"))
        f16 = str(z["joint_dtype"]) == "f16"
        costs, grads = run(case, z["cost_scale"], joint_dtype="f16" if f16 else "f32")
        np.testing.assert_allclose(costs, z["costs"], rtol=1e-4)
        for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
            tol = (1e-3 if f16 else 1e-4) * max(1.0, np.abs(z[key]).max())
            assert np.abs(g - z[key]).max() <= tol, (os.path.basename(f), key)


def test_joint_f16_is_deterministic():
    case = make(2, 40, 40, 16, 128, 512, True, seed=11)
    scale = np.ones(2)
    c1, g1 = run(case, scale)
    c2, g2 = run(case, scale)
    assert np.array_equal(c1, c2)
    for a, b in zip(g1, g2):
        assert np.array_equal(a, b)


def test_joint_f16_odd_shapes_are_padded_exactly():
    """V = 1000 (not a multiple of 128) and J = 320 (BASELINE configs 3/4's joint width): the host layer pads to the
    kernels' shapes (1024, 384) with zero units / zero-probability symbols; results equal the oracle on the UNPADDED problem."""
    case = make(2, 12, 7, 16, 320, 1000, True, seed=21)
    scale = np.array([1.0, 0.5])
    costs, grads = run(case, scale)
    ref = orc.joint_loss_and_grads_f16(*case, cost_scale=scale)
    np.testing.assert_allclose(costs, ref["costs"], rtol=1e-4)
    for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
        assert g.shape == ref[key].shape
        assert np.abs(g - ref[key]).max() <= 1e-3 * max(1.0, np.abs(ref[key]).max()), key


def test_joint_f16_limits_are_reported():
    dev = torch.device("cuda:0")
    enc, pred = torch.zeros(1, 4, 8, device=dev), torch.zeros(1, 3, 8, device=dev)
    W1, b1 = torch.zeros(8, 128, device=dev), torch.zeros(128, device=dev)
    W2, b2 = torch.zeros(128, 9000, device=dev), torch.zeros(9000, device=dev)  # beyond the f16 joint's 8192 symbols
    with pytest.raises(ValueError, match="at most 8192"):
        pkg.rnnt_joint_loss(enc, pred, W1, b1, W2, b2, torch.ones(1, 2, dtype=torch.int32, device=dev),
                            torch.tensor([4], device=dev), torch.tensor([2], device=dev))
    # the raw C ABI still rejects shapes no path implements (V = 200: beyond the f32-grade joint's 128 symbols, not a multiple of
    # 128 for the f16 one); V = 100 is the f32-grade joint's since round 5 (four vocabulary tiles)
    from rnnt_speech_recognition_amd import _lib
    with pytest.raises(RuntimeError, match="invalid value"):
        _lib.joint_workspace_bytes(4, 3, 1, 128, 200)
    assert _lib.joint_workspace_bytes(4, 3, 1, 128, 100) > 0


def test_joint_f16_second_backward_recomputes_the_logits():
    """The first backward call turns the parked softmax numerators into dlogits in place; a second backward over the same
    forward (retain_graph) finds none and takes the recompute route (the J x V product again, one binary16 rounding less):
    each matches the oracle statement of its own route."""
    case = make(2, 20, 40, 24, 256, 1024, True, seed=5)
    enc, pred, W1, b1, W2, b2, labels, il, ll = case
    scale = np.array([1.0, 0.75])
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    params = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
    costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll), joint_dtype="f16")
    loss = (costs * t(scale.astype(np.float32))).sum()
    got = []
    for _ in range(3):
        for p in params:
            p.grad = None
        loss.backward(retain_graph=True)
        torch.cuda.synchronize()
        got.append([p.grad.cpu().numpy() for p in params])
    refs = [orc.joint_loss_and_grads_f16(*case, cost_scale=scale, parked=pk) for pk in (True, False, False)]
    for grads, ref in zip(got, refs):
        for g, key in zip(grads, ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")):
            assert np.abs(g - ref[key]).max() <= 1e-3 * max(1.0, np.abs(ref[key]).max()), key
    for a, b in zip(got[1], got[2]):  # the recompute route is repeatable
        assert np.array_equal(a, b)


def test_joint_f16_costs_only_call_matches_the_training_forward():
    """Under no_grad the costs-only entry runs (nothing parked); same costs as the forward of a training step."""
    case = make(2, 20, 40, 24, 256, 1024, True, seed=6)
    enc, pred, W1, b1, W2, b2, labels, il, ll = case
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    args = [t(x) for x in (enc, pred, W1, b1, W2, b2)]
    with torch.no_grad():
        c0 = pkg.rnnt_joint_loss(*args, t(labels), t(il), t(ll), joint_dtype="f16").cpu().numpy()
    c1, _ = run(case, np.ones(2))
    ref = orc.joint_loss_and_grads_f16(*case)
    np.testing.assert_allclose(c0, ref["costs"], rtol=1e-4)
    np.testing.assert_allclose(c0, c1, rtol=2e-6)
