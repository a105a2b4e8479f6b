"""GPU parity tests of the FUSED joint + loss paths at BASELINE.json's full configurations -- the shapes bench.py
times -- through the C ABI (compute_rnnt_joint_loss_fwd / _bwd behind joint._JointLossFunction):

  C2 fused   B=32 T=600  U=150 J=640 V=28    f32-grade products (split-precision f16 MFMAs)   model.py:158-166, hparams.py:18,23
  C5         B=16 T=1500 U=300 J=640 V=1024  f16 MFMA joint / f32 lattice                      BASELINE.json configs[4]
  C3 shape   B=64 T'=300 U=100 H=J=320 V=28  the end-to-end model's joint (configs[2])
  defaults   B=16 T'=300 U=100 J=640 V=4096  the reference's own default hyper-parameters (hparams.py:4,18,23: 4096 word
                                             pieces, joint size 640), f16 MFMA joint -- the `fused_joint_refdefault` bench leg

The float64 oracle cannot hold a whole batch at these sizes, so each test checks
  * C2: EVERY utterance of the batch against oracle.joint_utterance_streamed (evaluated side by side on the host's cores):
    cost, d enc_proj rows, d pred_proj rows, and the batch's dW2 / db2;
    C5 / C3: a few utterances (one full-length, the others ragged) the same way, with dW2 / db2 through a second call whose
    cost_scale is zero for every other utterance (the weight gradients are then exactly those utterances' share);
  * size-independent properties on the full batch: padded rows of d enc_proj / d pred_proj are exactly zero, fused
    costs equal rnnt_loss on materialised logits for a sub-batch, two runs are bit-identical.
Tolerances: f32-grade path 1e-4 (costs relative, gradients relative to max(1, max|ref|)); f16 path costs 1e-4 against
BOTH the rounding-aware oracle and the unrounded one, gradients 1e-3 * max(1, max|ref|) (binary16 dlogits, see
tests/test_joint_f16_gpu.py)."""
import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc
from rnnt_speech_recognition_amd.joint import JOINT_DTYPES, _JointLossFunction

pytestmark = pytest.mark.gpu


def make_proj_case(B, T, U, J, V, seed, w2_gain=1.0):
    """The bench's fused workload: enc_proj / pred_proj ~ N(0,1), glorot W2, ragged lengths with
    utterance 0 at full length."""
    g = torch.Generator().manual_seed(seed)
    ep = torch.randn(B, T, J, generator=g)
    pp = torch.randn(B, U, J, generator=g)
    lim = math.sqrt(6.0 / (J + V)) * w2_gain
    W2 = (torch.rand(J, V, generator=g) * 2 - 1) * lim
    b2 = 0.1 * torch.randn(V, generator=g)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32)
    il = torch.randint((T + 1) // 2, T + 1, (B,), generator=g, dtype=torch.int32)
    ll = torch.randint((U - 1) // 2, U, (B,), generator=g, dtype=torch.int32)
    il[0], ll[0] = T, U - 1
    return ep, pp, W2, b2, labels, il, ll


def run_fused(case, scale, dtype):
    dev = torch.device("cuda:0")
    ep, pp, W2, b2, labels, il, ll = (x.to(dev) for x in case)
    ps = [x.clone().requires_grad_(True) for x in (ep, pp, W2, b2)]
    costs = _JointLossFunction.apply(*ps, labels, il, ll, 0, JOINT_DTYPES[dtype])
    (costs * scale.to(dev)).sum().backward()
    torch.cuda.synchronize()
    return costs.detach(), [p.grad for p in ps]


def check_against_oracle(case, dtype, picks, scale, costs, grads, grads_masked, gtol, ctol=1e-4, also_exact=False):
    ep, pp, W2, b2, labels, il, ll = (x.numpy() for x in case)
    f16 = dtype == "f16"
    B = ep.shape[0]
    S = orc.dl_scale_f16(scale.numpy(), B)
    dW2_ref, db2_ref = 0.0, 0.0
    c = costs.cpu().numpy().astype(np.float64)
    d_ep, d_pp = grads[0].cpu().numpy(), grads[1].cpu().numpy()

    def one(b):  # (NumPy releases the GIL in its kernels: the utterances are evaluated side by side on the host's cores)
        Tb, Ub = int(il[b]), int(ll[b]) + 1
        o = orc.joint_utterance_streamed(ep[b, :Tb], pp[b, :Ub], W2, b2, labels[b, : Ub - 1], cost_scale=float(scale[b]),
                                         f16=f16, dl_scale=S)
        ex = None
        if also_exact:  # distance to the UNROUNDED joint: the price of binary16 operands, bounded at the same bar
            ex = orc.joint_utterance_streamed(ep[b, :Tb], pp[b, :Ub], W2, b2, labels[b, : Ub - 1], f16=False, want_grads=False)["cost"]
        return o, ex

    with ThreadPoolExecutor(max_workers=max(1, min(len(picks), (os.cpu_count() or 2) // 2, 16))) as pool:
        results = list(pool.map(one, picks))
    for b, (o, ex) in zip(picks, results):
        Tb, Ub = int(il[b]), int(ll[b]) + 1
        assert abs(c[b] - o["cost"]) <= ctol * max(1.0, abs(o["cost"])), (b, c[b], o["cost"])
        if ex is not None:
            assert abs(c[b] - ex) <= ctol * max(1.0, abs(ex)), (b, c[b], ex)
        for name, got, ref in (("d_enc_proj", d_ep[b, :Tb], o["d_enc_proj"]), ("d_pred_proj", d_pp[b, :Ub], o["d_pred_proj"])):
            assert np.abs(got - ref).max() <= gtol * max(1.0, np.abs(ref).max()), (b, name, np.abs(got - ref).max())
            if not f16:  # the f32-grade joint also meets north_star's ABSOLUTE figure on the two activation gradients
                _abs_report[name] = max(_abs_report.get(name, 0.0), float(np.abs(got - ref).max()))
                assert np.abs(got - ref).max() <= 1e-4, (b, name, np.abs(got - ref).max(), np.abs(ref).max())
        dW2_ref, db2_ref = dW2_ref + o["dW2"], db2_ref + o["db2"]
    for name, got, ref in (("dW2", grads_masked[2].cpu().numpy(), dW2_ref), ("db2", grads_masked[3].cpu().numpy(), db2_ref)):
        assert np.abs(got - ref).max() <= gtol * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())


_abs_report = {}  # max |d| of d enc_proj / d pred_proj over the f32-grade cases of this file -> gpurun_out/r05_accuracy_fused.json


@pytest.fixture(scope="module", autouse=True)
def _write_abs_report():
    yield
    import json
    import os

    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r05_accuracy_fused.json"), "w") as f:
        json.dump({"fused_f32_max_abs_d_activation_gradients": _abs_report, "bar": 1e-4}, f, indent=1)


def check_properties(case, costs, grads, rerun):
    _, _, _, _, _, il, ll = case
    d_ep, d_pp = grads[0], grads[1]
    for b in range(d_ep.shape[0]):  # padded frames / label positions: exact zeros
        assert not bool(d_ep[b, int(il[b]):].any()) and not bool(d_pp[b, int(ll[b]) + 1:].any()), b
    assert all(bool(torch.isfinite(g).all()) for g in grads) and bool(torch.isfinite(costs).all())
    costs2, grads2 = rerun()
    assert torch.equal(costs, costs2)  # bitwise determinism (fixed-order reductions, no floating-point atomics)
    for a, b in zip(grads, grads2):
        assert torch.equal(a, b)


def unfused_costs(case, nsub, f16):
    """The reference's own composition on a sub-batch: materialised logits (model.py:162-166) -> rnnt_loss."""
    dev = torch.device("cuda:0")
    ep, pp, W2, b2, labels, il, ll = case
    ep, pp, labels, il, ll = (x[:nsub].to(dev) for x in (ep, pp, labels, il, ll))
    W2, b2 = W2.to(dev), b2.to(dev)
    h = torch.tanh(ep[:, :, None, :] + pp[:, None, :, :])
    if f16:
        h, W2 = h.half().float(), W2.half().float()
    logits = h @ W2 + b2
    del h
    return pkg.rnnt_loss(logits, labels, il, ll).cpu().numpy()


def test_c2_fused_joint_at_bench_size():
    B, T, U, J, V = 32, 600, 150, 640, 28
    case = make_proj_case(B, T, U, J, V, seed=2024)
    scale = torch.linspace(0.5, 1.5, B) / B
    costs, grads = run_fused(case, scale, "f32")
    picks = list(range(B))  # EVERY utterance of the batch (round 3 checked three): the weight gradients are then the call's own
    check_against_oracle(case, "f32", picks, scale, costs, grads, grads, gtol=1e-4)
    check_properties(case, costs, grads, lambda: run_fused(case, scale, "f32"))
    np.testing.assert_allclose(costs[:4].cpu().numpy(), unfused_costs(case, 4, False), rtol=2e-5)


def test_c3_joint_shape_at_size():
    """configs[2]'s joint: B=64, T'=300 (600 frames, x2 time reduction), U=100, H=J=320, V=28 -- five 64-wide J slabs,
    four u-tiles, row splits; through the model-level entry (the first Dense layer inside the library too, as train steps do)."""
    B, T, U, H, J, V = 64, 300, 100, 320, 320, 28
    rng = np.random.default_rng(33)
    enc = rng.normal(size=(B, T, H)).astype(np.float32)
    pred = rng.normal(size=(B, U, H)).astype(np.float32)
    lim1, lim2 = np.sqrt(6.0 / (H + J)), np.sqrt(6.0 / (J + V))
    W1 = rng.uniform(-lim1, lim1, size=(H, J)).astype(np.float32)
    b1 = (0.1 * rng.normal(size=J)).astype(np.float32)
    W2 = (rng.uniform(-lim2, lim2, size=(J, V)) * 3.0).astype(np.float32)
    b2 = (0.1 * rng.normal(size=V)).astype(np.float32)
    labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    il = rng.integers(T // 2, T + 1, size=B).astype(np.int32)
    ll = rng.integers(U // 2, U, size=B).astype(np.int32)
    il[0], ll[0] = T, U - 1
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)

    def run(scale):
        ps = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
        costs = pkg.rnnt_joint_loss(*ps, t(labels), t(il), t(ll))
        (costs * t(scale.astype(np.float32))).sum().backward()
        torch.cuda.synchronize()
        return costs.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in ps]

    scale = np.full(B, 1.0 / B)
    costs, grads = run(scale)
    picks = [0, 21, 63]
    mask = np.zeros(B)
    mask[picks] = 1.0
    _, gm = run(scale * mask)
    sub = [x[picks] for x in (enc, pred)]
    ref = orc.joint_loss_and_grads(sub[0], sub[1], W1, b1, W2, b2, labels[picks], il[picks], ll[picks], cost_scale=scale[picks])
    np.testing.assert_allclose(costs[picks], ref["costs"], rtol=1e-4)
    for k, key in ((0, "d_enc"), (1, "d_pred")):
        assert np.abs(grads[k][picks] - ref[key]).max() <= 1e-4 * max(1.0, np.abs(ref[key]).max()), key
    for k, key in ((2, "dW1"), (3, "db1"), (4, "dW2"), (5, "db2")):
        assert np.abs(gm[k] - ref[key]).max() <= 1e-4 * max(1.0, np.abs(ref[key]).max()), key
    for b in range(B):
        assert not grads[0][b, il[b]:].any() and not grads[1][b, ll[b] + 1:].any()
    c2, g2 = run(scale)
    assert np.array_equal(costs, c2) and all(np.array_equal(a, b) for a, b in zip(grads, g2))


def test_c5_fused_f16_joint_at_full_size():
    B, T, U, J, V = 16, 1500, 300, 640, 1024
    case = make_proj_case(B, T, U, J, V, seed=555, w2_gain=3.0)
    case[5][1], case[6][1] = 420, 140  # short utterances keep the other oracle passes cheap
    case[5][2], case[6][2] = 701, 97
    case[5][3], case[6][3] = 233, 299
    # utterance 0 carries the largest upstream gradient: the masked call below then derives the same power-of-two dlogits
    # scale (from max|cost_scale| of the call) as the full call
    scale = torch.linspace(1.5, 0.5, B) / B
    costs, grads = run_fused(case, scale, "f16")
    picks = list(range(8))  # half the batch: one FULL-length utterance (450,000 cells x 1024 symbols, streamed in float64), one short, six ragged
    mask = torch.zeros(B)
    mask[picks] = 1.0
    _, grads_masked = run_fused(case, scale * mask, "f16")
    check_against_oracle(case, "f16", picks, scale, costs, grads, grads_masked, gtol=1e-3, also_exact=True)
    check_properties(case, costs, grads, lambda: run_fused(case, scale, "f16"))
    np.testing.assert_allclose(costs[:1].cpu().numpy(), unfused_costs(case, 1, True), rtol=5e-5)


def test_reference_default_hparams_f16_joint_at_size():
    """The reference's default hyper-parameters (hparams.py:4 vocab_size 4096, :23 joint_net_size 640) on a realistic lattice --
    600 frames after x2 time reduction, 99 word pieces: the shape bench.py's `fused_joint_refdefault` leg times.  Three
    utterances (one full-length, two ragged) against the streamed float64 joint with the f16 path's roundings restated, costs
    also against the unrounded joint; properties on the whole batch."""
    B, T, U, J, V = 16, 300, 100, 640, 4096
    case = make_proj_case(B, T, U, J, V, seed=4096, w2_gain=3.0)
    case[5][1], case[6][1] = 171, 60
    case[5][2], case[6][2] = 290, 31
    scale = torch.linspace(1.5, 0.5, B) / B  # (utterance 0 carries the largest upstream gradient: see the C5 test)
    costs, grads = run_fused(case, scale, "f16")
    picks = [0, 1, 2]
    mask = torch.zeros(B)
    mask[picks] = 1.0
    _, grads_masked = run_fused(case, scale * mask, "f16")
    check_against_oracle(case, "f16", picks, scale, costs, grads, grads_masked, gtol=1e-3, also_exact=True)
    check_properties(case, costs, grads, lambda: run_fused(case, scale, "f16"))
    np.testing.assert_allclose(costs[:1].cpu().numpy(), unfused_costs(case, 1, True), rtol=5e-5)
