"""GPU parity of the fused joint at the MID-SIZED vocabularies bench.py times, at the headline lattice size (T = 600, U = 150,
J = 640), through the C ABI -- the shapes whose code changed most in rounds 5 and 6 and whose parity cases stopped at T <= 70:

  f32-grade  V = 64  (two vocabulary tiles, what joint_dtype "auto" picks)      `fused_joint_v64`
  f32-grade  V = 100 (four tiles, the last one partly padded; joint_dtype "f32")
  f16        V = 128 (K4's own 128-column instantiation)                         `fused_joint_v128`
  f16        V = 256

each on N(0,1) projections AND on trained-like posteriors (pkg.synthetic_trained_like_joint: one dominant symbol per cell along a
monotone alignment, every second utterance emitting late), ragged lengths, B = 8.  V is a free hyper-parameter of the reference
(hparams.py:4, utils/encoding.py:72-90).  Three utterances per case (one full-length, two ragged) against
oracle.joint_utterance_streamed -- cost, d enc_proj, d pred_proj -- and the batch's dW2 / db2 through a second call whose
cost_scale is zero for every other utterance (tests/test_baseline_sizes_gpu.py).  Bars: f32-grade 1e-4 (costs relative, gradients
relative to max(1, max|ref|), d enc_proj / d pred_proj also 1e-4 absolute); f16 costs 1e-4 against the rounding-aware AND the
unrounded oracle, gradients 1e-3 max(1, max|ref|).
On the trained-like cases the backward must have visited less than 30 % of the lattice rows (f32-grade: the few-items-per-workgroup
branch of joint_rowplan_kernel; f16, groups of four rows: 35 %), and RNNT_VISIT_ALL must give the same numbers to 1e-6.
The f16 engine again at V = 640 / 1024 on trained-like posteriors (test_large_vocabulary_...): several chunks per K4 tile, most rows skipped."""
import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import joint as joint_mod
from rnnt_speech_recognition_amd.joint import JOINT_DTYPES, _JointLossFunction
from rnnt_speech_recognition_amd import _lib
from tests.test_baseline_sizes_gpu import check_against_oracle, check_properties, make_proj_case, run_fused

pytestmark = pytest.mark.gpu
joint_mod.TRACK_BACKWARD_ROWS = True  # every backward of this file reports how many lattice rows it visited
B, T, U, J = 8, 600, 150, 640
PICKS = [0, 3, 6]  # 0: full length; 3: an odd (late-emitting) utterance; 6: even


def _case(kind, V, seed, nb=B):
    case = list(make_proj_case(nb, T, U, J, V, seed=seed))
    if kind == "trained":
        ep, pp, W2, b2, labels = pkg.synthetic_trained_like_joint(nb, T, U, V, J, seed=seed, input_lengths=case[5], label_lengths=case[6])
        case[0], case[1], case[2], case[3], case[4] = ep, pp, W2, b2, labels
    return tuple(case)


def _run(case, scale, dtype, visit_all=False):
    dev = torch.device("cuda:0")
    ep, pp, W2, b2, labels, il, ll = (x.to(dev) for x in case)
    ps = [x.clone().requires_grad_(True) for x in (ep, pp, W2, b2)]
    word = JOINT_DTYPES[dtype] | (_lib.RNNT_VISIT_ALL if visit_all else 0)
    costs = _JointLossFunction.apply(*ps, labels, il, ll, 0, word)
    (costs * scale.to(dev)).sum().backward()
    torch.cuda.synchronize()
    return costs.detach(), [p.grad for p in ps]


@pytest.mark.parametrize("kind", ["n01", "trained"])
@pytest.mark.parametrize("dtype,V", [("f32", 64), ("f32", 100), ("f16", 128), ("f16", 256)])
def test_mid_vocabulary_joint_at_headline_lattice(dtype, V, kind):
    case = _case(kind, V, seed=600 + V)
    scale = torch.linspace(1.5, 0.5, B) / B  # (utterance 0 carries the largest upstream gradient: the masked call derives the same dlogits scale)
    costs, grads = _run(case, scale, dtype)
    rows = joint_mod.last_backward_rows()
    mask = torch.zeros(B)
    mask[PICKS] = 1.0
    _, grads_masked = _run(case, scale * mask, dtype)
    f16 = dtype == "f16"
    check_against_oracle(case, dtype, PICKS, scale, costs, grads, grads_masked, gtol=1e-3 if f16 else 1e-4, also_exact=f16)
    check_properties(case, costs, grads, lambda: _run(case, scale, dtype))
    # the backward's row pruning (both engines since round 6): how many rows it visited, and that switching it off changes nothing
    assert rows is not None and rows[1] > 0, rows
    frac = rows[0] / rows[1]
    if kind == "trained":
        assert frac < (0.35 if f16 else 0.3), frac  # the alignment band: most rows carry no mass (f16: groups of four rows)
    # no occupancy floor: the same numbers (what the skipped rows would have added are exact zeros; a few f32 sums run in another order)
    costs_all, grads_all = _run(case, scale, dtype, visit_all=True)
    rows_all = joint_mod.last_backward_rows()
    assert rows_all[0] == rows_all[1] == rows[1], (rows_all, rows)
    assert torch.equal(costs, costs_all)
    for a, b in zip(grads, grads_all):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("V", [640, 1024])
def test_large_vocabulary_f16_joint_on_trained_like_posteriors(V):
    """The f16 engine's row pruning where a vocabulary spans several 32-symbol chunks per K4 tile and most rows are skipped: V = 1024 (two full
    512-column tiles of K4, 32 chunks per K3 iteration) and V = 640 (the partly filled last tile), trained-like posteriors, ragged, B = 4 at the
    headline lattice.  (The random sweep's lattices are too small for any row to fall below the floor; BASELINE configs[4]'s at-size test runs
    N(0,1) projections, a third of whose rows are visited.)  Same checks as above on two utterances."""
    nb, picks = 4, [0, 3]
    case = _case("trained", V, seed=900 + V, nb=nb)
    scale = torch.linspace(1.5, 0.5, nb) / nb
    costs, grads = _run(case, scale, "f16")
    rows = joint_mod.last_backward_rows()
    mask = torch.zeros(nb)
    mask[picks] = 1.0
    _, grads_masked = _run(case, scale * mask, "f16")
    check_against_oracle(case, "f16", picks, scale, costs, grads, grads_masked, gtol=1e-3, also_exact=True)
    assert rows is not None and rows[1] > 0 and rows[0] / rows[1] < 0.35, rows
    costs_all, grads_all = _run(case, scale, "f16", visit_all=True)
    assert torch.equal(costs, costs_all)
    for a, b in zip(grads, grads_all):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
