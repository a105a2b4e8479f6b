"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libwarprnnt.so), against the
float64 oracle, the committed goldens and the C restatement of the reference CPU path.

Tolerances (BASELINE.json north_star: "within 1e-4 fp32"):
  costs  |d| <= 1e-4 * max(1, |cost|)   (a cost of ~2200 has an fp32 ulp of 2.4e-4, so the bar is relative)
  grads  max |d| <= 1e-4 absolute       (gradients live in [-1, 1])
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg
from oracle import cpu_oracle, rnnt_oracle as orc

pytestmark = pytest.mark.gpu

GTOL = 1e-4
CTOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    pkg.build()


def run_hip(acts, labels, il, ll, blank=0):
    dev = torch.device("cuda:0")
    costs, grads = pkg.rnnt_loss_and_grad(
        torch.as_tensor(acts, dtype=torch.float32, device=dev),
        torch.as_tensor(np.asarray(labels), dtype=torch.int32, device=dev),
        torch.as_tensor(np.asarray(il), dtype=torch.int32, device=dev),
        torch.as_tensor(np.asarray(ll), dtype=torch.int32, device=dev), blank)
    torch.cuda.synchronize()
    return costs.cpu().numpy().astype(np.float64), grads.cpu().numpy()


def check(acts, labels, il, ll, blank=0, gtol=GTOL, ctol=CTOL):
    c_ref, g_ref = orc.rnnt_loss_and_grad(acts, labels, il, ll, blank=blank)
    c, g = run_hip(acts, labels, il, ll, blank)
    assert np.isfinite(c).all() and np.isfinite(g).all()
    np.testing.assert_array_less(np.abs(c - c_ref), ctol * np.maximum(1.0, np.abs(c_ref)))
    assert np.abs(g - g_ref).max() <= gtol
    # padded cells: exact zeros
    for b in range(acts.shape[0]):
        assert not g[b, int(il[b]):].any() and not g[b, :, int(ll[b]) + 1:].any()
    return c, g


def make_case(B, T, U, V, ragged, seed, blank=0):
    rng = np.random.default_rng(seed)
    acts = rng.normal(size=(B, T, U, V)).astype(np.float32)
    pool = [v for v in range(V) if v != blank] or [0]
    labels = rng.choice(pool, size=(B, max(U - 1, 1))).astype(np.int32)[:, : max(U - 1, 0)]
    if ragged:
        il = rng.integers((T + 1) // 2, T + 1, size=B)
        ll = rng.integers(U // 2, U, size=B)
        il[0], ll[0] = T, U - 1
    else:
        il, ll = np.full(B, T), np.full(B, U - 1)
    return acts, labels, il.astype(np.int32), ll.astype(np.int32)


def test_kat(golden_dir):
    with open(os.path.join(golden_dir, "kat_small.json")) as f:
        k = json.load(f)
    c, g = run_hip(np.array(k["logits"], np.float32), k["labels"], k["input_lengths"], k["label_lengths"], k["blank"])
    assert abs(c[0] - k["cost_f64"]) < 1e-5
    np.testing.assert_allclose(g, np.array(k["grads_wrt_logits"]), atol=1e-5, rtol=0)


def test_kat_b2(golden_dir):
    """Upstream's two-utterance known-answer vector (tests/golden/kat_b2.json): published costs and gradients."""
    with open(os.path.join(golden_dir, "kat_b2.json")) as f:
        k = json.load(f)
    x = np.array(k["logits_flat"], np.float32).reshape(k["B"], k["T"], k["U"], k["V"])
    c, g = run_hip(x, k["labels"], k["input_lengths"], k["label_lengths"], k["blank"])
    np.testing.assert_allclose(c, k["costs_published"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(g.reshape(-1), k["grads_wrt_logits_flat_published"], atol=1e-5, rtol=0)


def test_goldens(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "*.npz")))
    assert len(files) >= 5
    for f in files:
        d = np.load(f)
        c, g = run_hip(d["acts"], d["labels"], d["input_lengths"], d["label_lengths"], int(d["blank"]))
        np.testing.assert_array_less(np.abs(c - d["costs"]), CTOL * np.maximum(1.0, np.abs(d["costs"])), err_msg=f)
        assert np.abs(g - d["grads"]).max() <= GTOL, f


SHAPES = [
    # B, T, U, V            exercises
    (1, 1, 1, 3),           # single cell
    (2, 1, 5, 7),           # one frame
    (2, 9, 1, 6),           # blanks only
    (2, 5, 1, 1),           # V = 1 (only the blank exists)
    (3, 17, 64, 28),        # K=1 sweep, full lane use
    (2, 33, 65, 28),        # K=2
    (2, 40, 130, 28),       # K=3
    (1, 20, 200, 16),       # K=4
    (1, 12, 300, 8),        # K=6
    (1, 8, 500, 5),         # K=8   (V%4 != 0 but total%4 == 0)
    (1, 6, 700, 4),         # K=12
    (1, 4, 1000, 4),        # K=16
    (1, 7, 5, 31),          # reference char vocab (31), total % 4 != 0 -> wave path, scalar loads
    (2, 7, 6, 31),          # same V, total % 4 == 0 -> lane-per-cell path without float4 LDS reads
    (4, 37, 70, 31),        # ... several patches per utterance, ragged patch edges in t and u
    (4, 20, 33, 29),        # the 29-symbol character set
    (4, 12, 9, 30),         # V % 4 == 2
    (2, 6, 5, 60),          # largest lane-per-cell vocabulary
    (2, 6, 5, 61),          # first wave-per-cell vocabulary
    (2, 6, 5, 64),
    (1, 5, 4, 257),
    (2, 9, 7, 1024),        # BPE-sized vocabulary (config 5's V)
    (3, 100, 30, 28),       # several chunks + several rebase blocks
]


@pytest.mark.parametrize("B,T,U,V", SHAPES)
@pytest.mark.parametrize("ragged", [False, True])
def test_shapes(B, T, U, V, ragged):
    acts, labels, il, ll = make_case(B, T, U, V, ragged, seed=B * 1000 + T * 7 + U * 3 + V)
    check(acts, labels, il, ll)


def test_extreme_raggedness():
    acts, labels, il, ll = make_case(6, 37, 21, 28, False, seed=9)
    il = np.array([37, 1, 1, 19, 37, 2], np.int32)
    ll = np.array([20, 0, 20, 0, 7, 1], np.int32)
    check(acts, labels, il, ll)


def test_nonzero_blank_and_label_equal_to_blank():
    acts, labels, il, ll = make_case(3, 11, 6, 12, True, seed=5, blank=11)
    check(acts, labels, il, ll, blank=11)
    # labels that equal the blank id are legal inputs for the op (both corrections hit one entry)
    acts, labels, il, ll = make_case(2, 8, 5, 9, False, seed=6)
    labels[:, 1] = 0
    check(acts, labels, il, ll, blank=0)


def test_char_vocab_31_edge_symbols():
    """The reference's 31-symbol character set with the blank / labels among the last symbols of the row."""
    for blank in (30, 28, 27):
        acts, labels, il, ll = make_case(4, 26, 35, 31, True, seed=70 + blank, blank=blank)
        labels[:, ::3] = 29 if blank != 29 else 28
        labels[:, 1::3] = 27 if blank != 27 else 26
        check(acts, labels, il, ll, blank=blank)


def test_large_magnitude_logits():
    """Peaked distributions: log-probs down to about -60, long chains of near-zero probability."""
    acts, labels, il, ll = make_case(2, 60, 20, 28, True, seed=21)
    check(acts * 8.0, labels, il, ll)


def test_headline_config_c2_full_size():
    """BASELINE.json configs[1]: B=32 T=600 U=150 V=28 at full size."""
    B, T, U, V = 32, 600, 150, 28
    acts, labels, il, ll = make_case(B, T, U, V, False, seed=1234)
    c, g = run_hip(acts, labels, il, ll)
    assert np.isfinite(c).all() and np.isfinite(g).all()
    # (1) float64 oracle on a subset of utterances (0.6 s each)
    for b in (0, 13, 31):
        c_ref, g_ref, _, _, _ = orc.utterance_cost_and_grad(acts[b], labels[b])
        assert abs(c[b] - c_ref) <= CTOL * abs(c_ref)
        assert np.abs(g[b] - g_ref).max() <= GTOL
    # (2) C restatement of the reference CPU path on ALL utterances.  That path keeps alpha/beta as
    # raw float32 of magnitude ~2e3 (ulp 2.4e-4), so it is itself ~1e-3 away from exact: looser bar.
    lp = torch.log_softmax(torch.from_numpy(acts), dim=-1).numpy()
    c32, glp = cpu_oracle.rnnt_cpu(lp, labels, il, ll)
    g32 = glp - np.exp(lp) * glp.sum(-1, keepdims=True)
    np.testing.assert_allclose(c, c32, rtol=2e-5)
    assert np.abs(g - g32).max() <= 3e-3
    # (3) size-independent properties: every cell's fused-softmax gradient sums to zero, and the
    # blank+label mass leaving each anti-diagonal is exactly one path's worth
    assert np.abs(g.sum(-1)).max() <= 2e-5


def test_headline_config_c2_ragged():
    B, T, U, V = 8, 600, 150, 28
    acts, labels, il, ll = make_case(B, T, U, V, True, seed=77)
    c, g = run_hip(acts, labels, il, ll)
    for b in (0, 3):
        Tb, Ub = int(il[b]), int(ll[b]) + 1
        c_ref, g_ref, _, _, _ = orc.utterance_cost_and_grad(acts[b, :Tb, :Ub], labels[b, : Ub - 1])
        assert abs(c[b] - c_ref) <= CTOL * abs(c_ref)
        assert np.abs(g[b, :Tb, :Ub] - g_ref).max() <= GTOL
        assert not g[b, Tb:].any() and not g[b, :, Ub:].any()


def test_large_vocab_slice_of_c5():
    """BASELINE.json configs[4] shape family (U=300, V=1024), reduced B and T to keep the oracle fast."""
    acts, labels, il, ll = make_case(2, 120, 300, 1024, True, seed=55)
    check(acts, labels, il, ll)


def test_bitwise_determinism():
    acts, labels, il, ll = make_case(4, 90, 40, 28, True, seed=31)
    c1, g1 = run_hip(acts, labels, il, ll)
    c2, g2 = run_hip(acts, labels, il, ll)
    assert np.array_equal(c1, c2) and np.array_equal(g1, g2)


def test_every_sweep_width():
    """Every column-width instantiation of the sweep kernel (K = 1, 2, 3, 4, 6, 8 columns per lane) on ragged batches."""
    acts, labels, il, ll = make_case(3, 200, 150, 28, True, seed=41)
    check(acts, labels, il, ll)
    for U in (40, 100, 250, 330, 500):  # 1, 2, 4, 6, 8 column groups
        acts, labels, il, ll = make_case(2, 30, U, 8, True, seed=U)
        check(acts, labels, il, ll)


@pytest.mark.parametrize("shape", [(1, 40, 700, 4), (2, 24, 1024, 4), (2, 3000, 20, 8), (1, 1, 130, 8), (3, 90, 1, 8)])
def test_widest_and_longest_lattices(shape):
    """Up to 16 columns per lane (U = 1024, the documented limit), thousands of diagonals, and the degenerate
    single-row / single-column lattices."""
    B, T, U, V = shape
    acts, labels, il, ll = make_case(B, T, U, V, True, seed=T + U)
    check(acts, labels, il, ll)


@pytest.mark.parametrize("shape", [(2, 20, 1100, 4), (1, 9, 2049, 8), (2, 300, 1030, 4)])
def test_label_sequences_beyond_1024_take_the_wide_sweep(shape):
    """1024 < maxU <= 8192: the multi-wave sweep (previous diagonal in LDS, a barrier per diagonal) behind the same
    lsm / gradient kernels; upstream has no limit on U, this is what keeps the boundary a drop-in there."""
    B, T, U, V = shape
    acts, labels, il, ll = make_case(B, T, U, V, True, seed=T + U)
    check(acts, labels, il, ll)


def test_back_to_back_calls_on_one_stream():
    """Several calls in flight on one stream, different workspaces/inputs: stream order is the only dependency."""
    dev = torch.device("cuda:0")
    cases = [make_case(5, 80, 33, 28, True, seed=100 + i) for i in range(4)]
    outs = []
    for acts, labels, il, ll in cases:  # enqueue everything before any synchronisation
        outs.append(pkg.rnnt_loss_and_grad(torch.tensor(acts, device=dev), torch.tensor(labels, device=dev),
                                           torch.tensor(il, device=dev), torch.tensor(ll, device=dev)))
    torch.cuda.synchronize()
    for (acts, labels, il, ll), (c, g) in zip(cases, outs):
        c_ref, g_ref = orc.rnnt_loss_and_grad(acts, labels, il, ll)
        np.testing.assert_allclose(c.cpu().numpy(), c_ref, rtol=CTOL)
        assert np.abs(g.cpu().numpy() - g_ref).max() <= GTOL


def test_out_of_range_lengths_are_contained():
    """Lengths are device data: the kernels clamp them into the tensor (no out-of-bounds access) and report the
    offending utterance as NaN -- never a plausible number; the other utterances of the batch are unaffected."""
    acts, labels, il, ll = make_case(5, 40, 17, 28, True, seed=77)
    il_bad, ll_bad = il.copy(), ll.copy()
    il_bad[1] = 41          # T_b > maxT
    ll_bad[2] = 17          # L_b > maxU - 1
    il_bad[3] = 0           # T_b < 1
    c, g = run_hip(acts, labels, il_bad, ll_bad)
    c_ref, g_ref = orc.rnnt_loss_and_grad(acts, labels, il, ll)
    for b in (1, 2, 3):
        assert np.isnan(c[b])
    for b in (0, 4):
        assert abs(c[b] - c_ref[b]) <= CTOL * max(1.0, abs(c_ref[b]))
        assert np.abs(g[b] - g_ref[b]).max() <= GTOL
    # a second, well-formed call right behind it sees no leftovers
    check(acts, labels, il, ll)


def test_autograd_folds_upstream_gradient():
    """run_rnnt.py:278: loss = sum(costs) / global_batch; gradient reaches the logits scaled."""
    acts, labels, il, ll = make_case(4, 30, 12, 28, True, seed=61)
    dev = torch.device("cuda:0")
    x = torch.tensor(acts, device=dev, requires_grad=True)
    w = torch.tensor([0.25, 0.5, 1.0, 2.0], device=dev)
    costs = pkg.rnnt_loss(x, torch.tensor(labels, device=dev), torch.tensor(il, device=dev),
                          torch.tensor(ll, device=dev))
    (costs * w).sum().backward()
    c_ref, g_ref = orc.rnnt_loss_and_grad(acts, labels, il, ll)
    g_ref = g_ref * w.cpu().numpy()[:, None, None, None]
    assert np.abs(x.grad.cpu().numpy() - g_ref).max() <= GTOL * 2.0
    np.testing.assert_allclose(costs.detach().cpu().numpy(), c_ref, rtol=CTOL)


def test_get_loss_fn_mirrors_reference_adapter():
    """utils/loss.py:24-36: labels cast to int32, T_b = ceil(spec_len / reduction_factor)."""
    acts, labels, _, ll = make_case(3, 20, 8, 28, True, seed=71)
    spec = np.array([40, 39, 21])  # -> 20, 20, 11
    dev = torch.device("cuda:0")
    fn = pkg.get_loss_fn(2)
    costs = fn(torch.tensor(labels, device=dev).to(torch.int64), torch.tensor(acts, device=dev),
               torch.tensor(spec, device=dev), torch.tensor(ll, device=dev))
    c_ref, _ = orc.rnnt_loss_and_grad(acts, labels, [20, 20, 11], ll)
    np.testing.assert_allclose(costs.cpu().numpy(), c_ref, rtol=CTOL)
    # keyword form, as run_rnnt.py:405-407 calls it
    costs_kw = fn(y_true=torch.tensor(labels, device=dev), y_pred=torch.tensor(acts, device=dev),
                  spec_lengths=torch.tensor(spec, device=dev), label_lengths=torch.tensor(ll, device=dev))
    assert torch.equal(costs, costs_kw)


def test_module_and_input_checks():
    dev = torch.device("cuda:0")
    acts, labels, il, ll = make_case(2, 6, 4, 9, False, seed=81)
    m = pkg.RNNTLoss(reduction="mean")
    out = m(torch.tensor(acts, device=dev), torch.tensor(labels, device=dev), torch.tensor(il, device=dev),
            torch.tensor(ll, device=dev))
    c_ref, _ = orc.rnnt_loss_and_grad(acts, labels, il, ll)
    assert abs(out.item() - c_ref.mean()) < 1e-3
    with pytest.raises(TypeError):
        pkg.rnnt_loss(torch.tensor(acts, device=dev).half(), torch.tensor(labels, device=dev),
                      torch.tensor(il, device=dev), torch.tensor(ll, device=dev))
    with pytest.raises(ValueError):
        pkg.rnnt_loss(torch.tensor(acts, device=dev), torch.tensor(labels[:, :1], device=dev),
                      torch.tensor(il, device=dev), torch.tensor(ll, device=dev))


def test_calls_can_be_captured_in_a_hip_graph():
    """The library allocates nothing, never synchronises and enqueues everything on the caller's stream: a whole
    loss + gradient call (and the fused joint) records into a HIP graph and replays with new inputs in the same buffers."""
    from rnnt_speech_recognition_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    B, T, U, V = 3, 37, 21, 28
    acts = torch.zeros(B, T, U, V, device=dev)
    labels = torch.tensor(rng.integers(1, V, size=(B, U - 1)), dtype=torch.int32, device=dev)
    il = torch.tensor([T, T - 5, T - 11], dtype=torch.int32, device=dev)
    ll = torch.tensor([U - 1, U - 4, U - 2], dtype=torch.int32, device=dev)
    costs = torch.empty(B, device=dev)
    grads = torch.empty_like(acts)
    ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)

    def call(stream):
        opts = _lib.make_options(stream.cuda_stream, 0, T, U)
        _lib.check(lib.compute_rnnt_loss(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(),
                                         V, B, costs.data_ptr(), ws.data_ptr(), opts), "compute_rnnt_loss")

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):  # warm-up outside the capture (first-use function attributes)
        call(side)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream())
    for seed in (1, 2):
        x = np.random.default_rng(seed).normal(size=(B, T, U, V)).astype(np.float32)
        acts.copy_(torch.from_numpy(x))
        graph.replay()
        torch.cuda.synchronize()
        cr, gr = orc.rnnt_loss_and_grad(x, labels.cpu().numpy(), il.cpu().numpy(), ll.cpu().numpy())
        assert np.abs(costs.cpu().numpy() - cr).max() <= CTOL * max(1.0, np.abs(cr).max())
        assert np.abs(grads.cpu().numpy() - gr).max() <= GTOL


def test_wide_sweeps_and_the_whole_joint_network_record_into_a_hip_graph():
    """Launch paths the small capture above does not reach: sweeps whose LDS ring exceeds 64 KB (hipFuncSetAttribute on the
    launch path, K >= 3 columns per lane at U = 150), the persistent joint forward / backward kernels and the dense layer's
    GEMMs (dynamic LDS above 64 KB, device-attribute queries).  One compute_rnnt_loss and one compute_rnnt_joint_net_loss
    call are captured; replays on new inputs must equal direct calls on the same inputs bit for bit.
    (This test found that hipMemsetAsync NODES replay a garbage pattern on this stack: the library fills through its own
    kernel, launch_fill, since.)"""
    from rnnt_speech_recognition_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, T, U, V, H, J = 2, 40, 150, 28, 64, 128
    g = torch.Generator().manual_seed(3)
    acts = torch.zeros(B, T, U, V, device=dev)
    enc, pred = torch.zeros(B, T, H, device=dev), torch.zeros(B, U, H, device=dev)
    W1 = ((torch.rand(H, J, generator=g) * 2 - 1) * 0.2).to(dev)
    b1 = (0.1 * torch.randn(J, generator=g)).to(dev)
    W2 = ((torch.rand(J, V, generator=g) * 2 - 1) * 0.3).to(dev)
    b2 = (0.1 * torch.randn(V, generator=g)).to(dev)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
    il = torch.tensor([T, T - 7], dtype=torch.int32, device=dev)
    ll = torch.tensor([U - 1, U - 30], dtype=torch.int32, device=dev)
    scale = torch.full((B,), 0.5, device=dev)
    costs, costs_j = torch.empty(B, device=dev), torch.empty(B, device=dev)
    grads = torch.empty_like(acts)
    outs = [torch.empty_like(x) for x in (enc, pred, W1, b1, W2, b2)]
    ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
    wsj = torch.empty(_lib.joint_net_workspace_bytes(T, U, B, H, J, V), dtype=torch.uint8, device=dev)
    # the f16 joint through its autograd split: _fwd parks the softmax numerators, the first _bwd consumes them in place
    # (streaming kernel), the second one finds the workspace state changed and recomputes the logits
    V16 = 512
    ep16, pp16 = torch.zeros(B, T, J, device=dev), torch.zeros(B, U, J, device=dev)
    W216 = ((torch.rand(J, V16, generator=g) * 2 - 1) * 0.3).to(dev)
    b216 = (0.1 * torch.randn(V16, generator=g)).to(dev)
    labels16 = torch.randint(1, V16, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
    costs16 = torch.empty(B, device=dev)
    outs16 = [[torch.empty_like(x) for x in (ep16, pp16, W216, b216)] for _ in range(2)]
    ws16 = torch.empty(_lib.joint_workspace_bytes(T, U, B, J, V16), dtype=torch.uint8, device=dev)

    def call(stream):
        opts = _lib.make_options(stream.cuda_stream, 0, T, U)
        _lib.check(lib.compute_rnnt_loss(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(),
                                         V, B, costs.data_ptr(), ws.data_ptr(), opts), "compute_rnnt_loss")
        _lib.check(lib.compute_rnnt_joint_net_loss(enc.data_ptr(), pred.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(),
                                                   b2.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(), scale.data_ptr(),
                                                   H, J, V, B, costs_j.data_ptr(), *(o.data_ptr() for o in outs), 0, wsj.data_ptr(),
                                                   opts), "compute_rnnt_joint_net_loss")
        args16 = (ep16.data_ptr(), pp16.data_ptr(), W216.data_ptr(), b216.data_ptr(), labels16.data_ptr(), ll.data_ptr(), il.data_ptr())
        _lib.check(lib.compute_rnnt_joint_loss_fwd(*args16, J, V16, B, costs16.data_ptr(), 1, ws16.data_ptr(), opts),
                   "compute_rnnt_joint_loss_fwd")
        for o in outs16:
            _lib.check(lib.compute_rnnt_joint_loss_bwd(*args16, scale.data_ptr(), J, V16, B, *(x.data_ptr() for x in o), 1,
                                                       ws16.data_ptr(), opts), "compute_rnnt_joint_loss_bwd")

    everything = lambda: (costs, grads, costs_j, *outs, costs16, *outs16[0], *outs16[1])

    def snapshot():  # (the clones run on the default stream, the direct calls on `side`: fence both ways)
        torch.cuda.synchronize()
        snap = [x.clone() for x in everything()]
        torch.cuda.synchronize()
        return snap

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        call(side)  # first use outside the capture
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream())
    for seed in (11, 12):
        gg = torch.Generator().manual_seed(seed)
        acts.copy_(torch.randn(B, T, U, V, generator=gg))
        enc.copy_(torch.randn(B, T, H, generator=gg))
        pred.copy_(torch.randn(B, U, H, generator=gg))
        ep16.copy_(torch.randn(B, T, J, generator=gg))
        pp16.copy_(torch.randn(B, U, J, generator=gg))
        graph.replay()
        replayed = snapshot()
        for x in everything():
            x.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            call(side)
        side.synchronize()
        direct = snapshot()
        assert all(bool(torch.isfinite(x).all()) for x in direct)
        names = ("costs", "grads", "costs_joint", "d_enc", "d_pred", "dW1", "db1", "dW2", "db2", "costs_f16",
                 "f16_bwd1_d_enc_proj", "f16_bwd1_d_pred_proj", "f16_bwd1_dW2", "f16_bwd1_db2",
                 "f16_bwd2_d_enc_proj", "f16_bwd2_d_pred_proj", "f16_bwd2_dW2", "f16_bwd2_db2")
        bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, replayed, direct) if not torch.equal(a, b)]
        assert not bad, bad
        # the two backward routes agree to binary16 rounding noise, and are not the same computation
        d1, d2 = direct[10:14], direct[14:18]
        for a, b in zip(d1, d2):
            assert float((a - b).abs().max()) <= 2e-3 * max(1.0, float(b.abs().max()))
        assert any(not torch.equal(a, b) for a, b in zip(d1, d2))


def test_occupancy_floor_and_its_opt_out():
    """Vocabularies above 60 symbols: a cell whose occupancy alpha.beta/L is at most 2^-50 gets exact zeros and its logits are not
    read (include/rnnt.h).  RNNT_VISIT_ALL (compute_rnnt_loss_flags) switches the floor off: the same numbers to 2^-44.  A NaN
    logit inside such a cell is NOT hidden by the floor: the forward pass reads every cell, so the utterance's lattice, cost and
    gradients are NaN with or without the flag (costs never depend on the floor)."""
    import rnnt_speech_recognition_amd as pkg

    B, T, U, V = 2, 60, 20, 64
    rng = np.random.default_rng(5)
    acts = rng.normal(size=(B, T, U, V)).astype(np.float32)
    labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    # a trained-like band: blank dominant until the label is due at t = 3 u, the label afterwards (bonus 30 nats): cells far
    # off the band have occupancies of e^-60 and less
    for b in range(B):
        for u in range(U):
            te = 3 * u if u < U - 1 else T
            acts[b, :te, u, 0] += 30.0
            if u < U - 1:
                acts[b, te:, u, labels[b, u]] += 30.0
    il = np.full(B, T, np.int32)
    ll = np.full(B, U - 1, np.int32)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(x, device=dev)
    c0, g0 = pkg.rnnt_loss_and_grad(t(acts), t(labels), t(il), t(ll))
    c1, g1 = pkg.rnnt_loss_and_grad(t(acts), t(labels), t(il), t(ll), visit_all=True)
    torch.cuda.synchronize()
    assert torch.equal(c0, c1)
    assert float((g0 - g1).abs().max()) <= 2.0 ** -44
    dead = (g0.abs().amax(dim=-1) == 0)  # cells the default call did not visit
    assert bool(dead.any()) and not bool((g1.abs().amax(dim=-1) == 0)[dead].all())  # ... hold ~1e-30's when visited
    # a NaN logit in one of them
    b, tt, uu = [int(v[0]) for v in torch.nonzero(dead, as_tuple=True)]
    bad = acts.copy()
    bad[b, tt, uu, 5] = np.nan
    c2, g2 = pkg.rnnt_loss_and_grad(t(bad), t(labels), t(il), t(ll))
    c3, g3 = pkg.rnnt_loss_and_grad(t(bad), t(labels), t(il), t(ll), visit_all=True)
    torch.cuda.synchronize()
    # the floor hides nothing: the forward pass reads every cell, a NaN logit makes the cell's edge weights -- and with them the
    # utterance's lattice, cost and occupancies -- NaN, and a NaN occupancy counts as occupied
    assert bool(torch.isnan(c2[b])) and bool(torch.isnan(c3[b]))
    assert bool(torch.isnan(g2[b, tt, uu]).any()) and bool(torch.isnan(g3[b, tt, uu]).any())
    other = 1 - b
    assert bool(torch.isfinite(c2[other])) and torch.equal(g2[other], g0[other])  # ... of that utterance only
