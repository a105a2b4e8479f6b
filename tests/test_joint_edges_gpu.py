"""Shape extremes of the fused joint entry points (limits of include/rnnt.h) against the float64 oracle."""
import math

import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc

pytestmark = pytest.mark.gpu


def run(B, T, U, H, J, V, dtype, seed=0):
    dev = torch.device("cuda:0")
    t = lambda x: torch.tensor(x, device=dev)
    rng = np.random.default_rng(seed)
    enc, pred = rng.normal(size=(B, T, H)).astype(np.float32), rng.normal(size=(B, U, H)).astype(np.float32)
    W1, b1 = (rng.normal(size=(H, J)) * 0.3).astype(np.float32), (rng.normal(size=J) * 0.1).astype(np.float32)
    W2, b2 = (rng.normal(size=(J, V)) * 0.1).astype(np.float32), (rng.normal(size=V) * 0.1).astype(np.float32)
    labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    il = np.full(B, T, np.int32)
    ll = np.full(B, U - 1, np.int32)
    if B > 1:
        il[1], ll[1] = max(1, T // 2), (U - 1) // 2
    params = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
    costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll), joint_dtype=dtype)
    costs.sum().backward()
    torch.cuda.synchronize()
    f = orc.joint_loss_and_grads_f16 if dtype == "f16" else orc.joint_loss_and_grads
    ref = f(*(x.astype(np.float64) for x in (enc, pred, W1, b1, W2, b2)), labels, il, ll)
    dc = np.abs(costs.detach().cpu().numpy() - ref["costs"]).max() / max(1.0, np.abs(ref["costs"]).max())
    out = {"cost": dc}
    for name, p in zip(("d_enc", "d_pred", "dW1", "db1", "dW2", "db2"), params):
        g, r = p.grad.cpu().numpy(), ref[name]
        out[name] = float(np.abs(g - r).max() / max(1e-30, np.abs(r).max()))
    return out


CASES = [
    (1, 3, 1024, 8, 64, 8, "f32"),      # fused joint's widest lattice (maxU = 1024)
    (2, 5, 1000, 8, 128, 28, "f32"),
    (1, 2000, 3, 8, 64, 28, "f32"),     # long and narrow
    (2, 11, 40, 8, 704, 32, "f32"),     # widest f32 joint, full 32-symbol tile
    (2, 6, 5, 8, 640, 8192, "f16"),     # largest vocabulary, widest native joint
    (1, 4, 3, 8, 128, 8192, "f16"),
    (1, 3, 1024, 8, 128, 512, "f16"),
    (2, 7, 9, 8, 100, 1000, "f16"),     # padded J and V
    (1, 1, 1, 8, 64, 5, "f32"),         # degenerate lattice
    (1, 1, 1, 8, 128, 512, "f16"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_fused_joint_at_its_limits(case):
    pkg.build()
    out = run(*case)
    tol = 2e-3 if case[-1] == "f16" else 1e-4  # relative to max |reference| per tensor; f16: binary16 dlogits (test_joint_f16_gpu.py)
    assert all(v <= tol for v in out.values()), out


@pytest.mark.parametrize("route", ["single_bwd_J640", "two_kernel_J704", "f32_fallback_W2_out_of_binary16", "f16_joint", "engine_first_layer"])
def test_backward_reads_nothing_it_did_not_write(route, monkeypatch):
    """The partial buffers of the backward are not zero-filled any more (the reductions know which rows exist), and the dense
    layer's operand images, scales and abs-max entries are written on the way: with the WHOLE workspace pre-filled with 0xFF
    (NaN in every float, -1 in every int) each backward path must return finite gradients, bit-identical to the run on a
    workspace of unspecified contents."""
    import rnnt_speech_recognition_amd as pkg
    from rnnt_speech_recognition_amd import joint as jmod

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    B, T, U = 3, 41, 37
    if route == "f16_joint":
        H, J, V, gain = 64, 128, 512, 1.0
    else:
        H, J, V, gain = 64, 704 if route == "two_kernel_J704" else 640, 28, (1.0e6 if route == "f32_fallback_W2_out_of_binary16" else 1.0)
    enc, pred = torch.randn(B, T, H, generator=g), torch.randn(B, U, H, generator=g)
    W1 = (torch.rand(H, J, generator=g) * 2 - 1) * math.sqrt(6.0 / (H + J))
    b1 = 0.1 * torch.randn(J, generator=g)
    W2 = (torch.rand(J, V, generator=g) * 2 - 1) * math.sqrt(6.0 / (J + V)) * gain
    b2 = 0.1 * torch.randn(V, generator=g)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32)
    il = torch.tensor([T, T - 9, 5], dtype=torch.int32)   # ragged: rows / u-tiles / slots that no workgroup writes
    ll = torch.tensor([U - 1, 3, U - 6], dtype=torch.int32)
    first = "engine" if route == "engine_first_layer" else "torch"

    def run():
        ps = [x.clone().to(dev).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
        costs = pkg.rnnt_joint_loss(*ps, labels.to(dev), il.to(dev), ll.to(dev), first_layer=first)
        (costs.sum() / B).backward()
        torch.cuda.synchronize()
        return costs.detach().cpu(), [p.grad.cpu() for p in ps]

    c0, g0 = run()
    monkeypatch.setattr(jmod, "_WORKSPACE_FILL", 0xFF)
    c1, g1 = run()
    assert bool(torch.isfinite(c1).all()) and all(bool(torch.isfinite(x).all()) for x in g1)
    assert torch.equal(c0, c1) and all(torch.equal(a, b) for a, b in zip(g0, g1))


@pytest.mark.parametrize("J,w2_gain", [(704, 1.0), (128, 1.0), (128, 7.0e5)], ids=["two-kernel-J704", "single-kernel-J128", "W2-outside-binary16"])
def test_f32_joint_backward_can_be_repeated(J, w2_gain):
    """include/rnnt.h: compute_rnnt_joint_loss_bwd may be called more than once per _fwd.  The two-kernel backward (J = 704, or
    any J when some |W2| leaves the binary16 range) used to overwrite the parked logits with dlogits: a second backward read
    dlogits as logits.  Both calls must give the same gradients, and the right ones (float64 oracle)."""
    pkg.build()
    dev = torch.device("cuda:0")
    B, T, U, H, V = 2, 9, 6, 8, 12
    rng = np.random.default_rng(J)
    enc, pred = rng.normal(size=(B, T, H)).astype(np.float32), rng.normal(size=(B, U, H)).astype(np.float32)
    W1, b1 = (rng.normal(size=(H, J)) * 0.3).astype(np.float32), (rng.normal(size=J) * 0.1).astype(np.float32)
    W2, b2 = (rng.normal(size=(J, V)) * 0.1).astype(np.float32), (rng.normal(size=V) * 0.1).astype(np.float32)
    if w2_gain != 1.0:
        W2[0, 3] = np.float32(w2_gain)  # one weight beyond 65504: the plain-f32 MFMA kernels take the call
        W1[:, 0] = 0.0                  # (its unit's activation is tanh(0) = 0 exactly: the weight amplifies no rounding of h)
        b1[0] = 0.0
    labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    il, ll = np.array([T, T - 3], np.int32), np.array([U - 1, 2], np.int32)
    t = lambda x: torch.tensor(x, device=dev)
    params = [t(x).requires_grad_(True) for x in (enc, pred, W1, b1, W2, b2)]
    costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll), joint_dtype="f32", first_layer="torch")
    loss = costs.sum()
    loss.backward(retain_graph=True)
    g1 = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    (2.0 * loss).backward()  # another upstream gradient: the stored dlogits of the first call could not serve it
    torch.cuda.synchronize()
    ref = orc.joint_loss_and_grads(*(x.astype(np.float64) for x in (enc, pred, W1, b1, W2, b2)), labels, il, ll)
    for name, a, b_, p in zip(("d_enc", "d_pred", "dW1", "db1", "dW2", "db2"), g1, [p.grad for p in params], params):
        r = ref[name]
        s = max(1e-30, np.abs(r).max())
        assert np.abs(a.cpu().numpy() - r).max() / s <= 1e-4, name
        assert np.abs(b_.cpu().numpy() - 2.0 * r).max() / s <= 2e-4, name


def test_backward_says_how_many_lattice_rows_it_visited():
    """joint_bwd_kernel skips the rows (x 32-column tiles) none of whose cells has an occupancy above 2^-40 (round 5: 2^-50) and divides the rest among
    its workgroups by weight (include/rnnt.h get_rnnt_joint_backward_rows).  Through the C ABI: on a long lattice with unstructured
    logits a good part of the rows goes, on a tiny one nothing does; the count and every gradient are the same from call to call
    (the division of the work depends on the data, not on timing).  Parity of what is left: every other test of the fused joint."""
    import ctypes
    import rnnt_speech_recognition_amd as pkg
    from rnnt_speech_recognition_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()

    def run(B, T, U, J, V, seed):
        g = torch.Generator().manual_seed(seed)
        ep, pp = torch.randn(B, T, J, generator=g).to(dev), torch.randn(B, U, J, generator=g).to(dev)
        W2 = ((torch.rand(J, V, generator=g) * 2 - 1) * math.sqrt(6.0 / (J + V)) * 3.0).to(dev)
        b2 = torch.zeros(V, device=dev)
        labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
        il = torch.tensor([T] + [max(1, T - 7 * (i + 1)) for i in range(B - 1)], dtype=torch.int32, device=dev)
        ll = torch.tensor([U - 1] + [max(0, U - 2 - 3 * i) for i in range(B - 1)], dtype=torch.int32, device=dev)
        scale = torch.full((B,), 1.0 / B, device=dev)
        costs = torch.empty(B, device=dev)
        outs = [torch.empty_like(x) for x in (ep, pp, W2, b2)]
        ws = torch.empty(_lib.joint_workspace_bytes(T, U, B, J, V), dtype=torch.uint8, device=dev)
        opts = _lib.make_options(stream.cuda_stream, 0, T, U)
        _lib.check(lib.compute_rnnt_joint_loss(ep.data_ptr(), pp.data_ptr(), W2.data_ptr(), b2.data_ptr(), labels.data_ptr(),
                                               ll.data_ptr(), il.data_ptr(), scale.data_ptr(), J, V, B, costs.data_ptr(),
                                               *[o.data_ptr() for o in outs], 0, ws.data_ptr(), opts), "joint")
        rows = (ctypes.c_int * 2)(-7, -7)
        _lib.check(lib.get_rnnt_joint_backward_rows(ws.data_ptr(), J, V, B, opts, rows), "rows")
        torch.cuda.synchronize()
        inside = sum(int(t) * ((int(l) + 1 + 31) // 32) for t, l in zip(il.tolist(), ll.tolist()))
        return (rows[0], rows[1]), inside, [o.cpu() for o in outs], costs.cpu()

    (vis, tot), inside, g0, c0 = run(3, 400, 130, 128, 28, seed=3)
    assert tot == inside and 0 < vis < 0.8 * tot, (vis, tot, inside)
    (vis1, tot1), _, g1, c1 = run(3, 400, 130, 128, 28, seed=3)
    assert (vis1, tot1) == (vis, tot) and torch.equal(c0, c1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    assert all(bool(torch.isfinite(x).all()) for x in g0)
    (vis, tot), inside, _, _ = run(2, 6, 5, 64, 12, seed=4)
    assert vis == tot == inside, (vis, tot, inside)
