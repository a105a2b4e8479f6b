"""The HIP engine in a TWO-rank process group (round 6).  No multi-GPU node was available in any round, and RCCL refuses two ranks on one
device -- so both ranks share cuda:0 and the group's transport is gloo (torch.distributed reduces CUDA tensors through it).  What this
adds to tests/test_parallel.py (oracle engine, CPU): parallel.dp_loss_step with the PRODUCT engine (rnnt_joint_loss through the C ABI)
in every rank -- utterance shards of different sizes, the 1 / GLOBAL_batch factor folded into the kernels' cost_scale, one flat-bucket
SUM all-reduce of real dW1, db1, dW2, db2 -- must reproduce the one-process full-batch step (run_rnnt.py:87-88, 278, 288, 293-294)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GB, T, U, H, J, V = 5, 24, 9, 12, 64, 28


def _problem():
    g = torch.Generator().manual_seed(11)
    enc = torch.randn(GB, T, H, generator=g)
    pred = torch.randn(GB, U, H, generator=g)
    labels = torch.randint(1, V, (GB, U - 1), generator=g, dtype=torch.int32)
    il = torch.tensor([24, 17, 24, 9, 20], dtype=torch.int32)
    ll = torch.tensor([8, 5, 0, 8, 3], dtype=torch.int32)
    params = [0.4 * torch.randn(*s, generator=g) for s in ((H, J), (J,), (J, V), (V,))]
    return enc, pred, labels, il, ll, params


def _step(rank, world):
    import rnnt_speech_recognition_amd as pkg
    from rnnt_speech_recognition_amd import parallel

    dev = torch.device("cuda:0")
    enc, pred, labels, il, ll, params = _problem()
    params = [p.to(dev).requires_grad_(True) for p in params]
    shard = [x.to(dev) for x in parallel.shard_batch([enc, pred, labels, il, ll], world, rank)]
    costs_fn = lambda: pkg.rnnt_joint_loss(shard[0], shard[1], *params, shard[2], shard[3], shard[4], joint_dtype="f32")  # noqa: E731
    logged = parallel.dp_loss_step(costs_fn, params, GB)
    torch.cuda.synchronize()
    return float(logged), [p.grad.detach().cpu().numpy() for p in params]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = _step(rank, world)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_of_the_hip_engine_reproduce_the_full_batch_step():
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    import rnnt_speech_recognition_amd as pkg

    pkg.build()  # (the workers find the library built)
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ref_logged, ref_grads = _step(0, 1)  # one process, the whole batch, no group
    assert np.isfinite(ref_logged)
    for r in range(world):
        logged, grads = out[r]
        assert abs(logged - ref_logged) <= 1e-5 * max(1.0, abs(ref_logged))
        for g, gr in zip(grads, ref_grads):  # (f32 sums in a different order: shards of 3 + 2 utterances against 5)
            assert np.abs(g - gr).max() <= 1e-5 * max(1.0, np.abs(gr).max())
    # both ranks hold the same reduced gradients, bit for bit
    for g0, g1 in zip(out[0][1], out[1][1]):
        assert np.array_equal(g0, g1)
