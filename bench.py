#!/usr/bin/env python
"""bench.py -- transducer loss+grad throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch resident in HBM: ONE call of
compute_rnnt_loss_ex through the C ABI of libwarprnnt.so (log-softmax denominators + alpha/beta
sweeps -> costs, then the gradient w.r.t. the logits scaled by 1/global_batch, run_rnnt.py:278),
all buffers pre-allocated.  The logits of consecutive steps come from DIFFERENT buffers (a ring of
`--rotate` tensors, ~1 GB in all), as in training where every step's logits are freshly produced: the
256 MiB Infinity Cache cannot carry one step's logits into the next.  `warm_ms_per_step` (same buffer
every step) is printed beside it.
Workload at N=1: BASELINE.json configs[1]  B=32 T=600 U=150 V=28, acts ~ N(0,1), full lengths.
N>1 (one rank per GPU, RCCL): utterances shard across ranks (weak scaling: every rank owns a full B=32
batch, cost_scale = 1/global_batch) and every timed step ends with the path's one exchange step, ONE
RCCL SUM all-reduce of the flat parameter-gradient bucket of the reference's joint network
((H*J + J + J*V + V) fp32 = 1.7 MB at hparams.py:18,23; run_rnnt.py:288).  `fused_dp_step` times the
complete data-parallel step of the fused engine (joint + loss + gradients + that all-reduce on the real
dW1, db1, dW2, db2) on the same ranks.

`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N
ranks; under a launcher (WORLD_SIZE set) --gpus must equal WORLD_SIZE.

Prints ONE JSON line (rank 0).  Extra objects: `roofline`, `cpu_baseline` (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (~2.5 PFLOP/s)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--shape", type=str, default="32,600,150,28", help="B,T,U,V per GPU")
    ap.add_argument("--rotate", type=int, default=3, help="number of logits buffers the timed steps cycle through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the fused joint+loss measurements")
    ap.add_argument("--no-ragged", action="store_true", help="skip the ragged-batch leg (profiling runs: full-length launches only)")
    ap.add_argument("--joint-size", type=int, default=640, help="H = J of the fused joint (hparams.py:18,23)")
    ap.add_argument("--fused-only", type=str, default="",
                    help="B,T,U,V: time only the fused joint+loss on this shape (e.g. 16,1500,300,1024 = BASELINE "
                         "config 5) and print its JSON object")
    ap.add_argument("--fused-leg", choices=["all", "n01", "n01_all_rows", "trained", "trained_all_rows"], default="all",
                    help="with --fused-only: time just one of the four legs (profiling runs: the kernel trace then holds one kind of launch)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip BASELINE configs[2] (end-to-end train step, 2x320 LSTM encoder / 1x320 decoder, B=64 per GPU)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the two BASELINE configs[4] legs (f16 fused joint and the op on materialised logits; N=1 only)")
    ap.add_argument("--cpu-reps", type=int, default=10)
    ap.add_argument("--engine", choices=["hip", "stub"], default="hip",
                    help="stub = CPU/gloo stand-in with no kernels: exercises the launch / sharding / collective / reporting "
                         "path only (tests/test_bench_launch.py); it reports value null")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launch: one process per GPU
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ensure_world(a):
    """Returns (world, rank, local_rank).  Re-executes under torch.distributed.run when --gpus N > 1 was asked for
    without a launcher; refuses a launcher whose world size contradicts --gpus."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if a.gpus > 1:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            sys.stdout.flush()
            os.execv(sys.executable, cmd)
        return 1, 0, 0
    world = int(env_world)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} contradicts the launcher's WORLD_SIZE={world}; pass --gpus {world}")
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle/cpu_rnnt.c = the C restatement of the reference's CPU path; kind = "port")
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(B, T, U, V, reps):
    """loss + gradient w.r.t. the logits on the host cores, the way the reference runs on a non-CUDA build
    (utils/loss.py:29-30): log_softmax (PyTorch CPU) -> warp-transducer CPU lattice (C restatement, OpenMP over utterances
    only, input log-probs, gradient w.r.t. log-probs) -> log_softmax backward.  Every buffer is allocated and touched
    before the timed region; 3 warm-ups, median of `reps` runs; the three stages are timed separately."""
    import ctypes

    from oracle import cpu_oracle

    cpu_oracle.build()
    lib = cpu_oracle._load()
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)
    acts = torch.randn(B, T, U, V, generator=g, dtype=torch.float32)
    labels = np.ascontiguousarray(torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).numpy())
    il = np.full(B, T, np.int32)
    ll = np.full(B, U - 1, np.int32)
    lp = torch.empty_like(acts)
    glp = torch.zeros_like(acts)
    grad = torch.empty_like(acts)
    rowsum = torch.empty(B, T, U, 1)
    costs = np.zeros(B, np.float32)

    def one(nthreads, nb):
        """one loss+grad step on the first `nb` utterances; returns (softmax fwd, lattice, softmax bwd) seconds"""
        x, l, gl, gr, rs = acts[:nb], lp[:nb], glp[:nb], grad[:nb], rowsum[:nb]
        t0 = time.perf_counter()
        torch._log_softmax(x, -1, False, out=l)
        t1 = time.perf_counter()
        rc = lib.oracle_rnnt_cpu(l.data_ptr(), gl.data_ptr(), labels.ctypes.data, ll.ctypes.data, il.ctypes.data, V, nb, T, U,
                                 0, nthreads, costs.ctypes.data)
        assert rc == 0
        t2 = time.perf_counter()
        # log_softmax backward, in place: d logits = g - softmax * sum_v g, scaled by 1/B (run_rnnt.py:278)
        torch.sum(gl, dim=-1, keepdim=True, out=rs)
        torch.exp(l, out=gr)
        gr.mul_(rs).neg_().add_(gl).mul_(1.0 / B)
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    def measure(nthreads, nb, n):
        torch.set_num_threads(nthreads)
        for _ in range(3):
            one(nthreads, nb)
        runs = np.array([one(nthreads, nb) for _ in range(n)])
        tot = runs.sum(1)
        k = int(np.argsort(tot)[len(tot) // 2])
        return float(tot[k]), [float(v) for v in runs[k]]

    use = min(ncpu, B)
    t_all, split_all = measure(use, B, reps)
    nb1 = min(4, B)
    t_one, split_one = measure(1, nb1, 3)
    torch.set_num_threads(ncpu)
    return {
        "value": B * T * U / t_all, "unit": "cells/s", "cores": use, "host_cores": ncpu, "kind": "port",
        "seconds_per_step": t_all,
        "stage_seconds": {"log_softmax_fwd": split_all[0], "lattice_alpha_beta_grad": split_all[1],
                          "log_softmax_bwd": split_all[2]},
        "single_thread": {"value": nb1 * T * U / t_one, "unit": "cells/s", "cores": 1,
                          "sample": f"{nb1} utterances of the same batch, median of 3 after 3 warm-ups",
                          "stage_seconds": {"log_softmax_fwd": split_one[0], "lattice_alpha_beta_grad": split_one[1],
                                            "log_softmax_bwd": split_one[2]}},
        "thread_speedup": (B * T * U / t_all) / (nb1 * T * U / t_one),
        "sample": f"full headline batch B={B} T={T} U={U} V={V}, all buffers pre-allocated and touched, 3 warm-ups, "
                  f"median of {reps} runs; lattice = OpenMP over utterances only (like the reference: at most B={B} threads "
                  f"can work), log_softmax fwd/bwd = PyTorch CPU ops with {use} threads",
    }


# ----------------------------------------------------------------------------------------------------------------
# fused joint + loss (SURVEY.md 8d "P2")
# ----------------------------------------------------------------------------------------------------------------
def bench_fused_joint(lib, _lib, dev, B, T, U, V, J, stream, reps, only="all"):
    """compute_rnnt_joint_loss (costs + d_enc_proj, d_pred_proj, dW2, db2) from enc_proj / pred_proj.  Four timings where the
    backward skips work (the f32-grade joint: lattice rows x 32-column tiles without mass, include/rnnt.h RNNT_VISIT_ALL):
        ms_per_step           N(0,1) projections, glorot W2 -- the pruned default, as a caller gets it
        all_rows              the same input with RNNT_VISIT_ALL: every row visited, what the reference's autodiff does (run_rnnt.py:284)
        trained_like          posteriors of a trained model (one dominant symbol per cell along a monotone alignment:
                              pkg.synthetic_trained_like_joint), pruned and with every row visited
    `roofline.frac` = EXECUTED matrix-core flops / time / peak in every one of them; the 8*J*V convention of SURVEY.md 8(d)
    (which counts a backward recompute that is not executed, and every row) is reported as `convention_*`."""
    import math

    import rnnt_speech_recognition_amd as pkg

    f16 = V > 64  # large vocabularies run the J x V products on the f16 MFMA units (joint_dtype = 1); up to 64 symbols: f32-grade
    cells = B * T * U
    try:
        ws = torch.empty(_lib.joint_workspace_bytes(T, U, B, J, V), dtype=torch.uint8, device=dev)
    except RuntimeError as e:
        return {"error": str(e)}
    opts = _lib.make_options(stream.cuda_stream, 0, T, U)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
    scale = torch.full((B,), 1.0 / B, device=dev)
    costs = torch.empty(B, device=dev)

    def inputs(kind):
        if kind == "trained_like":
            ep, pp, W2, b2, labels = pkg.synthetic_trained_like_joint(B, T, U, V, J, seed=7)
        else:
            g = torch.Generator(device="cpu").manual_seed(4321)
            ep = torch.randn(B, T, J, generator=g)
            pp = torch.randn(B, U, J, generator=g)
            lim = math.sqrt(6.0 / (J + V))
            W2 = (torch.rand(J, V, generator=g) * 2 - 1) * lim
            b2 = torch.zeros(V)
            labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32)
        return [x.to(dev) for x in (ep, pp, W2, b2, labels)]

    def run(kind, visit_all):
        ep, pp, W2, b2, labels = inputs(kind)
        d_ep, d_pp, dW2, db2 = (torch.empty_like(x) for x in (ep, pp, W2, b2))
        word = (1 if f16 else 0) | (_lib.RNNT_VISIT_ALL if visit_all else 0)

        def step():
            _lib.check(lib.compute_rnnt_joint_loss(ep.data_ptr(), pp.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                                   labels.data_ptr(), ll.data_ptr(), il.data_ptr(), scale.data_ptr(),
                                                   J, V, B, costs.data_ptr(), d_ep.data_ptr(), d_pp.data_ptr(),
                                                   dW2.data_ptr(), db2.data_ptr(), word, ws.data_ptr(), opts), "joint")

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        rows = (ctypes.c_int * 2)(-1, -1)
        _lib.check(lib.get_rnnt_joint_backward_rows(ws.data_ptr(), J, V, B, opts, rows), "backward rows")
        visited = rows[0] / rows[1] if rows[1] > 0 else 1.0
        return dt, visited, bool(torch.isfinite(costs).all()), float(costs.mean())

    conv = 8.0 * J * V * cells  # SURVEY.md 8(d) convention (includes a backward recompute of the logits GEMM)
    # forward product (2 J V per cell) on every row, dh + dW2 (4 J V) on the visited rows.  f16: one MFMA per product; f32-grade:
    # split-precision products, three f16 MFMAs each -> ceiling = dense f16 peak / 3, issued on V padded to vocabulary tiles of 32
    peak, vp = (MFMA_F16_PEAK_TFLOPS, V) if f16 else (MFMA_F16_PEAK_TFLOPS / 3.0, 32 * ((V + 31) // 32))
    executed_of = lambda visited: (2.0 + 4.0 * visited) * J * vp * cells  # noqa: E731

    def leg(kind, visit_all):
        dt, visited, finite, mean_cost = run(kind, visit_all)
        ex = executed_of(visited)
        return {"ms_per_step": dt * 1e3, "cells_per_s": cells / dt, "backward_rows_visited": visited, "costs_finite": finite,
                "mean_cost_nats": mean_cost, "executed_tflops": ex / dt / 1e12, "frac": ex / dt / 1e12 / peak}

    if only != "all":  # one leg (profiling)
        kind, va = {"n01": ("n01", False), "n01_all_rows": ("n01", True), "trained": ("trained_like", False),
                    "trained_all_rows": ("trained_like", True)}[only]
        one = leg(kind, va)
        one["leg"] = only
        return one
    main = leg("n01", False)
    out = {"workload": f"joint+loss+grads from enc_proj/pred_proj, B={B} T={T} U={U} V={V} J={J}, "
                       + ("f16 MFMA joint / f32 lattice" if f16 else "f32-grade products on split-precision f16 MFMAs"),
           "dtype": ("f16 products (binary16 operands, f32 accumulation), f32 lattice" if f16 else
                     "f16x3-split (binary16 hi+lo operands, three f16 MFMAs per product, f32 accumulation), f32 lattice"),
           "ms_per_step": main["ms_per_step"], "cells_per_s": main["cells_per_s"],
           "roofline": {"bound": "mfma", "achieved": main["executed_tflops"], "peak": peak, "unit": "TFLOP/s", "frac": main["frac"],
                        "backward_rows_visited": main["backward_rows_visited"],
                        "convention_flops_per_step": conv, "convention_tflops": conv / (main["ms_per_step"] * 1e-3) / 1e12,
                        "convention_frac": conv / (main["ms_per_step"] * 1e-3) / 1e12 / peak,
                        "note": "achieved / frac = matrix-core flops actually ISSUED (the forward product on every row + dh and dW2 on the visited "
                                "fraction of the lattice rows; f32-grade: V padded to 32-symbol tiles and peak = dense f16 peak / 3 because a "
                                "product is three f16 MFMAs) over time; convention_* = SURVEY.md 8(d)'s "
                                "8*J*V per cell, which counts a backward recompute nobody executes and every row whether visited or not"},
           "workspace_GB": ws.numel() / 1e9}
    if not f16:
        out["roofline"]["f32_mfma_peak_for_reference"] = MFMA_F32_PEAK_TFLOPS
        out["roofline"]["issue_bound"] = fused_issue_bound(B, T, U, V, J, main["ms_per_step"] * 1e-3)
    out["all_rows"] = leg("n01", True)
    out["all_rows"]["note"] = "RNNT_VISIT_ALL (include/rnnt.h): no occupancy floor -- the backward visits every lattice row, as the reference's autodiff does"
    out["trained_like"] = {"pruned": leg("trained_like", False), "all_rows": leg("trained_like", True),
                           "input": "pkg.synthetic_trained_like_joint(seed=7): one dominant symbol per cell along a monotone alignment "
                                    "(bonus 10 nats), every second utterance emitting its labels in the last 40 % of the frames"}
    return out


def bench_fused_full(dev, B, T, U, V, J, reps):
    """The complete fused step as a training graph sees it (model.py:158-166 + run_rnnt.py:269-288): enc [B,T,H], pred [B,U,H]
    -> compute_rnnt_joint_net_loss_fwd / _bwd through autograd: the first Dense layer (split-precision MFMA GEMMs,
    csrc/dense_kernels.hip), tanh, the J x V products, the lattice, and every gradient (dW1, db1, dW2, db2, d enc, d pred)
    behind the C ABI.  Everything the timed region of `fused_joint` leaves out (it starts from enc_proj / pred_proj) is inside
    this one.  `torch_w1_gemms_fwd_bwd_ms` times the same first layer + backward as torch.matmul / autograd (hipBLASLt f32),
    for reference.  H = J (hparams.py:18,23)."""
    import rnnt_speech_recognition_amd as pkg

    torch.manual_seed(99)
    joint = pkg.JointLoss(J, J, V).to(dev)
    g = torch.Generator(device="cpu").manual_seed(4321)
    enc = torch.randn(B, T, J, generator=g).to(dev).requires_grad_(True)
    pred = torch.randn(B, U, J, generator=g).to(dev).requires_grad_(True)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
    leaves = list(joint.parameters()) + [enc, pred]

    def step():
        for x in leaves:
            x.grad = None
        costs = joint(enc, pred, labels, il, ll)
        (costs.sum() * (1.0 / B)).backward()  # run_rnnt.py:278

    def w1_only():  # the part of the step that is NOT behind the C ABI: the two projections and their backward
        for x in leaves:
            x.grad = None
        ep = torch.matmul(enc, joint.W1) + joint.b1
        pp = torch.matmul(pred, joint.W1)
        torch.autograd.backward([ep, pp], [g_ep, g_pp])

    def timeit(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    from rnnt_speech_recognition_amd import joint as joint_mod

    g_ep = torch.randn(B, T, J, device=dev)
    g_pp = torch.randn(B, U, J, device=dev)
    dt = timeit(step, reps)
    joint_mod.TRACK_BACKWARD_ROWS = True  # one more step, outside the timed loop, that asks the library how many rows it visited
    step()
    joint_mod.TRACK_BACKWARD_ROWS = False
    rows = joint_mod.last_backward_rows()
    visited = rows[0] / rows[1] if rows and rows[1] > 0 else 1.0
    joint.visit_all = True  # RNNT_VISIT_ALL: the same step with every lattice row visited (what the reference's autodiff does)
    dt_all = timeit(step, reps)
    joint.visit_all = False
    dt_w1 = timeit(w1_only, reps)
    cells = B * T * U
    vp = 32 * ((V + 31) // 32)
    w1_flops = 6.0 * B * (T + U) * J * J
    conv = 8.0 * J * V * cells + w1_flops  # joint products + the factored first layer
    executed = lambda vis: (2.0 + 4.0 * vis) * J * vp * cells + w1_flops  # noqa: E731
    split_peak = MFMA_F16_PEAK_TFLOPS / 3.0
    return {"workload": f"fused step from enc/pred: W1 GEMMs + joint + loss + gradients (dW1, db1, dW2, db2, d enc, d pred), "
                        f"B={B} T={T} U={U} V={V} H=J={J}",
            "dtype": "f16x3-split products (binary16 hi+lo operands, f32 accumulation) in both Dense layers, f32 lattice",
            "ms_per_step": dt * 1e3, "cells_per_s": cells / dt,
            "all_rows": {"ms_per_step": dt_all * 1e3, "cells_per_s": cells / dt_all, "executed_tflops": executed(1.0) / dt_all / 1e12,
                         "frac": executed(1.0) / dt_all / 1e12 / split_peak,
                         "note": "RNNT_VISIT_ALL: every lattice row visited (JointLoss(visit_all=True))"},
            "torch_w1_gemms_fwd_bwd_ms": dt_w1 * 1e3,
            "roofline": {"bound": "mfma", "achieved": executed(visited) / dt / 1e12, "peak": split_peak, "unit": "TFLOP/s",
                         "frac": executed(visited) / dt / 1e12 / split_peak, "backward_rows_visited": visited,
                         "convention_flops_per_step": conv, "convention_frac": conv / dt / 1e12 / split_peak,
                         "note": "achieved / frac = matrix-core flops ISSUED (W1 GEMMs 6*B*(T+U)*H*J + the joint's forward on every row + "
                                 "dh / dW2 on the visited rows, V padded to 32) over time; peak = dense f16 MFMA peak / 3 (three f16 MFMAs per "
                                 "f32-grade product); convention_* = 8*J*V per cell + the W1 term"}}


def bench_fused_dp_step(dev, world, rank, B, T, U, V, J, reps, sync):
    """The complete data-parallel step of the fused engine on every rank: enc/pred [B,T|U,H] -> W1 (hipBLASLt) -> fused
    joint + loss + gradients (libwarprnnt.so) -> autograd to dW1, db1 -> ONE RCCL SUM all-reduce of the flat
    (dW1, db1, dW2, db2) bucket -> logged-loss scalar all-reduce (parallel.dp_loss_step; run_rnnt.py:278,288,293-294)."""
    import rnnt_speech_recognition_amd as pkg
    from rnnt_speech_recognition_amd import parallel

    torch.manual_seed(99)  # the same joint weights on every rank
    joint = pkg.JointLoss(J, J, V).to(dev)
    g = torch.Generator(device="cpu").manual_seed(777 + rank)
    enc = torch.randn(B, T, J, generator=g).to(dev)
    pred = torch.randn(B, U, J, generator=g).to(dev)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
    params = list(joint.parameters())
    bucket_bytes = sum(p.numel() for p in params) * 4

    def step():
        return parallel.dp_loss_step(lambda: joint(enc, pred, labels, il, ll), params, B * world)

    for _ in range(2):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        logged = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dt /= reps
    return {"workload": f"fused joint + loss + gradients, B={B} T={T} U={U} V={V} H=J={J} per GPU, then one RCCL SUM "
                        "all-reduce of the flat (dW1, db1, dW2, db2) bucket",
            "ms_per_step": dt * 1e3, "cells_per_s": world * B * T * U / dt, "rccl_ranks": world,
            "all_reduce_bytes": bucket_bytes, "logged_loss": float(logged)}


def bench_e2e(dev, world, rank, steps=8):
    """BASELINE.json configs[2]/[3]: synthetic log-mel [B, 600, 240] -> BatchNorm -> 2 x LSTM(320, proj 320) with
    x2 time reduction after layer 0 -> 1 x LSTM(320) prediction net -> fused joint (J=320, V=28) -> SGD step;
    B=64 per GPU, global batch 64*world, gradients summed with one RCCL all-reduce."""
    import rnnt_speech_recognition_amd as pkg

    hp = pkg.HParams(vocab_size=28, embedding_size=320, encoder_layers=2, encoder_size=320, projection_size=320,
                     time_reduction_index=0, pred_net_layers=1, pred_net_size=320, joint_net_size=320)
    torch.manual_seed(1234)
    model = pkg.Transducer(hp).to(dev)
    batch = pkg.synthetic_batch(hp, batch=64, frames=600, max_labels=100, device=dev, seed=1234 + rank)
    step = pkg.TrainStep(model, global_batch=64 * world)
    for _ in range(2):
        step(*batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        log = step(*batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": "configs[2]: B=64/GPU, 600 frames x 240 feats, enc 2x320 (x2 time reduction), pred 1x320, "
                        "J=320, V=28, SGD(1e-4, 0.9); synthetic features",
            "ms_per_step": dt * 1e3, "utterances_per_s": 64 * world / dt, "loss": log["loss"]}


def bench_op_shape(lib, _lib, dev, B, T, U, V, stream, reps):
    """P1 (the warp-transducer op contract: loss + gradient on MATERIALISED f32 logits) at another shape -- BASELINE
    configs[4] B=16 T=1500 U=300 V=1024: 29.5 GB of logits in, 29.5 GB of gradients out, generated on the device."""
    gd = torch.Generator(device=dev).manual_seed(4321)
    acts = torch.randn(B, T, U, V, generator=gd, dtype=torch.float32, device=dev)
    grads = torch.empty_like(acts)
    g = torch.Generator(device="cpu").manual_seed(4321)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
    scale = torch.full((B,), 1.0 / B, dtype=torch.float32, device=dev)
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
    opts = _lib.make_options(stream.cuda_stream, 0, T, U)

    def step(flags=0):
        _lib.check(lib.compute_rnnt_loss_flags(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(),
                                               il.data_ptr(), scale.data_ptr(), V, B, costs.data_ptr(), ws.data_ptr(), opts, flags),
                   "compute_rnnt_loss_flags")

    def timeit(flags):
        for _ in range(2):
            step(flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step(flags)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    dt_all = timeit(_lib.RNNT_VISIT_ALL)  # every cell's logits read, every gradient row formed (the reference's op)
    dt = timeit(0)                        # the default: cells with occupancy <= 2^-50 get zeros without their logits being read
    cells = B * T * U
    alg = 8.0 * V * cells
    finite = bool(torch.isfinite(costs).all())
    picks = [0, B - 1]
    ref = f64_costs(acts, labels, picks)
    got = costs.cpu().numpy().astype(np.float64)
    dcost = max(abs(got[b] - r) / max(1.0, abs(r)) for b, r in zip(picks, ref))
    del acts, grads, ws
    torch.cuda.empty_cache()
    return {"workload": f"transducer loss+grad on given f32 logits, B={B} T={T} U={U} V={V}, full lengths, acts~N(0,1)",
            "dtype": "f32", "ms_per_step": dt * 1e3, "cells_per_s": cells / dt, "costs_finite": finite,
            "max_rel_dcost": dcost, "max_rel_dcost_note": f"utterances {picks} against a float64 evaluation of the same logits (bar 1e-4)",
            "all_cells": {"ms_per_step": dt_all * 1e3, "cells_per_s": cells / dt_all,
                          "note": "RNNT_VISIT_ALL (compute_rnnt_loss_flags): no occupancy floor, every cell's logits read twice and its gradients written"},
            "roofline": {"bound": "hbm", "achieved": alg / dt_all / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / dt_all / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "skipping_speedup_equivalent_GBs": alg / dt / 1e9,
                         "note": "whole op (lsm + sweeps + gradient pass), 8*V bytes per cell, on the ALL-CELLS timing: the default call "
                                 "(ms_per_step) does not read the logits of cells below the occupancy floor, so 8*V per cell over ITS time "
                                 "(skipping_speedup_equivalent_GBs) is not a bandwidth"}}


def joint_bucket_floats(H, J, V):
    return H * J + J + J * V + V


def csrc_sha16():
    """Fingerprint of the kernel sources (scripts/summarize_trace.py stamps the committed profiles with the same one): a
    profile is quoted only while the kernels it was taken from are the ones in the tree."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "rnnt-speech-recognition_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_profile(name):
    """profiles/<name> if it was taken from the kernel sources of this tree, else None (a stale profile is not quoted)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        return d if d.get("csrc_sha16") == csrc_sha16() else None
    except Exception:
        return None


def rocprof_grad_ms():
    """Average duration of the full-length gradient-pass launches from the committed rocprofv3 kernel trace of this same
    command (profiles/kernel_stats_latest.json, written by scripts/summarize_trace.py), if it matches the tree."""
    d = committed_profile("kernel_stats_latest.json")
    try:
        return d["headline_kernels"]["grad_pass"]["avg_ms"] if d else None
    except Exception:
        return None


def sweep_ns_per_diagonal(n_diag):
    """The alpha / beta sweeps are a lone wave's dependent chain of T + U - 1 steps per utterance: kernel time / steps, from the
    committed rocprofv3 trace (None when the kernels changed since)."""
    d = committed_profile("kernel_stats_latest.json")
    try:
        return d["headline_kernels"]["sweeps"]["avg_ms"] * 1e6 / n_diag if d else None
    except Exception:
        return None


def fused_issue_bound(B, T, U, V, J, dt):
    """What actually limits the f32-grade fused joint: the issue slots of the SIMDs.  Its two big kernels spend ~18 VALU
    instructions per MFMA (tanh from the e^{2x} tables, binary16 hi / lo splits, DPP broadcasts), and on one SIMD VALU and MFMA
    work of the resident waves add up instead of overlapping.  From the committed instruction counters of this
    shape (profiles/fused_valu_latest.json: rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_MFMA / SQ_VALU_MFMA_BUSY_CYCLES; None when
    the kernels changed since or the shape differs): time at 1024 SIMDs x 2.4 GHz if every VALU instruction cost the 2 clocks of
    a SIMD-32 (the hardware's issue floor) and if it cost the 3.8 clocks measured for independent v_fma_f32 of resident waves
    (scripts/probes/probe_issue.hip), plus the matrix pipe's busy time."""
    d = committed_profile("fused_valu_latest.json")
    if not d or (B, T, U, V, J) != (32, 600, 150, 28, 640):
        return None
    try:
        k = d["kernels"]
        fwd = next(v for n, v in k.items() if "joint_fwd_kernel" in n)
        bwd = next(v for n, v in k.items() if "joint_bwd_kernel" in n)
        simd_hz = 1024 * 2.4e9
        out = {}
        for name, c in (("forward", fwd), ("backward", bwd)):
            nv, nm = c["SQ_INSTS_VALU"]["avg"], c["SQ_INSTS_MFMA"]["avg"]
            mfma_ms = c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / simd_hz * 1e3
            out[name] = {"valu_instructions": nv, "mfma_instructions": nm, "valu_per_mfma": nv / nm,
                         "valu_ms_at_2_clocks": nv * 2.0 / simd_hz * 1e3, "valu_ms_at_3.8_clocks": nv * 3.8 / simd_hz * 1e3,
                         "mfma_busy_ms": mfma_ms}
        floor = sum(v["valu_ms_at_3.8_clocks"] + v["mfma_busy_ms"] for v in out.values())
        out["valu_plus_mfma_ms"] = floor
        out["frac_of_issue_bound"] = floor / (dt * 1e3)
        out["note"] = ("valu_plus_mfma_ms = VALU instructions at 3.8 clocks + matrix-pipe busy time of the two big kernels: the time "
                       "this instruction mix needs when nothing overlaps on a SIMD; frac_of_issue_bound = that / the measured step "
                       "(the rest: prep / records / reductions / sweeps and stalls).  The MFMA roofline above overstates the headroom.")
        return out
    except Exception:
        return None


def f64_costs(acts, labels, picks):
    """Checker (outside every timed region): float64 transducer costs of the full-length utterances `picks` -- log-softmax in
    float64 on the device, the forward recurrence alpha(t,u) = lse(alpha(t-1,u) + lp_blank(t-1,u), alpha(t,u-1) +
    lp_label(t,u-1)) by anti-diagonals on the host (Graves 2012 eq. 16)."""
    out = []
    for b in picks:
        T, U, V = acts.shape[1:]
        lpb = torch.empty(T, U, dtype=torch.float64)
        lpl = torch.empty(T, max(U - 1, 1), dtype=torch.float64)
        lab = labels[b].long()
        for t0 in range(0, T, 100):
            x = acts[b, t0:t0 + 100].double()
            lse = torch.logsumexp(x, dim=-1)
            lpb[t0:t0 + 100] = (x[:, :, 0] - lse).cpu()
            if U > 1:
                lpl[t0:t0 + 100] = (torch.gather(x[:, :U - 1], 2, lab[None, :, None].expand(x.shape[0], -1, 1))[:, :, 0] - lse[:, :U - 1]).cpu()
        lpb, lpl = lpb.numpy(), lpl.numpy()
        a = np.full((T, U), -np.inf)
        a[0, 0] = 0.0
        for n in range(1, T + U - 1):
            u = np.arange(max(0, n - T + 1), min(n, U - 1) + 1)
            t = n - u
            up = np.where(t >= 1, a[np.maximum(t - 1, 0), u] + lpb[np.maximum(t - 1, 0), u], -np.inf)
            lf = np.where(u >= 1, a[t, np.maximum(u - 1, 0)] + lpl[t, np.maximum(u - 1, 0)], -np.inf)
            a[t, u] = np.logaddexp(up, lf)
        out.append(-(a[T - 1, U - 1] + lpb[T - 1, U - 1]))
    return out


# ----------------------------------------------------------------------------------------------------------------
# stub engine: launch-path test only
# ----------------------------------------------------------------------------------------------------------------
def main_stub(a, world, rank):
    """No GPU, no kernels: gloo ranks run the sharding + barrier + max-over-ranks timing + collective + JSON path of this
    file with a trivial CPU step.  value is null on purpose -- nothing here is a measurement."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B, T, U, V = (int(v) for v in a.shape.split(","))
    bucket = torch.ones(joint_bucket_floats(a.joint_size, a.joint_size, V))

    def step():
        if world > 1:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
            bucket.div_(world)

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "rnnt_loss_grad_lattice_cells_per_sec", "value": None, "unit": "cells/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub engine: no kernels ran (launch-path test)",
            "engine": "stub", "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "config": {"workload": f"stub, B={B} T={T} U={U} V={V} per rank", "global_batch": B * world,
                       "parallelism": f"utterance-sharded x{world}"},
            "collective": {"op": "all_reduce SUM", "bytes": bucket.numel() * 4, "bucket_ok": bool(torch.all(bucket == 1.0))}}))
    if world > 1:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------
def main():
    a = parse()
    world, rank, local = ensure_world(a)
    if a.engine == "stub":
        return main_stub(a, world, rank)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # The contract is ONE JSON line on stdout.  Native code prints there too (RCCL's version banner when a communicator is
    # created): keep the real stdout aside for the JSON line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import rnnt_speech_recognition_amd as pkg
    from rnnt_speech_recognition_amd import _lib

    # one rank compiles (if the in-tree library is stale at all), the others wait for it
    if rank == 0:
        pkg.build()
    if world > 1:
        dist.barrier()
    lib = _lib.load()

    if a.fused_only:
        fB, fT, fU, fV = (int(v) for v in a.fused_only.split(","))
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            out = bench_fused_joint(lib, _lib, dev, fB, fT, fU, fV, a.joint_size, st, max(2, min(a.steps, 5)), only=a.fused_leg)
        if rank == 0:
            real_stdout.write(json.dumps({"fused_joint": out}) + "\n")
            real_stdout.flush()
        return

    B, T, U, V = (int(v) for v in a.shape.split(","))
    nrot = max(1, a.rotate)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    if B * T * U * V <= (1 << 28):
        acts_ring = [torch.randn(B, T, U, V, generator=g, dtype=torch.float32).to(dev) for _ in range(nrot)]
    else:  # tens of GB (config 5): generate on the device, no ring
        gd = torch.Generator(device=dev).manual_seed(1234 + rank)
        acts_ring = [torch.randn(B, T, U, V, generator=gd, dtype=torch.float32, device=dev)]
        nrot = 1
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
    scale = torch.full((B,), 1.0 / (B * world), dtype=torch.float32, device=dev)
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    grads = torch.empty_like(acts_ring[0])
    ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    opts = _lib.make_options(stream.cuda_stream, 0, T, U)
    # the flat parameter-gradient bucket a data-parallel step exchanges (reference joint: W1 [H,J], b1, W2 [J,V], b2)
    bucket = torch.zeros(joint_bucket_floats(a.joint_size, a.joint_size, V), dtype=torch.float32, device=dev)

    def fwd(x):
        _lib.check(lib.compute_rnnt_loss_fwd(x.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(),
                                             V, B, costs.data_ptr(), ws.data_ptr(), opts), "fwd")

    def bwd(x):
        _lib.check(lib.compute_rnnt_loss_bwd(x.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(),
                                             il.data_ptr(), scale.data_ptr(), V, B, ws.data_ptr(), opts), "bwd")

    def step(i):
        # ONE call = the reference op's contract (costs + grads) with the 1/global_batch factor folded in
        x = acts_ring[i % nrot]
        _lib.check(lib.compute_rnnt_loss_ex(x.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(),
                                            il.data_ptr(), scale.data_ptr(), V, B, costs.data_ptr(), ws.data_ptr(),
                                            opts), "compute_rnnt_loss_ex")
        if world > 1:  # the path's exchange step: replicas' parameter gradients are summed (run_rnnt.py:288)
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        sync()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    for i in range(a.warmup):
        step(i)
    dt = timed(step, a.steps)
    cells = B * T * U
    value = world * cells * a.steps / dt
    # the same step on ONE buffer (what round 1 reported): part of the logits survives in the Infinity Cache
    dt_warm = timed(lambda i: step(0), a.steps) if nrot > 1 else dt

    # ---- per-stage durations, HIP events on the launch stream (the kernels run on `stream`) ----
    roof = None
    if rank == 0:
        n_ev = max(10, min(a.steps, 50))
        ef = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for k, (e0, e1, e2) in enumerate(ef):
            x = acts_ring[k % nrot]
            e0.record(stream)
            fwd(x)
            e1.record(stream)
            bwd(x)
            e2.record(stream)
        torch.cuda.synchronize()
        t_f = float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in ef])) * 1e-3
        t_b = float(np.mean([e1.elapsed_time(e2) for _, e1, e2 in ef])) * 1e-3
        alg = 8.0 * V * cells  # read each f32 logit once + write each f32 gradient once
        ach_grad = alg / t_b / 1e9
        pmc = committed_profile("pmc_latest.json")  # None when the kernels changed since it was taken
        traffic = pmc.get("grad_kernel_hbm_bytes_per_launch") if pmc else None
        roof = {
            "bound": "hbm",
            "kernel": "gradient pass rnnt::cell_tile_kernel<32, true, true, true>: reads V logits + writes V grads per cell",
            "achieved": ach_grad, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_grad / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg,
            "kernel_avg_ms": t_b * 1e3,
            "kernel_avg_ms_source": "HIP events on the launch stream around compute_rnnt_loss_bwd = ONE launch of this kernel + the "
                                    "hand-back launch of the linear lattice, whose workgroups read two words and return (~4 us: the "
                                    "figure is that much on the safe side; full-length utterances, rotating logits buffers)",
            "profiles_csrc_sha16": csrc_sha16(),
            "step_traffic_over_algorithmic": (pmc.get("step_traffic_over_algorithmic") if pmc else None),
            "rocprof_avg_ms": rocprof_grad_ms(),
            "sweep_ns_per_diagonal": sweep_ns_per_diagonal(T + U - 1),
            "whole_op": {"note": "same algorithmic bytes over the whole timed step (lsm + sweeps + gradient pass + hand-back launch)",
                         "achieved": alg / (dt / a.steps) / 1e9, "frac": alg / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                         "fwd_ms": t_f * 1e3, "bwd_ms": t_b * 1e3},
        }

    # ---- ragged batch of the same padded shape (also report sum_b T_b*U_b per second) ----
    ragged = None
    if rank == 0 and not a.no_ragged:
        gr = torch.Generator(device="cpu").manual_seed(4242)
        il_r = torch.randint((T + 1) // 2, T + 1, (B,), generator=gr, dtype=torch.int32)
        ll_r = torch.randint((U - 1 + 1) // 2, U, (B,), generator=gr, dtype=torch.int32)
        il_r[0], ll_r[0] = T, U - 1
        valid = int((il_r.long() * (ll_r.long() + 1)).sum())
        il_d, ll_d = il_r.to(dev), ll_r.to(dev)

        def step_r(i):
            x = acts_ring[i % nrot]
            _lib.check(lib.compute_rnnt_loss_ex(x.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll_d.data_ptr(),
                                                il_d.data_ptr(), scale.data_ptr(), V, B, costs.data_ptr(),
                                                ws.data_ptr(), opts), "compute_rnnt_loss_ex")

        for i in range(3):
            step_r(i)
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        nr = max(5, min(a.steps, 20))
        for i in range(nr):
            step_r(i)
        torch.cuda.synchronize()
        dtr = (time.perf_counter() - t0r) / nr
        ragged = {"ms_per_step": dtr * 1e3, "valid_cells_per_s": valid / dtr, "padded_cells_per_s": B * T * U / dtr,
                  "valid_fraction": valid / float(B * T * U),
                  "lengths": "T_b ~ U{T/2..T}, L_b ~ U{(U-1)/2..U-1}, one full-length utterance (SURVEY.md 8d)"}

    # ---- the same op on TRAINED-LIKE logits (one dominant symbol per cell along a monotone alignment, bonus 10 nats -- the posteriors
    # tests/test_peaky_gpu.py checks against the oracle): the headline's N(0,1) logits are the one distribution where neither a narrow
    # alignment band nor the hand-back to the log-domain kernels can show.  Same shape, same call; reported beside the headline.
    op_trained = None
    if rank == 0 and not a.no_ragged and B * T * U * V <= (1 << 28):
        gt = torch.Generator(device=dev).manual_seed(77)
        xt = torch.randn(B, T, U, V, generator=gt, dtype=torch.float32, device=dev)
        emit = torch.empty(B, U - 1, device=dev)
        for b in range(B):
            lo = int(0.6 * T) if b % 2 else 0  # odd utterances: every label is emitted late
            emit[b] = torch.sort(torch.randint(lo, T, (U - 1,), generator=gt, device=dev)).values.float()
        tt = torch.arange(T, device=dev, dtype=torch.float32)[None, :, None]
        due = torch.cat([emit, torch.full((B, 1), float(T), device=dev)], 1)[:, None, :]  # [B,1,U]: frame at which column u's label is due
        before = tt < due                                                                    # [B,T,U]
        xt[..., 0] += 10.0 * before
        lab_idx = torch.cat([labels.long(), torch.zeros(B, 1, dtype=torch.long, device=dev)], 1)[:, None, :, None].expand(B, T, U, 1)
        bonus = (10.0 * (~before)).unsqueeze(-1)
        bonus[:, :, U - 1] = 0.0  # the last column has no label
        xt.scatter_add_(3, lab_idx, bonus)
        del bonus, lab_idx, before

        def step_t(i):
            _lib.check(lib.compute_rnnt_loss_ex(xt.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(),
                                                il.data_ptr(), scale.data_ptr(), V, B, costs.data_ptr(), ws.data_ptr(),
                                                opts), "compute_rnnt_loss_ex")

        for i in range(3):
            step_t(i)
        torch.cuda.synchronize()
        t0t = time.perf_counter()
        nt_ = max(5, min(a.steps, 20))
        for i in range(nt_):
            step_t(i)
        torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0t) / nt_
        ct = costs.float().cpu()
        op_trained = {"ms_per_step": dtt * 1e3, "cells_per_s": B * T * U / dtt, "costs_finite": bool(torch.isfinite(ct).all()),
                      "mean_cost_nats": float(ct.mean()),
                      "input": "N(0,1) logits + 10 nats on blank before a cell's label is due and on the label afterwards; odd "
                               "utterances emit every label in the last 40 % of the frames (tests/test_peaky_gpu.py 'trained'); ONE buffer"}
        del xt

    # ---- fused joint + loss (SURVEY.md 8d "P2"): reported beside the headline, not as `value` ----
    fused = fused_full = fused_c5 = fused_mid = fused_ref = fused_v64 = op_c5 = fused_dp = None
    headline_shape = (B, T, U, V) == (32, 600, 150, 28)
    if not a.no_fused:
        nf = max(3, min(a.steps, 10))
        if rank == 0:
            fused = bench_fused_joint(lib, _lib, dev, B, T, U, V, a.joint_size, stream, nf)
        if rank == 0 and V <= 32:
            try:
                fused_full = bench_fused_full(dev, B, T, U, V, a.joint_size, nf)
            except Exception as e:  # a leg must not take the headline line down with it
                fused_full = {"error": repr(e)}
        if V <= 32:
            # the complete data-parallel step of the fused engine through parallel.dp_loss_step and RCCL.  At N = 1 this is a
            # ONE-rank RCCL group (the collective degenerates to a copy): the point is that the driver's single-GPU run
            # executes exactly the code the N > 1 runs execute.
            own_group = False
            try:
                if world == 1 and not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", str(_free_port()))
                    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                    own_group = True
                fused_dp = bench_fused_dp_step(dev, world, rank, B, T, U, V, a.joint_size, nf, sync)
            except Exception as e:
                fused_dp = {"error": repr(e)}
            finally:
                if own_group and dist.is_initialized():
                    dist.destroy_process_group()
        if rank == 0 and world == 1 and headline_shape and not a.no_config5:
            # BASELINE configs[4] (large-vocabulary stress): parity-test cases, timed here too because they are the path's
            # MFMA-bound corner (fused f16 joint, ~16 GB of workspace) and its largest HBM-bound one (op on 29.5 GB of logits)
            del acts_ring[1:]
            torch.cuda.empty_cache()
            fused_c5 = bench_fused_joint(lib, _lib, dev, 16, 1500, 300, 1024, 640, stream, 3)
            torch.cuda.empty_cache()
            # a mid-sized vocabulary (vocab_size and joint_net_size are free hyper-parameters, hparams.py:4,23): V = 128 word pieces
            # at the headline lattice, native since round 4 (before: padded to 512 columns)
            fused_mid = bench_fused_joint(lib, _lib, dev, B, T, U, 128, 640, stream, 5)
            torch.cuda.empty_cache()
            # 64 symbols, f32-grade (round 5: two vocabulary tiles of the split-precision joint; before: binary16 products on 128 columns)
            fused_v64 = bench_fused_joint(lib, _lib, dev, B, T, U, 64, 640, stream, 5)
            torch.cuda.empty_cache()
            # the reference's own default hyper-parameters (hparams.py:4 vocab_size 4096, :18,23 joint / hidden size 640) on a
            # realistic lattice: 600 frames after x2 time reduction, up to 99 word pieces (tests/test_baseline_sizes_gpu.py checks
            # this shape against the streamed float64 joint)
            fused_ref = bench_fused_joint(lib, _lib, dev, 16, 300, 100, 4096, 640, stream, 3)
            torch.cuda.empty_cache()
            try:
                op_c5 = bench_op_shape(lib, _lib, dev, 16, 1500, 300, 1024, stream, 3)
            except Exception as e:
                op_c5 = {"error": repr(e)}

    e2e = None
    if not a.no_e2e and headline_shape:
        try:
            e2e = bench_e2e(dev, world, rank)
        except Exception as e:
            e2e = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(B, T, U, V, a.cpu_reps)

    if rank == 0:
        out = {
            "metric": "rnnt_loss_grad_lattice_cells_per_sec", "value": value, "unit": "cells/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"transducer loss+grad on given logits (warp-transducer op contract), "
                                   f"B={B} T={T} U={U} V={V} per GPU, full lengths, acts~N(0,1), "
                                   f"logits rotate through {nrot} buffers",
                       "global_batch": B * world,
                       "parallelism": (f"utterance-sharded x{world}; each step ends with one RCCL SUM all-reduce of the "
                                       f"{bucket.numel() * 4} B joint parameter-gradient bucket") if world > 1
                                      else "single GPU (no collective)"},
            "warm_ms_per_step": dt_warm / a.steps * 1e3,
            "warm_value": world * cells * a.steps / dt_warm,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "roofline": roof, "cpu_baseline": cpu, "ragged_batch": ragged, "op_trained_like": op_trained, "fused_joint": fused,
            "fused_joint_full": fused_full, "fused_joint_config5": fused_c5, "fused_joint_v128": fused_mid, "fused_joint_v64": fused_v64, "fused_joint_refdefault": fused_ref,
            "op_config5": op_c5,
            "fused_dp_step": fused_dp, "e2e_train_step": e2e,
        }
        if world > 1:
            out["collective"] = {"op": "all_reduce SUM (RCCL)", "bytes": bucket.numel() * 4, "per_step": 1,
                                 "inside_timed_region": True,
                                 "note": "latency probe: the op-level step produces no parameter gradients, so the bucket it "
                                         "reduces has the SIZE of the reference joint's (W1, b1, W2, b2) but carries zeros; "
                                         "fused_dp_step reduces the real dW1, db1, dW2, db2"}
        if cpu:
            out["gpu_over_cpu"] = value / cpu["value"]
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
