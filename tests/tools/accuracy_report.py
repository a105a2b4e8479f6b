"""Parity margins on the GPU box: HIP path (through the C ABI) vs the float64 oracle at BASELINE shapes.
Writes profiles/r01_accuracy.json.  (Test infrastructure: uses oracle/.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc

dev = torch.device("cuda:0")
out = {}


def one(name, B, T, U, V, utts, ragged=False, seed=1234):
    rng = np.random.default_rng(seed)
    acts = rng.normal(size=(B, T, U, V)).astype(np.float32)
    labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    il = np.full(B, T, np.int32); ll = np.full(B, U - 1, np.int32)
    if ragged:
        il = rng.integers(T // 2, T + 1, size=B).astype(np.int32); ll = rng.integers(U // 2, U, size=B).astype(np.int32)
    c, g = pkg.rnnt_loss_and_grad(torch.tensor(acts, device=dev), torch.tensor(labels, device=dev),
                                  torch.tensor(il, device=dev), torch.tensor(ll, device=dev))
    c, g = c.cpu().numpy().astype(np.float64), g.cpu().numpy()
    dc, dg, cs = 0.0, 0.0, 0.0
    for b in utts:
        Tb, Ub = int(il[b]), int(ll[b]) + 1
        cr, gr, _, _, _ = orc.utterance_cost_and_grad(acts[b, :Tb, :Ub], labels[b, :Ub - 1])
        dc = max(dc, abs(c[b] - cr) / abs(cr)); dg = max(dg, float(np.abs(g[b, :Tb, :Ub] - gr).max()))
        cs = max(cs, float(np.abs(g[b].sum(-1)).max()))
    out[name] = {"shape": [B, T, U, V], "utterances_checked": list(utts), "max_rel_cost_err": dc,
                 "max_abs_grad_err": dg, "max_abs_cell_grad_sum": cs}
    print(name, out[name])


one("C2_B32_T600_U150_V28", 32, 600, 150, 28, (0, 7, 13, 22, 31))
one("C2_ragged", 16, 600, 150, 28, (0, 3, 9), ragged=True, seed=77)
one("C1_B4_T50_U20_V28", 4, 50, 20, 28, (0, 1, 2, 3))
one("C5_slice_B2_T300_U300_V1024", 2, 300, 300, 1024, (0, 1), seed=5)
one("long_T1500_U300_V28", 2, 1500, 300, 28, (0,), seed=9)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "accuracy.json"), "w"), indent=1)
