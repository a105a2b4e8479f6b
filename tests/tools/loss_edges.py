"""Dev tool (uses oracle/): shape extremes of the loss op (large V, large B, degenerate lattices) against the float64 oracle."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc
dev = torch.device("cuda:0")
t = lambda x: torch.tensor(x, device=dev)
bad = 0
for (B, T, U, V) in [(1, 6, 5, 4096), (2, 4, 3, 10000), (700, 5, 4, 28), (300, 9, 7, 31), (1, 1, 1, 1), (3, 1, 1, 7), (2, 2, 1, 28), (1, 5000, 2, 4), (64, 33, 17, 60), (5, 40, 30, 61)]:
    rng = np.random.default_rng(B + T + U + V)
    acts = rng.normal(size=(B, T, U, V)).astype(np.float32)
    labels = rng.integers(1, max(V, 2), size=(B, max(U - 1, 1))).astype(np.int32)[:, : max(U - 1, 0)] if V > 1 else np.zeros((B, max(U - 1, 0)), np.int32)
    il = rng.integers(max(1, T // 2), T + 1, size=B).astype(np.int32); il[0] = T
    ll = rng.integers((U - 1) // 2, U, size=B).astype(np.int32); ll[0] = U - 1
    lab_t = t(labels) if labels.size else torch.zeros((B, 1), dtype=torch.int32, device=dev)
    try:
        c, g = pkg.rnnt_loss_and_grad(t(acts), lab_t, t(il), t(ll))
        cr, gr = orc.rnnt_loss_and_grad(acts, labels, il, ll)
        dc = float(np.abs(c.cpu().numpy() - cr).max() / max(1.0, np.abs(cr).max())); dg = float(np.abs(g.cpu().numpy() - gr).max())
        ok = dc <= 1e-4 and dg <= 1e-4
        bad += not ok
        print("ok " if ok else "BAD", (B, T, U, V), f"{dc:.1e} {dg:.1e}")
    except Exception as e:
        bad += 1; print("ERR", (B, T, U, V), type(e).__name__, str(e)[:150])
print("failures", bad)
