"""Dev tool (uses oracle/): non-finite inputs must stay inside their own utterance and never hang a kernel."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc

dev = torch.device("cuda:0")
t = lambda x: torch.tensor(x, device=dev)
rng = np.random.default_rng(3)
B, T, U, V = 4, 23, 12, 28
acts = rng.normal(size=(B, T, U, V)).astype(np.float32)
labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
il, ll = np.full(B, T, np.int32), np.full(B, U - 1, np.int32)
clean_c, clean_g = orc.rnnt_loss_and_grad(acts, labels, il, ll)
bad = acts.copy()
bad[1, 3, 2, 5] = np.nan
bad[2, 7, 4, 0] = np.inf
bad[3, 9, 1, 3] = -np.inf
c, g = pkg.rnnt_loss_and_grad(t(bad), t(labels), t(il), t(ll))
torch.cuda.synchronize()
c, g = c.cpu().numpy(), g.cpu().numpy()
print("costs:", c)
print("utterance 0 untouched:", float(abs(c[0] - clean_c[0])), float(np.abs(g[0] - clean_g[0]).max()))
print("NaN utterance -> NaN cost:", bool(np.isnan(c[1])), " +inf logit:", c[2], " -inf logit (a label nobody needs unless forced):", c[3])

J = 64
ep, pp = rng.normal(size=(B, T, J)).astype(np.float32), rng.normal(size=(B, U, J)).astype(np.float32)
W1, b1 = np.eye(J, dtype=np.float32), np.zeros(J, np.float32)
W2, b2 = (rng.normal(size=(J, V)) * 0.2).astype(np.float32), np.zeros(V, np.float32)
ep[1, 2, 3] = np.nan
ep[2, 5, 7] = np.inf
params = [t(x).requires_grad_(True) for x in (ep, pp, W1, b1, W2, b2)]
costs = pkg.rnnt_joint_loss(*params, t(labels), t(il), t(ll))
costs.sum().backward()
torch.cuda.synchronize()
print("joint costs:", costs.detach().cpu().numpy())
ref = orc.joint_loss_and_grads(ep[:1].astype(np.float64), pp[:1].astype(np.float64), W1.astype(np.float64), b1.astype(np.float64),
                               W2.astype(np.float64), b2.astype(np.float64), labels[:1], il[:1], ll[:1])
print("joint utterance 0 cost untouched:", float(abs(costs[0].item() - ref["costs"][0])))
print("done")
