"""Dev tool (uses oracle/): max |d grad| against the float64 oracle over several seeds, to compare numerics variants
beyond the extreme-value noise of a single draw.  usage: accuracy_seeds.py T U V nseeds"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc

T, U, V, ns = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda:0")
errs, rms = [], []
for seed in range(ns):
    rng = np.random.default_rng(1000 + seed)
    acts = rng.normal(size=(1, T, U, V)).astype(np.float32)
    labels = rng.integers(1, V, size=(1, U - 1)).astype(np.int32)
    il, ll = np.array([T], np.int32), np.array([U - 1], np.int32)
    c, g = pkg.rnnt_loss_and_grad(torch.tensor(acts, device=dev), torch.tensor(labels, device=dev),
                                  torch.tensor(il, device=dev), torch.tensor(ll, device=dev))
    cr, gr, _, _, _ = orc.utterance_cost_and_grad(acts[0], labels[0])
    d = g.cpu().numpy()[0] - gr
    errs.append(float(np.abs(d).max()))
    rms.append(float(np.sqrt((d * d).sum() / (np.abs(gr) > 1e-3).sum())))
print(f"T={T} U={U} V={V}: max|dgrad| per seed " + " ".join(f"{e:.2e}" for e in errs) +
      f" | mean {np.mean(errs):.2e} worst {np.max(errs):.2e} | rms over significant entries {np.mean(rms):.2e}")
