"""Dev tool: where does the f32 lattice lose precision on very wide lattices (U >> T)?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc

dev = torch.device("cuda:0")
for (T, U, V, seed, sc) in [(23, 533, 8, 0, 4.0), (84, 574, 8, 1, 4.0), (600, 150, 28, 2, 4.0), (600, 150, 28, 3, 8.0), (300, 100, 28, 4, 4.0), (84, 574, 8, 5, 1.0)]:
    rng = np.random.default_rng(seed)
    acts = (sc * rng.normal(size=(1, T, U, V))).astype(np.float32)
    labels = rng.integers(1, V, size=(1, U - 1)).astype(np.int32)
    il, ll = np.array([T], np.int32), np.array([U - 1], np.int32)
    c, g = pkg.rnnt_loss_and_grad(torch.tensor(acts, device=dev), torch.tensor(labels, device=dev),
                                  torch.tensor(il, device=dev), torch.tensor(ll, device=dev))
    cr, gr, al, be, lp = orc.utterance_cost_and_grad(acts[0], labels[0])
    err = np.abs(g.cpu().numpy()[0] - gr)
    t, u, v = np.unravel_index(err.argmax(), err.shape)
    n = t + u
    # natural-log alpha along diagonal n, relative to the straight-line ridge cell of that diagonal
    us = np.arange(max(0, n - T + 1), min(U - 1, n) + 1)
    a_diag = al[n - us, us]
    ur = int(round(n * (U - 1) / (T + U - 2)))
    ur = min(max(ur, us[0]), us[-1])
    print(f"T={T} U={U} V={V} x{sc}: max|dgrad|={err.max():.2e} at t={t} u={u} (grad there {gr[t,u,v]:.3f}); "
          f"alpha(t,u)-alpha(ridge cell u={ur}) = {(al[t,u]-al[n-ur,ur])/np.log(2):.1f} bits; "
          f"alpha+beta-ll there = {(al[t,u]+be[t,u]+cr)/np.log(2):.2f} bits; diag max - ridge = {(a_diag.max()-al[n-ur,ur])/np.log(2):.1f} bits")
