import sys, numpy as np, torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
from oracle import rnnt_oracle as orc
sys.path.insert(0, "tests")
from test_peaky_wide_gpu import make_logits
dev = torch.device("cuda:0")
pkg.build()
for (B, T, U, V) in ((2, 120, 60, 64), (2, 200, 90, 100), (1, 60, 300, 1024), (1, 30, 1100, 4), (1, 1200, 1100, 4), (2, 150, 70, 31)):
    for kind in ("sigma1", "sigma4", "sigma8", "trained"):
        x, labels, il, ll = make_logits(kind, B, T, U, V, seed=T + U)
        xt = torch.from_numpy(x).to(dev)
        if V == 31:  # unaligned view: the wave-per-cell kernels
            buf = torch.empty(x.size + 1, dtype=torch.float32, device=dev)
            xt = buf[1:].view(B, T, U, V); xt.copy_(torch.from_numpy(x).to(dev))
        c, g = pkg.rnnt_loss_and_grad(xt, torch.from_numpy(labels).to(dev), torch.from_numpy(il).to(dev), torch.from_numpy(ll).to(dev))
        c = c.cpu().numpy().astype(np.float64); g = g.cpu().numpy()
        dc = dg = 0.0
        for b in range(B):
            cr, gr, _, _, _ = orc.utterance_cost_and_grad(x[b], labels[b])
            dc = max(dc, abs(c[b] - cr) / max(1, abs(cr))); dg = max(dg, float(np.abs(g[b] - gr).max()))
        print(f"B{B} T{T} U{U} V{V} {kind:8s} dcost {dc:.1e} dgrad {dg:.2e}", flush=True)
