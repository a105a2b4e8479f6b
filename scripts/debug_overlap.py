"""Diagnostic (GPU): where did the overlap-mode hand-offs run?  Prints the per-XCD lsm patch counters and,
for every sweep wave, its XCD and whether it took the same-L2 fast path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RNNT_OVERLAP"] = "1"
import numpy as np, torch
import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import _lib
pkg.build(); lib = _lib.load()
B, T, U, V = 32, 600, 150, 28
dev = torch.device("cuda:0")
acts = torch.randn(B, T, U, V, device=dev)
labels = torch.randint(1, V, (B, U - 1), dtype=torch.int32, device=dev)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
costs = torch.empty(B, device=dev); grads = torch.empty_like(acts)
n = _lib.workspace_bytes(T, U, B)
ws = torch.zeros(n, dtype=torch.uint8, device=dev)
opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, 0, T, U)
for _ in range(3):
    _lib.check(lib.compute_rnnt_loss_ex(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(),
                                        None, V, B, costs.data_ptr(), ws.data_ptr(), opts), "ex")
torch.cuda.synchronize()
words = 13 * B + 32
fbytes = (words * 4 + 255) // 256 * 256
flags = ws[n - fbytes:n - fbytes + words * 4].view(torch.int32).cpu().numpy()
cnt = flags[: 8 * B].reshape(B, 8)
print("lsm patches per (utterance, XCD):")
print(cnt[:8]); print("...", cnt.sum(1)[:8], "nonzero XCDs per utterance:", (cnt > 0).sum(1))
print("sweep done:", flags[8 * B: 9 * B][:8], "err:", flags[9 * B], "lsm_done:", flags[9 * B + 1])
diag = flags[9 * B + 16: 11 * B + 16].reshape(B, 2)
print("sweep (xcd, fast) alpha:", [(int(d & 255), int(d >> 8)) for d in diag[:, 0]][:16])
print("sweep (xcd, fast) beta :", [(int(d & 255), int(d >> 8)) for d in diag[:, 1]][:16])
print("fast fraction:", float(((diag >> 8) & 1).mean()))

