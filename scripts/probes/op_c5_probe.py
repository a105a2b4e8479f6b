"""Dev probe: the loss op at BASELINE configs[4]'s shape (B16 T1500 U300 V1024, 29.5 GB of f32 logits) -- two calls as a caller
makes them (cells below the occupancy floor get zeros, their logits are not read), then two with RNNT_VISIT_ALL, for counter passes
(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: the kernel trace separates the two pairs by order)."""
import sys
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rnnt_speech_recognition_amd as pkg  # noqa: E402
from rnnt_speech_recognition_amd import _lib  # noqa: E402

pkg.build()
lib = _lib.load()
B, T, U, V = 16, 1500, 300, 1024
dev = torch.device("cuda:0")
gd = torch.Generator(device=dev).manual_seed(4321)
acts = torch.randn(B, T, U, V, generator=gd, dtype=torch.float32, device=dev)
grads = torch.empty_like(acts)
g = torch.Generator().manual_seed(4321)
labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32).to(dev)
il = torch.full((B,), T, dtype=torch.int32, device=dev)
ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
scale = torch.full((B,), 1.0 / B, device=dev)
costs = torch.empty(B, device=dev)
ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
opts = _lib.make_options(torch.cuda.current_stream().cuda_stream, 0, T, U)
for flags in (0, 0, _lib.RNNT_VISIT_ALL, _lib.RNNT_VISIT_ALL):
    _lib.check(lib.compute_rnnt_loss_flags(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(),
                                           scale.data_ptr(), V, B, costs.data_ptr(), ws.data_ptr(), opts, flags), "loss")
torch.cuda.synchronize()
print("costs finite:", bool(torch.isfinite(costs).all()))
