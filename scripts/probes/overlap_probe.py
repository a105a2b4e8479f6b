"""Dev probe: how long do the linear sweeps take while ANOTHER stream saturates HBM with the gradient pass?
Stream A: compute_rnnt_loss_bwd in a loop on problem 1 (the hog); stream B: compute_rnnt_loss_fwd on problem 2.
Run under rocprofv3 --kernel-trace --stats; argv[1] = 0 (no hog) / 1 (hog)."""
import sys
import torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import _lib

hog = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pkg.build()
lib = _lib.load()
dev = torch.device("cuda:0")
B, T, U, V = 32, 600, 150, 28
g = torch.Generator(device=dev).manual_seed(1)
probs = []
for i in range(2):
    x = torch.randn(B, T, U, V, generator=g, device=dev)
    labels = torch.randint(1, V, (B, U - 1), generator=g, device=dev, dtype=torch.int32)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
    costs = torch.empty(B, device=dev)
    grads = torch.empty_like(x)
    ws = torch.empty(_lib.workspace_bytes(T, U, B), dtype=torch.uint8, device=dev)
    probs.append((x, labels, il, ll, costs, grads, ws))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
oa, ob = _lib.make_options(sa.cuda_stream, 0, T, U), _lib.make_options(sb.cuda_stream, 0, T, U)
torch.cuda.synchronize()


def fwd(pr, o):
    x, labels, il, ll, costs, grads, ws = pr
    _lib.check(lib.compute_rnnt_loss_fwd(x.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(), V, B, costs.data_ptr(), ws.data_ptr(), o), "fwd")


def bwd(pr, o):
    x, labels, il, ll, costs, grads, ws = pr
    _lib.check(lib.compute_rnnt_loss_bwd(x.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), il.data_ptr(), None, V, B, ws.data_ptr(), o), "bwd")


fwd(probs[0], oa)
torch.cuda.synchronize()
for it in range(40):
    if hog:
        bwd(probs[0], oa)
        bwd(probs[0], oa)
    fwd(probs[1], ob)
torch.cuda.synchronize()
print("costs", probs[1][4][:3].tolist())
