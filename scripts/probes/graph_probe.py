"""Probe: the headline op and the fused joint step, direct launches against HIP-graph replay (launch gaps between the step's kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rnnt_speech_recognition_amd as pkg
pkg.build(); dev = torch.device("cuda:0")
B, T, U, V = 32, 600, 150, 28
g = torch.Generator(device=dev).manual_seed(3)
labels = torch.randint(1, V, (B, U - 1), generator=g, device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
xs = [torch.randn(B, T, U, V, generator=g, device=dev) for _ in range(3)]
def step(i): return pkg.rnnt_loss_and_grad(xs[i % 3], labels, il, ll)
def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for i in range(6): step(i)
print("op direct        %.4f ms/step" % timeit(step, 60))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(3): step(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        outs = [step(i) for i in range(3)]
torch.cuda.synchronize()
for _ in range(3): gr.replay()
print("op graph replay  %.4f ms/step" % (timeit(lambda i: gr.replay(), 20) / 3))
