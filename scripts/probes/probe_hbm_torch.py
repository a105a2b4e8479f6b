"""Dev probe: what plain PyTorch kernels reach on this part's HBM (read-only reduction, copy, fill), for scale next to the
loss kernels' rates.  python scripts/probes/probe_hbm_torch.py"""
import time
import torch

dev = torch.device("cuda:0")
n = 1 << 30  # 4 GiB of f32: far beyond the 256 MiB Infinity Cache
x = torch.randn(n, device=dev)
y = torch.empty_like(x)


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


b = n * 4
print("read  (sum)   %.2f TB/s" % (b / timed(lambda: x.sum()) / 1e12))
print("read  (max)   %.2f TB/s" % (b / timed(lambda: x.max()) / 1e12))
print("copy  (r + w) %.2f TB/s" % (2 * b / timed(lambda: y.copy_(x)) / 1e12))
print("scale (r + w) %.2f TB/s" % (2 * b / timed(lambda: torch.mul(x, 2.0, out=y)) / 1e12))
print("fill  (write) %.2f TB/s" % (b / timed(lambda: y.fill_(1.0)) / 1e12))
