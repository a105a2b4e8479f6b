import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rnnt_speech_recognition_amd as pkg
pkg.build(); dev = torch.device("cuda:0")
B, T, U, V = 32, 600, 150, 28
g = torch.Generator(device=dev).manual_seed(3)
labels = torch.randint(1, V, (B, U - 1), generator=g, device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
for sigma in [float(x) for x in os.environ.get('SIGMAS', '1,4,6,8').split(',')]:
    x = torch.randn(B, T, U, V, generator=g, device=dev) * sigma
    for _ in range(3): pkg.rnnt_loss_and_grad(x, labels, il, ll)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): pkg.rnnt_loss_and_grad(x, labels, il, ll)
    torch.cuda.synchronize(); print(f"sigma {sigma}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step")
# one flagged utterance in an otherwise ordinary batch (the cliff the team of workgroups is for)
if os.environ.get('ONE', '1') == '1':
    x = torch.randn(B, T, U, V, generator=g, device=dev)
    x[5] *= 8.0
    for _ in range(3): pkg.rnnt_loss_and_grad(x, labels, il, ll)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): pkg.rnnt_loss_and_grad(x, labels, il, ll)
    torch.cuda.synchronize(); print(f"one utterance at sigma 8: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step")
