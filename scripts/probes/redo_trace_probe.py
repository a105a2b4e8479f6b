import sys, torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
pkg.build(); dev = torch.device("cuda:0")
B, T, U, V = 32, 600, 150, 28
g = torch.Generator(device=dev).manual_seed(3)
labels = torch.randint(1, V, (B, U - 1), generator=g, device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U - 1, dtype=torch.int32, device=dev)
x = torch.randn(B, T, U, V, generator=g, device=dev) * 8.0
for _ in range(3):
    pkg.rnnt_loss_and_grad(x, labels, il, ll); torch.cuda.synchronize()
