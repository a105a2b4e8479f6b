"""Dev probe: the configs[2] train step (bench.py bench_e2e) launched eagerly vs replayed from ONE HIP graph
(forward + backward + SGD update captured with static input buffers)."""
import os, sys, time, torch
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg

# BLAS=cublas|cublaslt|default: which library torch's GEMMs (the LSTM's) go to -- on ROCm "cublas" = rocBLAS, "cublaslt" = hipBLASLt
blas = os.environ.get("BLAS", "default")
if blas != "default":
    torch.backends.cuda.preferred_blas_library(blas)
print("preferred_blas_library:", torch.backends.cuda.preferred_blas_library())

dev = torch.device("cuda:0")
hp = pkg.HParams(vocab_size=28, embedding_size=320, encoder_layers=2, encoder_size=320, projection_size=320,
                 time_reduction_index=0, pred_net_layers=1, pred_net_size=320, joint_net_size=320)
torch.manual_seed(1234)
model = pkg.Transducer(hp).to(dev)
batch = pkg.synthetic_batch(hp, batch=64, frames=600, max_labels=100, device=dev, seed=1234)
step = pkg.TrainStep(model, global_batch=64)
for _ in range(2):
    log = step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    log = step(*batch)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3, "loss", log["loss"])

params = step.params
opt = step.optimizer
static = [b.clone() for b in batch]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        costs = model.loss(*static)
        (costs.sum() * (1.0 / 64)).backward()
        opt.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
try:
    with torch.cuda.graph(g):
        costs = model.loss(*static)
        loss = costs.sum() * (1.0 / 64)
        loss.backward()
        opt.step()
except Exception as e:
    print("capture failed:", repr(e)[:600])
    sys.exit(0)
torch.cuda.synchronize()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 5 * 1e3, "loss", float(loss))
