import sys, torch
sys.path.insert(0, ".")
import bench
import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import _lib
pkg.build(); lib = _lib.load(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
for (V, J) in ((128, 640), (512, 640), (256, 384), (384, 640)):
    r = bench.bench_fused_joint(lib, _lib, dev, 32, 600, 150, V, J, st, 5)
    print(V, J, r.get("ms_per_step"), r.get("error"))
r = bench.bench_fused_joint(lib, _lib, dev, 16, 1500, 300, 1024, 640, st, 3)
print("c5", r.get("ms_per_step"))
