"""Dev probe: the fused f16 joint at mid-sized vocabularies / joint widths (B32 T600 U150), and config 5 as the control."""
import sys, torch
sys.path.insert(0, ".")
import bench
import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import _lib
pkg.build(); lib = _lib.load(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
shapes = ((128, 640), (256, 384)) if len(sys.argv) > 1 else ((128, 640), (512, 640), (256, 384), (384, 640))
for (V, J) in shapes:
    r = bench.bench_fused_joint(lib, _lib, dev, 32, 600, 150, V, J, st, 5)
    print(V, J, r.get("ms_per_step"), r.get("error"))
if len(sys.argv) == 1:
    r = bench.bench_fused_joint(lib, _lib, dev, 16, 1500, 300, 1024, 640, st, 3)
    print("c5", r.get("ms_per_step"))
