"""Dev tool: per-row timeline of three same-SIMD consumer waves of joint_bwd_kernel from a -DJH_TRACE build
(JH_TRACE_FILE32): stamps 0 row start, 1 after tanh, 2 after the dh MFMAs were issued, 3 after the next row's poll + reads,
4 after the dW2 MFMAs were issued, 5 after dz / sums.  Prints stage durations in shader cycles."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64)[:768].reshape(3, 32, 8)[:, :, :6] if False else np.fromfile(sys.argv[1], dtype=np.int64)
w = a[:768].reshape(3, 256)[:, :192].reshape(3, 32, 6)
for k in range(3):
    t = w[k]
    if not t.any():
        continue
    d = np.diff(t, axis=1)
    row = t[1:, 0] - t[:-1, 0]
    print(f"consumer {4 * k}: stage cycles (tanh, dh, poll+reads, dW2, dz) median", np.median(d, axis=0).astype(int).tolist(),
          " row period median", int(np.median(row)), " first rows", row[:8].tolist())
    print("   row-start offset vs consumer 0:", (t[:8, 0] - w[0][:8, 0]).tolist())
