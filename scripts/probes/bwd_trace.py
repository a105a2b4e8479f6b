"""Print the s_memtime stamps a -DJH_TRACE build of joint_bwd_kernel leaves (joint_kernels.hip bwd_consumer BT(k), bwd_producer PT(k)):
consumers 0 / 4 / 8 (one SIMD) of workgroup 20, rows 40..71, six stamps per row; the two producers, four stamps per own row.
usage: bwd_trace.py /path/to/j32_trace.bin      (s_memtime ticks: 100 MHz constant clock -> 10 ns each)"""
import sys
import numpy as np

h = np.fromfile(sys.argv[1], dtype=np.int64)
t0 = h[h > 0].min()
print("consumer stamps (ticks since first stamp): row | wave: BT0 BT1-0 BT2-1 BT3-2 BT4-3 BT5-4 | period")
for w in range(3):
    c = h[w * 256: w * 256 + 192].reshape(32, 6)
    per = np.diff(c[:, 0])
    d = np.diff(c, axis=1)
    print(f"consumer {4 * w}: mean deltas tanh {d[:,0].mean():.1f} dh {d[:,1].mean():.1f} poll+loads {d[:,2].mean():.1f} "
          f"dW {d[:,3].mean():.1f} dz {d[:,4].mean():.1f} | row period {per.mean():.1f} (min {per.min()}, max {per.max()})")
for pw in range(2):
    q = h[768 + pw * 64: 768 + pw * 64 + 64].reshape(16, 4)
    per = np.diff(q[:, 0])
    d = np.diff(q, axis=1)
    tail = q[1:, 0] - q[:-1, 3]
    print(f"producer {pw}: issue loads {d[:,0].mean():.1f} slot wait {d[:,1].mean():.1f} build+publish {d[:,2].mean():.1f} "
          f"end-of-iteration (wait for the next row's loads + copies) {tail.mean():.1f} | own-row period {per.mean():.1f}")
    print("   rows:", " ".join(f"{int(x - t0)}" for x in q[:, 0]))
