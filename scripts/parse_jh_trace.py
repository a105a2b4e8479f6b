"""Dev tool: print the K1 timeline recorded by a -DJH_TRACE build (scripts/build_variant.sh trace -DJH_TRACE)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/jh_trace.bin", dtype=np.int64).reshape(4, 8, 160)
for blk in range(4):
    t = a[blk]
    if not t.any():
        continue
    t0 = t[:, 0].min()
    print("block", blk, "prologue cycles per wave", (t[:, 1] - t[:, 0]).tolist())
    for w in (0, 4):
        rows = []
        for vc in (0, 1, 2, 10, 20, 31):
            b = 2 + 4 * vc
            rows.append((vc, int(t[w, b + 1] - t[w, b]), int(t[w, b + 2] - t[w, b + 1]), int(t[w, b + 3] - t[w, b + 2]),
                         int(t[w, b + 4] - t[w, b + 3])))
        print("  wave", w, "(chunk, wait+barrier, epilogue-before, mfma, epilogue-after):", rows)
    print("  tile total", int(t[:, 2 + 128].max() - t0), " chunk period avg", float(np.mean(t[0, 6:130:4] - t[0, 2:126:4])))
