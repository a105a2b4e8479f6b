"""Dev tool: print the K1 timeline recorded by a -DJH_TRACE build (scripts/build_variant.sh trace -DJH_TRACE)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/jh_trace.bin", dtype=np.int64)[: 4 * 8 * 160].reshape(4, 8, 160)
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
    for w in (0, 4):  # phase averages by position of the chunk in its group of four (park mode flushes once per group)
        for r in range(4):
            vcs = [vc for vc in range(4, 32) if vc % 4 == r]
            ph = np.array([[t[w, 2 + 4 * vc + k + 1] - t[w, 2 + 4 * vc + k] for k in range(4)] for vc in vcs], float).mean(0)
            print("  wave %d chunk%%4=%d  wait+barrier %.0f  before %.0f  mfma %.0f  after %.0f" % (w, r, *ph))
    print("  tile total", int(t[:, 2 + 128].max() - t0), " chunk period avg", float(np.mean(t[0, 6:130:4] - t[0, 2:126:4])))

raw = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/jh_trace.bin", dtype=np.int64)
if raw.size >= 6 * 8 * 160:
    k4 = raw.reshape(6, 8, 160)[4]
    k3 = raw.reshape(6, 8, 160)[5]
    if k4.any():
        for w in (0, 1, 7):
            t = k4[w, : 4 * 39].reshape(39, 4)
            print("K4 wave", w, "per step (wait+barrier, dma issue, build_h, mfma+rest -> next step):")
            print("   ", [(int(t[i, 1] - t[i, 0]), int(t[i, 2] - t[i, 1]), int(t[i, 3] - t[i, 2]), int(t[i + 1, 0] - t[i, 3])) for i in (2, 3, 10, 20, 30)])
        print("K4 step period avg", float(np.mean(np.diff(k4[0, 0 : 4 * 39 : 4])[2:])))
    if k3.any():
        for w in (0, 7):
            t = k3[w, : 3 * 52].reshape(52, 3)
            print("K3 wave", w, "per chunk (wait+barrier, dma issue, mfma (+epilogue every 16th) -> next chunk):")
            print("   ", [(int(t[i, 1] - t[i, 0]), int(t[i, 2] - t[i, 1]), int(t[i + 1, 0] - t[i, 2])) for i in (2, 3, 10, 15, 20, 31, 40)])
        print("K3 chunk period avg", float(np.mean(np.diff(k3[0, 0 : 3 * 52 : 3])[2:])))
