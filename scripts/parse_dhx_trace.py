"""Dev tool: print the jh_dhx_kernel timeline recorded by a -DJH_TRACE build (scripts/build_variant.sh trace -DJH_TRACE):
per step the shader clocks between the stamps {top, after the vmcnt/lgkm wait, after the barrier, after the fragment reads + store +
A-DMA issue, after the first k-step (W2-DMA issue inside), after the second k-step, after the conversion} of some waves of one workgroup."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
k3 = raw.reshape(6, 8, 160)[5]
for w in (0, 4, 1, 7):
    t = k3[w, : 7 * 22].reshape(22, 7)
    if not t.any():
        continue
    print("wave", w, "per step: wait, barrier, reads+store+dmaA, kstep0(+dmaB), kstep1, convert, -> next top")
    for i in list(range(0, 14)) + [18, 19, 20]:
        print("   step %2d: %6d %6d %6d %6d %6d %6d %6d" % (i, t[i, 1] - t[i, 0], t[i, 2] - t[i, 1], t[i, 5] - t[i, 2], t[i, 6] - t[i, 5], t[i, 3] - t[i, 6],
                                                        t[i, 4] - t[i, 3], t[i + 1, 0] - t[i, 4]))
    print("   step period avg (steps 2..20)", float(np.mean(np.diff(t[2:21, 0]))))
    e = k3[w, 154:159]
    print("   epilogue of the second iteration: first unit tile %d, second %d, tiles 2..%d, next factors %d" % (e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]))
