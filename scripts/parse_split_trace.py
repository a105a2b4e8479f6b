"""Dev tool: print the per-chunk timeline written by a -DSPLIT_TRACE build (RNNT_SWEEP_MODE=4, workgroup 0)."""
import struct, sys
d = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/split_trace.bin", "rb").read()
v = struct.unpack("<%dq" % (len(d) // 8), d)
names = ["alpha low (producer)", "alpha high (consumer)", "beta low (consumer)", "beta high (producer)"]
t0 = min(x for x in v if x)
for w in range(4):
    row = v[w * 256:(w + 1) * 256]
    n = max((i for i, x in enumerate(row) if x), default=-1) + 1
    print(names[w], "stamps", n)
    prev = None
    out = []
    for i in range(0, n, 2):
        a, b = row[i], row[i + 1]
        if not a:
            continue
        out.append("%d:+%d(w%d)" % (i // 2, (a - prev) if prev else a - t0, b - a))
        prev = a
    print("  chunk:start-delta(wait) ", " ".join(out))
