#!/bin/bash
# Dev: fused legs + the sweep kernels' rocprofv3 averages (the f32-grade joint's sweeps run the float64 recurrence)
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pq
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o pq --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-ragged --no-config5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for k in ('fused_joint','fused_joint_full','fused_dp_step'): print(k, d[k].get('ms_per_step') if d.get(k) else None)" )
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pq/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "sweep" in r["Name"]:
        print("   %9.1f us x %4s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:70]))
PY
