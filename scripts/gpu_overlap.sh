#!/bin/bash
# Dev: duration of the linear sweeps with / without a second stream saturating HBM with the gradient pass
# (scripts/probes/overlap_probe.py; result in profiles/r04_notes.md "overlap").
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for hog in 0 1; do
  echo "== hog $hog"
  rm -rf /tmp/ov
  ( cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ov -o ov --output-format csv -- python scripts/probes/overlap_probe.py $hog 2>&1 | grep costs )
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ov/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'rnnt' in r['Name']:
        print(f"   {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']:>4}  {r['Name'][:60]}")
PY
done
