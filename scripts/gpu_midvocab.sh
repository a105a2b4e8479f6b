cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mv
( cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/mv -o mv --output-format csv -- python scripts/probes/midvocab_probe.py only128 2>&1 | tail -2 )
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/mv/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(f"   {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']:>4}  {r['Name'][:70]}")
PY
