for pa in 0 1; do
  echo "== RNNT_PRECISE_ALL=$pa"
  RNNT_PRECISE_ALL=$pa timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-ragged 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for k in ('fused_joint','fused_joint_full','fused_joint_config5','fused_joint_v128','op_config5'):
            print(k, d[k].get('ms_per_step') if d.get(k) else None)"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp; ( cd $GRAFT_REPO_ROOT && RNNT_PRECISE_ALL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o pp --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-ragged > /dev/null 2>&1 )
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'sweep' in r['Name']: print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4}  {r['Name'][:70]}")
PY
