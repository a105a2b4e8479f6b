#!/bin/bash
# kernel statistics of BASELINE configs[2] (end-to-end train step): where do the 49 ms go?
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2e -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused --no-ragged --no-config5 > /tmp/log_e2e 2>/tmp/err_e2e)
python - <<'PY'
import csv,glob,json
f=glob.glob('/tmp/prof_e2e/**/*kernel_stats.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms over the run' % (tot/1e6))
for r in rows[:22]:
    print('%8.2f ms %6d calls %8.1f us avg  %s' % (float(r['TotalDurationNs'])/1e6, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:110]))
try:
    d=json.loads(open('/tmp/log_e2e').readline()); print('e2e', d['e2e_train_step'])
except Exception as e: print('no json', e)
PY
grep -i "miopen\|warn" /tmp/err_e2e | head -5
