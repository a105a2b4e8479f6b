export TMPDIR=/tmp
for shape in ${SHAPES:-32,600,150,28 32,600,150,31 32,600,150,32}; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$shape -o b -- python $GRAFT_REPO_ROOT/bench.py --shape $shape --steps 30 --warmup 5 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > /tmp/log_$shape 2>/dev/null)
  python - /tmp/prof_$shape $shape /tmp/log_$shape <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'rnnt::' in n: out.append('%s %.1f us' % (n.split('rnnt::')[1][:34], float(r['AverageNs'])/1e3))
try: ms=json.loads(open(sys.argv[3]).readline())['ms_per_step']
except Exception: ms=float('nan')
print('%-14s step %.4f ms | %s' % (sys.argv[2], ms, ' | '.join(out)))
PY
done
