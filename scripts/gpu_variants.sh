#!/bin/bash
# usage: gpu_variants.sh variant...   -> op-level kernel times (rocprofv3 --stats) + bench ms/step for each library variant
# (lib/libwarprnnt_<variant>.so built by scripts/build_variant.sh; "product" = the shipped library)
export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  [[ $v == product ]] && L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt.so
  (cd /tmp && RNNT_LIBWARPRNNT=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > /tmp/log_$v 2>/dev/null)
  python - /tmp/prof_$v $v /tmp/log_$v <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
t={}
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'cell_tile_kernel' in n: t['grad' if 'true, true' in n or 'true, false' in n.split('<')[1][:12] and False else ('grad' if ', true,' in n else 'lsm')]=float(r['AverageNs'])/1e3
    if 'sweep' in n: t['sweep']=float(r['AverageNs'])/1e3
try: ms=json.loads(open(sys.argv[3]).readline())['ms_per_step']
except Exception: ms=float('nan')
print('%-10s step %.4f ms (profiled run)  %s' % (sys.argv[2], ms, '  '.join('%s %.1f us' % kv for kv in sorted(t.items()))))
PY
done
