#!/bin/bash
# usage: gpu_c5_variants.sh variant...  -> per-kernel times (rocprofv3 --stats) of the f16 joint at BASELINE config 5 for each
# library variant (lib/libwarprnnt_<variant>.so from scripts/build_variant.sh; "product" = the shipped library)
export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  [[ $v == product ]] && L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt.so
  (cd /tmp && RNNT_LIBWARPRNNT=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --fused-only 16,1500,300,1024 --steps 3 > /tmp/log_$v 2>/dev/null)
  python - /tmp/prof_$v $v /tmp/log_$v <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
t=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    n=n.replace('void ','').replace('(rnnt::JhParams)','')
    if n.startswith('rnnt::jh_') and float(r['AverageNs'])>1e5: t.append((n.replace('rnnt::jh_','').replace('_kernel',''), float(r['AverageNs'])/1e6))
try: ms=json.loads(open(sys.argv[3]).readline())['fused_joint']['ms_per_step']
except Exception: ms=float('nan')
print('%-10s step %.2f ms (profiled run)  %s' % (sys.argv[2], ms, '  '.join('%s %.2f' % kv for kv in sorted(t))))
PY
done
