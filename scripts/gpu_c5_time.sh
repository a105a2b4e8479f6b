#!/bin/bash
# usage: gpu_c5_time.sh TAG variant...   -> config-5 kernel times (rocprofv3 kernel stats) for each library variant, no tests
OUT=gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  [[ $v == product ]] && L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt.so
  echo "== variant $v"
  (cd /tmp && RNNT_LIBWARPRNNT=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$v -o c5 -- python $GRAFT_REPO_ROOT/bench.py --fused-only 16,1500,300,1024 --steps 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_$v.log 2>&1)
  python - $OUT/prof_$v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    if 'jh_' in r['Name'] and 'prep' not in r['Name']:
        print('   %-60s %.2f ms' % (r['Name'][:60], float(r['AverageNs'])/1e6)); tot+=float(r['AverageNs'])/1e6
print('   sum of the four kernels %.2f ms' % tot)
PY
done
