#!/bin/bash
# usage: gpu_j32_variant.sh TAG variant...   -> kernel times of the fused f32-grade joint at C2 for each library variant
# (knock-out variants give wrong results on purpose: no parity test here)
OUT=gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  [[ $v == product ]] && L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt.so
  echo "== variant $v"
  (cd /tmp && RNNT_LIBWARPRNNT=$L timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$v -o j -- python $GRAFT_REPO_ROOT/bench.py --fused-only 32,600,150,28 --steps 4 > $GRAFT_REPO_ROOT/$OUT/rocprof_$v.log 2>&1)
  python - $OUT/prof_$v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'joint_fwd' in r['Name'] or 'joint_bwd' in r['Name'] or 'cellrec' in r['Name']:
        print('   %-40s %.1f us' % (r['Name'][:40], float(r['AverageNs'])/1e3))
PY
done
