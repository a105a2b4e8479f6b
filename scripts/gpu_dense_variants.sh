#!/bin/bash
# usage: gpu_dense_variants.sh variant...  -> dense-layer parity tests + dense / joint kernel times in the fused bench, per library variant
export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  [[ $v == product ]] && L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt.so
  echo "== $v: $(RNNT_LIBWARPRNNT=$L timeout 300 python -m pytest tests/test_dense_gpu.py -m gpu -q -x 2>&1 | tail -1)"
  (cd /tmp && RNNT_LIBWARPRNNT=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd_$v -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-e2e --no-config5 > /tmp/logd_$v 2>/dev/null)
  python - /tmp/profd_$v /tmp/logd_$v <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'dense_' in n: print('   %-44s %7.1f us x %s' % (n.split('rnnt::')[-1][:44] if 'rnnt::' in n else n[:44], float(r['AverageNs'])/1e3, r['Calls']))
try:
    d=json.loads(open(sys.argv[2]).readline()); print('   fused_joint_full %.4f ms   fused_joint %.4f ms   fused_dp_step %.4f ms' % (d['fused_joint_full']['ms_per_step'], d['fused_joint']['ms_per_step'], d['fused_dp_step']['ms_per_step']))
except Exception as e: print('   no json', e)
PY
done
