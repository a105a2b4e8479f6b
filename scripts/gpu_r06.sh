#!/bin/bash
# Round-6 GPU session driver (runs on the GPU box via gpurun).  usage: scripts/gpu_r06.sh TAG "steps..." [variants...]
#   f16       tests/test_joint_f16_gpu.py per library variant
#   c5        per-kernel times (rocprofv3 --stats) of the f16 joint at BASELINE config 5 per variant; LEG=n01|n01_all_rows|trained|trained_all_rows
#             picks the input / RNNT_VISIT_ALL leg (default: N(0,1) projections, every row visited)
#   mid       the same at B32 T600 U150 V128 / V256 and at the reference-default shape B16 T300 U100 V4096
#   sizes     tests/test_baseline_sizes_gpu.py (at-size parity)
#   test      full `pytest -m gpu`
#   bench     python bench.py (default line)
TAG=${1:-r06}; WHAT=${2:-"f16 c5"}; shift 2
VARS=${@:-product}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
lib() { local v=$1; [[ $v == product ]] && echo $R/rnnt-speech-recognition_amd/lib/libwarprnnt.so || echo $R/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so; }
if has f16; then
  for v in $VARS; do echo "== f16 tests $v"; RNNT_LIBWARPRNNT=$(lib $v) timeout 900 python -m pytest tests/test_joint_f16_gpu.py -m gpu -q --maxfail=10 2>&1 | tee $OUT/pytest_f16_$v.log | tail -15; done
fi
shape_prof() {  # name shape
  for v in $VARS; do
    (cd /tmp && RNNT_LIBWARPRNNT=$(lib $v) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$v -o b -- python $R/bench.py --fused-only $2 --steps 3 --fused-leg ${LEG:-n01_all_rows} > /tmp/log_$1_$v 2>/tmp/err_$1_$v)
    python - /tmp/prof_$1_$v $v /tmp/log_$1_$v $1 <<'PY' | tee -a $OUT/shapes.txt
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
t=[]
if f:
  for r in csv.DictReader(open(f[0])):
    n=r['Name'].replace('void ','').replace('(rnnt::JhParams)','')
    if n.startswith('rnnt::') and float(r['AverageNs'])>5e4: t.append((n.replace('rnnt::','').replace('_kernel',''), float(r['AverageNs'])/1e6))
try: ms=json.loads(open(sys.argv[3]).readline())['fused_joint']['ms_per_step']
except Exception as e: ms=float('nan')
print('%-8s %-10s step %.3f ms (profiled run)  %s' % (sys.argv[4], sys.argv[2], ms, '  '.join('%s %.3f' % kv for kv in sorted(t, key=lambda kv:-kv[1])[:8])))
PY
  done
}
if has c5; then shape_prof c5 16,1500,300,1024; fi
if has mid; then shape_prof v128 32,600,150,128; shape_prof v256 32,600,150,256; shape_prof v4096 16,300,100,4096; fi
if has sizes; then
  timeout 1500 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q --maxfail=10 --durations=8 > $OUT/pytest_sizes.log 2>&1; echo "sizes rc=$?"; tail -25 $OUT/pytest_sizes.log
fi
if has test; then
  timeout 3000 python -m pytest tests -m gpu -q --maxfail=40 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -40 $OUT/pytest_gpu.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
fi
