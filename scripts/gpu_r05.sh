#!/bin/bash
# Round-5 GPU session driver (runs on the GPU box via gpurun).  usage: scripts/gpu_r05.sh TAG "steps..." [variants...]
#   test      full `pytest -m gpu` (all failures listed, not -x)
#   quick     the parity tests of the paths this round touched
#   ab        bench.py (no CPU baseline / e2e) for each library variant, fused legs summarised (gpu_bench_variants.sh)
#   handback  scripts/probes/handback_probe.py for each variant
#   fused     bench.py --fused-only 32,600,150,28 for each variant
TAG=${1:-r05}; WHAT=${2:-"quick"}; shift 2
VARS=${@:-product}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
lib() { local v=$1; [[ $v == product ]] && echo $R/rnnt-speech-recognition_amd/lib/libwarprnnt.so || echo $R/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so; }
if has quick; then
  timeout 1500 python -m pytest tests/test_loss_gpu.py tests/test_lin_gpu.py tests/test_joint_gpu.py tests/test_joint_edges_gpu.py tests/test_dense_gpu.py tests/test_peaky_gpu.py tests/test_baseline_sizes_gpu.py -m gpu -q --maxfail=40 --durations=8 ${PYTEST_ARGS} > $OUT/pytest_quick.log 2>&1; echo "quick rc=$?"
  tail -60 $OUT/pytest_quick.log
fi
if has test; then
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -40 $OUT/pytest_gpu.log
fi
if has handback; then
  for v in $VARS; do echo "== handback $v"; RNNT_LIBWARPRNNT=$(lib $v) timeout 300 python scripts/probes/handback_probe.py 2>&1 | tee $OUT/handback_$v.log | grep -v Warning; done
fi
if has hbprof; then
  for sg in 4 8; do
    (cd /tmp && SIGMAS=$sg ONE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/hbprof$sg -o hb -- python $R/scripts/probes/handback_probe.py > $R/$OUT/hbprof$sg.log 2>&1)
    python scripts/summarize_trace.py stats $OUT/hbprof$sg $OUT/hb${sg}_kernel_stats.json $OUT/hb${sg}_kernel_stats.csv; echo "== handback sigma $sg"; head -8 $OUT/hb${sg}_kernel_stats.csv | cut -c1-150
  done
fi
if has profj; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profj -o j -- python $R/bench.py --fused-only 32,600,150,28 --steps 5 > $R/$OUT/rocprofj.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/profj $OUT/joint_kernel_stats.json $OUT/joint_kernel_stats.csv; cut -c1-150 $OUT/joint_kernel_stats.csv | head -30
fi
if has lintests; then
  timeout 900 python -m pytest tests/test_lin_gpu.py tests/test_joint_edges_gpu.py tests/test_joint_gpu.py -m gpu -q --maxfail=40 > $OUT/pytest_lin.log 2>&1; echo "lintests rc=$?"; tail -30 $OUT/pytest_lin.log
fi
if has fused; then
  for rep in 1 2; do for v in $VARS; do RNNT_LIBWARPRNNT=$(lib $v) timeout 300 python bench.py --fused-only 32,600,150,28 --steps 20 2>$OUT/fused_$v.err > $OUT/fused_$v.$rep.json; python - $OUT/fused_$v.$rep.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); f=d.get('fused_joint') or d
print('fused-only %-10s %.4f ms' % (sys.argv[2], f.get('ms_per_step', float('nan'))))
PY
  done; done
fi
if has profjv; then
  for v in $VARS; do
    (cd /tmp && RNNT_LIBWARPRNNT=$(lib $v) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profj_$v -o j -- python $R/bench.py --fused-only 32,600,150,28 --steps 10 > $R/$OUT/rocprofj_$v.log 2>&1)
    python scripts/summarize_trace.py stats $OUT/profj_$v $OUT/joint_kernel_stats_$v.json $OUT/joint_kernel_stats_$v.csv > /dev/null; echo "== kernels $v"; sed -n 2,5p $OUT/joint_kernel_stats_$v.csv | cut -c1-90
  done
fi
if has ab; then
  bash scripts/gpu_bench_variants.sh $TAG $VARS
fi
