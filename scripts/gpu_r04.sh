#!/bin/bash
# Round-4 GPU pass (via gpurun): scripts/gpu_r04.sh TAG "steps..."   steps: test peaky bench prof profj pmc c5
# Everything lands in gpurun_out/TAG/; every step has its own timeout.
TAG=${1:-r04a}
WHAT=${2:-"test bench prof"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has smoke; then
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
fi
if has loss; then
  echo "== loss-op parity (tests/test_loss_gpu.py + peaked logits)"
  timeout 1200 python -m pytest tests/test_loss_gpu.py tests/test_lin_gpu.py tests/test_peaky_gpu.py -m gpu -q --durations=5 ${PYTEST_ARGS} > $OUT/pytest_loss.log 2>&1; echo "loss rc=$?"
  tail -25 $OUT/pytest_loss.log
fi
if has test; then
  echo "== pytest -m gpu (without the peaked-logit file)"
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_peaky_gpu.py --deselect tests/test_peaky_wide_gpu.py ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -25 $OUT/pytest_gpu.log
fi
if has peaky; then
  echo "== peaked-logit parity"
  timeout 900 python -m pytest tests/test_peaky_gpu.py -m gpu -q > $OUT/pytest_peaky.log 2>&1; echo "peaky rc=$?"
  tail -15 $OUT/pytest_peaky.log; cp gpurun_out/r04_accuracy*.json $OUT/ 2>/dev/null; cat $OUT/r04_accuracy.json
fi
if has wide; then
  echo "== peaked logits on wide lattices + the op at configs[4]'s shape at full size"
  timeout 1500 python -m pytest tests/test_peaky_wide_gpu.py -m gpu -q --durations=5 > $OUT/pytest_wide.log 2>&1; echo "wide rc=$?"
  tail -25 $OUT/pytest_wide.log; cp gpurun_out/r04_accuracy_wide.json $OUT/ 2>/dev/null; cat $OUT/r04_accuracy_wide.json
fi
if has bench; then
  echo "== bench"; timeout 900 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  cat $OUT/bench.json; tail -3 $OUT/bench.err
fi
if has prof; then
  echo "== rocprof kernel trace of the op-level bench (full-length launches only)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/prof $OUT/kernel_stats.json $OUT/kernel_stats.csv
  cp $OUT/kernel_stats.json $OUT/kernel_stats_latest.json
fi
if has profj; then
  echo "== rocprof kernel trace, fused f32-grade joint at C2"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profj -o j -- python $R/bench.py --fused-only 32,600,150,28 --steps 5 > $R/$OUT/rocprofj.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/profj $OUT/joint_kernel_stats.json $OUT/joint_kernel_stats.csv
fi
if has proff; then
  echo "== rocprof kernel trace of the bench WITH the fused legs (dense + joint kernels)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/proff -o f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-e2e --no-config5 > $R/$OUT/rocproff.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/proff $OUT/fused_kernel_stats.json $OUT/fused_kernel_stats.csv
fi
if has dense; then
  echo "== dense-layer tests"
  timeout 600 python -m pytest tests/test_dense_gpu.py -m gpu -q -x > $OUT/pytest_dense.log 2>&1; echo "dense rc=$?"; tail -15 $OUT/pytest_dense.log
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc/$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > $R/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
  done
  python scripts/summarize_trace.py pmc $OUT/pmc $OUT/pmc_op.json
  python scripts/summarize_trace.py latest $OUT/pmc_op.json $OUT/pmc_latest.json 645120000
fi
if has pmcv; then
  echo "== PMC, fused f32-grade joint at C2: VALU / MFMA instruction counts (the VALU ceiling of bench.py's fused legs)"
  for c in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_')
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmcv/$n -o pmc -- python $R/bench.py --fused-only 32,600,150,28 --steps 3 > $R/$OUT/pmcv_$n.log 2>&1); echo "pmcv $n rc=$?"
  done
  python scripts/summarize_trace.py pmc $OUT/pmcv $OUT/pmc_fused_valu.json | grep -i "joint_fwd\|joint_bwd\|cellrec" 
fi
if has c5; then
  echo "== config 5 fused f16 joint: time + kernel trace"
  timeout 600 python bench.py --fused-only 16,1500,300,1024 --steps 3 > $OUT/c5.json 2> $OUT/c5.err; echo "c5 rc=$?"; cat $OUT/c5.json
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profc5 -o c5 -- python $R/bench.py --fused-only 16,1500,300,1024 --steps 3 > $R/$OUT/rocprofc5.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/profc5 $OUT/c5_kernel_stats.json $OUT/c5_kernel_stats.csv
fi
