#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench, rocprofv3 kernel stats.
# Everything lands in gpurun_out/<tag>/ ; each step has its own timeout so a hang cannot eat the box.
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -12 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== bench" ; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== bench sweep mode 0" ; RNNT_SWEEP_MODE=0 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_mode0.json 2>> $OUT/bench.err; cat $OUT/bench_mode0.json
echo "== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -12 $f; done
