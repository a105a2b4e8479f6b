#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench, rocprofv3 kernel stats.
# Everything lands in gpurun_out/<tag>/ ; each step has its own timeout so a hang cannot eat the box.
TAG=${1:-r01}
WHAT=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [[ $WHAT == all || $WHAT == *smoke* ]]; then
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
fi
if [[ $WHAT == all || $WHAT == *test* ]]; then
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
echo "== bench" ; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
for g in 1 2 8; do echo "== bench RNNT_GROUPS=$g"; RNNT_GROUPS=$g timeout 300 python bench.py --no-cpu-baseline 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['whole_op'])" | tee $OUT/bench_groups$g.txt; done
echo "== bench flat path"; RNNT_CELL_PATH=flat timeout 300 python bench.py --no-cpu-baseline 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])" | tee $OUT/bench_flat.txt
fi
if [[ $WHAT == all || $WHAT == *prof* ]]; then
echo "== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -12 $f; done
fi
