#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench, rocprofv3 kernel stats, PMC passes.
# Everything lands in gpurun_out/<tag>/ ; each step has its own timeout so a hang cannot eat the box.
#   scripts/gpu_check.sh TAG "smoke test bench prof pmc pmcj c5"
TAG=${1:-r02}
WHAT=${2:-"smoke test bench prof"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has smoke; then
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
fi
if has test; then
echo "== pytest -m gpu" ; timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
fi
if has bench; then
echo "== bench" ; timeout 900 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
fi
if has prof; then
echo "== rocprof kernel trace (full-length launches only)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-ragged > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
python scripts/summarize_trace.py stats $OUT/prof $OUT/kernel_stats.json $OUT/kernel_stats.csv
fi
if has profj; then
echo "== rocprof kernel trace, fused f32-grade joint at C2"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profj -o j -- python $R/bench.py --fused-only 32,600,150,28 --steps 5 > $R/$OUT/rocprofj.log 2>&1); echo "rocprof rc=$?"
python scripts/summarize_trace.py stats $OUT/profj $OUT/joint_kernel_stats.json $OUT/joint_kernel_stats.csv
fi
if has pmc; then
echo "== PMC, op-level path (separate passes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc/$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused --no-ragged > $R/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
done
python scripts/summarize_trace.py pmc $OUT/pmc $OUT/pmc_op.json
fi
if has pmcj; then
echo "== PMC, fused f32-grade joint at C2 (joint_phase1s / dl / phase2s)"
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmcj/$n -o pmc -- python $R/bench.py --fused-only 32,600,150,28 --steps 3 > $R/$OUT/pmcj_$n.log 2>&1); echo "pmcj $n rc=$?"
done
python scripts/summarize_trace.py pmc $OUT/pmcj $OUT/pmc_joint.json
fi
if has c5; then
echo "== config 5 fused f16 joint: time + kernel trace"
timeout 600 python bench.py --fused-only 16,1500,300,1024 --steps 3 > $OUT/c5.json 2> $OUT/c5.err; echo "c5 rc=$?"; cat $OUT/c5.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profc5 -o c5 -- python $R/bench.py --fused-only 16,1500,300,1024 --steps 3 > $R/$OUT/rocprofc5.log 2>&1); echo "rocprof rc=$?"
python scripts/summarize_trace.py stats $OUT/profc5 $OUT/c5_kernel_stats.json $OUT/c5_kernel_stats.csv
fi
