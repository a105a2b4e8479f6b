#!/bin/bash
# f16 fused joint at BASELINE config 5: parity tests, bench, rocprof kernel stats, optional counter passes.
TAG=${1:-c5}
WHAT=${2:-test,bench,prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [[ $WHAT == *test* ]]; then timeout 300 python -m pytest tests/test_joint_f16_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log; fi
if [[ $WHAT == *bench* ]]; then timeout 300 python bench.py --fused-only 16,1500,300,1024 --steps 3 > $OUT/c5.json 2> $OUT/c5.err; python -c "
import json; d=json.load(open('$OUT/c5.json'))['fused_joint']; print('C5 fused f16:', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['achieved'],1), 'TFLOP/s')"; tail -2 $OUT/c5.err; fi
if [[ $WHAT == *prof* ]]; then (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --fused-only 16,1500,300,1024 --steps 2 > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -8 $f | cut -d, -f1-4 | cut -c1-120; done; fi
if [[ $WHAT == *pmc* ]]; then bash scripts/gpu_pmc_c5.sh $TAG; fi
