#!/bin/bash
# SQ counter passes over the fused f32-grade joint at C2 (separate --pmc runs, kernel-trace only): where do the waves' cycles go?
# usage (on the GPU box): scripts/gpu_pmc_sq.sh TAG [B,T,U,V] [kernel-name filter]
TAG=${1:-pmcsq}; SHAPE=${2:-32,600,150,28}; FILT=${3:-joint_}; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
         "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH" "SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/$n -o pmc -- python $R/bench.py --fused-only $SHAPE --steps 3 > $R/$OUT/log_$n.txt 2>&1); echo "pmc $n rc=$?"
done
python scripts/summarize_trace.py pmc $OUT $OUT/pmc_sq.json
python - $OUT/pmc_sq.json $FILT <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in d:
    if sys.argv[2] in k: print(k, json.dumps(d[k]))
PY
