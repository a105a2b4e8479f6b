#!/bin/bash
# Counter passes over the f16 fused joint at BASELINE config 5 (separate --pmc runs; no trace domains besides kernel-trace).
TAG=${1:-c5pmc}
SHAPE=${2:-16,1500,300,1024}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/bench.py --fused-only $SHAPE --steps 2 > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1
  echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for f in sorted(glob.glob(out+'/pmc_*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:48], r['Counter_Name'])
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(agg.items()):
        if 'jh_' in k[0]: print(k[0], k[1], 'launches',v[0],'avg %.4g' % (v[1]/v[0]))
PY
