#!/bin/bash
# HBM traffic per kernel: separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1
  echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/*/pmc_*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:60], r['Counter_Name'])
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    print(f)
    for k,v in sorted(agg.items()):
        if 'rnnt' in k[0] or 'fill' in k[0]: print('  ',k, 'launches',v[0],'avg',v[1]/v[0])
PY
