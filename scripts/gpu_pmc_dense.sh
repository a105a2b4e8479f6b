#!/bin/bash
# PMC passes over the fused bench legs: HBM fetch / write bytes and L2 hit rates of the dense-layer and joint kernels
TAG=${1:-pmcd}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ragged --no-e2e --no-config5 > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1
  echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]; res=collections.defaultdict(dict)
for f in sorted(glob.glob(out+'/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'], r['Counter_Name']); agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for (kn,cn),v in agg.items():
        if 'rnnt::' in kn: res[kn.split('rnnt::')[1][:40]][cn]=v[1]/v[0]
json.dump(res, open(out+'/pmc_dense.json','w'), indent=1)
for k,v in sorted(res.items()):
    if any(s in k for s in ('dense_gemm','joint_fwd','joint_bwd','cell_tile','sweep')):
        print(k, {a: ('%.4g' % b) for a,b in v.items()})
PY
