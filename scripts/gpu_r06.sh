#!/bin/bash
# Round-6 GPU session driver (runs on the GPU box via gpurun).  usage: scripts/gpu_r06.sh TAG "steps..." [variants...]
#   f16       tests/test_joint_f16_gpu.py per library variant
#   c5        per-kernel times (rocprofv3 --stats) of the f16 joint at BASELINE config 5 per variant; LEG=n01|n01_all_rows|trained|trained_all_rows
#             picks the input / RNNT_VISIT_ALL leg (default: N(0,1) projections, every row visited)
#   mid       the same at B32 T600 U150 V128 / V256 and at the reference-default shape B16 T300 U100 V4096
#   sizes     tests/test_baseline_sizes_gpu.py (at-size parity)
#   test      full `pytest -m gpu`
#   bench     python bench.py (default line)
TAG=${1:-r06}; WHAT=${2:-"f16 c5"}; shift 2
VARS=${@:-product}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
lib() { local v=$1; [[ $v == product ]] && echo $R/rnnt-speech-recognition_amd/lib/libwarprnnt.so || echo $R/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so; }
if has f16; then
  for v in $VARS; do echo "== f16 tests $v"; RNNT_LIBWARPRNNT=$(lib $v) timeout 900 python -m pytest tests/test_joint_f16_gpu.py -m gpu -q --maxfail=10 2>&1 | tee $OUT/pytest_f16_$v.log | tail -15; done
fi
shape_prof() {  # name shape
  for v in $VARS; do
    (cd /tmp && RNNT_LIBWARPRNNT=$(lib $v) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$v -o b -- python $R/bench.py --fused-only $2 --steps 3 --fused-leg ${LEG:-n01_all_rows} > /tmp/log_$1_$v 2>/tmp/err_$1_$v)
    python - /tmp/prof_$1_$v $v /tmp/log_$1_$v $1 <<'PY' | tee -a $OUT/shapes.txt
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
t=[]
if f:
  for r in csv.DictReader(open(f[0])):
    n=r['Name'].replace('void ','').replace('(rnnt::JhParams)','')
    if n.startswith('rnnt::') and float(r['AverageNs'])>5e4: t.append((n.replace('rnnt::','').replace('_kernel',''), float(r['AverageNs'])/1e6))
try: ms=json.loads(open(sys.argv[3]).readline())['fused_joint']['ms_per_step']
except Exception as e: ms=float('nan')
print('%-8s %-10s step %.3f ms (profiled run)  %s' % (sys.argv[4], sys.argv[2], ms, '  '.join('%s %.3f' % kv for kv in sorted(t, key=lambda kv:-kv[1])[:8])))
PY
  done
}
if has c5; then shape_prof c5 16,1500,300,1024; fi
if has mid; then shape_prof v128 32,600,150,128; shape_prof v256 32,600,150,256; shape_prof v4096 16,300,100,4096; fi
if has sizes; then
  timeout 1500 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q --maxfail=10 --durations=8 > $OUT/pytest_sizes.log 2>&1; echo "sizes rc=$?"; tail -25 $OUT/pytest_sizes.log
fi
if has test; then
  timeout 3000 python -m pytest tests -m gpu -q --maxfail=40 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -40 $OUT/pytest_gpu.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
fi
# ---- the profiles the round commits (profiles/r06_*), one pass: scripts/gpu_r06.sh r06final "final"
if has final; then
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
  echo "== bench (default line)"; timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err; echo "bench rc=$?"
  echo "== kernel trace of the op-level bench (full-length launches only)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/prof $OUT/kernel_stats.json $OUT/kernel_stats.csv; cp $OUT/kernel_stats.json $OUT/kernel_stats_latest.json
  echo "== PMC, op-level path"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc/$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > $R/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
  done
  python scripts/summarize_trace.py pmc $OUT/pmc $OUT/pmc_op.json; python scripts/summarize_trace.py latest $OUT/pmc_op.json $OUT/pmc_latest.json 645120000
  echo "== kernel trace of the bench WITH the fused legs"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/proff -o f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-e2e --no-config5 > $R/$OUT/rocproff.log 2>&1); echo "rocprof rc=$?"
  python scripts/summarize_trace.py stats $OUT/proff $OUT/fused_kernel_stats.json $OUT/fused_kernel_stats.csv
  echo "== PMC, fused f32-grade joint at C2 (VALU / MFMA instruction counts)"
  for c in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_')
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmcv/$n -o pmc -- python $R/bench.py --fused-only 32,600,150,28 --fused-leg n01_all_rows --steps 3 > $R/$OUT/pmcv_$n.log 2>&1); echo "pmcv $n rc=$?"
  done
  python scripts/summarize_trace.py pmc $OUT/pmcv $OUT/pmc_fused_valu.json > /dev/null
  echo "== config 5: kernel traces of the three legs, counters of the all-rows leg"
  for leg in n01_all_rows n01 trained; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profc5_$leg -o c5 -- python $R/bench.py --fused-only 16,1500,300,1024 --fused-leg $leg --steps 3 > $R/$OUT/rocprofc5_$leg.log 2>&1); echo "rocprof c5 $leg rc=$?"
    python scripts/summarize_trace.py stats $OUT/profc5_$leg $OUT/c5_${leg}_kernel_stats.json $OUT/c5_${leg}_kernel_stats.csv
  done
  (cd /tmp; for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/c5pmc/pmc_$n -o pmc -- python $R/bench.py --fused-only 16,1500,300,1024 --fused-leg n01_all_rows --steps 2 > $R/$OUT/c5pmc_$n.log 2>&1; echo "c5 pmc $c rc=$?"
  done)
  python - $OUT/c5pmc > $OUT/c5_pmc.txt <<'PY'
import csv,glob,collections,sys
for f in sorted(glob.glob(sys.argv[1]+'/pmc_*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:48], r['Counter_Name'])
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(agg.items()):
        if 'jh_' in k[0]: print(k[0], k[1], 'launches',v[0],'avg %.4g' % (v[1]/v[0]))
PY
  cat $OUT/c5_pmc.txt | head -60
  echo "== PMC of the op at configs[4]'s shape (default call and RNNT_VISIT_ALL)"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmcop5/$c -o pmc -- python $R/scripts/probes/op_c5_probe.py > $R/$OUT/pmcop5_$c.log 2>&1); echo "pmc op5 $c rc=$?"
  done
  python scripts/summarize_trace.py pmc $OUT/pmcop5 $OUT/pmc_op_config5.json > /dev/null
  echo "== randomised parity sweep of every path through the C ABI (tests/tools/fuzz_parity.py), on the tree the profiles above are of"
  timeout 1200 python tests/tools/fuzz_parity.py 606 2400 > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -8 $OUT/fuzz.txt
  timeout 900 python tests/tools/fuzz_parity.py 616 600 joint16 > $OUT/fuzz_joint16.txt 2>&1; echo "fuzz joint16 rc=$?"; tail -6 $OUT/fuzz_joint16.txt
fi
