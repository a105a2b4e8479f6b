#!/bin/bash
# usage: gpu_sweep_variants.sh TAG variant...  -> rocprofv3 kernel averages of the op-level bench for each library variant
# ("product" = the shipped library; others: rnnt-speech-recognition_amd/lib/libwarprnnt_<variant>.so from scripts/build_variant.sh)
export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  L=$R/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  [[ $v == product ]] && L=$R/rnnt-speech-recognition_amd/lib/libwarprnnt.so
  (cd /tmp && RNNT_LIBWARPRNNT=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof_$v -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-ragged --no-e2e --no-config5 > $R/gpurun_out/$TAG/rocprof_$v.log 2>&1)
  echo "== $v"; python scripts/summarize_trace.py stats gpurun_out/$TAG/prof_$v gpurun_out/$TAG/stats_$v.json gpurun_out/$TAG/stats_$v.csv | head -4
done
