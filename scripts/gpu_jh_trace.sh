#!/bin/bash
# usage: gpu_jh_trace.sh TAG variant...  -> K1 / K3 / K4 timelines (s_memtime stamps) of -DJH_TRACE library variants at config 5
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib/libwarprnnt_$v.so
  JH_TRACE_FILE=/tmp/jh_$v.bin RNNT_LIBWARPRNNT=$L timeout 300 python bench.py --fused-only 16,1500,300,1024 --steps 1 > /tmp/log_$v 2>&1
  echo "== $v"; python scripts/parse_jh_trace.py /tmp/jh_$v.bin | tee gpurun_out/$TAG/trace_$v.txt
done
