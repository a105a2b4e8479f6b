#!/bin/bash
# Copy the summaries of a `scripts/gpu_r06.sh TAG "final"` pass from gpurun_out/TAG (scratch) into profiles/ (tracked).
# usage: scripts/collect_profiles.sh TAG [ROUND]      e.g. scripts/collect_profiles.sh r06final r06
set -e
SRC=gpurun_out/${1:-r06final}; R=${2:-r06}
for f in bench_n1.json kernel_stats.json kernel_stats.csv pmc_op.json fused_kernel_stats.json fused_kernel_stats.csv pmc_fused_valu.json \
         c5_n01_all_rows_kernel_stats.json c5_n01_all_rows_kernel_stats.csv c5_n01_kernel_stats.json c5_n01_kernel_stats.csv \
         c5_trained_kernel_stats.json c5_trained_kernel_stats.csv c5_pmc.txt pmc_op_config5.json fuzz.txt fuzz_joint16.txt; do
  [ -f $SRC/$f ] && cp $SRC/$f profiles/${R}_$f || echo "missing: $SRC/$f"
done
cp $SRC/kernel_stats_latest.json profiles/kernel_stats_latest.json
cp $SRC/pmc_latest.json profiles/pmc_latest.json
cp $SRC/pmc_fused_valu.json profiles/fused_valu_latest.json
grep -o '"csrc_sha16": "[0-9a-f]*"' profiles/kernel_stats_latest.json profiles/pmc_latest.json profiles/fused_valu_latest.json | sort | uniq -c
