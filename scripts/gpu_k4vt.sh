#!/bin/bash
# Dev: K4 with 256-column V tiles (two workgroups per CU) against the 512-column product: parity of the variant, then step times
L=$GRAFT_REPO_ROOT/rnnt-speech-recognition_amd/lib
RNNT_LIBWARPRNNT=$L/libwarprnnt_vt256.so timeout 900 python -m pytest tests/test_joint_f16_gpu.py -x -q -m gpu 2>&1 | tail -3
for lib in libwarprnnt.so libwarprnnt_vt256.so; do
  echo "== $lib"
  RNNT_LIBWARPRNNT=$L/$lib timeout 300 python scripts/probes/midvocab_probe.py 2>&1 | tail -5
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kv; ( cd $GRAFT_REPO_ROOT && RNNT_LIBWARPRNNT=$L/libwarprnnt_vt256.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kv -o kv --output-format csv -- python bench.py --fused-only 16,1500,300,1024 --steps 2 > /dev/null 2>&1 )
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kv/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:6]:
    print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4}  {r['Name'][:70]}")
PY
