#!/bin/bash
# Dev tool: build an experimental variant of libwarprnnt.so with extra -D flags, e.g.
#   scripts/build_variant.sh trace -DJH_TRACE      -> rnnt-speech-recognition_amd/lib/libwarprnnt_trace.so
# and run anything with RNNT_LIBWARPRNNT=<that path> to load it instead of the product library.
# (Same per-source flags as rnnt-speech-recognition_amd/build.py.)
set -e
NAME=$1; shift
D=$(cd "$(dirname "$0")/.." && pwd)/rnnt-speech-recognition_amd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-inline-asm"
T=$(mktemp -d)
for s in rnnt_kernels rnnt_lin_kernels joint_kernels joint_f16_kernels dense_kernels rnnt_entrypoint; do
  X="-fno-slp-vectorize"; [[ $s == rnnt_kernels || $s == rnnt_entrypoint ]] && X=""
  /opt/rocm/bin/hipcc $F $X "$@" -c $D/csrc/$s.hip -o $T/$s.o &
done
wait
/opt/rocm/bin/hipcc $F -shared $T/*.o -o $D/lib/libwarprnnt_$NAME.so
rm -rf $T
echo $D/lib/libwarprnnt_$NAME.so
