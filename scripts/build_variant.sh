#!/bin/bash
# Dev tool: build an experimental variant of libwarprnnt.so with extra -D flags, e.g.
#   scripts/build_variant.sh trace -DJH_TRACE      -> rnnt-speech-recognition_amd/lib/libwarprnnt_trace.so
# and run anything with RNNT_LIBWARPRNNT=<that path> to load it instead of the product library.
set -e
NAME=$1; shift
D=$(cd "$(dirname "$0")/.." && pwd)/rnnt-speech-recognition_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-inline-asm "$@" $D/csrc/rnnt_kernels.hip $D/csrc/rnnt_lin_kernels.hip $D/csrc/joint_kernels.hip $D/csrc/joint_f16_kernels.hip $D/csrc/dense_kernels.hip $D/csrc/rnnt_entrypoint.hip -o $D/lib/libwarprnnt_$NAME.so
echo $D/lib/libwarprnnt_$NAME.so
