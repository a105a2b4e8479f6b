#!/bin/sh
# MI355X counterpart of the reference's scripts/build_rnnt.sh: builds libwarprnnt.so for gfx950
# in-tree (no cmake, no install step; the Python package loads it from its own lib/ directory).
set -e
cd "$(dirname "$0")/.."
python "rnnt-speech-recognition_amd/build.py"
