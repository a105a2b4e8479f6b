"""Build driver for libwarprnnt.so (the MI355X counterpart of the reference's
scripts/build_rnnt.sh:1-13, which runs cmake+make on warp-transducer and installs the binding).

hipcc cross-compiles for gfx950 without a GPU.  The library is built IN-TREE
(`rnnt-speech-recognition_amd/lib/libwarprnnt.so`) so that it travels with the source snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwarprnnt.so")
SOURCES = ["rnnt_kernels.hip", "rnnt_lin_kernels.hip", "joint_kernels.hip", "joint_f16_kernels.hip", "dense_kernels.hip", "rnnt_entrypoint.hip"]
# -fvisibility=hidden: the library exports exactly the entry points include/rnnt.h marks RNNT_API (tests/test_abi.py)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-inline-asm"]
# per-source extras.  -fno-slp-vectorize: no packed-f32 instructions (v_pk_fma_f32 ...) from the compiler -- in the linear sweeps
# they cost more register moves than they save (rnnt_lin_kernels.hip lin_alpha_step); in the MFMA kernels a packed-f32
# instruction does not overlap with the matrix pipe (scripts/probes/probe_pk.hip; fused step -0.6 %, config 5 -1.2 %).  The
# HBM-bound cell kernels of rnnt_kernels.hip keep the vectoriser (the op at config 5's shape: 16.4 against 17.0 ms).
_NO_SLP = ["-fno-slp-vectorize"]
EXTRA_FLAGS = {"rnnt_lin_kernels.hip": _NO_SLP, "joint_kernels.hip": _NO_SLP, "joint_f16_kernels.hip": _NO_SLP, "dense_kernels.hip": _NO_SLP}


def _deps():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "rnnt.h"))
    return files


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _deps())


def _compile_one(args):
    hipcc, src, obj, verbose = args
    cmd = [hipcc] + [f for f in HIPCC_FLAGS if f != "-shared"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source (one hipcc per source, in parallel) and link lib/libwarprnnt.so; returns the path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libwarprnnt.so (ROCm toolchain required)")
    os.makedirs(LIB_DIR, exist_ok=True)
    tag = f".tmp{os.getpid()}"  # several ranks may arrive here at once
    jobs = [(hipcc, os.path.join(CSRC, s), os.path.join(LIB_DIR, s[:-4] + tag + ".o"), verbose) for s in SOURCES]
    objs = []
    try:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            objs = list(ex.map(_compile_one, jobs))
        tmp = LIB_PATH + tag
        cmd = [hipcc] + HIPCC_FLAGS + objs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB_PATH)
    finally:
        for _, _, obj, _ in jobs:
            if os.path.exists(obj):
                os.remove(obj)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
