"""ctypes wrapper around oracle/cpu_rnnt.c (TEST INFRASTRUCTURE ONLY; see that file's header).

`build()` compiles it with the reference's flags (-O2 -fopenmp); `rnnt_cpu()` runs the
restated reference CPU transducer path on host NumPy arrays (input = log-probs, gradient
w.r.t. log-probs)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_rnnt.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cpu_rnnt.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_rnnt_cpu.restype = ctypes.c_int
        _lib.oracle_rnnt_cpu.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
        _lib.oracle_rnnt_cpu_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(_load().oracle_rnnt_cpu_max_threads())


def rnnt_cpu(log_probs, labels, input_lengths, label_lengths, blank=0, num_threads=0, want_grad=True):
    """log_probs f32 [B,T,U,V] -> (costs f32 [B], grads f32 [B,T,U,V] w.r.t. log-probs or None)."""
    lib = _load()
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    B, T, U, V = lp.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(B, max(U - 1, 0))
    il = np.ascontiguousarray(input_lengths, dtype=np.int32)
    ll = np.ascontiguousarray(label_lengths, dtype=np.int32)
    costs = np.zeros(B, dtype=np.float32)
    grads = np.empty_like(lp) if want_grad else None
    if lab.size == 0:
        lab = np.zeros((B, 1), dtype=np.int32)
    rc = lib.oracle_rnnt_cpu(
        lp.ctypes.data, grads.ctypes.data if want_grad else None, lab.ctypes.data, ll.ctypes.data,
        il.ctypes.data, V, B, T, U, blank, num_threads, costs.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_rnnt_cpu failed with status {rc}")
    return costs, grads
