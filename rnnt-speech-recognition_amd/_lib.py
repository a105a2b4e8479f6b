"""ctypes binding of include/rnnt.h.  There is NO fallback: if the HIP library is missing or
fails to load, every entry point raises (the reference silently returns its logits instead,
utils/loss.py:14-22 -- deliberately not reproduced)."""
from __future__ import annotations

import ctypes
import os

from .build import LIB_PATH as _DEFAULT_LIB_PATH

# dev knob: load an experimental build of the library instead (scripts/build_variant.sh)
LIB_PATH = os.environ.get("RNNT_LIBWARPRNNT", _DEFAULT_LIB_PATH)

RNNT_CPU, RNNT_GPU = 0, 1
STATUS_SUCCESS = 0

# every symbol include/rnnt.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "get_warprnnt_version",
    "rnntGetStatusString",
    "get_workspace_size",
    "compute_rnnt_loss",
    "compute_rnnt_loss_fwd",
    "compute_rnnt_loss_bwd",
    "compute_rnnt_loss_ex",
    "compute_rnnt_loss_flags",
    "get_joint_workspace_size",
    "compute_rnnt_joint_loss",
    "compute_rnnt_joint_loss_fwd",
    "compute_rnnt_joint_loss_bwd",
    "compute_rnnt_joint_logits",
    "compute_rnnt_joint_net_logits",
    "get_rnnt_joint_backward_rows",
    "get_joint_net_workspace_size",
    "compute_rnnt_joint_net_loss",
    "compute_rnnt_joint_net_loss_fwd",
    "compute_rnnt_joint_net_loss_bwd",
]


class _LocUnion(ctypes.Union):
    _fields_ = [("num_threads", ctypes.c_uint), ("stream", ctypes.c_void_p)]


class rnntOptions(ctypes.Structure):
    _anonymous_ = ("u",)
    _fields_ = [
        ("loc", ctypes.c_int),
        ("u", _LocUnion),
        ("blank_label", ctypes.c_int),
        ("maxT", ctypes.c_int),
        ("maxU", ctypes.c_int),
        ("batch_first", ctypes.c_bool),
    ]


_lib = None


class RNNTLibraryError(RuntimeError):
    pass


RNNT_VISIT_ALL = 0x100  # include/rnnt.h: no occupancy floor -- the gradient kernels visit every lattice cell / row


def load():
    """Load libwarprnnt.so (once).  Raises RNNTLibraryError loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RNNTLibraryError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run scripts/build_rnnt.sh "
            "(or __graft_entry__.build()). There is no CPU/eager fallback for the transducer loss."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the ROCm runtime being present
        raise RNNTLibraryError(f"failed to load {LIB_PATH}: {e}") from e
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.get_warprnnt_version.restype = ci
    lib.rnntGetStatusString.restype = ctypes.c_char_p
    lib.rnntGetStatusString.argtypes = [ci]
    lib.get_workspace_size.restype = ci
    lib.get_workspace_size.argtypes = [ci, ci, ci, ctypes.c_bool, ctypes.POINTER(ctypes.c_size_t)]
    lib.compute_rnnt_loss.restype = ci
    lib.compute_rnnt_loss.argtypes = [vp, vp, vp, vp, vp, ci, ci, vp, vp, rnntOptions]
    lib.compute_rnnt_loss_fwd.restype = ci
    lib.compute_rnnt_loss_fwd.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, rnntOptions]
    lib.compute_rnnt_loss_bwd.restype = ci
    lib.compute_rnnt_loss_bwd.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, vp, rnntOptions]
    lib.compute_rnnt_loss_ex.restype = ci
    lib.compute_rnnt_loss_ex.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, rnntOptions]
    if LIB_PATH == _DEFAULT_LIB_PATH or hasattr(lib, "compute_rnnt_loss_flags"):
        lib.compute_rnnt_loss_flags.restype = ci
        lib.compute_rnnt_loss_flags.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, rnntOptions, ctypes.c_uint]
    lib.get_joint_workspace_size.restype = ci
    lib.get_joint_workspace_size.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_size_t)]
    lib.compute_rnnt_joint_loss.restype = ci
    lib.compute_rnnt_joint_loss.argtypes = [vp] * 8 + [ci, ci, ci] + [vp] * 5 + [ci, vp, rnntOptions]
    lib.compute_rnnt_joint_loss_fwd.restype = ci
    lib.compute_rnnt_joint_loss_fwd.argtypes = [vp] * 7 + [ci, ci, ci, vp, ci, vp, rnntOptions]
    lib.compute_rnnt_joint_loss_bwd.restype = ci
    lib.compute_rnnt_joint_loss_bwd.argtypes = [vp] * 8 + [ci, ci, ci] + [vp] * 4 + [ci, vp, rnntOptions]
    if LIB_PATH == _DEFAULT_LIB_PATH or hasattr(lib, "compute_rnnt_joint_logits"):  # (an older dev variant may lack it)
        lib.compute_rnnt_joint_logits.restype = ci
        lib.compute_rnnt_joint_logits.argtypes = [vp] * 4 + [ci, ci, ci, vp, ci, vp, rnntOptions]
    if LIB_PATH == _DEFAULT_LIB_PATH or hasattr(lib, "get_rnnt_joint_backward_rows"):
        lib.get_rnnt_joint_backward_rows.restype = ci
        lib.get_rnnt_joint_backward_rows.argtypes = [vp, ci, ci, ci, rnntOptions, ctypes.POINTER(ctypes.c_int)]
    if LIB_PATH == _DEFAULT_LIB_PATH or hasattr(lib, "compute_rnnt_joint_net_logits"):
        lib.compute_rnnt_joint_net_logits.restype = ci
        lib.compute_rnnt_joint_net_logits.argtypes = [vp] * 6 + [ci] * 4 + [vp, ci, vp, rnntOptions]
    if LIB_PATH == _DEFAULT_LIB_PATH or hasattr(lib, "compute_rnnt_joint_net_loss"):
        lib.get_joint_net_workspace_size.restype = ci
        lib.get_joint_net_workspace_size.argtypes = [ci] * 6 + [ctypes.POINTER(ctypes.c_size_t)]
        lib.compute_rnnt_joint_net_loss.restype = ci
        lib.compute_rnnt_joint_net_loss.argtypes = [vp] * 10 + [ci] * 4 + [vp] * 7 + [ci, vp, rnntOptions]
        lib.compute_rnnt_joint_net_loss_fwd.restype = ci
        lib.compute_rnnt_joint_net_loss_fwd.argtypes = [vp] * 9 + [ci] * 4 + [vp, ci, vp, rnntOptions]
        lib.compute_rnnt_joint_net_loss_bwd.restype = ci
        lib.compute_rnnt_joint_net_loss_bwd.argtypes = [vp] * 10 + [ci] * 4 + [vp] * 6 + [ci, vp, rnntOptions]
    _lib = lib
    return lib


def status_string(status: int) -> str:
    return load().rnntGetStatusString(status).decode()


def check(status: int, what: str):
    if status != STATUS_SUCCESS:
        raise RuntimeError(f"{what} failed: rnntStatus_t={status} ({status_string(status)})")


def make_options(stream: int, blank: int, maxT: int, maxU: int, loc: int = RNNT_GPU) -> rnntOptions:
    o = rnntOptions()
    o.loc = loc
    o.stream = stream
    o.blank_label = blank
    o.maxT = maxT
    o.maxU = maxU
    o.batch_first = True
    return o


def joint_workspace_bytes(maxT: int, maxU: int, minibatch: int, joint_size: int, alphabet_size: int) -> int:
    n = ctypes.c_size_t(0)
    check(load().get_joint_workspace_size(maxT, maxU, minibatch, joint_size, alphabet_size, ctypes.byref(n)),
          "get_joint_workspace_size")
    return int(n.value)


def joint_net_workspace_bytes(maxT: int, maxU: int, minibatch: int, hidden_size: int, joint_size: int, alphabet_size: int) -> int:
    n = ctypes.c_size_t(0)
    check(load().get_joint_net_workspace_size(maxT, maxU, minibatch, hidden_size, joint_size, alphabet_size, ctypes.byref(n)),
          "get_joint_net_workspace_size")
    return int(n.value)


def workspace_bytes(maxT: int, maxU: int, minibatch: int) -> int:
    n = ctypes.c_size_t(0)
    check(load().get_workspace_size(maxT, maxU, minibatch, True, ctypes.byref(n)), "get_workspace_size")
    return int(n.value)
