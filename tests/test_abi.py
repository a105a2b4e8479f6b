"""CPU tests of the C-ABI boundary: the library builds/loads without a GPU, exports every symbol
include/rnnt.h declares, and rejects bad arguments before touching the device."""
import ctypes
import os
import re

import pytest

import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    pkg.build()
    return _lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rnnt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_lib.SYMBOLS)


def test_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), name
        assert ctypes.cast(getattr(lib, name), ctypes.c_void_p).value


def test_dynamic_symbol_table_is_the_abi_and_nothing_else(lib):
    """The library is built with -fvisibility=hidden: its dynamic symbol table holds the entry points of include/rnnt.h and,
    besides them, only what the HIP toolchain itself emits for device code -- the kernel handles (mangled, inside namespace
    rnnt) and the __hip_cuid_* translation-unit tags.  No host-side helper (validate, run_forward, rnnt::launch_* ...) is
    exported, so nothing of this library can be interposed in a TensorFlow / PyTorch process."""
    import shutil
    import subprocess

    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("binutils nm not available")
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    plain = sorted(n for n in names if not n.startswith("_Z") and not n.startswith("__hip_cuid_"))
    assert plain == sorted(_lib.SYMBOLS)
    for n in names:
        if n.startswith("_Z"):
            assert n.startswith("_ZN4rnnt") and "kernel" in n and "launch" not in n and "device_stub" not in n, n


def test_version_and_status_strings(lib):
    assert lib.get_warprnnt_version() >= 1
    assert _lib.status_string(0) == "no error"
    assert "invalid" in _lib.status_string(2)
    assert _lib.status_string(99) == "unknown error"


def test_workspace_size(lib):
    n = _lib.workspace_bytes(600, 150, 32)
    assert n % 256 == 0
    # lse + W (2 planes) + alpha + beta on the skewed grid: must at least hold 5 floats per cell
    assert n >= 32 * 600 * 150 * 4 * 5
    assert _lib.workspace_bytes(600, 150, 64) > n
    bad = ctypes.c_size_t(0)
    assert lib.get_workspace_size(0, 150, 32, True, ctypes.byref(bad)) == 2
    assert lib.get_workspace_size(600, 150, 32, False, ctypes.byref(bad)) == 2  # CPU location not provided
    assert lib.get_workspace_size(600, 150, 32, True, None) == 2


def test_options_struct_has_upstream_layout(lib):
    """rnntOptions is passed BY VALUE: its layout must be upstream's (bool batch_first = 1 byte at offset 28, 32 bytes in
    all), so that a caller compiled against upstream's rnnt.h can link against this library."""
    O = _lib.rnntOptions
    assert ctypes.sizeof(O) == 32
    assert (O.loc.offset, O.u.offset, O.blank_label.offset, O.maxT.offset, O.maxU.offset) == (0, 8, 16, 20, 24)
    assert (O.batch_first.offset, O.batch_first.size) == (28, 1)
    header = open(os.path.join(ROOT, "include", "rnnt.h")).read()
    assert "#include <stdbool.h>" in header and re.search(r"\bbool\s+batch_first\b", header)
    assert re.search(r"get_workspace_size\(int maxT, int maxU, int minibatch, bool gpu", header)


def test_padding_bytes_of_options_are_not_read(lib):
    """A caller built against upstream's header leaves the three bytes behind `bool batch_first` uninitialised."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("passes validation and would enqueue on the fake pointers")
    fake = ctypes.c_void_p(256)
    o = _lib.make_options(0, 0, 10, 5)
    raw = (ctypes.c_ubyte * 32).from_buffer(o)
    raw[29] = raw[30] = raw[31] = 0xFF
    ok_garbage = lib.compute_rnnt_loss(fake, None, fake, fake, fake, 28, 4, fake, fake, o)
    assert ok_garbage != 2  # passed validation (then failed for lack of a device)
    raw[28] = 0  # batch_first = false
    assert lib.compute_rnnt_loss(fake, None, fake, fake, fake, 28, 4, fake, fake, o) == 2


def test_argument_validation_needs_no_device(lib):
    o = _lib.make_options(0, 0, 10, 5)
    # null pointers
    assert lib.compute_rnnt_loss(None, None, None, None, None, 28, 4, None, None, o) == 2
    fake = ctypes.c_void_p(256)  # never dereferenced: rejected before any launch
    cpu = _lib.make_options(0, 0, 10, 5, loc=_lib.RNNT_CPU)
    assert lib.compute_rnnt_loss(fake, None, fake, fake, fake, 28, 4, fake, fake, cpu) == 2  # no CPU fallback
    big = _lib.make_options(0, 0, 10, 9000)
    assert lib.compute_rnnt_loss(fake, None, fake, fake, fake, 28, 4, fake, fake, big) == 2  # maxU > 8192
    n1, n2 = _lib.workspace_bytes(10, 1024, 2), _lib.workspace_bytes(10, 1100, 2)  # the wide-sweep layout (row stride U rounded to 64)
    assert n2 > n1
    blank_oob = _lib.make_options(0, 28, 10, 5)
    assert lib.compute_rnnt_loss(fake, None, fake, fake, fake, 28, 4, fake, fake, blank_oob) == 2
    assert lib.compute_rnnt_loss(fake, None, fake, fake, fake, 0, 4, fake, fake, o) == 2
    misaligned = ctypes.c_void_p(260)
    assert lib.compute_rnnt_loss(fake, None, fake, fake, fake, 28, 4, fake, misaligned, o) == 2
    # the build-only flags: anything but RNNT_VISIT_ALL is refused, for the op and -- OR-ed into joint_dtype -- for the fused joint
    assert _lib.RNNT_VISIT_ALL == 0x100 and re.search(r"#define\s+RNNT_VISIT_ALL\s+0x100", open(os.path.join(ROOT, "include", "rnnt.h")).read())
    assert lib.compute_rnnt_loss_flags(fake, fake, fake, fake, fake, None, 28, 4, fake, fake, o, 0x2) == 2
    assert lib.compute_rnnt_loss_flags(fake, fake, fake, fake, fake, None, 28, 4, fake, fake, o, 0x101) == 2
    jargs = [fake] * 7 + [None, 640, 28, 4] + [fake] * 5
    assert lib.compute_rnnt_joint_loss(*jargs, 0x200, fake, o) == 2   # unknown flag bit
    assert lib.compute_rnnt_joint_loss(*jargs, 0x102, fake, o) == 2   # unknown arithmetic type under the flag


def test_python_surface_fails_loudly_on_cpu_tensors(lib):
    import torch

    acts = torch.zeros(1, 2, 2, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        pkg.rnnt_loss(acts, torch.ones(1, 1, dtype=torch.int32), torch.tensor([2]), torch.tensor([1]))
    fn = pkg.get_loss_fn(2)
    with pytest.raises(RuntimeError):
        fn(torch.ones(1, 1), acts, torch.tensor([4]), torch.tensor([1]))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libwarprnnt.so"))
    with pytest.raises(_lib.RNNTLibraryError):
        _lib.load()


def test_reduced_lengths_matches_reference_ceil():
    import torch

    # utils/loss.py:31-33: ceil(spec_lengths / reduction_factor) as int32
    out = pkg.reduced_lengths(torch.tensor([1, 2, 3, 4, 599, 600]), 2)
    assert out.dtype == torch.int32
    assert out.tolist() == [1, 1, 2, 2, 300, 300]
