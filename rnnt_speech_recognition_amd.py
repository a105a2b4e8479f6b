"""Import shim: the package directory is named ``rnnt-speech-recognition_amd`` (not a valid Python
identifier), so ``import rnnt_speech_recognition_amd`` lands here and is redirected to that
directory, which then behaves as a normal package (sub-modules import as
``rnnt_speech_recognition_amd.loss`` etc.)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rnnt-speech-recognition_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
