"""Importable alias of the `multi-modal-emotion_amd/` package (hyphenated directory names cannot be imported).

`import tav_amd` gives the package; submodules resolve as `tav_amd.ops`, `tav_amd.models.tav`, ...
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-modal-emotion_amd")
_spec = importlib.util.spec_from_file_location(
    "tav_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tav_amd"] = _mod
_spec.loader.exec_module(_mod)
