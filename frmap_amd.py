"""Import alias for the product package.

The package directory is named ``facerecognition-multiarchitecture-pipeline_amd`` (the name the
build contract fixes); hyphens are not legal in a Python module name, so this loader registers that
directory under the importable name ``frmap_amd``.  ``import frmap_amd`` from the repository root
(or with the root on ``sys.path``) therefore yields the real package, sub-modules included
(``import frmap_amd.face_models`` works).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                         "facerecognition-multiarchitecture-pipeline_amd")

_spec = _ilu.spec_from_file_location(
    "frmap_amd", _os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["frmap_amd"] = _mod
_spec.loader.exec_module(_mod)
