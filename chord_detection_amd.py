"""Import shim: the package lives in ``chord-detection_amd/`` (a directory name
Python cannot import directly); this module loads it under the importable name
``chord_detection_amd``.  Use ``import chord_detection_amd as chord_detection``
for a drop-in of the reference's package."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "chord-detection_amd")
_spec = importlib.util.spec_from_file_location(
    "chord_detection_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["chord_detection_amd"] = _mod
_spec.loader.exec_module(_mod)
