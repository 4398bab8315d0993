"""Importable alias of the package directory `gaussian-splatting_amd/` (a hyphen cannot appear in
an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("gaussian-splatting_amd")
sys.modules[__name__] = _pkg
