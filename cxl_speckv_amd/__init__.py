"""Importable alias of the ``cxl-speckv_amd/`` package directory.

The package directory carries the project's name (with a hyphen), which Python
cannot import directly; this stub makes ``import cxl_speckv_amd`` resolve to it.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cxl-speckv_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
