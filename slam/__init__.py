"""Import shim: makes ``import slam.net_amd`` resolve to the on-disk package directory
``slam.net_amd/`` (the directory name contains a dot, so the normal finder cannot see it)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "slam.net_amd")
if "slam.net_amd" not in _sys.modules:
    _spec = _ilu.spec_from_file_location("slam.net_amd", _os.path.join(_dir, "__init__.py"),
                                         submodule_search_locations=[_dir])
    _mod = _ilu.module_from_spec(_spec)
    _sys.modules["slam.net_amd"] = _mod
    _spec.loader.exec_module(_mod)
net_amd = _sys.modules["slam.net_amd"]
