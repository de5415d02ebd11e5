"""Loader: makes the hyphen-named package directory importable as ``mpcqp``."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd")
_spec = _u.spec_from_file_location("mpcqp", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["mpcqp"] = _mod
_spec.loader.exec_module(_mod)
