"""Import shim: exposes the package directory ``circuitscape.jl_amd/`` (not a valid Python identifier because of
the dot) as the module ``circuitscape_jl_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "circuitscape.jl_amd")
_spec = importlib.util.spec_from_file_location("circuitscape_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["circuitscape_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
