"""The C-ABI shared library loads and exports every symbol include/csgpu.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "circuitscape.jl_amd", "libcsgpu.so")


def declared_symbols():
    with open(os.path.join(ROOT, "include", "csgpu.h")) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(csgpu_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ["csgpu_setup", "csgpu_solve_pairs", "csgpu_solve_rhs", "csgpu_free", "csgpu_last_error",
              "csgpu_raster_setup", "csgpu_get_info", "csgpu_device_count", "csgpu_default_opts"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import subprocess, sys
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    L = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(L, s), s
    L.csgpu_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.csgpu_version()


def test_binding_struct_sizes_match_library():
    """csgpu_default_opts writes struct_size = sizeof(csgpu_opts): the ctypes mirror must agree."""
    import sys
    sys.path.insert(0, ROOT)
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    L = ctypes.CDLL(LIB)
    o = lib.Opts()
    L.csgpu_default_opts(ctypes.byref(o))
    assert o.struct_size == ctypes.sizeof(lib.Opts)
    assert o.batch == 8 and o.max_coarse == 100 and abs(o.rtol - 1e-6) < 1e-20 and o.nu_coarse == 2


def test_product_loader_has_no_fallback(tmp_path):
    import sys
    sys.path.insert(0, ROOT)
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    with pytest.raises(lib.CsgpuError):
        lib.load(str(tmp_path / "missing_libcsgpu.so"))
