import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# (CSGPU_EMU_LIB: another build of the same emulator library, e.g. one compiled with -fsanitize=address -- tools/README.md)
EMU_LIB = os.environ.get("CSGPU_EMU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden_cases():
    """pairwise cases (raster + network) of the reference's regression suite"""
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and f.startswith("sg"))


def advanced_cases():
    """network advanced-mode cases (multiple_solve path)"""
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and f.startswith("mgNetwork"))


def _stem(f):
    return f[:-8] if f.endswith(".json.gz") else f[:-5]


def raster_advanced_cases():
    """raster advanced-mode cases (mgVerify1..6: voltage / current map goldens; mgVerify7, a 355 x 481 landscape with
    5574 finite grounds whose golden current map the reference ships but its own suite does not run
    (test/test_utils.jl:117 stops at 6), is stored gzipped)"""
    return sorted(_stem(f) for f in os.listdir(GOLDEN)
                  if (f.endswith(".json") or f.endswith(".json.gz")) and f.startswith("mgVerify"))


def onetoall_cases():
    """raster one-to-all (13) and all-to-one (12) cases"""
    return sorted(f[:-5] for f in os.listdir(GOLDEN)
                  if f.endswith(".json") and (f.startswith("oneToAllVerify") or f.startswith("allToOneVerify")))


def compare_aagrid(expected, got, tol=1e-6):
    """the reference's map criterion (test/test_utils.jl:196): sum of squared differences below tol"""
    return float(np.sum((np.asarray(expected, dtype=float) - np.asarray(got, dtype=float)) ** 2)) < tol


def load_case(name):
    path = os.path.join(GOLDEN, name + ".json")
    if not os.path.exists(path):
        with gzip.open(path + ".gz", "rt") as f:
            return json.load(f)
    with open(path) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (checker). Built on demand from oracle/cs_oracle.cpp."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "libcs_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import refsolve
    refsolve.lib()
    return refsolve


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel sources compiled against the CPU fiber emulator (tests/emu): exercises kernel LOGIC without a GPU.
    Test infrastructure only -- the product loader never picks this library up by itself."""
    if not os.path.exists(EMU_LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")])
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    lib.load(EMU_LIB)
    return lib


@pytest.fixture(scope="session")
def gpu_lib():
    """The real hipcc-built library on a real device. No fallback: missing library or device is a failure."""
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    lib.load()
    assert lib.loaded_path().endswith("libcsgpu.so")
    assert lib.device_count() >= 1, "no HIP device visible"
    return lib


def compare_resistances(expected, got, rtol=1e-6, atol=1e-9):
    """-1 / 0 pattern identical, finite entries within rtol (the goldens carry 10 significant digits)."""
    expected = np.asarray(expected, dtype=float)
    got = np.asarray(got, dtype=float)
    assert expected.shape == got.shape, (expected.shape, got.shape)
    assert np.array_equal(expected == -1, got == -1)
    err = np.abs(expected - got)
    assert np.all(err <= atol + rtol * np.abs(expected)), float(err.max())
