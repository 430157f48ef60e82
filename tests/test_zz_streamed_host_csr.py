"""Host matrices with 2^31 stored entries and more (the reference's use_64bit_indexing, src/run.jl:34, src/config.jl:28):
csgpu_setup streams them to the device in blocks of rows (csgpu.hip, setup_from_host_streamed). The path is exercised at
test sizes through CSGPU_STREAM_HOST_CSR; the argument checks at the real threshold need no memory (the calls are refused
before anything is read). Sorted last: the newest device test runs after every older one."""
import ctypes

import numpy as np
import pytest


def test_streamed_host_csr_matches_ordinary_path(emu_lib, oracle):
    from helpers import check_streamed_host_csr
    check_streamed_host_csr(emu_lib, oracle)


def test_matrices_above_2_31_entries_are_admitted_only_with_coordinates(emu_lib):
    """nnz >= 2^31 used to be refused by the argument check (status 4, "too large for int32 device indexing"); it now reaches
    the streamed set-up, which refuses -- before reading a single array element beyond the row pointers' ends -- what it
    cannot take, and says why. n >= 2^31 - 1 is still outside the device's int32 node ids."""
    L = emu_lib
    lib = L.lib()
    n, nnz = 1000, (1 << 31) + 5
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[-1] = nnz
    dummy = np.zeros(16, dtype=np.float64)
    h = ctypes.c_void_p(0)
    o = L.default_opts()
    rc = lib.csgpu_setup(rp.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, n, nnz, 8, 8, 0, ctypes.byref(o), ctypes.byref(h))
    assert rc == 4 and b"node_row / node_col missing" in lib.csgpu_last_error(), lib.csgpu_last_error()
    rc = lib.csgpu_setup(rp.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, (1 << 31) - 1, nnz, 8, 8, 0, ctypes.byref(o),
                         ctypes.byref(h))
    assert rc == 4 and b"int32 device indexing" in lib.csgpu_last_error()


@pytest.mark.gpu
def test_streamed_host_csr_matches_ordinary_path_gpu(gpu_lib, oracle):
    from helpers import check_streamed_host_csr
    check_streamed_host_csr(gpu_lib, oracle, exact=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", __import__("conftest").golden_cases())
def test_golden_fixtures_in_single_precision_gpu(gpu_lib, name):
    """runtests(precision = "single") of test/test_utils.jl through the product path on the device (helpers.
    check_golden_single_precision; emulator twin in test_emu_solver.py: worst case 1.8e-3 against the reference's 1e-2)"""
    from helpers import check_golden_single_precision
    check_golden_single_precision(gpu_lib, name)
