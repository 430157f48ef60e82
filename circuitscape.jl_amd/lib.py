"""ctypes binding of the C ABI declared in include/csgpu.h.

This is plumbing only: it loads ``libcsgpu.so`` (the hipcc-built gfx950 library that lives next to this file)
and exposes its entry points 1:1. There is NO fallback: if the library is missing, cannot be loaded, or no HIP
device is visible, every entry point raises -- the product path never silently computes on the CPU.

A different shared object can only be substituted explicitly through ``load(path)``; the test-suite uses that to
run the *same* kernel sources compiled against the CPU fiber emulator (tests/emu), which is test infrastructure.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libcsgpu.so")

CSGPU_OK, CSGPU_NOT_CONVERGED, CSGPU_HIP_ERROR, CSGPU_OOM, CSGPU_BAD_ARGS, CSGPU_INTERNAL = range(6)
CRIT_KRYLOV, CRIT_TRUE_RESIDUAL, CRIT_BOTH = 0, 1, 2
AGG_AUTO, AGG_MIS2, AGG_GRID = 0, 1, 2

EXPORTS = [
    "csgpu_device_count", "csgpu_default_opts", "csgpu_setup", "csgpu_raster_setup", "csgpu_get_info",
    "csgpu_solve_pairs", "csgpu_solve_pairs_currents", "csgpu_solve_rhs", "csgpu_solve_grounded", "csgpu_solve_sources", "csgpu_solve_region_pairs", "csgpu_spmv_bench", "csgpu_spmv_host", "csgpu_level_spmv_host",
    "csgpu_get_level_matrix", "csgpu_raster_nodemap", "csgpu_components", "csgpu_raster_setup_grounded", "csgpu_raster_setup_poly",
    "csgpu_solve_raster", "csgpu_dia_product_host",
    "csgpu_multi_setup", "csgpu_multi_raster_setup", "csgpu_multi_solve_pairs", "csgpu_multi_solve_pairs_currents",
    "csgpu_multi_solve_grounded", "csgpu_multi_solve_sources", "csgpu_multi_device_count",
    "csgpu_multi_handle", "csgpu_multi_last_busy", "csgpu_multi_free",
    "csgpu_free", "csgpu_trim_memory", "csgpu_last_error", "csgpu_version",
]


class CsgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


# csgpu_opts, round-6 block (include/csgpu.h): the decisions that used to be environment switches; 0 = library default
OPTS_INT_FIELDS = ("last_level_sweeps", "enrich", "enrich_steps", "dia25_min_rows", "dia25_prefetch", "dia25_waves",
                   "dia25_fused_j0", "stream", "tail_rows", "poly_lattice", "cellspace", "cellspace_from_csr", "lattice_level1",
                   "lattice_level1_min_rows", "lattice_setup", "lattice_s", "lattice_q", "direct_tiles", "tile_pieces",
                   "direct_at", "dirichlet_coarse", "deflation", "tail_projection", "coarse_smoother", "nu_l1", "nu_deep",
                   "wide_csr", "fixed_k", "recompute_ap", "longrow", "narrow_tile", "spmv_grid_cap", "dia_seg", "restrict_seg",
                   "collapse_min", "verbose", "expander_probe", "fused_restrict", "sparse_init", "fused_level1")
OPTS_DOUBLE_FIELDS = ("enrich_tau", "hetero_fp64_frac", "poly_strength", "poly_coef", "poly_smin", "poly_smax",
                      "cellspace_min_frac", "tile_theta", "tile_split_min")


class Opts(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("device", ctypes.c_int32), ("max_levels", ctypes.c_int32),
        ("max_coarse", ctypes.c_int32), ("aggregation", ctypes.c_int32), ("nu_pre", ctypes.c_int32),
        ("nu_post", ctypes.c_int32), ("criterion", ctypes.c_int32), ("itmax", ctypes.c_int32),
        ("batch", ctypes.c_int32), ("check_every", ctypes.c_int32), ("nu_coarse", ctypes.c_int32),
        ("theta", ctypes.c_double), ("omega_p", ctypes.c_double), ("omega_s", ctypes.c_double),
        ("rtol", ctypes.c_double), ("atol", ctypes.c_double),
        ("node_row", ctypes.c_void_p), ("node_col", ctypes.c_void_p),
        ("precond_bytes", ctypes.c_int32), ("use_graph", ctypes.c_int32),
        ("two_product", ctypes.c_int32), ("stencil", ctypes.c_int32),
        ("explicit_check", ctypes.c_int32), ("reserved3", ctypes.c_int32),
    ] + [(name, ctypes.c_int32) for name in OPTS_INT_FIELDS] + [
        ("stream_min", ctypes.c_int64), ("host_stream_block", ctypes.c_int64),
    ] + [(name, ctypes.c_double) for name in OPTS_DOUBLE_FIELDS]


class Info(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int64), ("nnz", ctypes.c_int64), ("levels", ctypes.c_int32), ("val_bytes", ctypes.c_int32),
        ("precond_bytes", ctypes.c_int32), ("lattice_period", ctypes.c_int32),
        ("operator_complexity", ctypes.c_double), ("grid_complexity", ctypes.c_double),
        ("setup_ms", ctypes.c_double), ("upload_ms", ctypes.c_double), ("device_bytes", ctypes.c_int64),
        ("level_n", ctypes.c_int64 * 32), ("level_nnz", ctypes.c_int64 * 32),
        ("spmv_bytes_fine", ctypes.c_int64), ("bytes_per_iteration", ctypes.c_int64),
        ("level_form", ctypes.c_int32 * 32), ("hierarchy_rebuilt_fp64", ctypes.c_int32),
        ("enrich_vectors", ctypes.c_int32), ("host_blocks", ctypes.c_int32), ("reserved_info", ctypes.c_int32),
        ("batch_width", ctypes.c_int32), ("stream_mode", ctypes.c_int32), ("tail_first_level", ctypes.c_int32),
        ("last_level_sweeps", ctypes.c_int32), ("coarse_chebyshev", ctypes.c_int32), ("cellspace", ctypes.c_int32),
        ("poly_lattice", ctypes.c_int32), ("enrich_on", ctypes.c_int32), ("enrich_tau", ctypes.c_double),
        ("expander_probe_hit", ctypes.c_int32), ("fused_restrict_solves", ctypes.c_int32), ("virtual_rhs_solves", ctypes.c_int32), ("reserved_info3", ctypes.c_int32),
    ]


# csgpu_info.level_form (include/csgpu.h, CSGPU_FORM_*)
FORM_CSR, FORM_LATTICE9, FORM_LATTICE25, FORM_TAIL = 0, 1, 2, 3


class Stats(ctypes.Structure):
    _fields_ = [
        ("nrhs", ctypes.c_int32), ("max_iters", ctypes.c_int32), ("total_iters", ctypes.c_int64),
        ("max_relres", ctypes.c_double), ("solve_ms", ctypes.c_double), ("device_ms", ctypes.c_double),
        ("cg_spmv_ms", ctypes.c_double), ("cg_spmv_calls", ctypes.c_int64), ("batch", ctypes.c_int32),
        ("not_converged", ctypes.c_int32), ("graph_launches", ctypes.c_int64),
        ("polished_batches", ctypes.c_int64), ("cg_spmv_bytes", ctypes.c_int64), ("stream_slots", ctypes.c_int64),
        ("resid_ms", ctypes.c_double), ("resid_calls", ctypes.c_int64), ("resid_bytes", ctypes.c_int64),
        ("resid_fused", ctypes.c_int32), ("reserved_stats", ctypes.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_LIB = None
_LIB_PATH = None


def _bind(L):
    vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    L.csgpu_device_count.restype = i32
    L.csgpu_default_opts.argtypes = [ctypes.POINTER(Opts)]
    L.csgpu_default_opts.restype = None
    L.csgpu_setup.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, ctypes.POINTER(Opts), ctypes.POINTER(vp)]
    L.csgpu_raster_setup.argtypes = [vp, i64, i64, i32, i32, i32, i32, ctypes.POINTER(Opts), ctypes.POINTER(vp)]
    L.csgpu_get_info.argtypes = [vp, ctypes.POINTER(Info)]
    L.csgpu_solve_pairs.argtypes = [vp, vp, vp, i64, vp, vp, i64, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_solve_pairs_currents.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_solve_rhs.argtypes = [vp, vp, i64, vp, ctypes.POINTER(Stats)]
    L.csgpu_solve_grounded.argtypes = [vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_solve_sources.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_solve_region_pairs.argtypes = [vp, vp, vp, i64, vp, vp, i64, vp, ctypes.POINTER(Stats)]
    L.csgpu_spmv_bench.argtypes = [vp, i32, i32, ctypes.POINTER(dbl)]
    L.csgpu_spmv_host.argtypes = [vp, vp, vp, i32]
    L.csgpu_level_spmv_host.argtypes = [vp, i32, i32, vp, vp, i32, vp]
    L.csgpu_raster_nodemap.argtypes = [vp, vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.csgpu_components.argtypes = [vp, vp, ctypes.POINTER(i64)]
    L.csgpu_raster_setup_grounded.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32, ctypes.POINTER(Opts), ctypes.POINTER(vp)]
    L.csgpu_raster_setup_poly.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32, ctypes.POINTER(Opts), ctypes.POINTER(vp)]
    L.csgpu_solve_raster.argtypes = [vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_get_level_matrix.argtypes = [vp, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64),
                                         vp, vp, vp]
    L.csgpu_dia_product_host.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
    L.csgpu_multi_setup.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, ctypes.POINTER(Opts), vp, i32, ctypes.POINTER(vp)]
    L.csgpu_multi_raster_setup.argtypes = [vp, i64, i64, i32, i32, i32, i32, ctypes.POINTER(Opts), vp, i32, ctypes.POINTER(vp)]
    L.csgpu_multi_solve_pairs.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_multi_solve_pairs_currents.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_multi_solve_pairs_currents.restype = ctypes.c_int
    L.csgpu_multi_solve_grounded.argtypes = [vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_multi_solve_sources.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Stats)]
    L.csgpu_multi_device_count.argtypes = [vp]
    L.csgpu_multi_handle.argtypes = [vp, i32]
    L.csgpu_multi_handle.restype = vp
    L.csgpu_multi_last_busy.argtypes = [vp, vp, vp]
    L.csgpu_multi_free.argtypes = [vp]
    L.csgpu_multi_free.restype = None
    L.csgpu_free.argtypes = [vp]
    L.csgpu_free.restype = None
    L.csgpu_trim_memory.argtypes = [i32]
    L.csgpu_trim_memory.restype = i64
    L.csgpu_last_error.restype = ctypes.c_char_p
    L.csgpu_version.restype = ctypes.c_char_p
    return L


def load(path=None):
    """Load the shared library (default: the in-tree hipcc build). Raises if it is missing -- no fallback."""
    global _LIB, _LIB_PATH
    path = os.path.abspath(path or DEFAULT_LIB)
    if _LIB is not None and _LIB_PATH == path:
        return _LIB
    if not os.path.exists(path):
        raise CsgpuError(CSGPU_INTERNAL,
                         "HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; "
                         "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = ctypes.CDLL(path)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise CsgpuError(CSGPU_INTERNAL, "%s does not export %s" % (path, name))
    _LIB, _LIB_PATH = _bind(L), path
    return _LIB


def lib():
    return _LIB if _LIB is not None else load()


def loaded_path():
    return _LIB_PATH


def _check(rc):
    if rc != 0:
        raise CsgpuError(rc, (lib().csgpu_last_error() or b"").decode("utf-8", "replace"))


def device_count():
    return lib().csgpu_device_count()


def trim_memory(device=-1):
    """Return the library's pooled device blocks (of one device, or of all) to the driver; bytes released."""
    return lib().csgpu_trim_memory(device)


def default_opts(**kw):
    o = Opts()
    lib().csgpu_default_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown option %r" % k)
        setattr(o, k, v)
    return o


def _ragged(lists):
    """(ptr[len + 1], idx) int64 arrays of a list of lists of node ids"""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(q) for q in lists])
    idx = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.int64).ravel() for q in lists])
                               if ptr[-1] > 0 else np.zeros(1, dtype=np.int64))
    return ptr, idx


def _sources_args(dtype, n, sources, grounds, values, check, want_voltages, want_currents, cum, mx):
    """Marshal the arguments of csgpu_[multi_]solve_sources. sources / grounds: one list of 0-based node ids per column
    (a bare int = that one node); values: None (all ones) or one list of numbers per column, matching `sources`;
    check: None or one node id per column (< 0: none)."""
    sources = [[q] if np.isscalar(q) else list(q) for q in sources]
    nrhs = len(sources)
    assert len(grounds) == nrhs
    sptr, sidx = _ragged(sources)
    gptr, gidx = _ragged(grounds)
    sval = None
    if values is not None:
        values = [[v] if np.isscalar(v) else list(v) for v in values]
        assert [len(v) for v in values] == [len(q) for q in sources]
        sval = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=dtype) for v in values])
                                    if sptr[-1] > 0 else np.zeros(1, dtype=dtype))
    chk = np.ascontiguousarray(check, dtype=np.int64) if check is not None else None
    assert chk is None or chk.shape == (nrhs,)
    cout = np.zeros(nrhs, dtype=dtype) if chk is not None else None
    X = np.zeros((n, nrhs), dtype=dtype, order="F") if want_voltages else None
    C = np.zeros((n, nrhs), dtype=dtype, order="F") if want_currents else None
    for a in (cum, mx):
        assert a is None or (a.dtype == dtype and a.flags["C_CONTIGUOUS"] and a.shape == (n,))
    ptr = lambda a: a.ctypes.data if a is not None else None
    args = [nrhs, sptr.ctypes.data, sidx.ctypes.data, ptr(sval), gptr.ctypes.data, gidx.ctypes.data, ptr(chk), ptr(cout),
            ptr(X), ptr(C), ptr(cum), ptr(mx)]
    return args, (sptr, sidx, sval, gptr, gidx, chk), cout, X, C


class Handle:
    """Owns a csgpu_handle* (device-resident matrix + AMG hierarchy); freed on close()/GC like the reference's
    factor objects (ext/CircuitscapePardisoExt.jl:8-13)."""

    def __init__(self, ptr, dtype, keepalive=None):
        self._p = ptr
        self.dtype = np.dtype(dtype)
        self._keep = keepalive

    def close(self):
        if self._p:
            lib().csgpu_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def info(self):
        i = Info()
        _check(lib().csgpu_get_info(self._p, ctypes.byref(i)))
        d = {k: getattr(i, k) for k, _ in Info._fields_ if k not in ("level_n", "level_nnz", "level_form")}
        d["level_n"] = [i.level_n[l] for l in range(min(i.levels, 32))]
        d["level_nnz"] = [i.level_nnz[l] for l in range(min(i.levels, 32))]
        d["level_form"] = [i.level_form[l] for l in range(min(i.levels, 32))]
        return d

    def solve_pairs(self, src, dst, gather=None, want_voltages=False):
        """0-based node ids. Returns (resistances[npairs], gathered[npairs, ngather] or None,
        voltages[n, npairs] (Fortran order) or None, stats dict)."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        npairs = len(src)
        n = self.info["n"]
        res = np.zeros(npairs, dtype=self.dtype)
        g = np.ascontiguousarray(gather, dtype=np.int64) if gather is not None and len(gather) else None
        gathered = np.zeros((npairs, len(g)), dtype=self.dtype) if g is not None else None
        volt = np.zeros((n, npairs), dtype=self.dtype, order="F") if want_voltages else None
        st = Stats()
        rc = lib().csgpu_solve_pairs(self._p, src.ctypes.data, dst.ctypes.data, npairs,
                                     volt.ctypes.data if volt is not None else None,
                                     g.ctypes.data if g is not None else None, len(g) if g is not None else 0,
                                     gathered.ctypes.data if gathered is not None else None, res.ctypes.data,
                                     ctypes.byref(st))
        _check(rc)
        return res, gathered, volt, st.as_dict()

    def solve_pairs_currents(self, src, dst, weights=None, want_voltages=False, want_currents=True, cum=None, mx=None,
                             want_branch=False):
        """Pair solves + node currents (scope row N1). cum / mx: optional length-n arrays updated in place
        (cum += sum_p w_p * curr_p, mx = max(mx, curr_p)). Returns (R, voltages or None, currents or None, stats)."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        npairs = len(src)
        n = self.info["n"]
        res = np.zeros(npairs, dtype=self.dtype)
        volt = np.zeros((n, npairs), dtype=self.dtype, order="F") if want_voltages else None
        curr = np.zeros((n, npairs), dtype=self.dtype, order="F") if want_currents else None
        branch = np.zeros((self.info["nnz"], npairs), dtype=self.dtype, order="F") if want_branch else None
        w = np.ascontiguousarray(weights, dtype=np.int32) if weights is not None else None
        for a in (cum, mx):
            assert a is None or (a.dtype == self.dtype and a.flags["C_CONTIGUOUS"] and a.shape == (n,))
        st = Stats()
        rc = lib().csgpu_solve_pairs_currents(self._p, src.ctypes.data, dst.ctypes.data, npairs,
                                              w.ctypes.data if w is not None else None,
                                              volt.ctypes.data if volt is not None else None,
                                              curr.ctypes.data if curr is not None else None,
                                              cum.ctypes.data if cum is not None else None,
                                              mx.ctypes.data if mx is not None else None,
                                              branch.ctypes.data if branch is not None else None, res.ctypes.data,
                                              ctypes.byref(st))
        _check(rc)
        if want_branch:
            return res, volt, curr, st.as_dict(), branch
        return res, volt, curr, st.as_dict()

    def solve_rhs(self, rhs):
        rhs = np.asarray(rhs, dtype=self.dtype)
        one = rhs.ndim == 1
        B = np.asfortranarray(rhs.reshape(rhs.shape[0], -1))
        X = np.zeros_like(B, order="F")
        st = Stats()
        _check(lib().csgpu_solve_rhs(self._p, B.ctypes.data, B.shape[1], X.ctypes.data, ctypes.byref(st)))
        return (X[:, 0] if one else X), st.as_dict()

    def solve_grounded(self, rhs, grounds, want_currents=False):
        """csgpu_solve_grounded: rhs (n, nrhs) or (n,), grounds = one list of 0-based node ids per column (x = 0
        there). Returns (x, currents or None, stats)."""
        rhs = np.asarray(rhs, dtype=self.dtype)
        one = rhs.ndim == 1
        B = np.asfortranarray(rhs.reshape(rhs.shape[0], -1))
        if one:
            grounds = [grounds] if (len(grounds) == 0 or np.isscalar(grounds[0])) else grounds
        assert len(grounds) == B.shape[1]
        gptr = np.zeros(B.shape[1] + 1, dtype=np.int64)
        gptr[1:] = np.cumsum([len(g) for g in grounds])
        gidx = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int64) for g in grounds])
                                    if gptr[-1] > 0 else np.zeros(1, dtype=np.int64))
        X = np.zeros_like(B, order="F")
        C = np.zeros_like(B, order="F") if want_currents else None
        st = Stats()
        _check(lib().csgpu_solve_grounded(self._p, B.ctypes.data, B.shape[1], gptr.ctypes.data, gidx.ctypes.data,
                                          X.ctypes.data, C.ctypes.data if C is not None else None, ctypes.byref(st)))
        if one:
            return X[:, 0], (C[:, 0] if C is not None else None), st.as_dict()
        return X, C, st.as_dict()

    def solve_sources(self, sources, grounds, values=None, check=None, want_voltages=False, want_currents=False,
                      cum=None, mx=None):
        """csgpu_solve_sources: sparse right-hand sides (column c = `values[c]` -- default ones -- at the nodes
        `sources[c]`), x = 0 on `grounds[c]`. check: one node per column whose voltage is returned (the one-to-all
        drivers' `res[i] = v[1]`). cum / mx: optional length-n arrays updated in place with the columns' node currents.
        Returns (check voltages or None, voltages (n, nrhs) or None, currents or None, stats)."""
        args, keep, cout, X, C = _sources_args(self.dtype, self.info["n"], sources, grounds, values, check, want_voltages,
                                               want_currents, cum, mx)
        st = Stats()
        _check(lib().csgpu_solve_sources(self._p, *args, ctypes.byref(st)))
        del keep
        return cout, X, C, st.as_dict()

    def solve_region_pairs(self, sets, src_set, dst_set):
        """csgpu_solve_region_pairs: `sets` = list of lists of 0-based node ids; effective resistance between the
        short-circuited sets sets[src_set[p]] and sets[dst_set[p]] for every p (-1: no current flows). Returns
        (resistances, stats)."""
        sptr = np.zeros(len(sets) + 1, dtype=np.int64)
        sptr[1:] = np.cumsum([len(q) for q in sets])
        snodes = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.int64) for q in sets])
                                      if sptr[-1] > 0 else np.zeros(1, dtype=np.int64))
        a = np.ascontiguousarray(src_set, dtype=np.int64)
        b = np.ascontiguousarray(dst_set, dtype=np.int64)
        R = np.zeros(len(a), dtype=np.float64)
        st = Stats()
        _check(lib().csgpu_solve_region_pairs(self._p, sptr.ctypes.data, snodes.ctypes.data, len(sets), a.ctypes.data,
                                              b.ctypes.data, len(a), R.ctypes.data, ctypes.byref(st)))
        return R, st.as_dict()

    def spmv_bench(self, k=1, reps=20):
        ms = ctypes.c_double(0)
        _check(lib().csgpu_spmv_bench(self._p, k, reps, ctypes.byref(ms)))
        return ms.value

    def spmv(self, x):
        """y = A x for x of shape (n,) or (n, k) with k in {1,2,4,8,16} (host arrays; test helper)."""
        x = np.asarray(x, dtype=self.dtype)
        k = 1 if x.ndim == 1 else x.shape[1]
        xi = np.ascontiguousarray(x.reshape(x.shape[0], k))  # C order == interleaved [n][k]
        y = np.zeros_like(xi)
        _check(lib().csgpu_spmv_host(self._p, xi.ctypes.data, y.ctypes.data, k))
        return y[:, 0] if x.ndim == 1 else y

    def raster_nodemap(self):
        """Node map (1-based ids, 0 = NODATA) of a handle built by raster_setup, shape (nrows, ncols)."""
        r, c = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().csgpu_raster_nodemap(self._p, None, ctypes.byref(r), ctypes.byref(c)))
        nm = np.zeros((r.value, c.value), dtype=np.int32)
        _check(lib().csgpu_raster_nodemap(self._p, nm.ctypes.data, None, None))
        return nm

    def solve_raster(self, source, want_currents=True, want_voltages=False):
        """Advanced-mode solve, rasters in and out (csgpu_solve_raster). Returns (current raster or None, voltage
        raster or None, stats)."""
        nm_r, nm_c = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().csgpu_raster_nodemap(self._p, None, ctypes.byref(nm_r), ctypes.byref(nm_c)))
        src = np.ascontiguousarray(source, dtype=self.dtype)
        assert src.shape == (nm_r.value, nm_c.value)
        cur = np.zeros(src.shape, dtype=self.dtype) if want_currents else None
        vol = np.zeros(src.shape, dtype=self.dtype) if want_voltages else None
        st = Stats()
        _check(lib().csgpu_solve_raster(self._p, src.ctypes.data, cur.ctypes.data if cur is not None else None,
                                        vol.ctypes.data if vol is not None else None, ctypes.byref(st)))
        return cur, vol, st.as_dict()

    def components(self):
        """(labels, count): dense 0-based component index per node, ordered by smallest node id (device CC)."""
        lab = np.zeros(self.info["n"], dtype=np.int32)
        nc = ctypes.c_int64(0)
        _check(lib().csgpu_components(self._p, lab.ctypes.data, ctypes.byref(nc)))
        return lab, nc.value

    def level_spmv(self, lvl, which, x):
        """y = (level operator) x through the V-cycle's launcher for that operator; returns (y, dots) where dots is
        the fused x[:n].y per column for which == "M", else None. x has shape (ncols,) or (ncols, k)."""
        w = {"A": 0, "P": 1, "R": 2, "Q": 3, "QT": 4, "M": 5}[which]
        info = self.info
        dt = np.float32 if (info["precond_bytes"] or info["val_bytes"]) == 4 else np.float64
        nr, nc, nz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().csgpu_get_level_matrix(self._p, lvl, w, ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nz),
                                            None, None, None))
        x = np.asarray(x, dtype=dt)
        k = 1 if x.ndim == 1 else x.shape[1]
        assert x.shape[0] == nc.value
        xi = np.ascontiguousarray(x.reshape(nc.value, k))
        y = np.zeros((nr.value, k), dtype=dt)
        dots = np.zeros(k, dtype=np.float64)
        _check(lib().csgpu_level_spmv_host(self._p, lvl, w, xi.ctypes.data, y.ctypes.data, k, dots.ctypes.data))
        return (y[:, 0] if x.ndim == 1 else y), (dots if which == "M" else None)

    def poly_project_norm(self, x):
        """Test hook (csgpu_level_spmv_host, which = 6) of a polygon handle on the lattice path: returns (Pi x, node-space
        squared norms per column) for a cell-space array x of shape (R * C, k), rows = column-major cell ids."""
        info = self.info
        dt = np.float32 if (info["precond_bytes"] or info["val_bytes"]) == 4 else np.float64
        x = np.ascontiguousarray(x, dtype=dt)
        k = x.shape[1]
        y = np.zeros_like(x)
        dots = np.zeros(k, dtype=np.float64)
        _check(lib().csgpu_level_spmv_host(self._p, 0, 6, x.ctypes.data, y.ctypes.data, k, dots.ctypes.data))
        return y, dots

    def dia_product(self, z, p_in, beta):
        """Fused lattice-form CG product (test hook): returns (p_out, y, dots) with p_out = z + beta * p_in,
        y = A p_out, dots = column-wise p_out . y. z, p_in: (n, k) arrays, beta: k values."""
        info = self.info
        dt = np.float32 if (info["precond_bytes"] or info["val_bytes"]) == 4 else np.float64
        z = np.ascontiguousarray(z, dtype=dt)
        p_in = np.ascontiguousarray(p_in, dtype=dt)
        k = z.shape[1]
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        assert z.shape == p_in.shape == (info["n"], k) and beta.shape == (k,)
        p_out = np.zeros_like(z)
        y = np.zeros((info["n"], k), dtype=self.dtype)
        dots = np.zeros(k, dtype=np.float64)
        _check(lib().csgpu_dia_product_host(self._p, z.ctypes.data, p_in.ctypes.data, beta.ctypes.data,
                                            p_out.ctypes.data, y.ctypes.data, k, dots.ctypes.data))
        return p_out, y, dots

    def level_matrix(self, lvl, which="A"):
        import scipy.sparse as sp
        w = {"A": 0, "P": 1, "R": 2, "Q": 3, "QT": 4, "M": 5}[which]
        nr, nc, nz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().csgpu_get_level_matrix(self._p, lvl, w, ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nz),
                                            None, None, None))
        rp = np.zeros(nr.value + 1, dtype=np.int32)
        ci = np.zeros(max(nz.value, 1), dtype=np.int32)
        va = np.zeros(max(nz.value, 1), dtype=self.dtype)
        _check(lib().csgpu_get_level_matrix(self._p, lvl, w, None, None, None, rp.ctypes.data, ci.ctypes.data,
                                            va.ctypes.data))
        return sp.csr_matrix((va[:nz.value], ci[:nz.value], rp), shape=(nr.value, nc.value))


class MultiHandle:
    """Owns a csgpu_multi* (one replicated handle per GPU of the node, chunks of pairs dealt from a shared queue)."""

    def __init__(self, ptr, dtype):
        self._p = ptr
        self.dtype = np.dtype(dtype)

    def close(self):
        if self._p:
            lib().csgpu_multi_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def ndevices(self):
        return lib().csgpu_multi_device_count(self._p)

    def info(self, slot=0):
        i = Info()
        _check(lib().csgpu_get_info(lib().csgpu_multi_handle(self._p, slot), ctypes.byref(i)))
        return {k: getattr(i, k) for k, _ in Info._fields_ if k not in ("level_n", "level_nnz", "level_form")}

    def solve_pairs(self, src, dst, gather=None):
        """As Handle.solve_pairs (no voltages). Returns (resistances, gathered or None, stats dict incl. the per-device
        busy seconds and pair counts of this call)."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        npairs = len(src)
        res = np.zeros(npairs, dtype=self.dtype)
        g = np.ascontiguousarray(gather, dtype=np.int64) if gather is not None and len(gather) else None
        gathered = np.zeros((npairs, len(g)), dtype=self.dtype) if g is not None else None
        st = Stats()
        rc = lib().csgpu_multi_solve_pairs(self._p, src.ctypes.data, dst.ctypes.data, npairs,
                                           g.ctypes.data if g is not None else None, len(g) if g is not None else 0,
                                           gathered.ctypes.data if gathered is not None else None, res.ctypes.data,
                                           ctypes.byref(st))
        _check(rc)
        nd = self.ndevices
        busy = np.zeros(nd)
        done = np.zeros(nd, dtype=np.int64)
        lib().csgpu_multi_last_busy(self._p, busy.ctypes.data, done.ctypes.data)
        d = st.as_dict()
        d["device_busy_s"] = busy.tolist()
        d["device_pairs"] = done.tolist()
        return res, gathered, d

    def solve_pairs_currents(self, src, dst, weights=None, cum=None, mx=None):
        """csgpu_multi_solve_pairs_currents: pair solves dealt over the devices with the reference's cumulative / maximum
        node-current vectors (src/out.jl:96-107) accumulated per device and combined on return. cum / mx: optional length-n
        arrays of the handle's value type, updated in place. Returns (resistances, stats dict)."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        npairs = len(src)
        n = self.info(0)["n"]
        res = np.zeros(npairs, dtype=self.dtype)
        w = np.ascontiguousarray(weights, dtype=np.int32) if weights is not None else None
        for a in (cum, mx):
            assert a is None or (a.dtype == self.dtype and a.flags["C_CONTIGUOUS"] and a.shape == (n,))
        st = Stats()
        rc = lib().csgpu_multi_solve_pairs_currents(self._p, src.ctypes.data, dst.ctypes.data, npairs,
                                                    w.ctypes.data if w is not None else None,
                                                    cum.ctypes.data if cum is not None else None,
                                                    mx.ctypes.data if mx is not None else None, res.ctypes.data,
                                                    ctypes.byref(st))
        _check(rc)
        nd = self.ndevices
        busy = np.zeros(nd)
        done = np.zeros(nd, dtype=np.int64)
        lib().csgpu_multi_last_busy(self._p, busy.ctypes.data, done.ctypes.data)
        d = st.as_dict()
        d["device_busy_s"] = busy.tolist()
        d["device_pairs"] = done.tolist()
        return res, d

    def solve_grounded(self, rhs, grounds, want_currents=False):
        """csgpu_multi_solve_grounded: Handle.solve_grounded with the columns dealt over the devices (contiguous ranges).
        Returns (x, currents or None, stats incl. per-device busy seconds and column counts)."""
        B = np.asfortranarray(np.asarray(rhs, dtype=self.dtype).reshape(np.shape(rhs)[0], -1))
        assert len(grounds) == B.shape[1] and B.shape[0] == self.info(0)["n"]
        gptr, gidx = _ragged(grounds)
        X = np.zeros_like(B, order="F")
        C = np.zeros_like(B, order="F") if want_currents else None
        st = Stats()
        _check(lib().csgpu_multi_solve_grounded(self._p, B.ctypes.data, B.shape[1], gptr.ctypes.data, gidx.ctypes.data,
                                                X.ctypes.data, C.ctypes.data if C is not None else None, ctypes.byref(st)))
        return X, C, _multi_busy(self, st)

    def solve_sources(self, sources, grounds, values=None, check=None, want_voltages=False, want_currents=False,
                      cum=None, mx=None):
        """csgpu_multi_solve_sources: Handle.solve_sources with the columns dealt over the devices. Returns (check
        voltages or None, voltages or None, currents or None, stats incl. per-device busy seconds / columns)."""
        args, keep, cout, X, C = _sources_args(self.dtype, self.info(0)["n"], sources, grounds, values, check,
                                               want_voltages, want_currents, cum, mx)
        st = Stats()
        _check(lib().csgpu_multi_solve_sources(self._p, *args, ctypes.byref(st)))
        del keep
        return cout, X, C, _multi_busy(self, st)


def _multi_busy(mh, st):
    nd = mh.ndevices
    busy = np.zeros(nd)
    done = np.zeros(nd, dtype=np.int64)
    lib().csgpu_multi_last_busy(mh._p, busy.ctypes.data, done.ctypes.data)
    d = st.as_dict()
    d["device_busy_s"] = busy.tolist()
    d["device_pairs"] = done.tolist()
    return d


def _device_list(devices):
    if devices is None:
        return None, 0
    if isinstance(devices, int):
        return None, devices
    arr = np.ascontiguousarray(devices, dtype=np.int32)
    return arr, len(arr)


def multi_raster_setup(cond, opts=None, devices=None, four_neighbors=False, avg_resistances=False, reg=True):
    """csgpu_multi_raster_setup: one replicated handle per device. devices: None = all visible, an int = the first
    that many, or a list of ordinals."""
    cond = np.ascontiguousarray(cond)
    dtype = np.float32 if cond.dtype == np.float32 else np.float64
    cond = np.ascontiguousarray(cond, dtype=dtype)
    o = opts if opts is not None else default_opts()
    arr, nd = _device_list(devices)
    h = ctypes.c_void_p(0)
    _check(lib().csgpu_multi_raster_setup(cond.ctypes.data, cond.shape[0], cond.shape[1], np.dtype(dtype).itemsize,
                                          int(four_neighbors), int(avg_resistances), int(reg), ctypes.byref(o),
                                          arr.ctypes.data if arr is not None else None, nd, ctypes.byref(h)))
    return MultiHandle(h, dtype)


def multi_setup(matrix, opts=None, devices=None, index_dtype=np.int64, index_base=1):
    """csgpu_multi_setup on a scipy sparse symmetric matrix (arrays handed over the way Julia would)."""
    m = matrix.tocsr()
    m.sort_indices()
    dtype = np.float32 if m.dtype == np.float32 else np.float64
    rp = np.ascontiguousarray(m.indptr.astype(index_dtype) + index_base)
    ci = np.ascontiguousarray(m.indices.astype(index_dtype) + index_base)
    va = np.ascontiguousarray(m.data, dtype=dtype)
    o = opts if opts is not None else default_opts()
    arr, nd = _device_list(devices)
    h = ctypes.c_void_p(0)
    _check(lib().csgpu_multi_setup(rp.ctypes.data, ci.ctypes.data, va.ctypes.data, m.shape[0], m.nnz,
                                   np.dtype(index_dtype).itemsize, np.dtype(dtype).itemsize, index_base, ctypes.byref(o),
                                   arr.ctypes.data if arr is not None else None, nd, ctypes.byref(h)))
    return MultiHandle(h, dtype)


def setup(matrix, opts=None, node_row=None, node_col=None, index_dtype=np.int64, index_base=1):
    """csgpu_setup on a scipy sparse symmetric matrix. By default the arrays are handed over the way Julia's
    SparseMatrixCSC{T,Int64} would hand them (Int64, 1-based) so the conversion path is the one production uses."""
    m = matrix.tocsr()
    m.sort_indices()
    dtype = np.float32 if m.dtype == np.float32 else np.float64
    n = m.shape[0]
    rp = np.ascontiguousarray(m.indptr.astype(index_dtype) + index_base)
    ci = np.ascontiguousarray(m.indices.astype(index_dtype) + index_base)
    va = np.ascontiguousarray(m.data, dtype=dtype)
    o = opts if opts is not None else default_opts()
    keep = []
    if node_row is not None:
        nr = np.ascontiguousarray(node_row, dtype=np.int32)
        nc = np.ascontiguousarray(node_col, dtype=np.int32)
        o.node_row = nr.ctypes.data
        o.node_col = nc.ctypes.data
        keep = [nr, nc]
    h = ctypes.c_void_p(0)
    rc = lib().csgpu_setup(rp.ctypes.data, ci.ctypes.data, va.ctypes.data, n, m.nnz, np.dtype(index_dtype).itemsize,
                           np.dtype(dtype).itemsize, index_base, ctypes.byref(o), ctypes.byref(h))
    o.node_row = None
    o.node_col = None
    del keep
    _check(rc)
    return Handle(h, dtype)


def setup_arrays(rowptr, colidx, vals, n, nnz, opts=None, index_base=1):
    """csgpu_setup on raw CSR/CSC arrays (any of Int32/Int64 indices, Float32/Float64 values) without copies."""
    assert rowptr.flags["C_CONTIGUOUS"] and colidx.flags["C_CONTIGUOUS"] and vals.flags["C_CONTIGUOUS"]
    assert rowptr.dtype == colidx.dtype and rowptr.dtype.itemsize in (4, 8) and vals.dtype.itemsize in (4, 8)
    o = opts if opts is not None else default_opts()
    h = ctypes.c_void_p(0)
    rc = lib().csgpu_setup(rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, n, nnz, rowptr.dtype.itemsize,
                           vals.dtype.itemsize, index_base, ctypes.byref(o), ctypes.byref(h))
    _check(rc)
    return Handle(h, np.float32 if vals.dtype.itemsize == 4 else np.float64)


def raster_setup(cond, opts=None, four_neighbors=False, avg_resistances=False, reg=True, ground=None, polymap=None):
    """csgpu_raster_setup[_grounded|_poly]: Laplacian of a conductance raster (NODATA = values <= 0) built directly in
    HBM; `ground`: optional raster of finite ground conductances added to the diagonal (advanced mode); `polymap`:
    optional raster of short-circuit polygon ids (> 0; cells of a polygon share one node)."""
    cond = np.ascontiguousarray(cond)
    dtype = np.float32 if cond.dtype == np.float32 else np.float64
    cond = np.ascontiguousarray(cond, dtype=dtype)
    o = opts if opts is not None else default_opts()
    h = ctypes.c_void_p(0)
    if polymap is not None and np.size(polymap) > 0:
        assert ground is None, "polygons and finite grounds together are not supported on the device path"
        pm = np.ascontiguousarray(polymap, dtype=np.int32)
        assert pm.shape == cond.shape
        rc = lib().csgpu_raster_setup_poly(cond.ctypes.data, pm.ctypes.data, cond.shape[0], cond.shape[1],
                                           np.dtype(dtype).itemsize, int(four_neighbors), int(avg_resistances), int(reg),
                                           ctypes.byref(o), ctypes.byref(h))
        _check(rc)
        return Handle(h, dtype)
    gnd = None
    if ground is not None:
        gnd = np.ascontiguousarray(ground, dtype=dtype)
        assert gnd.shape == cond.shape and np.all(np.isfinite(gnd)), "finite ground conductances only"
    rc = lib().csgpu_raster_setup_grounded(cond.ctypes.data, gnd.ctypes.data if gnd is not None else None,
                                           cond.shape[0], cond.shape[1], np.dtype(dtype).itemsize, int(four_neighbors),
                                           int(avg_resistances), int(reg), ctypes.byref(o), ctypes.byref(h))
    _check(rc)
    return Handle(h, dtype)
