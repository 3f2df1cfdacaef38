"""ctypes face of libsparseharness_host.so (no HIP): the product's MatrixMarket
loader (host/src/sparse_matrix.cpp) and the seeded synthetic generators
(host/src/synth.cpp) that define the benchmark configs of BASELINE.json."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsparseharness_host.so")
HOST_DIR = os.path.join(_HERE, "host")

# seeds and shapes fixed by SURVEY.md 8d / BASELINE.md
SEED_RMAT = 0x5EED0023
SEED_POWERLAW = 0x5EED1000
SEED_SCIRCUIT = 0x5EED5C1C


class _HostCsr(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("header_nnz", C.c_int32), ("nnz", C.c_int64),
                ("row_ptr", C.POINTER(C.c_int32)), ("col_idx", C.POINTER(C.c_int32)), ("val", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HOST_DIR, "../libsparseharness_host.so"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {HOST_DIR}`")
        os.environ.setdefault("SH_QUIET_TIMERS", "1")
        _lib = C.CDLL(LIB_PATH)
        _lib.sh_synth_powerlaw.restype = C.c_int
        _lib.sh_synth_powerlaw.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int64, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.sh_synth_rmat.restype = C.c_int
        _lib.sh_synth_rmat.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.sh_mm_load.restype = C.c_int
        _lib.sh_mm_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(_HostCsr)]
        _lib.sh_mm_load_ex.restype = C.c_int
        _lib.sh_mm_load_ex.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(_HostCsr)]
        _lib.sh_host_csr_release.restype = None
        _lib.sh_host_csr_release.argtypes = [C.POINTER(_HostCsr)]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def powerlaw(rows, nnz, cols=None, exponent=2.1, dmax=1_000_000, seed=SEED_POWERLAW):
    """Power-law row degrees, uniform columns, integer weights in [1,16] (config 5)."""
    cols = rows if cols is None else cols
    rp = np.empty(rows + 1, np.int32)
    ci = np.empty(nnz, np.int32)
    va = np.empty(nnz, np.float32)
    rc = load().sh_synth_powerlaw(rows, cols, nnz, exponent, min(dmax, cols), seed, _p(rp), _p(ci), _p(va))
    if rc:
        raise RuntimeError(f"sh_synth_powerlaw failed: {rc}")
    return rp, ci, va


def rmat(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=SEED_RMAT, permute=True):
    """Graph500 R-MAT (configs 3/4): 2^scale rows, edge_factor*2^scale entries, duplicates kept."""
    n = 1 << scale
    m = n * edge_factor
    rp = np.empty(n + 1, np.int32)
    ci = np.empty(m, np.int32)
    va = np.empty(m, np.float32)
    rc = load().sh_synth_rmat(scale, edge_factor, a, b, c, seed, int(permute), _p(rp), _p(ci), _p(va))
    if rc:
        raise RuntimeError(f"sh_synth_rmat failed: {rc}")
    return rp, ci, va


def scircuit_like(seed=SEED_SCIRCUIT):
    """Stand-in of SuiteSparse scircuit's shape (config 2): 170 998 rows, 958 936 entries, max row 353."""
    return powerlaw(170_998, 958_936, dmax=353, seed=seed)


NORM_NONE, NORM_PAGERANK, NORM_SCC = 0, 1, 2


def mm_load(path, elem_is_int=False, truncate=True, normalise=NORM_NONE, damping=0.85):
    """MatrixMarket -> (rows, cols, header_nnz, row_ptr, col_idx, val) through the product loader;
    normalise applies SparseMatrix::pagerank_normalise / scc_normalise as the pr / scc apps do."""
    m = _HostCsr()
    rc = load().sh_mm_load_ex(os.fsencode(path), int(elem_is_int), int(truncate), int(normalise), float(damping),
                              C.byref(m))
    if rc:
        raise RuntimeError(f"sh_mm_load({path}) failed: {rc}")
    try:
        rp = np.ctypeslib.as_array(m.row_ptr, (m.rows + 1,)).copy()
        ci = np.ctypeslib.as_array(m.col_idx, (max(m.nnz, 1),))[:m.nnz].copy()
        vt = C.c_int32 if elem_is_int else C.c_float
        va = np.ctypeslib.as_array(C.cast(m.val, C.POINTER(vt)), (max(m.nnz, 1),))[:m.nnz].copy()
    finally:
        load().sh_host_csr_release(C.byref(m))
    return m.rows, m.cols, m.header_nnz, rp, ci, va
