"""ctypes/numpy wrapper around oracle/libsh_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by sparseharness_amd (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsh_oracle.so")

PLUS_TIMES_F32, MIN_PLUS_F32, OR_AND_I32, MAX_MIN_I32 = 0, 1, 2, 3
NORM_NONE, NORM_PAGERANK, NORM_SCC = 0, 1, 2
INT_MIN, INT_MAX = -2**31, 2**31 - 1
FLT_MAX = np.float32(3.4028235e38)
CORRECT, NOT_CHECKED, BAD_LENGTH, BAD_VALUES = 0, 1, 3, 4

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsh_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "sh_oracle.c")
        if (not os.path.exists(_LIB_PATH)
                or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_mm_load.restype = C.c_int
        _lib.oracle_mm_load_ex.restype = C.c_int
        _lib.oracle_kernel.restype = C.c_int
        _lib.oracle_iterate.restype = C.c_int
        _lib.oracle_check_result_f32.restype = C.c_int
        _lib.oracle_gold_spmv_f32.restype = None
        _lib.oracle_gold_dot_f32.restype = None
        _lib.oracle_free.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def elem_dtype(semiring):
    return np.int32 if semiring in (OR_AND_I32, MAX_MIN_I32) else np.float32


def mm_load(path, elem_is_int=False, normalise=NORM_NONE, damping=0.85):
    """MatrixMarket -> (rows, cols, hdr_nnz, row_ptr, col_idx, val) with the reference's quirks;
    normalise = NORM_PAGERANK / NORM_SCC applies the app's normaliser before the int narrowing."""
    rows, cols, hdr = C.c_int32(), C.c_int32(), C.c_int32()
    nnz = C.c_int64()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib().oracle_mm_load_ex(os.fsencode(path), C.c_int(int(elem_is_int)), C.c_int(normalise),
                                 C.c_double(damping), C.byref(rows), C.byref(cols), C.byref(hdr),
                                 C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(va))
    if rc != 0:
        raise RuntimeError(f"oracle_mm_load({path}) failed: {rc}")
    n = nnz.value
    row_ptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int32)), (rows.value + 1,)).copy()
    col_idx = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), (max(n, 1),))[:n].copy()
    vt = C.c_int32 if elem_is_int else C.c_float
    val = np.ctypeslib.as_array(C.cast(va, C.POINTER(vt)), (max(n, 1),))[:n].copy()
    for ptr in (rp, ci, va):
        lib().oracle_free(ptr)
    return rows.value, cols.value, hdr.value, row_ptr, col_idx, val


def gold_spmv(row_ptr, col_idx, val, x, y_const=0.0, alpha=1.0, beta=0.0, zero=0.0):
    rows = len(row_ptr) - 1
    out = np.empty(rows, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    lib().oracle_gold_spmv_f32(C.c_int32(rows), _p(row_ptr), _p(col_idx), _p(val), _p(x),
                               C.c_float(y_const), C.c_float(alpha), C.c_float(beta),
                               C.c_float(zero), _p(out))
    return out


def gold_dot(row_ptr, col_idx, val, x, alpha=1.0, out=None):
    rows = len(row_ptr) - 1
    if out is None:
        out = np.empty(rows, np.float32)
    lib().oracle_gold_dot_f32(C.c_int64(rows), _p(row_ptr), _p(col_idx), _p(val), _p(x),
                              C.c_float(alpha), _p(out))
    return out


def gold_dot_all_cores(row_ptr, col_idx, val, x, alpha=1.0, out=None, threads=None):
    """The same dot loop row-parallel under OpenMP (rows still summed sequentially: same bits).
    Returns (result, threads that ran)."""
    import os
    rows = len(row_ptr) - 1
    if out is None:
        out = np.empty(rows, np.float32)
    fn = lib().oracle_gold_dot_f32_omp
    fn.restype = C.c_int
    used = fn(C.c_int64(rows), _p(row_ptr), _p(col_idx), _p(val), _p(x), C.c_float(alpha), _p(out),
              C.c_int(threads or os.cpu_count() or 1))
    return out, int(used)


def kernel(semiring, row_ptr, col_idx, val, x, y, alpha, beta, vlength=None):
    """One launch of the Lift glb-sdp kernel semantics for `semiring`."""
    dt = elem_dtype(semiring)
    rows = len(row_ptr) - 1
    val = np.ascontiguousarray(val, dt)
    x = np.ascontiguousarray(x, dt)
    y = np.ascontiguousarray(y, dt)
    a, b = np.array([alpha], dt), np.array([beta], dt)
    out = np.empty(rows, dt)
    rc = lib().oracle_kernel(C.c_int(semiring), C.c_int32(rows), _p(row_ptr), _p(col_idx),
                             _p(val), _p(x), _p(y), _p(a), _p(b),
                             C.c_int32(len(x) if vlength is None else vlength), _p(out))
    assert rc == 0
    return out


def iterate(semiring, row_ptr, col_idx, val, x0, y0, alpha, beta, delta=1e-4, max_iters=10000):
    """SSSP/BFS do-while loop; returns (final, iters, converged)."""
    dt = elem_dtype(semiring)
    rows = len(row_ptr) - 1
    val = np.ascontiguousarray(val, dt)
    x = np.array(x0, dt, copy=True)
    y = np.ascontiguousarray(y0, dt)
    scratch = np.zeros(rows, dt)
    a, b = np.array([alpha], dt), np.array([beta], dt)
    it, conv = C.c_int32(), C.c_int32()
    rc = lib().oracle_iterate(C.c_int(semiring), C.c_int32(rows), _p(row_ptr), _p(col_idx),
                              _p(val), _p(x), _p(y), _p(scratch), _p(a), _p(b),
                              C.c_double(delta), C.c_int32(max_iters), C.byref(it), C.byref(conv))
    assert rc == 0
    return x, it.value, bool(conv.value)


def check_result(gold, res):
    gold = np.ascontiguousarray(gold, np.float32)
    res = np.ascontiguousarray(res, np.float32)
    return lib().oracle_check_result_f32(_p(gold), C.c_int64(len(gold)), _p(res),
                                         C.c_int64(len(res)))


def initial_vector(semiring, n):
    """x0 == y0 of the iterative apps (app/sssp.cpp:179-209, app/bfs.cpp:177-207)."""
    if semiring == MIN_PLUS_F32:
        v = np.full(n, FLT_MAX, np.float32)
        v[0] = 0.0
    elif semiring == OR_AND_I32:
        v = np.zeros(n, np.int32)
        v[0] = 1
    elif semiring == MAX_MIN_I32:   # app/scc.cpp:176-185: x[i] = i (y0 is INT_MIN everywhere)
        v = np.arange(n, dtype=np.int32)
    else:
        v = np.ones(n, np.float32)
    return v
