"""Thin Python face of the C ABI (include/sparseharness_hip.h) for tests and bench.

Everything here forwards to the HIP engine through ctypes; there is no Python
or CPU compute path.  Names follow the reference's domain: matrix, x/y/output
vectors, semiring, run (launch geometry), trials.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import MAX_MIN_I32, MIN_PLUS_F32, OR_AND_I32, PLUS_TIMES_F32  # noqa: F401

FLT_MAX = np.float32(3.4028235e38)


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[sh error {code}] {msg}")
        self.code = code


def elem_dtype(semiring):
    return np.int32 if semiring in (OR_AND_I32, MAX_MIN_I32) else np.float32


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Vec:
    """Device vector of 4-byte elements (x, y, output of the harness)."""

    def __init__(self, engine, handle, owned=True):
        self.engine, self.h, self.owned = engine, handle, owned

    def __len__(self):
        return abi.load().sh_vec_len(self.h)

    @property
    def device_ptr(self):
        return abi.load().sh_vec_device_ptr(self.h)

    def upload(self, host):
        host = np.ascontiguousarray(host)
        assert host.dtype.itemsize == 4
        self.engine._chk(abi.load().sh_vec_upload(self.engine.h, self.h, _ptr(host), host.size))
        return self

    def download(self, dtype=np.float32, n=None):
        n = len(self) if n is None else n
        out = np.empty(n, dtype)
        self.engine._chk(abi.load().sh_vec_download(self.engine.h, self.h, _ptr(out), n))
        return out

    def fill(self, value, dtype=np.float32):
        pat = int(np.array([value], dtype).view(np.uint32)[0])
        self.engine._chk(abi.load().sh_vec_fill(self.engine.h, self.h, pat))
        return self

    def free(self):
        if self.h is not None:
            abi.load().sh_vec_free(self.engine.h, self.h)
            self.h = None


class CsrMatrix:
    """Device-resident CSR matrix + launch schedule (replaces cl_encode's buffers)."""

    def __init__(self, engine, handle, rows, cols, nnz):
        self.engine, self.h = engine, handle
        self.rows, self.cols, self.nnz = rows, cols, nnz

    def algorithmic_bytes(self, reads_y=False):
        b = C.c_uint64()
        self.engine._chk(abi.load().sh_csr_algorithmic_bytes(self.h, int(reads_y), C.byref(b)))
        return b.value

    def plan(self):
        """('stream'|'tiled'|'bits', HBM bytes one SpMV streams by construction)."""
        p, b = C.c_int32(), C.c_uint64()
        self.engine._chk(abi.load().sh_csr_plan(self.h, C.byref(p), C.byref(b)))
        return ("stream", "tiled", "bits")[p.value], b.value

    def describe(self):
        """One-line description of the device layout (plan, value coding, tile/bin counts)."""
        buf = C.create_string_buffer(256)
        self.engine._chk(abi.load().sh_csr_describe(self.h, buf, len(buf)))
        return buf.value.decode()

    def footprint(self):
        """Device bytes held by this matrix."""
        b = C.c_uint64()
        self.engine._chk(abi.load().sh_csr_footprint(self.h, C.byref(b)))
        return b.value

    def builder(self):
        """("host" | "device", note): who built the tiled layout, and why the device builder was not used if asked for."""
        w, note = C.c_int32(), C.create_string_buffer(256)
        self.engine._chk(abi.load().sh_csr_builder(self.h, C.byref(w), note, len(note)))
        return ("device" if w.value else "host"), note.value.decode()

    def placement(self):
        """(placements of the big arrays timed at upload, ms of the first, ms of the one kept)."""
        n, a, b = C.c_int32(), C.c_float(), C.c_float()
        self.engine._chk(abi.load().sh_csr_placement(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, round(a.value, 4), round(b.value, 4)

    def free(self):
        if self.h is not None:
            abi.load().sh_csr_free(self.engine.h, self.h)
            self.h = None


class Engine:
    """One HIP device + one stream (replaces Harness's OpenCL context/queue)."""

    def __init__(self, device=0, stream=None):
        lib = abi.load()
        h = C.c_void_p()
        if stream is None:
            rc = lib.sh_engine_create(device, C.byref(h))
        else:
            rc = lib.sh_engine_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if rc != abi.SH_OK:
            raise EngineError(rc, (lib.sh_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def _chk(self, rc):
        if rc != abi.SH_OK:
            raise EngineError(rc, (abi.load().sh_last_error(self.h) or b"").decode())

    def close(self):
        if self.h is not None:
            abi.load().sh_engine_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        self._chk(abi.load().sh_engine_device_name(self.h, buf, 256))
        return buf.value.decode()

    def max_alloc(self):
        b = C.c_uint64()
        self._chk(abi.load().sh_engine_max_alloc(self.h, C.byref(b)))
        return b.value

    def synchronize(self):
        self._chk(abi.load().sh_engine_synchronize(self.h))

    # ---- buffers
    def upload_csr(self, rows, cols, row_ptr, col_idx, val, **options):
        """options: fields of sh_plan_options (plan=0|1|2, value_coding=0|8|-1, fold=0, ...) on top of the
        SH_* environment; without any the environment alone decides (sh_csr_upload)."""
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        val = np.ascontiguousarray(val)
        assert val.dtype.itemsize == 4
        nnz = int(row_ptr[-1]) if len(row_ptr) else 0
        h = C.c_void_p()
        lib = abi.load()
        if options:
            opt = abi.sh_plan_options()
            lib.sh_plan_options_from_env(C.byref(opt))
            for k, v in options.items():
                setattr(opt, k, v)
            self._chk(lib.sh_csr_upload_ex(self.h, rows, cols, nnz, _ptr(row_ptr), _ptr(col_idx), _ptr(val),
                                           C.byref(opt), C.byref(h)))
        else:
            self._chk(lib.sh_csr_upload(self.h, rows, cols, nnz, _ptr(row_ptr), _ptr(col_idx), _ptr(val), C.byref(h)))
        return CsrMatrix(self, h, rows, cols, nnz)

    def alloc(self, n):
        h = C.c_void_p()
        self._chk(abi.load().sh_vec_alloc(self.h, n, C.byref(h)))
        return Vec(self, h)

    def wrap(self, device_ptr, n):
        h = C.c_void_p()
        self._chk(abi.load().sh_vec_wrap(self.h, C.c_void_p(device_ptr), n, C.byref(h)))
        return Vec(self, h, owned=False)

    def vector(self, host):
        host = np.ascontiguousarray(host)
        return self.alloc(host.size).upload(host)

    # ---- hot path
    def spmv(self, semiring, A, x, y, alpha, beta, out, timed=False, run=None):
        dt = elem_dtype(semiring)
        a, b = np.array([alpha], dt), np.array([beta], dt)
        ns = C.c_uint64()
        launch = None
        if run is not None:
            launch = abi.sh_launch()
            launch.global_[:] = run[:3]
            launch.local[:] = run[3:]
        self._chk(abi.load().sh_spmv(self.h, semiring, A.h, x.h, None if y is None else y.h, _ptr(a),
                                     _ptr(b), out.h, launch, C.byref(ns) if timed else None))
        return ns.value if timed else None

    def step(self, semiring, A, x, y, alpha, beta, out, x_row_offset=0, delta=1e-4, changed_ptr=None):
        dt = elem_dtype(semiring)
        a, b = np.array([alpha], dt), np.array([beta], dt)
        self._chk(abi.load().sh_spmv_step(self.h, semiring, A.h, x.h, None if y is None else y.h, _ptr(a),
                                          _ptr(b), out.h, x_row_offset, delta,
                                          None if changed_ptr is None else C.c_void_p(changed_ptr)))

    def iterate(self, semiring, A, x, y0, scratch, alpha, beta, delta=1e-4, max_iters=10000):
        dt = elem_dtype(semiring)
        a, b = np.array([alpha], dt), np.array([beta], dt)
        iters, conv, total = C.c_int32(), C.c_int32(), C.c_uint64()
        per = (C.c_uint64 * max_iters)()
        self._chk(abi.load().sh_iterate(self.h, semiring, A.h, x.h, y0.h, scratch.h, _ptr(a), _ptr(b),
                                        delta, max_iters, None, C.byref(iters), C.byref(conv), per,
                                        C.byref(total)))
        return iters.value, bool(conv.value), list(per[:iters.value]), total.value
