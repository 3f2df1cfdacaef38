"""The device-side builder of the x-tiled layout (csrc/plan_gpu.hip) against the host builder (plan_host.h::
build_tiled_plan): every array of the layout must come out the same, byte for byte -- the host builder is itself
checked by the host emulator (tests/test_plan_cpu.py) and by the parity tests, so equality carries all of that over.
The comparison lives in the tools build of the engine (sh_debug_compare_builds, `make -C sparseharness_amd/csrc emulate`);
the product-level tests below then upload through the ordinary ABI with sh_plan_options::build = 2."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sparseharness_amd import abi
from sparseharness_amd import hostlib as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sparseharness_amd", "csrc")
LIB = os.path.join(ROOT, "sparseharness_amd", "variants", "emulate.so")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def tools():
    """(library, engine handle) of the tools build: its own engine, since the comparison runs inside that library."""
    rc = subprocess.call(["make", "-s", "-C", CSRC, "emulate"])   # (rebuilds only when a source is newer than the library)
    assert rc == 0 or os.path.exists(LIB)
    lib = C.CDLL(LIB)
    lib.sh_plan_options_default.argtypes = [C.POINTER(abi.sh_plan_options)]
    lib.sh_engine_create.restype = C.c_int
    lib.sh_engine_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.sh_engine_destroy.argtypes = [C.c_void_p]
    lib.sh_debug_compare_builds.restype = C.c_int
    lib.sh_debug_compare_builds.argtypes = [C.c_void_p] + [C.c_int64] * 3 + [C.c_void_p] * 3 + [C.POINTER(abi.sh_plan_options), C.c_char_p, C.c_int64]
    lib.sh_debug_compare_bits_builds.restype = C.c_int
    lib.sh_debug_compare_bits_builds.argtypes = [C.c_void_p] + [C.c_int64] * 3 + [C.c_void_p] * 3 + [C.c_char_p, C.c_int64]
    e = C.c_void_p()
    assert lib.sh_engine_create(0, C.byref(e)) == 0
    yield lib, e
    lib.sh_engine_destroy(e)


def compare(tools, rows, cols, rp, ci, va, **options):
    lib, e = tools
    opt = abi.sh_plan_options()
    lib.sh_plan_options_default(C.byref(opt))
    opt.plan = 2
    for k, v in options.items():
        setattr(opt, k, v)
    rp, ci, va = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(va)
    buf = C.create_string_buffer(4096)
    rc = lib.sh_debug_compare_builds(e, rows, cols, len(ci), _p(rp), _p(ci), _p(va), C.byref(opt), buf, len(buf))
    return rc, buf.value.decode()


def random_matrix(rng, rows, cols, avg, heavy=0, hlen=3000, oob=False, local=False, values="int16"):
    deg = rng.poisson(avg, rows).astype(np.int64)
    deg[rng.integers(0, rows, max(1, rows // 10))] = 0                     # empty rows
    for h in rng.integers(0, rows, heavy):
        deg[h] = hlen + rng.integers(0, 500)                               # heavy rows (>= 512 entries and >= 8 per tile)
    for h in rng.integers(0, rows, max(1, rows // 50)):
        deg[h] = rng.integers(30, 400)                                     # medium rows: many entries per (row, tile)
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    nnz = int(rp[-1])
    ci = rng.integers(0, cols, nnz).astype(np.int32)
    if local:
        ci = ((np.repeat(np.arange(rows), deg) * (cols / rows)).astype(np.int64) + rng.integers(-50, 50, nnz)).clip(0, cols - 1).astype(np.int32)
    if oob and nnz:
        ci[rng.integers(0, nnz, max(1, nnz // 100))] = rng.choice([-1, cols, cols + 5, 2 ** 31 - 1])
    if values == "int16":
        va = rng.integers(1, 17, nnz).astype(np.float32)                   # 16 finite values, no zero: four-bit codes, padding borrows code 0
    elif values == "few":
        va = rng.integers(0, 9, nnz).astype(np.float32)                    # zero among them
    elif values == "bytes":
        va = rng.integers(0, 200, nnz).astype(np.float32)                  # one-byte codes
    elif values == "words":
        va = rng.integers(0, 3000, nnz).astype(np.float32)                 # two-byte codes (value_coding = 8 in OPTIONS: raw)
    elif values == "words_full":
        va = (1 + rng.integers(0, 4095, nnz)).astype(np.float32)           # 4095 values + the padding word: the table is full
        va[:4095] = 1 + np.arange(4095)
    else:
        va = rng.random(nnz).astype(np.float32)                            # raw
    return rp, ci, va


SHAPES = [  # rows, cols, avg degree, heavy rows, extra
    (3000, 3000, 8, 2, {}), (3000, 100_000, 12, 3, {}), (20_000, 200_000, 10, 4, {}), (500, 70_000, 40, 5, dict(hlen=9000)),
    (3000, 100_000, 12, 3, dict(oob=True)), (8000, 40_000, 15, 0, dict(local=True)), (100, 33_000, 3, 0, {}), (1, 5, 3, 0, {}),
    (40_000, 1_000_000, 14, 6, dict(hlen=20_000)), (3000, 100_000, 12, 3, dict(values="few")), (3000, 100_000, 12, 3, dict(values="bytes")),
    (3000, 100_000, 12, 3, dict(values="real")), (3000, 100_000, 12, 3, dict(values="words")), (20_000, 200_000, 10, 4, dict(values="words_full")),
]
OPTIONS = [dict(fold=1), dict(fold=0), dict(fold=1, value_coding=-1), dict(fold=1, value_coding=8), dict(fold=1, chunk=2048), dict(fold=1, heavy_per_tile=2)]


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_device_builder_equals_host_builder(tools, shape):
    rows, cols, avg, heavy, kw = SHAPES[shape]
    rng = np.random.default_rng(300 + shape)
    rp, ci, va = random_matrix(rng, rows, cols, avg, heavy, **kw)
    for options in OPTIONS:
        rc, report = compare(tools, rows, cols, rp, ci, va, **options)
        assert rc == 0, (shape, options, rc, report)


def test_device_builder_equals_host_builder_powerlaw_2m(tools):
    """A power-law matrix big enough for several bins per tile and heavy rows cut over wave boundaries."""
    rows, nnz = 400_000, 8_000_000
    rp, ci, va = H.powerlaw(rows, nnz)
    for options in (dict(fold=1), dict(fold=0, value_coding=-1)):
        rc, report = compare(tools, rows, rows, rp, ci, va, **options)
        assert rc == 0, (options, rc, report)


def test_device_builder_equals_host_builder_rmat(tools):
    rp, ci, va = H.rmat(18, seed=5)
    rc, report = compare(tools, 1 << 18, 1 << 18, rp, ci, va)
    assert rc == 0, (rc, report)


def test_device_bits_builder_equals_host_builder(tools):
    """The bit-blocked (or,and) layout: entries, work items and sub-range offsets of both builders are the same bytes."""
    lib, e = tools
    rng = np.random.default_rng(77)
    cases = [random_matrix(rng, 3000, 100_000, 12, 3, values="few") + (3000, 100_000),
             random_matrix(rng, 600_000, 1_500_000, 6, 2, oob=True, values="few") + (600_000, 1_500_000)]
    rp, ci, va = H.rmat(18, seed=5)
    cases.append((rp, ci, va, 1 << 18, 1 << 18))
    for rp, ci, va, rows, cols in cases:
        rp, ci, vi = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(va).astype(np.int32)
        buf = C.create_string_buffer(2048)
        rc = lib.sh_debug_compare_bits_builds(e, rows, cols, len(ci), _p(rp), _p(ci), _p(vi), buf, len(buf))
        assert rc == 0, (rows, cols, rc, buf.value.decode())


def test_device_builder_through_the_abi():
    """build = 2 through sh_csr_upload_ex: the matrix reports the device builder, has the same description and
    footprint as the host-built one, and all four semirings give the same bits."""
    from oracle import oracle as O
    from sparseharness_amd.engine import Engine
    rows, nnz = 300_000, 6_000_000
    rp, ci, va = H.powerlaw(rows, nnz)
    x = (1 + np.arange(rows) % 7).astype(np.float32)
    with Engine(0) as eng:
        Ah = eng.upload_csr(rows, rows, rp, ci, va, plan=2, build=1)
        Ad = eng.upload_csr(rows, rows, rp, ci, va, plan=2, build=2)
        assert Ah.builder()[0] == "host"
        assert Ad.builder() == ("device", ""), Ad.builder()
        assert Ad.describe() == Ah.describe()
        assert Ad.footprint() == Ah.footprint()
        for sem in (O.PLUS_TIMES_F32, O.MIN_PLUS_F32, O.OR_AND_I32, O.MAX_MIN_I32):
            integer = sem in (O.OR_AND_I32, O.MAX_MIN_I32)
            xs = (np.arange(rows) % 3 == 0).astype(np.int32) if integer else x
            vals = va.astype(np.int32) if integer else va
            if integer:
                Mh = eng.upload_csr(rows, rows, rp, ci, vals, plan=2, build=1)
                Md = eng.upload_csr(rows, rows, rp, ci, vals, plan=2, build=2)
            else:
                Mh, Md = Ah, Ad
            dt = np.int32 if integer else np.float32
            ys = (np.arange(rows) % 5).astype(dt)
            xv, yv, oh, od = eng.vector(xs), eng.vector(ys), eng.alloc(rows), eng.alloc(rows)
            alpha, beta = {O.PLUS_TIMES_F32: (1.0, 0.5), O.MIN_PLUS_F32: (0.0, 0.0), O.OR_AND_I32: (1, 1), O.MAX_MIN_I32: (1, 1)}[sem]
            eng.spmv(sem, Mh, xv, yv, alpha, beta, oh)
            eng.spmv(sem, Md, xv, yv, alpha, beta, od)
            got_h, got_d = oh.download(dt), od.download(dt)
            np.testing.assert_array_equal(got_d.view(np.uint32), got_h.view(np.uint32))
            want = O.kernel(sem, rp, ci, vals, xs, ys, alpha, beta)
            np.testing.assert_array_equal(got_d, want)
