"""The x-tiled two-phase plan, checked WITHOUT a GPU: a tools build of the engine (-DSH_PLAN_EMULATE) exports
sh_debug_emulate_plan, which builds the device layout exactly as sh_csr_upload does and then walks both phases on
the host through the very tables the kernels read (fold flags, obase, gdest, gblk / ptab, pslot, lrp) with the
kernels' indexing.  Comparing its result with a plain CSR product checks the layout builder: pair folding (pairs
laid out column-wise over an even / odd lane pair, singles and padding filling the other columns), piece tables,
heavy strips and their partial slots.  The device code itself is covered by the -m gpu parity tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sparseharness_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sparseharness_amd", "csrc")
LIB = os.path.join(ROOT, "sparseharness_amd", "variants", "emulate.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", CSRC, "emulate"])   # (rebuilds only when a source is newer)
    lib = C.CDLL(LIB)
    lib.sh_plan_options_default.argtypes = [C.POINTER(abi.sh_plan_options)]
    lib.sh_debug_emulate_plan.restype = C.c_int
    lib.sh_debug_emulate_plan.argtypes = [C.c_int64] * 3 + [C.c_void_p] * 3 + [C.POINTER(abi.sh_plan_options), C.c_int,
                                                                                C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sh_debug_emulate_bits.restype = C.c_int
    lib.sh_debug_emulate_bits.argtypes = [C.c_int64] * 3 + [C.c_void_p] * 6
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def emulate(lib, rows, cols, rp, ci, va, sem, **options):
    opt = abi.sh_plan_options()
    lib.sh_plan_options_default(C.byref(opt))
    opt.plan = 2
    for k, v in options.items():
        setattr(opt, k, v)
    x = (1 + np.arange(cols) % 7).astype(np.float32) if sem == 0 else (np.arange(cols) % 3 == 0).astype(np.int32)
    y = np.zeros(rows, np.float32 if sem == 0 else np.int32)
    st = np.zeros(8, np.int64)
    rc = lib.sh_debug_emulate_plan(rows, cols, len(ci), _p(rp), _p(ci), _p(va), C.byref(opt), sem, _p(x), _p(y), _p(st))
    return rc, y, dict(zip(("stream", "light", "products", "bins", "chunks", "heavy_rows", "poison_reads", "tiles"), st.tolist())), x


def exact(rows, cols, rp, ci, va, x, sem):
    ok = (ci >= 0) & (ci < cols)
    xv = np.where(ok, x[np.clip(ci, 0, cols - 1)], 0)
    row_of = np.repeat(np.arange(rows), np.diff(rp))
    if sem == 0:
        out = np.zeros(rows)
        np.add.at(out, row_of, xv.astype(np.float64) * va.astype(np.float64))
        return out.astype(np.float32)
    out = np.zeros(rows, np.int64)
    np.add.at(out, row_of, ((xv != 0) & (va != 0)).astype(np.int64))
    return (out > 0).astype(np.int32)


def random_matrix(rng, rows, cols, avg, heavy=0, hlen=3000, oob=False, local=False):
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
    va = rng.integers(1, 17, nnz).astype(np.float32)
    return rp, ci, va


SHAPES = [  # rows, cols, avg degree, heavy rows, extra
    (3000, 3000, 8, 2, {}), (3000, 100_000, 12, 3, {}), (20_000, 200_000, 10, 4, {}), (500, 70_000, 40, 5, dict(hlen=9000)),
    (3000, 100_000, 12, 3, dict(oob=True)), (8000, 40_000, 15, 0, dict(local=True)), (100, 33_000, 3, 0, {}), (1, 5, 3, 0, {}),
    (40_000, 1_000_000, 14, 6, dict(hlen=20_000)),
]
OPTIONS = [dict(fold=1), dict(fold=0), dict(fold=1, value_coding=-1), dict(fold=1, value_coding=8), dict(fold=1, chunk=2048)]


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_emulated_plan_equals_csr_product(emu, shape):
    rows, cols, avg, heavy, kw = SHAPES[shape]
    rng = np.random.default_rng(100 + shape)
    rp, ci, va = random_matrix(rng, rows, cols, avg, heavy, **kw)
    for sem in (0, 2):
        vals = va if sem == 0 else va.astype(np.int32)
        for options in OPTIONS:
            rc, y, st, x = emulate(emu, rows, cols, rp, ci, vals, sem, **options)
            assert rc == 0, (rc, options, st)
            assert st["poison_reads"] == 0, (options, st)            # every word phase 2 read had been written by phase 1
            np.testing.assert_array_equal(y, exact(rows, cols, rp, ci, vals, x, sem), err_msg=str((sem, options, st)))
            if options.get("fold") == 0:
                assert st["products"] >= st["light"]                  # one product per light entry (+ padding)


@pytest.mark.parametrize("distinct", [300, 4095, 4096, 6000])
def test_two_byte_value_codes(emu, distinct):
    """More than 256 distinct value words: up to 4096 (the padding word 0 included) travel as two-byte codes into a
    dictionary that sits in LDS beside the x tile (VC = 3 of spmv_tiled_phase1); beyond that the stream carries the raw
    values.  Light groups, folded pairs and heavy strips through either stream, walked on the host."""
    rng = np.random.default_rng(distinct)
    rp, ci, va = random_matrix(rng, 6000, 120_000, 12, 3)
    va = (1 + rng.integers(0, distinct, len(va))).astype(np.float32)
    va[:distinct] = 1 + np.arange(distinct)          # every value occurs
    for sem in (0, 2):
        vals = va if sem == 0 else va.astype(np.int32)
        for options in (dict(fold=1), dict(fold=0), dict(fold=1, value_coding=8)):
            rc, y, st, x = emulate(emu, 6000, 120_000, rp, ci, vals, sem, **options)
            assert rc == 0 and st["poison_reads"] == 0, (rc, options, st)
            want = exact(6000, 120_000, rp, ci, vals, x, sem)
            if sem == 0:   # (the three 3000-entry rows sum past 2^24 with values this large: float order matters there)
                big = np.diff(rp) * distinct * 7 >= 2 ** 24
                np.testing.assert_array_equal(y[~big], want[~big], err_msg=str((sem, options)))
                np.testing.assert_allclose(y[big], want[big], rtol=1e-6)
            else:
                np.testing.assert_array_equal(y, want, err_msg=str((sem, options)))


def test_folding_removes_the_duplicates_of_a_row_inside_a_tile(emu):
    """A matrix whose rows keep their columns within one tile: with folding a row of d entries travels through P as
    ceil(d / 2) products (+ padding), without as d."""
    rng = np.random.default_rng(7)
    rows, cols = 5000, 30_000          # one column tile
    deg = rng.integers(1, 33, rows).astype(np.int64)
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.integers(0, cols, int(rp[-1])).astype(np.int32)
    va = rng.integers(1, 17, int(rp[-1])).astype(np.float32)
    rc, y, st, x = emulate(emu, rows, cols, rp, ci, va, 0, fold=1)
    assert rc == 0 and st["poison_reads"] == 0
    np.testing.assert_array_equal(y, exact(rows, cols, rp, ci, va, x, 0))
    want = int(np.ceil(deg / 2).sum())
    assert want <= st["products"] <= want + 8 * st["bins"] * st["tiles"]
    rc, y0, st0, _ = emulate(emu, rows, cols, rp, ci, va, 0, fold=0)
    assert rc == 0 and st0["products"] >= int(deg.sum())
    np.testing.assert_array_equal(y0, y)


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_emulated_bit_plan_equals_boolean_product(emu, shape):
    """The (or,and) semiring on bits (bits.hip.h): blocks of 262144 rows x 524288 columns, 4-byte coordinate entries in
    8192-row sub-ranges padded to octets with copies, work items, partial bitmaps -- walked on the host and compared with the
    boolean product, entries with a zero value or a column out of range included (they never contribute)."""
    rows, cols, avg, heavy, kw = SHAPES[shape]
    rng = np.random.default_rng(300 + shape)
    rp, ci, va = random_matrix(rng, rows, cols, avg, heavy, **kw)
    vals = va.astype(np.int32)
    vals[rng.integers(0, max(len(vals), 1), len(vals) // 7)] = 0          # dead entries
    for density in (0.0, 0.02, 0.5, 1.0):
        x = (rng.random(cols) < density).astype(np.int32) * rng.integers(1, 9, cols).astype(np.int32)
        y = np.full(rows, -1, np.int32)
        st = np.zeros(8, np.int64)
        rc = emu.sh_debug_emulate_bits(rows, cols, len(ci), _p(rp), _p(ci), _p(vals), _p(x), _p(y), _p(st))
        assert rc == 0, rc
        np.testing.assert_array_equal(y, exact(rows, cols, rp, ci, vals, x, 2))
        live = int(((vals != 0) & (ci >= 0) & (ci < cols)).sum())
        assert st[1] == live and live <= st[0] <= live + 8 + 7 * 32 * st[3] * st[4]


def test_bit_plan_on_a_matrix_spanning_several_blocks(emu):
    """3 row ranges x 3 column blocks, a hub row, empty blocks (no work item for them)."""
    rng = np.random.default_rng(11)
    rows, cols = 600_000, 1_200_000
    deg = rng.poisson(3, rows).astype(np.int64)
    deg[123_456] = 300_000
    deg[300_000:400_000] = 0
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.integers(0, cols, int(rp[-1])).astype(np.int32)
    ci[rp[500_000]:] = rng.integers(0, 400_000, int(rp[-1] - rp[500_000])).astype(np.int32)   # the last rows touch one column block only
    vals = np.ones(int(rp[-1]), np.int32)
    x = (rng.random(cols) < 0.01).astype(np.int32)
    y = np.full(rows, -1, np.int32)
    st = np.zeros(8, np.int64)
    assert emu.sh_debug_emulate_bits(rows, cols, len(ci), _p(rp), _p(ci), _p(vals), _p(x), _p(y), _p(st)) == 0
    np.testing.assert_array_equal(y, exact(rows, cols, rp, ci, vals, x, 2))
    assert st[3] == 3 and st[4] == 3 and st[2] <= 9


def test_emulated_plan_on_drawn_small_shapes(emu):
    """Shapes nobody wrote down: hypothesis draws small matrices -- empty ones, a single row or column, all entries in
    one row, duplicate (row, column) entries, columns out of range, rows just below and above the heavy threshold -- and
    every one must come out of the walked layout exactly as out of the plain CSR product, under folding on and off."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import strategies as S

    @hyp.settings(max_examples=400, deadline=None, derandomize=True,
                  suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(rows=S.integers(1, 400), cols=S.sampled_from([1, 2, 7, 300, 32_759, 32_760, 32_761, 65_520, 70_001]),
               kind=S.sampled_from(["sparse", "dense_rows", "one_row", "duplicates", "empty", "near_heavy"]),
               seed=S.integers(0, 2 ** 31 - 1), fold=S.sampled_from([0, 1]), oob=S.booleans())
    def run(rows, cols, kind, seed, fold, oob):
        rng = np.random.default_rng(seed)
        if kind == "empty":
            deg = np.zeros(rows, np.int64)
        elif kind == "one_row":
            deg = np.zeros(rows, np.int64)
            deg[rng.integers(0, rows)] = rng.integers(1, 5000)
        elif kind == "dense_rows":
            deg = rng.integers(0, 60, rows).astype(np.int64)
        elif kind == "near_heavy":
            deg = rng.poisson(3, rows).astype(np.int64)
            thr = max(512, 8 * ((cols + 32_759) // 32_760))
            for h in rng.integers(0, rows, 3):
                deg[h] = thr + rng.integers(-2, 3)
        else:
            deg = rng.poisson(4, rows).astype(np.int64)
        rp = np.zeros(rows + 1, np.int32)
        rp[1:] = np.cumsum(deg)
        nnz = int(rp[-1])
        ci = rng.integers(0, cols, nnz).astype(np.int32)
        if kind == "duplicates" and nnz:
            ci = (ci % min(cols, 3)).astype(np.int32)       # many entries of a row share a column
        if oob and nnz:
            ci[rng.integers(0, nnz, max(1, nnz // 20))] = rng.choice([-1, cols, 2 ** 31 - 1])
        va = rng.integers(1, 17, nnz).astype(np.float32)
        for sem in (0, 2):
            vals = va if sem == 0 else va.astype(np.int32)
            rc, y, st, x = emulate(emu, rows, cols, rp, ci, vals, sem, fold=fold)
            assert rc == 0, (rc, st)
            assert st["poison_reads"] == 0, st
            np.testing.assert_array_equal(y, exact(rows, cols, rp, ci, vals, x, sem), err_msg=str((rows, cols, kind, seed, fold, oob, sem, st)))
        # the same matrix through the bit-blocked (or,and) layout, a sparse and a dense x
        vals = va.astype(np.int32)
        for density in (0.05, 1.0):
            xb = (rng.random(cols) < density).astype(np.int32)
            yb = np.full(rows, -1, np.int32)
            stb = np.zeros(8, np.int64)
            assert emu.sh_debug_emulate_bits(rows, cols, nnz, _p(rp), _p(ci), _p(vals), _p(xb), _p(yb), _p(stb)) == 0
            np.testing.assert_array_equal(yb, exact(rows, cols, rp, ci, vals, xb, 2), err_msg=str((rows, cols, kind, seed, oob, density)))

    run()
