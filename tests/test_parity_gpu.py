"""GPU parity: the HIP path, called through the C ABI, against
  (a) the committed golden vectors produced by the real reference,
  (b) the pinned CPU oracle on the same seeded inputs,
  (c) size-independent properties at BASELINE.json's full sizes.
Integer-valued data (all reference fixtures, all synthetic weights) must match
bit-for-bit (the reference's own criterion, inc/harness.h:134); general float
data within 1e-5 relative (north_star); BFS and SSSP vectors and iteration
counts bit-exact.
"""
import os

import numpy as np
import pytest

from conftest import golden, mtx
from oracle import oracle as O
from sparseharness_amd import hostlib as H
from sparseharness_amd.engine import Engine, EngineError

pytestmark = pytest.mark.gpu

REL = 1e-5  # north_star tolerance for float SpMV


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(params=["stream", "tiled", "tiled-8bit", "tiled-raw", "tiled-nofold"], autouse=True)
def plan(request, monkeypatch):
    """Every parity test runs under both execution plans of the engine; the tiled plan with its value
    coding as the data allows (four-bit codes for <= 16 values, one-byte codes for <= 256), with one-byte
    codes at most, with raw values, and without folding a row's entries inside a tile in phase 1."""
    monkeypatch.setenv("SH_PLAN", request.param.split("-")[0])
    monkeypatch.setenv("SH_VALCODE", {"raw": "off", "8bit": "8"}.get(request.param.split("-")[-1], "auto"))
    monkeypatch.setenv("SH_FOLD", "0" if request.param.endswith("nofold") else "1")
    return request.param.split("-")[0]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run_spmv(eng, sr, rp, ci, va, x, y, alpha, beta, cols=None):
    rows = len(rp) - 1
    cols = len(x) if cols is None else cols
    dt = O.elem_dtype(sr)
    A = eng.upload_csr(rows, cols, rp, ci, np.ascontiguousarray(va, dt))
    xv = eng.vector(np.ascontiguousarray(x, dt))
    yv = None if y is None else eng.vector(np.ascontiguousarray(y, dt))
    out = eng.alloc(rows).fill(0)
    ns = eng.spmv(sr, A, xv, yv, alpha, beta, out, timed=True)
    res = out.download(dt)
    for v in (xv, yv, out):
        if v is not None:
            v.free()
    A.free()
    assert ns > 0
    return res


def assert_close(got, want):
    tol = REL * np.maximum(1.0, np.abs(want.astype(np.float64)))
    bad = np.abs(got.astype(np.float64) - want.astype(np.float64)) > tol
    assert not bad.any(), f"{bad.sum()} elements off, first at {np.argmax(bad)}: {got[np.argmax(bad)]} vs {want[np.argmax(bad)]}"


# ------------------------------------------------------------------ (a) golden
def test_spmv_matches_reference_gold_and_kernel(eng, matrix_name):
    g = golden(matrix_name)
    rows, cols, _, rp, ci, va = H.mm_load(mtx(matrix_name))
    n = rows
    x1 = np.ones(n, np.float32)
    xm = (1 + np.arange(n) % 7).astype(np.float32)
    ym = (np.arange(n) % 5).astype(np.float32)
    got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, x1, None, 1.0, 0.0)
    np.testing.assert_array_equal(bits(got), bits(g["gold_x1"]))       # Gold<float>::spmv
    np.testing.assert_array_equal(bits(got), bits(g["kern_spmv_x1"]))  # Lift kernel
    assert O.check_result(g["gold_x1"], got) == O.CORRECT
    got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, xm, None, 1.0, 0.0)
    np.testing.assert_array_equal(bits(got), bits(g["gold_xmod"]))
    got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, xm, ym, 2.0, 0.5)
    np.testing.assert_array_equal(bits(got), bits(g["kern_spmv_ab"]))


@pytest.mark.parametrize("sr,tag,a,b", [(O.MIN_PLUS_F32, "sssp", 0.0, 0.0), (O.OR_AND_I32, "bfs", 1, 0)])
def test_iterative_apps_match_reference(eng, matrix_name, sr, tag, a, b):
    g = golden(matrix_name)
    dt = O.elem_dtype(sr)
    rows, cols, _, rp, ci, va = H.mm_load(mtx(matrix_name), elem_is_int=(sr == O.OR_AND_I32))
    x0 = O.initial_vector(sr, rows)
    A = eng.upload_csr(rows, cols, rp, ci, va)
    # one launch
    xv, yv, out = eng.vector(x0), eng.vector(x0), eng.alloc(rows).fill(0)
    eng.spmv(sr, A, xv, yv, a, b, out)
    np.testing.assert_array_equal(bits(out.download(dt)), bits(g[tag + "_first"]))
    # whole do/while on the device
    iters, conv, per, total = eng.iterate(sr, A, xv, yv, out, a, b, delta=1e-4, max_iters=2000)
    assert [iters, int(conv)] == g[tag + "_meta"].tolist()
    np.testing.assert_array_equal(bits(xv.download(dt)), bits(g[tag + "_final"]))
    assert len(per) == iters and total == sum(per)
    for v in (xv, yv, out):
        v.free()
    A.free()


@pytest.mark.parametrize("name", ["matrix2", "matrix3", "matrix4"])   # see tests/test_oracle.py PR_OK
def test_pagerank_matches_reference(eng, name):
    g = golden(name)
    rows, cols, _, rp, ci, va = H.mm_load(mtx(name), normalise=H.NORM_PAGERANK, damping=0.85)
    x0 = np.full(rows, np.float32(1.0) / np.float32(rows), np.float32)
    y0 = np.ones(rows, np.float32)
    beta = (np.float32(1.0) - np.float32(0.85)) / np.float32(rows)
    A = eng.upload_csr(rows, cols, rp, ci, va)
    xv, yv, out = eng.vector(x0), eng.vector(y0), eng.alloc(rows).fill(0)
    eng.spmv(O.PLUS_TIMES_F32, A, xv, yv, 1.0, beta, out)
    np.testing.assert_array_equal(bits(out.download(np.float32)), bits(g["pr_first"]))
    iters, conv, _, _ = eng.iterate(O.PLUS_TIMES_F32, A, xv, yv, out, 1.0, beta, delta=1e-4, max_iters=2000)
    assert [iters, int(conv)] == g["pr_meta"].tolist()
    np.testing.assert_array_equal(bits(xv.download(np.float32)), bits(g["pr_final"]))


def test_scc_matches_reference(eng, matrix_name):
    g = golden(matrix_name)
    rows, cols, _, rp, ci, va = H.mm_load(mtx(matrix_name), elem_is_int=True, normalise=H.NORM_SCC)
    x0 = O.initial_vector(O.MAX_MIN_I32, rows)
    y0 = np.full(rows, O.INT_MIN, np.int32)
    A = eng.upload_csr(rows, cols, rp, ci, va)
    xv, yv, out = eng.vector(x0), eng.vector(y0), eng.alloc(rows).fill(0)
    eng.spmv(O.MAX_MIN_I32, A, xv, yv, O.INT_MAX, O.INT_MIN, out)
    np.testing.assert_array_equal(out.download(np.int32), g["scc_first"])
    iters, conv, _, _ = eng.iterate(O.MAX_MIN_I32, A, xv, yv, out, O.INT_MAX, O.INT_MIN, delta=1e-4, max_iters=2000)
    assert [iters, int(conv)] == g["scc_meta"].tolist()
    np.testing.assert_array_equal(xv.download(np.int32), g["scc_final"])


# ------------------------------------------------------------------ (b) oracle on seeded synthetic inputs
def synth_cases():
    rng = np.random.default_rng(1234)
    cases = {}
    rp, ci, va = H.powerlaw(200_000, 4_000_000, seed=11)           # long rows + short rows
    cases["powerlaw_int"] = (rp, ci, va, 200_000)
    cases["powerlaw_real"] = (rp, ci, (va * rng.uniform(0.5, 1.5, len(va))).astype(np.float32) - 4.0, 200_000)
    rp, ci, va = H.rmat(15, seed=5)
    cases["rmat15"] = (rp, ci, va, 1 << 15)
    # ragged: many empty rows, one very long row, a row of exactly one block, tiny rows
    deg = np.zeros(5000, np.int64)
    deg[7] = 70_001; deg[100] = 4096; deg[101] = 4097; deg[200:300] = 1; deg[4999] = 3; deg[1000:1010] = 64
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    ci = rng.integers(0, 5000, rp[-1]).astype(np.int32)
    va = rng.integers(1, 17, rp[-1]).astype(np.float32)
    cases["ragged"] = (rp, ci, va, 5000)
    return cases


@pytest.fixture(scope="module")
def cases():
    return synth_cases()


@pytest.mark.parametrize("name", ["powerlaw_int", "powerlaw_real", "rmat15", "ragged"])
def test_spmv_matches_oracle(eng, cases, name):
    rp, ci, va, n = cases[name]
    xm = (1 + np.arange(n) % 7).astype(np.float32)
    ym = (np.arange(n) % 5).astype(np.float32)
    for x, y, a, b in [(np.ones(n, np.float32), None, 1.0, 0.0), (xm, None, 1.0, 0.0), (xm, ym, 2.0, 0.5)]:
        got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, x, y, a, b)
        want = O.kernel(O.PLUS_TIMES_F32, rp, ci, va, x, np.zeros(n) if y is None else y, a, b)
        assert_close(got, want)
        if name != "powerlaw_real" and np.abs(want).max() < 2 ** 24:
            np.testing.assert_array_equal(bits(got), bits(want))  # integer-valued: exact
        if y is None:
            assert_close(got, O.gold_spmv(rp, ci, va, x))           # and against the gold restatement


@pytest.mark.parametrize("name", ["powerlaw_int", "rmat15", "ragged"])
@pytest.mark.parametrize("sr", [O.MIN_PLUS_F32, O.OR_AND_I32])
def test_semiring_variants_match_oracle(eng, cases, name, sr):
    rp, ci, va, n = cases[name]
    dt = O.elem_dtype(sr)
    vals = va.astype(dt)
    x0 = O.initial_vector(sr, n)
    a, b = (0.0, 0.0) if sr == O.MIN_PLUS_F32 else (1, 0)
    got = run_spmv(eng, sr, rp, ci, vals, x0, x0, a, b)
    np.testing.assert_array_equal(bits(got), bits(O.kernel(sr, rp, ci, vals, x0, x0, a, b)))
    # general alpha/beta/y
    rng = np.random.default_rng(3)
    x = rng.integers(0, 3, n).astype(dt)
    y = rng.integers(0, 50, n).astype(dt)
    a2, b2 = (2.0, 1.0) if sr == O.MIN_PLUS_F32 else (1, 1)
    got = run_spmv(eng, sr, rp, ci, vals, x, y, a2, b2)
    np.testing.assert_array_equal(bits(got), bits(O.kernel(sr, rp, ci, vals, x, y, a2, b2)))


def scc_values(rp, ci):
    """scc_normalise on a CSR: the row index off the diagonal, INT_MIN on it (src/sparse_matrix.cpp:433-456)."""
    row_of = np.repeat(np.arange(len(rp) - 1, dtype=np.int32), np.diff(rp))
    return np.where(ci == row_of, np.int32(O.INT_MIN), row_of).astype(np.int32)


@pytest.mark.parametrize("name", ["powerlaw_int", "rmat15", "ragged"])
def test_max_min_semiring_matches_oracle(eng, cases, name):
    rp, ci, va, n = cases[name]
    vals = scc_values(rp, ci)
    x0 = O.initial_vector(O.MAX_MIN_I32, n)
    y0 = np.full(n, O.INT_MIN, np.int32)
    got = run_spmv(eng, O.MAX_MIN_I32, rp, ci, vals, x0, y0, O.INT_MAX, O.INT_MIN)
    np.testing.assert_array_equal(got, O.kernel(O.MAX_MIN_I32, rp, ci, vals, x0, y0, O.INT_MAX, O.INT_MIN))
    # general alpha/beta/y, negative labels included
    rng = np.random.default_rng(5)
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    y = rng.integers(-1000, 1000, n).astype(np.int32)
    v2 = rng.integers(-1000, 1000, len(ci)).astype(np.int32)
    for a2, b2 in ((500, -200), (O.INT_MAX, O.INT_MAX), (-5, O.INT_MIN)):
        got = run_spmv(eng, O.MAX_MIN_I32, rp, ci, v2, x, y, a2, b2)
        np.testing.assert_array_equal(got, O.kernel(O.MAX_MIN_I32, rp, ci, v2, x, y, a2, b2))


def test_scc_iterate_matches_oracle_on_rmat(eng, cases):
    rp, ci, va, n = cases["rmat15"]
    vals = scc_values(rp, ci)
    x0 = O.initial_vector(O.MAX_MIN_I32, n)
    y0 = np.full(n, O.INT_MIN, np.int32)
    want, w_it, w_conv = O.iterate(O.MAX_MIN_I32, rp, ci, vals, x0, y0, O.INT_MAX, O.INT_MIN, max_iters=80)
    A = eng.upload_csr(n, n, rp, ci, vals)
    xv, yv, sc = eng.vector(x0), eng.vector(y0), eng.alloc(n).fill(0)
    iters, conv, _, _ = eng.iterate(O.MAX_MIN_I32, A, xv, yv, sc, O.INT_MAX, O.INT_MIN, delta=1e-4, max_iters=80)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(xv.download(np.int32), want)


def test_pagerank_iterate_tracks_oracle_on_rmat(eng, cases):
    """The PageRank the app was meant to run (no int narrowing): column-stochastic weights x damping.
    Floating-point sums reassociate on the GPU, so the vectors agree to 1e-5 relative, not bitwise."""
    rp, ci, va, n = cases["rmat15"]
    sums = np.bincount(ci, weights=va.astype(np.float64), minlength=n)
    vals = (np.abs(va) / sums[ci] * 0.85).astype(np.float32)
    x0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    y0 = np.ones(n, np.float32)
    beta = (np.float32(1.0) - np.float32(0.85)) / np.float32(n)
    delta = 1e-9
    want, w_it, w_conv = O.iterate(O.PLUS_TIMES_F32, rp, ci, vals, x0, y0, 1.0, beta, delta=delta, max_iters=200)
    A = eng.upload_csr(n, n, rp, ci, vals)
    xv, yv, sc = eng.vector(x0), eng.vector(y0), eng.alloc(n).fill(0)
    iters, conv, _, _ = eng.iterate(O.PLUS_TIMES_F32, A, xv, yv, sc, 1.0, beta, delta=delta, max_iters=200)
    assert conv and w_conv and abs(iters - w_it) <= 2 and iters > 5
    np.testing.assert_allclose(xv.download(np.float32), want, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("sr", [O.MIN_PLUS_F32, O.OR_AND_I32])
def test_iterate_matches_oracle_on_rmat(eng, cases, sr):
    rp, ci, va, n = cases["rmat15"]
    dt = O.elem_dtype(sr)
    vals = va.astype(dt)
    x0 = O.initial_vector(sr, n)
    a, b = (0.0, 0.0) if sr == O.MIN_PLUS_F32 else (1, 0)
    want, w_it, w_conv = O.iterate(sr, rp, ci, vals, x0, x0, a, b, delta=1e-4, max_iters=60)
    A = eng.upload_csr(n, n, rp, ci, vals)
    xv, yv, sc = eng.vector(x0), eng.vector(x0), eng.alloc(n).fill(0)
    iters, conv, _, _ = eng.iterate(sr, A, xv, yv, sc, a, b, delta=1e-4, max_iters=60)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(bits(xv.download(dt)), bits(want))


def test_value_coding_is_chosen_by_the_data(eng, cases, plan, monkeypatch):
    """<= 256 distinct values -> one-byte codes in the tiled stream; <= 4096 -> two-byte codes; more -> raw values; always lossless."""
    if plan != "tiled":
        pytest.skip("value coding belongs to the tiled plan")
    monkeypatch.setenv("SH_VALCODE", "auto")
    rp, ci, va, n = cases["powerlaw_int"]
    A = eng.upload_csr(n, n, rp, ci, va)
    assert "values=dict4(16)" in A.describe()        # weights 1..16: four-bit codes, padding borrows code 0's value
    A.free()
    monkeypatch.setenv("SH_VALCODE", "8")
    A = eng.upload_csr(n, n, rp, ci, va)
    assert "values=dict8(17)" in A.describe()        # one-byte codes: the 16 weights plus the padding word 0
    monkeypatch.setenv("SH_VALCODE", "auto")
    A.free()
    rp, ci, va, n = cases["powerlaw_real"]
    A = eng.upload_csr(n, n, rp, ci, va)
    assert "values=raw" in A.describe()
    A.free()
    # exactly 255 distinct non-zero bit patterns (incl. -0.0, inf, a NaN payload) still fit one byte; 256 take two-byte
    # codes, as do 4095; 4096 (+ the padding word) do not fit the 4096-word dictionary: raw
    rng = np.random.default_rng(11)
    m, nnz = 4096, 200_000
    rp = np.linspace(0, nnz, m + 1).astype(np.int32)
    ci = rng.integers(0, m, nnz).astype(np.int32)
    pool = np.unique(np.concatenate([np.array([0x80000000, 0x7F800000, 0x7FC00123, 0x12345678], np.uint32),
                                     rng.integers(1, 2**31, 4200).astype(np.uint32)]))
    keep = np.array([0x80000000, 0x7F800000, 0x7FC00123], np.uint32)
    pool = np.concatenate([keep, pool[~np.isin(pool, keep)]])          # the special words among the first 255
    for k, want in ((255, "dict8(256)"), (256, "dict16(257)"), (4095, "dict16(4096)"), (4096, "raw")):
        vb = pool[:k]
        assert len(np.unique(vb)) == k
        vals = vb[rng.integers(0, k, nnz)]
        vals[:k] = vb
        A = eng.upload_csr(m, m, rp, ci, vals.view(np.float32))
        assert f"values={want}" in A.describe()
        x = rng.integers(0, 2, m).astype(np.int32)    # (or,and) on raw words: exact whatever the bits mean
        xv, out = eng.vector(x), eng.alloc(m).fill(0)
        eng.spmv(O.OR_AND_I32, A, xv, None, 1, 0, out)
        np.testing.assert_array_equal(out.download(np.int32),
                                      O.kernel(O.OR_AND_I32, rp, ci, vals.view(np.int32), x, x, 1, 0))
        for v in (xv, out):
            v.free()
        A.free()
    # four-bit codes: <= 16 words counting a zero; exactly 16 non-zero words only if all are finite (padding then
    # borrows code 0's value, which an inf or NaN would turn into a NaN product)
    small = np.unique(rng.integers(1, 2**30, 64).astype(np.uint32))[:16]
    cases4 = ((small[:15], "dict4(16)"), (small, "dict4(16)"), (np.concatenate([[0], small[:15]]).astype(np.uint32), "dict4(16)"),
              (np.concatenate([small[:15], [0x7F800000]]).astype(np.uint32), "dict8(17)"), (small[:1], "dict4(2)"))
    for words, want in cases4:
        vals = words[rng.integers(0, len(words), nnz)]
        vals[:len(words)] = words
        A = eng.upload_csr(m, m, rp, ci, vals.view(np.float32))
        assert f"values={want}" in A.describe(), (A.describe(), want)
        x = rng.integers(0, 2, m).astype(np.int32)
        xv, out = eng.vector(x), eng.alloc(m).fill(0)
        eng.spmv(O.OR_AND_I32, A, xv, None, 1, 0, out)
        np.testing.assert_array_equal(out.download(np.int32),
                                      O.kernel(O.OR_AND_I32, rp, ci, vals.view(np.int32), x, x, 1, 0))
        for v in (xv, out):
            v.free()
        A.free()
    # (+,x) and (min,+) with a FULL four-bit table (weights 1..16, negative ones too): the borrowed padding value
    # must leave every sum untouched, heavy rows included
    deg = rng.multinomial(nnz - 30_000, np.ones(m) / m)
    deg[5] += 30_000                                     # a heavy row: its padded groups are summed in phase 1
    rp2 = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    for sign in (1.0, -1.0):
        fv = (sign * rng.integers(1, 17, nnz)).astype(np.float32)
        xf = rng.integers(0, 3, m).astype(np.float32)
        A = eng.upload_csr(m, m, rp2, ci, fv)
        assert "values=dict4(16)" in A.describe()
        xv, out = eng.vector(xf), eng.alloc(m).fill(0)
        eng.spmv(O.PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        np.testing.assert_array_equal(bits(out.download(np.float32)), bits(O.gold_dot(rp2, ci, fv, xf)))
        x0 = O.initial_vector(O.MIN_PLUS_F32, m)
        xv2 = eng.vector(x0)
        eng.spmv(O.MIN_PLUS_F32, A, xv2, xv2, 0.0, 0.0, out)
        np.testing.assert_array_equal(bits(out.download(np.float32)), bits(O.kernel(O.MIN_PLUS_F32, rp2, ci, fv, x0, x0, 0.0, 0.0)))
        for v in (xv, xv2, out):
            v.free()
        A.free()


def test_upload_times_both_plans_for_large_matrices(eng, plan, monkeypatch):
    """Size alone says "tiled" for any large matrix; a banded one keeps its x window in L2 and is faster under
    the CSR-stream plan, a random one is not: the upload-time measurement must tell them apart."""
    if plan != "stream":
        pytest.skip("runs once, with SH_PLAN unset")
    monkeypatch.delenv("SH_PLAN")
    rng = np.random.default_rng(3)
    n, per_row = 3_000_000, 8
    rp = (np.arange(n + 1, dtype=np.int64) * per_row).astype(np.int32)
    va = rng.integers(1, 17, n * per_row).astype(np.float32)
    x = (1 + np.arange(n) % 7).astype(np.float32)
    band = (np.repeat(np.arange(n, dtype=np.int64), per_row) + rng.integers(-300, 301, n * per_row)).clip(0, n - 1).astype(np.int32)
    rand = rng.integers(0, n, n * per_row).astype(np.int32)
    for ci, want_plan in ((band, "stream"), (rand, "tiled")):
        A = eng.upload_csr(n, n, rp, ci, va)
        d = A.describe()
        # random columns: not even timed.  Banded: timed; CSR-stream is kept when > 10 % faster -- since phase 1 folds
        # a row's entries inside a tile (here all 8 of a row: a quarter of the products travel) the two plans tie
        assert ("tuned(stream=" in d) == (want_plan == "stream"), d
        assert A.plan()[0] == want_plan or (want_plan == "stream" and "folded" in d), d
        xv, out = eng.vector(x), eng.alloc(n).fill(0)
        eng.spmv(O.PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        np.testing.assert_array_equal(bits(out.download(np.float32)), bits(O.gold_dot(rp, ci, va, x)))
        for v in (xv, out):
            v.free()
        A.free()
    monkeypatch.setenv("SH_AUTOTUNE", "0")
    A = eng.upload_csr(n, n, rp, band, va)
    assert "tuned" not in A.describe() and A.plan()[0] == "tiled"
    A.free()


def test_rectangular_and_wide_matrices(eng):
    """rows != cols (the C ABI allows it; only the apps insist on square) and x spanning many tiles."""
    rng = np.random.default_rng(99)
    for rows, cols, nnz in [(50_000, 300_000, 1_500_000), (300_000, 40_000, 2_000_000), (20_000, 2_500_000, 3_000_000)]:
        deg = rng.multinomial(nnz, np.ones(rows) / rows)
        deg[:3] = [0, 40_000, 1]          # an empty row, a heavy row, a singleton
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        ci = rng.integers(0, cols, rp[-1]).astype(np.int32)
        va = rng.integers(1, 17, rp[-1]).astype(np.float32)
        x = (1 + np.arange(cols) % 7).astype(np.float32)
        y = (np.arange(rows) % 5).astype(np.float32)
        got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, x, y, 2.0, 0.5, cols=cols)
        want = O.kernel(O.PLUS_TIMES_F32, rp, ci, va, x, y, 2.0, 0.5, vlength=cols)
        np.testing.assert_array_equal(bits(got), bits(want))
        xi = rng.integers(0, 2, cols).astype(np.int32)
        got = run_spmv(eng, O.OR_AND_I32, rp, ci, va.astype(np.int32), xi, None, 1, 0, cols=cols)
        want = O.kernel(O.OR_AND_I32, rp, ci, va.astype(np.int32), xi, np.zeros(rows, np.int32), 1, 0, vlength=cols)
        np.testing.assert_array_equal(got, want)


def test_edge_cases(eng):
    # empty matrix rows, nnz == 0, single row, out-of-range / negative columns -> identity
    rp = np.zeros(11, np.int32)
    got = run_spmv(eng, O.PLUS_TIMES_F32, rp, np.zeros(0, np.int32), np.zeros(0, np.float32), np.ones(10), np.full(10, 3.0), 1.0, 2.0)
    assert got.tolist() == [6.0] * 10
    got = run_spmv(eng, O.MIN_PLUS_F32, rp, np.zeros(0, np.int32), np.zeros(0, np.float32), np.ones(10), np.full(10, 3.0), 0.0, 0.0)
    assert got.tolist() == [3.0] * 10
    rp = np.array([0, 3], np.int32)
    ci = np.array([0, -1, 5], np.int32)
    va = np.array([1, 10, 100], np.float32)
    assert run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, [2, 3], None, 1.0, 0.0, cols=2).tolist() == [2.0]
    assert run_spmv(eng, O.MIN_PLUS_F32, rp, ci, va, [2, 3], [O.FLT_MAX], 0.0, 0.0, cols=2).tolist() == [3.0]
    assert run_spmv(eng, O.OR_AND_I32, rp, ci, va.astype(np.int32), [0, 1], None, 1, 0, cols=2).tolist() == [0]
    assert run_spmv(eng, O.OR_AND_I32, rp, np.array([1, -1, 5], np.int32), va.astype(np.int32), [0, 1], None, 1, 0, cols=2).tolist() == [1]


def test_error_paths(eng):
    rp = np.array([0, 1, 2], np.int32)
    A = eng.upload_csr(2, 2, rp, np.array([0, 1], np.int32), np.ones(2, np.float32))
    x, out = eng.vector(np.ones(2, np.float32)), eng.alloc(2)
    with pytest.raises(EngineError) as ei:   # beta != 0 needs y
        eng.spmv(O.PLUS_TIMES_F32, A, x, None, 1.0, 1.0, out)
    assert ei.value.code == -1
    short = eng.alloc(1)
    with pytest.raises(EngineError) as ei:
        eng.spmv(O.PLUS_TIMES_F32, A, short, None, 1.0, 0.0, out)
    assert ei.value.code == -5
    with pytest.raises(EngineError):
        eng.spmv(O.PLUS_TIMES_F32, A, x, None, 1.0, 0.0, x)   # out aliases x
    with pytest.raises(EngineError) as ei:   # inconsistent row_ptr
        eng.upload_csr(2, 2, np.array([0, 2, 1], np.int32), np.array([0], np.int32), np.ones(1, np.float32))
    assert ei.value.code == -5


def test_step_sets_changed_flag(eng):
    import ctypes
    rp, ci, va = H.rmat(10, seed=9)
    n = 1 << 10
    A = eng.upload_csr(n, n, rp, ci, va)
    x0 = O.initial_vector(O.MIN_PLUS_F32, n)
    x, out, flag = eng.vector(x0), eng.alloc(n), eng.alloc(1).fill(0, np.int32)
    eng.step(O.MIN_PLUS_F32, A, x, x, 0.0, 0.0, out, 0, 1e-4, flag.device_ptr)
    assert flag.download(np.int32)[0] == 1
    final, _, _ = O.iterate(O.MIN_PLUS_F32, rp, ci, va, x0, x0, 0.0, 0.0)
    x.upload(final)
    flag.fill(0, np.int32)
    eng.step(O.MIN_PLUS_F32, A, x, x, 0.0, 0.0, out, 0, 1e-4, flag.device_ptr)
    assert flag.download(np.int32)[0] == 0
    np.testing.assert_array_equal(bits(out.download()), bits(final))


# ------------------------------------------------------------------ (c) full-size properties
@pytest.mark.parametrize("rows,nnz", [(10_000_000, 200_000_000)])
def test_full_size_powerlaw_properties(eng, rows, nnz, plan):
    """BASELINE.json config 5 at full size: checksum + linearity + spot rows vs the oracle."""
    rp, ci, va = H.powerlaw(rows, nnz)
    A = eng.upload_csr(rows, rows, rp, ci, va)
    assert A.plan()[0] == plan
    x1 = eng.alloc(rows).fill(1.0)
    out = eng.alloc(rows)
    eng.spmv(O.PLUS_TIMES_F32, A, x1, None, 1.0, 0.0, out)
    y1 = out.download()
    # checksum of checksums: sum_r y[r] == sum of all weights (x = 1)
    assert abs(y1.astype(np.float64).sum() - va.astype(np.float64).sum()) <= 1e-6 * va.astype(np.float64).sum()
    # every row whose exact sum fits in 24 bits must be exact: compare a strided sample + the longest rows to the oracle
    deg = np.diff(rp)
    sample = np.unique(np.concatenate([np.arange(0, rows, 9973), np.argsort(deg)[-8:]]))
    for r in sample:
        s = va[rp[r]:rp[r + 1]].astype(np.float64).sum()
        assert abs(float(y1[r]) - s) <= 1e-5 * max(1.0, abs(s)), (r, deg[r])
        if s < 2 ** 24:
            assert float(y1[r]) == s
    # linearity: A(3x) == 3 A(x) exactly for integer data below 2^24
    x3 = eng.alloc(rows).fill(3.0)
    out3 = eng.alloc(rows)
    eng.spmv(O.PLUS_TIMES_F32, A, x3, None, 1.0, 0.0, out3)
    y3 = out3.download()
    small = y1 < 2 ** 22
    np.testing.assert_array_equal(y3[small], 3 * y1[small])
    # alpha/beta epilogue against the first result
    eng.spmv(O.PLUS_TIMES_F32, A, x1, out, 2.0, 1.0, out3)
    np.testing.assert_array_equal(out3.download()[small], 3 * y1[small])


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("sr", [O.MIN_PLUS_F32, O.OR_AND_I32])
def test_sharded_driver_with_hip_local_step(cases, sr, chunks):
    """The multi-GPU iteration driver's device seam (HipLocalStep -> sh_spmv_step_pieces on torch memory,
    slotted / chunked vector layout, ONE matrix per rank whose launch reports its pieces one by one, fused
    changed flag) on one rank; two ranks: test_two_ranks_share_the_gpu_with_hip_local_step."""
    import torch
    from sparseharness_amd.distributed import HipLocalStep, ShardedIteration, ShardPlan
    rp, ci, va, n = cases["rmat15"]
    dt = O.elem_dtype(sr)
    vals = va.astype(dt)
    a, b = (0.0, 0.0) if sr == O.MIN_PLUS_F32 else (1, 0)
    x0 = O.initial_vector(sr, n)
    want, w_it, w_conv = O.iterate(sr, rp, ci, vals, x0, x0, a, b, 1e-4, 60)
    torch.cuda.set_device(0)
    plan = ShardPlan(rp, ci, vals, 0, 1, chunks)
    final, iters, conv = ShardedIteration(plan, sr, HipLocalStep(plan, sr, 0)).run(x0, x0, a, b, 1e-4, 60)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(bits(final), bits(want))


def test_iteration_cap_on_a_graph_that_never_settles(eng):
    """(or,and) BFS with beta = 0 on a 2-cycle oscillates for ever (the reference would spin: TODO.md:7-8);
    the engine stops at max_iters with converged = false and the same vector as the oracle."""
    rp = np.array([0, 1, 2, 2], np.int32)          # row0 <- col1, row1 <- col0, row2 empty
    ci = np.array([1, 0], np.int32)
    va = np.array([1, 1], np.int32)
    x0 = np.array([1, 0, 0], np.int32)
    want, w_it, w_conv = O.iterate(O.OR_AND_I32, rp, ci, va, x0, x0, 1, 0, 1e-4, 7)
    assert (w_it, w_conv) == (7, False)
    A = eng.upload_csr(3, 3, rp, ci, va)
    x, y, sc = eng.vector(x0), eng.vector(x0), eng.alloc(3)
    iters, conv, per, _ = eng.iterate(O.OR_AND_I32, A, x, y, sc, 1, 0, 1e-4, 7)
    assert (iters, conv, len(per)) == (7, False, 7)
    np.testing.assert_array_equal(x.download(np.int32), want)


# ------------------------------------------------------------------ (c) randomised shapes (hypothesis)
def _random_csr(rng, rows, cols, nnz, n_heavy, frac_empty, frac_oob):
    """Ragged CSR with a few very long rows, a share of empty rows and some out-of-range columns."""
    w = rng.random(rows) ** 3
    w[rng.random(rows) < frac_empty] = 0
    if w.sum() == 0:
        w[0] = 1
    deg = rng.multinomial(nnz, w / w.sum())
    for h in rng.integers(0, rows, n_heavy):
        deg[h] += int(rng.integers(3_000, 20_000))
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    ci = rng.integers(0, cols, rp[-1]).astype(np.int32)
    oob = rng.random(rp[-1]) < frac_oob
    ci[oob] = rng.choice(np.array([-1, -7, cols, cols + 12345], np.int32), int(oob.sum()))
    return rp, ci


def test_random_shapes_match_oracle_bit_for_bit(eng):
    """Property test over matrix shapes: integer-valued data, so every semiring is exact in any order."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2**31 - 1), rows=st.integers(1, 60_000), cols=st.integers(1, 400_000),
           density=st.floats(0.2, 40.0), n_heavy=st.integers(0, 3), frac_empty=st.floats(0.0, 0.6),
           frac_oob=st.sampled_from([0.0, 0.0, 0.01]))
    def check(seed, rows, cols, density, n_heavy, frac_empty, frac_oob):
        rng = np.random.default_rng(seed)
        nnz = max(1, min(int(rows * density), 1_500_000))
        rp, ci = _random_csr(rng, rows, cols, nnz, n_heavy, frac_empty, frac_oob)
        nvals = int(rng.choice([1, 16, 300]))             # pattern-like / coded / raw value layouts
        # multiples of 1/64 below 4.7, x in {0,1}: every partial sum is exact in float (< 2^18), so the
        # result does not depend on the summation order and must match bit for bit
        va = (rng.integers(1, nvals + 1, rp[-1]) / 64.0).astype(np.float32)
        x = rng.integers(0, 2, cols).astype(np.float32)
        y = rng.integers(0, 9, rows).astype(np.float32)
        got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, x, y, 2.0, 1.0, cols=cols)
        want = O.kernel(O.PLUS_TIMES_F32, rp, ci, va, x, y, 2.0, 1.0, vlength=cols)
        np.testing.assert_array_equal(bits(got), bits(want))
        xs = np.where(rng.random(cols) < 0.3, np.float32(3.4028235e38), x)          # (min,+) with unreached sources
        ys = np.where(rng.random(rows) < 0.5, np.float32(3.4028235e38), y)
        got = run_spmv(eng, O.MIN_PLUS_F32, rp, ci, va, xs, ys, 0.0, 0.0, cols=cols)
        want = O.kernel(O.MIN_PLUS_F32, rp, ci, va, xs, ys, 0.0, 0.0, vlength=cols)
        np.testing.assert_array_equal(bits(got), bits(want))
        xi, yi, vi = x.astype(np.int32), y.astype(np.int32), ((va * 64).astype(np.int32) - 1)  # zeros among the values
        got = run_spmv(eng, O.OR_AND_I32, rp, ci, vi, xi, yi, 1, 1, cols=cols)
        np.testing.assert_array_equal(got, O.kernel(O.OR_AND_I32, rp, ci, vi, xi, yi, 1, 1, vlength=cols))
        got = run_spmv(eng, O.MAX_MIN_I32, rp, ci, vi - 5, xi - 2, yi, 7, -3, cols=cols)
        np.testing.assert_array_equal(got, O.kernel(O.MAX_MIN_I32, rp, ci, vi - 5, xi - 2, yi, 7, -3, vlength=cols))

    check()


def test_config4_rmat23_sssp_and_bfs_to_convergence(eng, plan):
    """BASELINE.json config 4 at full size: (min,+) SSSP and (or,and) BFS on R-MAT scale 23 (134 M entries) run to
    convergence on the device and are compared bit for bit -- vector and launch count -- with the CPU oracle."""
    import os
    if plan != "tiled" or os.environ.get("SH_VALCODE") != "auto":
        pytest.skip("full size once, under the layout the engine picks for it")
    rp, ci, va = H.rmat(23)
    n = 1 << 23
    for sr, a, b in ((O.MIN_PLUS_F32, 0.0, 0.0), (O.OR_AND_I32, 1, 0)):
        dt = O.elem_dtype(sr)
        vals = va.astype(dt)
        x0 = O.initial_vector(sr, n)
        want, w_it, w_conv = O.iterate(sr, rp, ci, vals, x0, x0, a, b, 1e-4, 200)
        A = eng.upload_csr(n, n, rp, ci, vals)
        assert A.plan()[0] == "tiled"
        xv, yv, sc = eng.vector(x0), eng.vector(x0), eng.alloc(n)
        iters, conv, per, total = eng.iterate(sr, A, xv, yv, sc, a, b, 1e-4, 200)
        assert (iters, conv) == (w_it, w_conv) and conv
        np.testing.assert_array_equal(bits(xv.download(dt)), bits(want))
        A.free()
        if sr == O.OR_AND_I32:
            # ... and on the layout bfs_harness (harness.h) and HipLocalStep (distributed.py) pick BY THEMSELVES for a
            # matrix of this size: the bit-blocked plan alone (or_and_bits = 2) -- 32 row ranges x 16 column blocks at
            # scale 23.  Same vector, same launch count.
            A = eng.upload_csr(n, n, rp, ci, vals, or_and_bits=2)
            assert "or_and=bits(" in A.describe() and "only" in A.describe(), A.describe()
            xv.upload(x0)
            yv.upload(x0)
            iters, conv, per, total = eng.iterate(sr, A, xv, yv, sc, a, b, 1e-4, 200)
            assert (iters, conv) == (w_it, w_conv) and conv
            np.testing.assert_array_equal(bits(xv.download(dt)), bits(want))
            A.free()
        for v in (xv, yv, sc):
            v.free()


def test_config3_rmat23_float_spmv_full_size(eng, plan):
    """BASELINE.json config 3 at full size: float (+,x) SpMV on R-MAT scale 23 (134 M entries, x = 1 + (i mod 7),
    integer weights 1..16).  Every row whose |terms| sum stays below 2^24 is order-independent in float and must
    equal the sequential gold bit for bit (inc/harness.h:134, exact compare); the few hub rows beyond that are
    named and held to 1e-5 of the exact (float64) dot."""
    import os
    if plan != "tiled" or os.environ.get("SH_VALCODE") != "auto":
        pytest.skip("full size once, under the layout the engine picks for it")
    rp, ci, va = H.rmat(23)
    n = 1 << 23
    x = (1 + np.arange(n) % 7).astype(np.float32)
    A = eng.upload_csr(n, n, rp, ci, va)
    assert A.plan()[0] == "tiled", A.describe()
    xv, out = eng.vector(x), eng.alloc(n).fill(0)
    ns = eng.spmv(O.PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out, timed=True)
    got = out.download(np.float32)
    want = O.gold_dot(rp, ci, va, x, 1.0)
    assert ns > 0 and O.check_result(want, got) in (O.CORRECT, O.BAD_VALUES)
    # an upper bound of a row's |terms| sum: 16 * 7 * length
    big = np.nonzero(np.diff(rp).astype(np.int64) * 112 >= 2 ** 24)[0]
    small = np.ones(n, bool)
    small[big] = False
    np.testing.assert_array_equal(bits(got[small]), bits(want[small]))
    for r in big:   # hub rows: the sequential float gold itself is inexact there
        a, b = int(rp[r]), int(rp[r + 1])
        exact = float((x[ci[a:b]].astype(np.float64) * va[a:b].astype(np.float64)).sum())
        assert abs(float(got[r]) - exact) <= REL * max(1.0, abs(exact)), (r, got[r], exact)
    assert len(big) < 64
    for v in (xv, out):
        v.free()
    A.free()


def test_config2_scircuit_shaped_float_spmv(eng, plan, monkeypatch):
    """BASELINE.json config 2 (stand-in of SuiteSparse scircuit's shape: 170 998 rows, 958 936 entries; the file
    itself is not available offline): x fits the per-XCD L2, so the engine picks the CSR-stream plan BY ITSELF
    (no SH_PLAN), and the result equals the gold bit for bit (integer weights, sums far below 2^24)."""
    if plan != "stream":
        pytest.skip("once; the plan is chosen by the engine here")
    monkeypatch.delenv("SH_PLAN", raising=False)
    rp, ci, va = H.scircuit_like()
    n = len(rp) - 1
    assert (n, int(rp[-1])) == (170_998, 958_936)
    A = eng.upload_csr(n, n, rp, ci, va)
    assert A.plan()[0] == "stream" and "tuned" not in A.describe(), A.describe()
    for x in (np.ones(n, np.float32), (1 + np.arange(n) % 7).astype(np.float32)):
        xv, out = eng.vector(x), eng.alloc(n).fill(0)
        eng.spmv(O.PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        got = out.download(np.float32)
        want = O.gold_dot(rp, ci, va, x, 1.0)
        np.testing.assert_array_equal(bits(got), bits(want))
        assert O.check_result(want, got) == O.CORRECT
        for v in (xv, out):
            v.free()
    A.free()




def clustered_matrix(n=60_000, seed=11):
    """Rows whose columns crowd into a few column tiles, 2..600 entries each, duplicates included: nearly every
    entry shares its (row, tile) with others, so phase 1's folding -- runs of 2, 3, 4 entries, runs cut at 4,
    padding entries joining runs -- carries most of the matrix.  Two rows are long enough to be heavy."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 40, n).astype(np.int64)
    deg[rng.integers(0, n, n // 20)] = rng.integers(100, 600, n // 20)
    deg[[7, n - 3]] = (5000, 1200)
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    nnz = int(rp[-1])
    row_of = np.repeat(np.arange(n, dtype=np.int64), deg)
    centre = (row_of * 7919) % n                       # a row's columns lie within +-3000 of a pseudo-random centre
    ci = ((centre + rng.integers(-3000, 3000, nnz)) % n).astype(np.int32)
    va = rng.integers(1, 17, nnz).astype(np.float32)
    return rp, ci, va, n


@pytest.mark.parametrize("sr", [O.PLUS_TIMES_F32, O.MIN_PLUS_F32, O.OR_AND_I32, O.MAX_MIN_I32])
def test_folded_runs_match_oracle(eng, plan, sr):
    """Phase 1 of the tiled plan folds the entries of one row that fall into one column tile into ONE product
    (at most 4 entries per fold) before it travels through P.  On a matrix made of such runs: every semiring
    bit-exact against the oracle (integer-valued data), three different x through the same device matrix."""
    if plan != "tiled":
        pytest.skip("folding belongs to the tiled plan")
    import os
    rp, ci, va, n = clustered_matrix()
    dt = O.elem_dtype(sr)
    vals = scc_values(rp, ci) if sr == O.MAX_MIN_I32 else va.astype(dt)
    A = eng.upload_csr(n, n, rp, ci, vals)
    folded = os.environ.get("SH_FOLD") != "0"
    assert (" folded" in A.describe()) == folded, A.describe()
    light = float(A.describe().split("light=")[1].split("M")[0])
    prods = float(A.describe().split("products=")[1].split("M")[0])
    assert (prods < 0.62 * light) if folded else (prods >= light), A.describe()
    rng = np.random.default_rng(9)
    out = eng.alloc(n).fill(0)
    a, b = {O.PLUS_TIMES_F32: (2.0, 0.5), O.MIN_PLUS_F32: (1.0, 2.0), O.OR_AND_I32: (1, 1), O.MAX_MIN_I32: (700, -300)}[sr]
    for k in range(3):
        x = rng.integers(0, 4, n).astype(dt) if sr != O.MAX_MIN_I32 else rng.integers(-1000, 1000, n).astype(dt)
        y = rng.integers(0, 50, n).astype(dt)
        xv, yv = eng.vector(x), eng.vector(y)
        eng.spmv(sr, A, xv, yv, a, b, out)
        np.testing.assert_array_equal(bits(out.download(dt)), bits(O.kernel(sr, rp, ci, vals, x, y, a, b)))
        xv.free()
        yv.free()
    out.free()
    A.free()


@pytest.mark.parametrize("sr", [O.MIN_PLUS_F32, O.OR_AND_I32, O.MAX_MIN_I32])
@pytest.mark.parametrize("weights", ["coded", "raw", "huge"])
def test_tiles_of_absorbing_x_are_skipped_exactly(eng, plan, sr, weights):
    """Phase 1 of the tiled plan does not read the entries of a column tile whose x words are all absorbing (an
    unreached vertex of SSSP: |x| = FLT_MAX; outside the BFS frontier: 0; INT_MIN for (max,min)) and writes identity
    products instead.  x with no, one, a few and many live entries over 7 column tiles, heavy rows included, against
    the oracle bit for bit; for (min,+) also with raw weights and with a weight of 2^110 in the dictionary
    (FLT_MAX + 2^110 overflows: such a matrix is never skipped)."""
    if plan != "tiled":
        pytest.skip("belongs to the tiled plan")
    if weights != "coded" and sr != O.MIN_PLUS_F32:
        pytest.skip("the weight condition is (min,+)'s")
    rng = np.random.default_rng(31)
    rows, cols = 60_000, 7 * 32760 - 1000
    deg = rng.poisson(9, rows).astype(np.int64)
    deg[rng.integers(0, rows, 4)] = 3000          # heavy rows
    deg[rng.integers(0, rows, 600)] = 150         # rows with several entries per tile: folded pairs
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    nnz = int(rp[-1])
    ci = rng.integers(0, cols, nnz).astype(np.int32)
    dt = O.elem_dtype(sr)
    if weights == "raw":
        vals = (rng.random(nnz) * 30).astype(np.float32)
    else:
        vals = rng.integers(1, 17, nnz).astype(dt)
        if weights == "huge":
            vals[rng.integers(0, nnz, 5)] = np.float32(2.0 ** 110)
    if sr == O.MAX_MIN_I32:
        vals = rng.integers(-50, 50, nnz).astype(dt)
    A = eng.upload_csr(rows, cols, rp, ci, vals, plan=2)
    assert A.plan()[0] == "tiled" and "tiles=7" in A.describe(), A.describe()
    dead = {O.MIN_PLUS_F32: np.float32(np.finfo(np.float32).max), O.OR_AND_I32: 0, O.MAX_MIN_I32: O.INT_MIN}[sr]
    a, b = {O.MIN_PLUS_F32: (0.0, 0.0), O.OR_AND_I32: (1, 0), O.MAX_MIN_I32: (700, -300)}[sr]
    out = eng.alloc(rows).fill(0)
    for live in (0, 1, 6, 2000, cols):
        x = np.full(cols, dead, dt)
        idx = rng.choice(cols, size=min(live, cols), replace=False)
        x[idx] = (rng.integers(1, 9, len(idx)) if sr != O.MIN_PLUS_F32 else rng.random(len(idx)) * 9).astype(dt)
        if sr == O.MIN_PLUS_F32 and live:
            x[idx[0]] = -x[idx[0]]                 # absadd takes |x|: a negative finite x is live
            if live > 1:
                x[idx[1]] = -dead                  # ... and -FLT_MAX is as absorbing as FLT_MAX
        y = np.full(rows, dead, dt)
        xv, yv = eng.vector(x), eng.vector(y)
        eng.spmv(sr, A, xv, yv, a, b, out)
        np.testing.assert_array_equal(bits(out.download(dt)), bits(O.kernel(sr, rp, ci, vals, x, y, a, b)), err_msg=f"{live} live")
        xv.free()
        yv.free()
    out.free()
    A.free()


def test_folded_runs_real_values_within_tolerance(eng, plan):
    """The same with real-valued weights and x: a fold changes the order of a row's float additions, so the
    yardstick is the north-star tolerance (1e-5 relative) against the exact (float64) dot."""
    if plan != "tiled":
        pytest.skip("folding belongs to the tiled plan")
    rp, ci, va, n = clustered_matrix(seed=12)
    rng = np.random.default_rng(13)
    va = (va * np.float32(0.37) + rng.random(len(va), dtype=np.float32)).astype(np.float32)
    x = (rng.random(n, dtype=np.float32) + np.float32(0.5)).astype(np.float32)
    got = run_spmv(eng, O.PLUS_TIMES_F32, rp, ci, va, x, None, 1.0, 0.0)
    row_of = np.repeat(np.arange(n), np.diff(rp))
    exact = np.zeros(n)
    np.add.at(exact, row_of, x[ci].astype(np.float64) * va.astype(np.float64))
    assert np.all(np.abs(got - exact) <= REL * np.maximum(1.0, np.abs(exact)))


def test_many_launches_with_changing_inputs_midsize(eng, plan):
    """Twelve launches with twelve different x through one device matrix (2 M x 40 M power-law, ~65 column tiles,
    ~1700 bins): P, the heavy partials and the LDS images are rewritten by every launch, so a product left over
    from the launch before would surface as a wrong row.  Every row compared with the oracle."""
    if plan != "tiled":
        pytest.skip("the product array belongs to the tiled plan")
    import os
    if os.environ.get("SH_VALCODE") == "8":
        pytest.skip("coded and raw values are enough here")
    n = 2_000_000
    rp, ci, va = H.powerlaw(n, 40_000_000, seed=21)
    A = eng.upload_csr(n, n, rp, ci, va)
    out = eng.alloc(n).fill(0)
    for k in range(12):
        x = (1 + (np.arange(n) * (k + 3)) % (5 + k)).astype(np.float32)
        xv = eng.vector(x)
        eng.spmv(O.PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        got = out.download(np.float32)
        want = O.gold_dot(rp, ci, va, x, 1.0)
        big = np.diff(rp).astype(np.int64) * 16 * (5 + k) >= 2 ** 24       # rows whose float sum may round
        np.testing.assert_array_equal(bits(got[~big]), bits(want[~big]))
        for r in np.nonzero(big)[0]:   # hub rows: the sequential float gold itself rounds; the yardstick is the exact dot
            a, b = int(rp[r]), int(rp[r + 1])
            exact = float((x[ci[a:b]].astype(np.float64) * va[a:b].astype(np.float64)).sum())
            assert abs(float(got[r]) - exact) <= REL * max(1.0, abs(exact)), (r, got[r], exact)
        xv.free()
    out.free()
    A.free()


@pytest.mark.parametrize("distinct", [257, 1000, 4095])
def test_two_byte_value_codes_match_oracle(eng, distinct, plan):
    """A matrix with more than 256 distinct value words (what the reference's int-narrowed weights, src/sparse_matrix.cpp:107,
    or 1 / degree weights give): the tiled plan's stream carries two-byte codes into a 4096-word dictionary in LDS
    (values=dict16, VC = 3 of spmv_tiled_phase1; 4 instead of 6 bytes per entry).  All four semirings, epilogues with y,
    light groups with folded pairs, heavy strips; bit-exact against the oracle on integer-valued data, both builders."""
    if plan != "tiled":
        pytest.skip("value coding belongs to the tiled plan")
    import os
    if os.environ.get("SH_VALCODE") != "auto":
        pytest.skip("needs the default value-coding policy")
    rng = np.random.default_rng(distinct)
    rows, cols = 60_000, 200_000
    deg = rng.poisson(9, rows).astype(np.int64)
    deg[rng.integers(0, rows, 5)] = 4000          # heavy rows
    deg[rng.integers(0, rows, 300)] = rng.integers(30, 300, 300)
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    nnz = int(rp[-1])
    ci = rng.integers(0, cols, nnz).astype(np.int32)
    vi = (1 + rng.integers(0, distinct, nnz)).astype(np.int32)
    vi[:distinct] = 1 + np.arange(distinct)
    x = rng.integers(0, 4, cols).astype(np.int32)
    y = rng.integers(-3, 4, rows).astype(np.int32)
    for build in (1, 2):
        for sr, vals, xx, yy, a, b in ((O.PLUS_TIMES_F32, vi.astype(np.float32), x.astype(np.float32), y.astype(np.float32), 2.0, 1.0),
                                       (O.MIN_PLUS_F32, vi.astype(np.float32), x.astype(np.float32), y.astype(np.float32), 0.0, 0.0),
                                       (O.OR_AND_I32, vi, x, y, 1, 1), (O.MAX_MIN_I32, vi, x, y, O.INT_MAX, O.INT_MIN)):
            A = eng.upload_csr(rows, cols, rp, ci, vals, build=build)
            assert f"values=dict16({distinct + 1})" in A.describe(), A.describe()
            xv, yv, out = eng.vector(xx), eng.vector(yy), eng.alloc(rows).fill(0)
            eng.spmv(sr, A, xv, yv, a, b, out)
            want = O.kernel(sr, rp, ci, vals, xx, yy, a, b, vlength=cols)
            np.testing.assert_array_equal(bits(out.download(vals.dtype)), bits(want))
            for v in (xv, yv, out):
                v.free()
            A.free()


def _two_rank_worker(rank, world, rendezvous, sr, chunks, q, log_dir, exchange="collective"):
    """One rank of the two-rank tests.  Whatever goes wrong here reaches the parent: the traceback travels through the
    queue, stderr goes to a file the parent prints, and the process group has a short timeout, so a rank whose peer died
    (or whose piece report never came: HipLocalStep.wait_piece raises with the engine's counters) fails by itself instead
    of sitting in a collective -- round 3's record holds a 240-second stall of this test that named nothing."""
    import datetime
    import sys
    import traceback
    sys.stderr = open(os.path.join(log_dir, f"rank{rank}.stderr"), "w", buffering=1)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # (the container's hostname need not resolve)
    try:
        import torch
        import torch.distributed as dist
        # (rendezvous through a file: a TCP port picked by the parent could be taken by someone else before rank 0 binds it)
        dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=90))
        try:
            from sparseharness_amd.distributed import HipLocalStep, ShardedIteration, ShardPlan
            rp, ci, va = H.rmat(15, seed=5)
            n = len(rp) - 1
            vals = va.astype(O.elem_dtype(sr))
            a, b = (0.0, 0.0) if sr == O.MIN_PLUS_F32 else (1, 0)
            x0 = O.initial_vector(sr, n)
            torch.cuda.set_device(0)
            plan = ShardPlan(rp, ci, vals, rank, world, chunks)
            final, iters, conv = ShardedIteration(plan, sr, HipLocalStep(plan, sr, 0), exchange=exchange).run(x0, x0, a, b, 1e-4, 60)
            q.put((rank, final, iters, conv, plan.r0, plan.r1))
        finally:
            dist.destroy_process_group()
    except BaseException:   # noqa: BLE001 -- the parent must see it
        q.put((rank, "error", traceback.format_exc()))
        raise


def _run_two_ranks(sr, chunks, exchange="collective"):
    """Start the two workers and collect their results; an error tuple from either fails the test with the worker's own
    traceback and both stderr files."""
    import tempfile

    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="sh_two_ranks_")
    rendezvous = os.path.join(tmp, "store")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, rendezvous, sr, chunks, q, tmp, exchange)) for r in range(2)]
    for p in procs:
        p.start()

    def stderr_of_workers():
        out = []
        for r in range(2):
            try:
                out.append(f"--- rank {r} stderr ---\n" + open(os.path.join(tmp, f"rank{r}.stderr")).read()[-4000:])
            except OSError:
                pass
        return "\n".join(out)
    try:
        res = []
        for _ in procs:
            try:
                item = q.get(timeout=240)
            except Exception:   # noqa: BLE001 -- queue.Empty: nobody reported at all
                pytest.fail("a worker neither finished nor reported an error within 240 s\n" + stderr_of_workers())
            if isinstance(item[1], str) and item[1] == "error":
                pytest.fail(f"rank {item[0]} failed:\n{item[2]}\n" + stderr_of_workers())
            res.append(item)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, stderr_of_workers()
    finally:
        for p in procs:   # (a worker that hangs must not outlive the test: it would hold the GPU and the run's pipes)
            if p.is_alive():
                p.kill()
                p.join(timeout=30)
    return sorted(res, key=lambda t: t[0])


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("sr", [O.MIN_PLUS_F32, O.OR_AND_I32])
def test_two_ranks_share_the_gpu_with_hip_local_step(sr, chunks, plan):
    """N > 1 x HIP: two ranks, both on GPU 0 (the box has one), each with its own engine, its own half of an R-MAT-15
    graph with columns remapped to the slotted layout, rank 1 writing its pieces at non-zero offsets, the two
    changed flags travelling with the last piece; the all-gather goes through gloo (host-staged).  SSSP and BFS to
    convergence, every rank's final vector and launch count bit-identical to the single-process oracle loop."""
    if plan != "tiled":
        pytest.skip("once per run is enough (the engine picks the plan by size)")
    rp, ci, va = H.rmat(15, seed=5)
    n = len(rp) - 1
    vals = va.astype(O.elem_dtype(sr))
    a, b = (0.0, 0.0) if sr == O.MIN_PLUS_F32 else (1, 0)
    x0 = O.initial_vector(sr, n)
    want, w_it, w_conv = O.iterate(sr, rp, ci, vals, x0, x0, a, b, 1e-4, 60)
    res = _run_two_ranks(sr, chunks)
    assert res[1][4] > 0 and res[0][5] == res[1][4]          # rank 1 starts where rank 0 ends, past row 0
    for rank, final, iters, conv, _, _ in res:
        assert (iters, conv) == (w_it, w_conv), f"rank {rank}"
        np.testing.assert_array_equal(bits(final), bits(want))


def test_two_ranks_exchange_their_pieces_point_to_point(plan):
    """The same two ranks with exchange="p2p" (SH_EXCHANGE=p2p): every finished piece goes straight to the peer by a
    grouped isend / irecv pair instead of an all-gather (host-staged here, as the all-gather of the test above: gloo
    has no device transfers and two ranks cannot share one GPU under RCCL).  Three pieces per rank, SSSP."""
    if plan != "tiled":
        pytest.skip("once per run")
    sr = O.MIN_PLUS_F32
    rp, ci, va = H.rmat(15, seed=5)
    x0 = O.initial_vector(sr, len(rp) - 1)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va.astype(np.float32), x0, x0, 0.0, 0.0, 1e-4, 60)
    for rank, final, iters, conv, _, _ in _run_two_ranks(sr, 3, exchange="p2p"):
        assert (iters, conv) == (w_it, w_conv), f"rank {rank}"
        np.testing.assert_array_equal(bits(final), bits(want))


def _nccl_world1_worker(rendezvous, sr, chunks, exchange, q, log_dir):
    import datetime
    import sys
    import traceback
    sys.stderr = open(os.path.join(log_dir, "rank0.stderr"), "w", buffering=1)
    try:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"file://{rendezvous}", rank=0, world_size=1,
                                timeout=datetime.timedelta(seconds=120), device_id=torch.device("cuda", 0))
        try:
            from sparseharness_amd.distributed import HipLocalStep, ShardedIteration, ShardPlan
            rp, ci, va = H.rmat(17, seed=9)
            vals = va.astype(O.elem_dtype(sr))
            a, b = (0.0, 0.0) if sr == O.MIN_PLUS_F32 else (1, 0)
            x0 = O.initial_vector(sr, len(rp) - 1)
            plan = ShardPlan(rp, ci, vals, 0, 1, chunks)
            step = HipLocalStep(plan, sr, 0)
            final, iters, conv = ShardedIteration(plan, sr, step, exchange=exchange).run(x0, x0, a, b, 1e-4, 80)
            q.put((0, final, iters, conv, step.A.describe()))
        finally:
            dist.destroy_process_group()
    except BaseException:   # noqa: BLE001
        q.put((0, "error", traceback.format_exc()))
        raise


@pytest.mark.parametrize("chunks,exchange", [(1, "collective"), (4, "collective"), (4, "p2p")])
def test_nccl_branch_of_the_driver_runs_beside_the_live_launch(chunks, exchange, plan):
    """The branch of ShardedIteration that runs on real hardware -- backend nccl (= RCCL), in-place
    all_gather_into_tensor (or the grouped isend / irecv fan-out) of every finished piece on a side stream WHILE the
    persistent phase 2 of the same iteration is still computing the later pieces -- executed with a real RCCL process
    group of one rank (the box has one GPU): the collective is a copy to itself, but the stream semantics, the piece
    reports and the waits are the ones eight ranks execute.  R-MAT-17 SSSP under the x-tiled plan, bit-exact."""
    if plan != "tiled":
        pytest.skip("once per run (SH_PLAN=tiled is what gives the persistent phase 2)")
    import tempfile

    import torch.multiprocessing as mp
    sr = O.MIN_PLUS_F32
    rp, ci, va = H.rmat(17, seed=9)
    x0 = O.initial_vector(sr, len(rp) - 1)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va.astype(np.float32), x0, x0, 0.0, 0.0, 1e-4, 80)
    tmp = tempfile.mkdtemp(prefix="sh_nccl1_")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(os.path.join(tmp, "store"), sr, chunks, exchange, q, tmp))
    p.start()
    try:
        try:
            item = q.get(timeout=300)
        except Exception:   # noqa: BLE001
            pytest.fail("the worker neither finished nor reported an error within 300 s\n" + open(os.path.join(tmp, "rank0.stderr")).read()[-4000:])
        if isinstance(item[1], str) and item[1] == "error":
            pytest.fail(item[2] + "\n" + open(os.path.join(tmp, "rank0.stderr")).read()[-4000:])
        p.join(timeout=60)
    finally:
        if p.is_alive():
            p.kill()
            p.join(timeout=30)
    _, final, iters, conv, layout = item
    assert layout.startswith("tiled"), layout
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(bits(final), bits(want))


@pytest.mark.parametrize("name", ["powerlaw_int", "rmat15", "ragged"])
@pytest.mark.parametrize("mode", [1, 2])
def test_or_and_on_bits_matches_oracle(eng, cases, name, mode, plan):
    """sh_plan_options::or_and_bits: SH_OR_AND_I32 launches run on the bit-blocked layout (x as a bitmap, 4-byte
    coordinate entries, partial result bitmaps; bits.hip.h) -- beside the ordinary plan (1) or instead of it (2).
    Every epilogue form, entries with a zero value, and an iteration loop to convergence, bit-exact against the
    oracle; with mode 1 the other semirings still run on the ordinary plan, with mode 2 they are refused."""
    if plan != "tiled":
        pytest.skip("once per run (the bit layout does not depend on the float plans)")
    rp, ci, va, n = cases[name]
    rng = np.random.default_rng(17)
    vals = va.astype(np.int32)
    vals[rng.integers(0, len(vals), len(vals) // 9)] = 0
    A = eng.upload_csr(n, n, rp, ci, vals, or_and_bits=mode)
    assert "or_and=bits(" in A.describe() and ("only" in A.describe()) == (mode == 2), A.describe()
    out = eng.alloc(n).fill(0)
    for density, a, b in [(0.0, 1, 0), (0.01, 1, 0), (0.3, 1, 1), (1.0, 0, 1), (0.05, 3, 0)]:
        x = ((rng.random(n) < density) * rng.integers(1, 5, n)).astype(np.int32)
        y = rng.integers(0, 2, n).astype(np.int32)
        xv, yv = eng.vector(x), eng.vector(y)
        eng.spmv(O.OR_AND_I32, A, xv, yv, a, b, out)
        np.testing.assert_array_equal(out.download(np.int32), O.kernel(O.OR_AND_I32, rp, ci, vals, x, y, a, b))
        xv.free()
        yv.free()
    x0 = O.initial_vector(O.OR_AND_I32, n)
    want, w_it, w_conv = O.iterate(O.OR_AND_I32, rp, ci, vals, x0, x0, 1, 0, 1e-4, 100)
    xv, yv, sc = eng.vector(x0), eng.vector(x0), eng.alloc(n)
    iters, conv, _, _ = eng.iterate(O.OR_AND_I32, A, xv, yv, sc, 1, 0, 1e-4, 100)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(xv.download(np.int32), want)
    fx = eng.vector(np.ones(n, np.float32))
    if mode == 2:
        with pytest.raises(Exception, match="SH_OR_AND_I32 launches only"):
            eng.spmv(O.PLUS_TIMES_F32, A, fx, None, 1.0, 0.0, out)
    else:
        eng.spmv(O.MAX_MIN_I32, A, xv, yv, O.INT_MAX, O.INT_MIN, out)
        np.testing.assert_array_equal(out.download(np.int32), O.kernel(O.MAX_MIN_I32, rp, ci, vals, xv.download(np.int32), x0, O.INT_MAX, O.INT_MIN))
    for v in (xv, yv, sc, out, fx):
        v.free()
    A.free()


def test_or_and_on_bits_wide_matrix_and_sharded_driver(monkeypatch, plan):
    """More than one row range and column block (700 K x 1.3 M), then the multi-GPU iteration driver's device seam on
    the bit layout (row -> element mapping of the pieces, piece reports) on one rank."""
    if plan != "tiled":
        pytest.skip("once per run")
    import torch
    from sparseharness_amd.distributed import HipLocalStep, ShardedIteration, ShardPlan
    rng = np.random.default_rng(23)
    rows, cols = 700_000, 1_300_000
    deg = rng.poisson(4, rows).astype(np.int64)
    deg[5] = 200_000
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.integers(0, cols, int(rp[-1])).astype(np.int32)
    vals = rng.integers(0, 3, int(rp[-1])).astype(np.int32)
    x = (rng.random(cols) < 0.02).astype(np.int32)
    y = rng.integers(0, 2, rows).astype(np.int32)
    with Engine(0) as e2:
        A = e2.upload_csr(rows, cols, rp, ci, vals, or_and_bits=2)
        xv, yv, out = e2.vector(x), e2.vector(y), e2.alloc(rows).fill(0)
        e2.spmv(O.OR_AND_I32, A, xv, yv, 1, 1, out)
        np.testing.assert_array_equal(out.download(np.int32), O.kernel(O.OR_AND_I32, rp, ci, vals, x, y, 1, 1))
    monkeypatch.setenv("SH_OR_AND_BITS", "1")
    rp2, ci2, va2 = H.rmat(15, seed=5)
    n = len(rp2) - 1
    v2 = va2.astype(np.int32)
    x0 = O.initial_vector(O.OR_AND_I32, n)
    want, w_it, w_conv = O.iterate(O.OR_AND_I32, rp2, ci2, v2, x0, x0, 1, 0, 1e-4, 60)
    torch.cuda.set_device(0)
    sp = ShardPlan(rp2, ci2, v2, 0, 1, 3)
    step = HipLocalStep(sp, O.OR_AND_I32, 0)
    assert "or_and=bits(" in step.A.describe()
    final, iters, conv = ShardedIteration(sp, O.OR_AND_I32, step).run(x0, x0, 1, 0, 1e-4, 60)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(bits(final), bits(want))
