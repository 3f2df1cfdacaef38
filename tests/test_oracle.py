"""Pin the CPU restatement (oracle/sh_oracle.c) to the REAL reference.

Every expected value here was produced by reference code run in the authoring
container (oracle/ref/ref_driver.cpp -> tests/golden/*.npz): the reference's
MatrixMarket loader + row builder, Gold<float>::spmv, and its Lift glb-sdp
kernels for the three semirings driven by the apps' do-while loop.
All comparisons are bit-exact.
"""
import numpy as np
import pytest

from conftest import golden, mtx
from oracle import oracle as O


def test_mm_load_matches_reference_rows(matrix_name):
    g = golden(matrix_name)
    rows, cols, hdr, rp, ci, va = O.mm_load(mtx(matrix_name))
    assert [rows, cols, hdr] == g["dims"].tolist()
    np.testing.assert_array_equal(rp, g["f32_row_ptr"])
    np.testing.assert_array_equal(ci, g["f32_col_idx"])
    np.testing.assert_array_equal(va.view(np.uint32), g["f32_val"].view(np.uint32))
    _, _, _, rp, ci, va = O.mm_load(mtx(matrix_name), elem_is_int=True)
    np.testing.assert_array_equal(rp, g["i32_row_ptr"])
    np.testing.assert_array_equal(ci, g["i32_col_idx"])
    np.testing.assert_array_equal(va, g["i32_val"])


def test_known_sums_from_survey():
    # SURVEY.md 8c "golden values captured during this survey"
    expect = {"matrix": (1138, 2270.0, [1460, 1, 2, 2, 0]), "matrix3": (20, 777.0, [7, 9, 12, 14, 18]),
              "matrix4": (111, 193.0, [47, 2, 4, 13, 6]), "matrix5": (130, -4717836.0, [1, 0, 11, -28, -7]),
              "matrix2": (18772, 396160.0, [43, 2, 12, 1, 2])}
    for name, (n, s, head) in expect.items():
        _, _, _, rp, ci, va = O.mm_load(mtx(name))
        y = O.gold_spmv(rp, ci, va, np.ones(n, np.float32))
        assert len(y) == n and float(y.astype(np.float64).sum()) == s and y[:5].tolist() == head


def test_gold_matches_reference(matrix_name):
    g = golden(matrix_name)
    rp, ci, va = g["f32_row_ptr"], g["f32_col_idx"], g["f32_val"]
    n = len(rp) - 1
    x1 = np.ones(n, np.float32)
    xm = (1 + np.arange(n) % 7).astype(np.float32)
    for got, key in [(O.gold_spmv(rp, ci, va, x1), "gold_x1"),
                     (O.gold_spmv(rp, ci, va, xm), "gold_xmod"),
                     (O.gold_spmv(rp, ci, va, xm, y_const=3.0, alpha=2.0, beta=0.5), "gold_ab"),
                     (O.gold_dot(rp, ci, va, x1), "gold_x1")]:
        np.testing.assert_array_equal(got.view(np.uint32), g[key].view(np.uint32), err_msg=key)


def test_spmv_kernel_matches_reference(matrix_name):
    g = golden(matrix_name)
    rp, ci, va = g["f32_row_ptr"], g["f32_col_idx"], g["f32_val"]
    n = len(rp) - 1
    got = O.kernel(O.PLUS_TIMES_F32, rp, ci, va, np.ones(n), np.zeros(n), 1.0, 0.0)
    np.testing.assert_array_equal(got.view(np.uint32), g["kern_spmv_x1"].view(np.uint32))
    # the reference's own criterion: kernel output == gold exactly (inc/harness.h:134)
    assert O.check_result(g["gold_x1"], got) == O.CORRECT
    xm = 1 + np.arange(n) % 7
    ym = np.arange(n) % 5
    got = O.kernel(O.PLUS_TIMES_F32, rp, ci, va, xm, ym, 2.0, 0.5)
    np.testing.assert_array_equal(got.view(np.uint32), g["kern_spmv_ab"].view(np.uint32))


@pytest.mark.parametrize("sr,tag,a,b", [(O.MIN_PLUS_F32, "sssp", 0.0, 0.0), (O.OR_AND_I32, "bfs", 1, 0)])
def test_iterative_apps_match_reference(matrix_name, sr, tag, a, b):
    g = golden(matrix_name)
    pre = "i32" if sr == O.OR_AND_I32 else "f32"
    rp, ci, va = g[pre + "_row_ptr"], g[pre + "_col_idx"], g[pre + "_val"]
    n = len(rp) - 1
    x0 = O.initial_vector(sr, n)
    first = O.kernel(sr, rp, ci, va, x0, x0, a, b)
    np.testing.assert_array_equal(first.view(np.uint32), g[tag + "_first"].view(np.uint32))
    final, iters, conv = O.iterate(sr, rp, ci, va, x0, x0, a, b, delta=1e-4, max_iters=2000)
    assert [iters, int(conv)] == g[tag + "_meta"].tolist()
    np.testing.assert_array_equal(final.view(np.uint32), g[tag + "_final"].view(np.uint32))


def test_check_result_codes():
    g = np.array([1, 2, 3], np.float32)
    assert O.check_result(np.zeros(0, np.float32), g) == O.NOT_CHECKED
    assert O.check_result(g, g[:2]) == O.BAD_LENGTH
    assert O.check_result(g, np.array([1, 2, 4, 9], np.float32)) == O.BAD_VALUES
    assert O.check_result(g, np.array([1, 2, 3, 9], np.float32)) == O.CORRECT  # padded output ok


def test_out_of_range_index_substitutes_identity():
    # bounds ladder of example/*/kernel5.json:3 (idx < 0 or >= VLength -> identity)
    rp = np.array([0, 3], np.int32)
    ci = np.array([0, -1, 5], np.int32)
    x = np.array([2, 3], np.float32)
    out = O.kernel(O.PLUS_TIMES_F32, rp, ci, np.array([1, 10, 100], np.float32), x, np.zeros(1), 1.0, 0.0, vlength=2)
    assert out.tolist() == [2.0]
    out = O.kernel(O.MIN_PLUS_F32, rp, ci, np.array([1, 10, 100], np.float32), x, [O.FLT_MAX], 0.0, 0.0, vlength=2)
    assert out.tolist() == [3.0]


def test_mm_load_errors(tmp_path):
    p = tmp_path / "bad.mtx"
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(RuntimeError):
        O.mm_load(str(p))
    with pytest.raises(RuntimeError):
        O.mm_load(str(tmp_path / "missing.mtx"))


# ---- f3: PageRank and SCC ---------------------------------------------------------
# The reference's PageRank app divides by float column sums that are <= 0 for real-valued files and
# narrows inf/NaN through `int` (undefined behaviour): matrix and matrix5 produce machine-dependent
# garbage there (2000 iterations of NaN in the fixtures), so PageRank is pinned on the pattern /
# integer files only.  SCC is pinned on all five.
PR_OK = ["matrix2", "matrix3", "matrix4"]


@pytest.mark.parametrize("name", PR_OK)
def test_pagerank_matches_reference(name):
    g = golden(name)
    rows, cols, _, rp, ci, va = O.mm_load(mtx(name), normalise=O.NORM_PAGERANK, damping=0.85)
    np.testing.assert_array_equal(rp, g["pr_row_ptr"])
    np.testing.assert_array_equal(ci, g["pr_col_idx"])
    np.testing.assert_array_equal(va.view(np.uint32), g["pr_val"].view(np.uint32))
    n = rows
    x0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    y0 = np.ones(n, np.float32)
    beta = (np.float32(1.0) - np.float32(0.85)) / np.float32(n)
    first = O.kernel(O.PLUS_TIMES_F32, rp, ci, va, x0, y0, 1.0, beta)
    np.testing.assert_array_equal(first.view(np.uint32), g["pr_first"].view(np.uint32))
    final, iters, conv = O.iterate(O.PLUS_TIMES_F32, rp, ci, va, x0, y0, 1.0, beta, 1e-4, 2000)
    assert [iters, int(conv)] == g["pr_meta"].tolist()
    np.testing.assert_array_equal(final.view(np.uint32), g["pr_final"].view(np.uint32))


def test_scc_matches_reference(matrix_name):
    g = golden(matrix_name)
    rows, cols, _, rp, ci, va = O.mm_load(mtx(matrix_name), elem_is_int=True, normalise=O.NORM_SCC)
    np.testing.assert_array_equal(rp, g["scc_row_ptr"])
    np.testing.assert_array_equal(ci, g["scc_col_idx"])
    np.testing.assert_array_equal(va, g["scc_val"])
    x0 = O.initial_vector(O.MAX_MIN_I32, rows)
    y0 = np.full(rows, O.INT_MIN, np.int32)
    first = O.kernel(O.MAX_MIN_I32, rp, ci, va, x0, y0, O.INT_MAX, O.INT_MIN)
    np.testing.assert_array_equal(first, g["scc_first"])
    final, iters, conv = O.iterate(O.MAX_MIN_I32, rp, ci, va, x0, y0, O.INT_MAX, O.INT_MIN, 1e-4, 2000)
    assert [iters, int(conv)] == g["scc_meta"].tolist()
    np.testing.assert_array_equal(final, g["scc_final"])
