"""Host-side logic of the product (no GPU): the C++ mirror's MatrixMarket loader,
kernel-config/run-file/CLI surface, the in-harness gold, and the generators."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden, mtx
from oracle import oracle as O
from sparseharness_amd import hostlib as H

HOST = os.path.join(ROOT, "sparseharness_amd", "host")
KERNELS = os.path.join(HOST, "kernels")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", HOST, "../libsparseharness_host.so", "bin/spmv_harness"],
                          stderr=subprocess.DEVNULL)


def test_product_loader_matches_reference_rows(matrix_name):
    g = golden(matrix_name)
    rows, cols, hdr, rp, ci, va = H.mm_load(mtx(matrix_name))
    assert [rows, cols, hdr] == g["dims"].tolist()
    np.testing.assert_array_equal(rp, g["f32_row_ptr"])
    np.testing.assert_array_equal(ci, g["f32_col_idx"])
    np.testing.assert_array_equal(va.view(np.uint32), g["f32_val"].view(np.uint32))
    _, _, _, rp, ci, va = H.mm_load(mtx(matrix_name), elem_is_int=True)
    np.testing.assert_array_equal(ci, g["i32_col_idx"])
    np.testing.assert_array_equal(va, g["i32_val"])


@pytest.mark.parametrize("name", ["matrix2", "matrix3", "matrix4"])   # see tests/test_oracle.py PR_OK
def test_product_pagerank_normaliser_matches_reference_rows(name):
    g = golden(name)
    _, _, _, rp, ci, va = H.mm_load(mtx(name), normalise=H.NORM_PAGERANK, damping=0.85)
    np.testing.assert_array_equal(rp, g["pr_row_ptr"])
    np.testing.assert_array_equal(ci, g["pr_col_idx"])
    np.testing.assert_array_equal(va.view(np.uint32), g["pr_val"].view(np.uint32))


def test_product_scc_normaliser_matches_reference_rows(matrix_name):
    g = golden(matrix_name)
    _, _, _, rp, ci, va = H.mm_load(mtx(matrix_name), elem_is_int=True, normalise=H.NORM_SCC)
    np.testing.assert_array_equal(rp, g["scc_row_ptr"])
    np.testing.assert_array_equal(ci, g["scc_col_idx"])
    np.testing.assert_array_equal(va, g["scc_val"])


def test_pagerank_normaliser_without_truncation_is_column_stochastic():
    # what the app was meant to compute (SH_NO_TRUNCATE): every column with entries sums to the damping factor
    rows, cols, _, rp, ci, va = H.mm_load(mtx("matrix2"), truncate=False, normalise=H.NORM_PAGERANK, damping=0.85)
    sums = np.bincount(ci, weights=va.astype(np.float64), minlength=cols)
    has = np.bincount(ci, minlength=cols) > 0
    np.testing.assert_allclose(sums[has], 0.85, rtol=1e-4)
    # and the truncated (reference-faithful) load of the same file is all zeros, quirk A-3
    assert not H.mm_load(mtx("matrix2"), normalise=H.NORM_PAGERANK)[5].any()


def test_no_truncate_mode_keeps_real_values():
    _, _, _, rp, ci, va = H.mm_load(mtx("matrix"), truncate=False)
    assert abs(float(va[0]) - 1474.779) < 1e-2  # first entry of HB/1138_bus


def run_app(app, matrix, kernel, *extra):
    cmd = [os.path.join(HOST, "bin", app), "-m", mtx(matrix), "-f", matrix, "-k", os.path.join(KERNELS, kernel),
           "-r", os.path.join(KERNELS, "runfile.csv"), "-n", "testhost", "-e", "exp1", *extra]
    env = {k: v for k, v in os.environ.items() if k != "SH_QUIET_TIMERS"}
    return subprocess.run(cmd, capture_output=True, text=True, env=env)


def test_spmv_app_gold_only_is_config1(matrix_name):
    # BASELINE.json config 1: example/matrix.mtx float SpMV on the CPU gold path, no GPU
    g = golden(matrix_name)["gold_x1"]
    r = run_app("spmv_harness", matrix_name, "spmv.json", "--gold_only")
    assert r.returncode == 0, r.stderr[-400:]
    line = [l for l in r.stdout.splitlines() if l.startswith("GOLD")][0]
    assert f"rows={len(g)}" in line
    assert float(line.split("sum=")[1].split()[0]) == float(g.astype(np.float64).sum())
    assert 'PROFILING_DATUM("spmv", "gold"' in r.stderr  # timer line format kept


def test_app_error_conventions(tmp_path):
    r = subprocess.run([os.path.join(HOST, "bin", "spmv_harness"), "-m", "x"], capture_output=True, text=True)
    assert r.returncode == 255 and 'Required argument "matrix_name" not set.' in r.stdout  # exit(-1)
    bad = tmp_path / "rect.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 3 1\n1 1 1.0\n")
    r = subprocess.run([os.path.join(HOST, "bin", "spmv_harness"), "-m", str(bad), "-f", "m", "-k",
                        os.path.join(KERNELS, "spmv.json"), "-r", os.path.join(KERNELS, "runfile.csv"), "-n", "h",
                        "-e", "e"], capture_output=True, text=True)
    assert r.returncode == 2 and "Matrix is not square" in r.stdout
    r = run_app("spmv_harness", "matrix3", "spmv.json")  # no GPU here -> log + exit(1), never a CPU fallback
    if H.load() and __import__("sparseharness_amd").abi.load().sh_device_count() == 0:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def test_shipped_kernel_configs_parse():
    for k in ["spmv.json", "sssp.json", "bfs.json", "pr.json", "scc.json", "spmv_chunk128.json"]:
        d = json.load(open(os.path.join(KERNELS, k)))
        assert {"name", "source", "properties", "inputArgs", "outputArg", "tempGlobals", "tempLocals",
                "paramVars"} <= set(d)


def test_powerlaw_generator_properties():
    rp, ci, va = H.powerlaw(50_000, 1_000_000)
    rp2, ci2, va2 = H.powerlaw(50_000, 1_000_000)
    assert rp[-1] == 1_000_000 and (np.diff(rp) >= 0).all()
    np.testing.assert_array_equal(ci, ci2)
    np.testing.assert_array_equal(va, va2)
    np.testing.assert_array_equal(rp, rp2)
    assert ci.min() >= 0 and ci.max() < 50_000 and va.min() >= 1 and va.max() <= 16
    deg = np.diff(rp)
    assert deg.max() > 50 * np.median(deg)  # heavy tail
    assert (va == np.floor(va)).all()


def test_rmat_generator_properties():
    rp, ci, va = H.rmat(12)
    assert len(rp) == 4097 and rp[-1] == 16 * 4096 and ci.min() >= 0 and ci.max() < 4096
    rp2, ci2, va2 = H.rmat(12)
    np.testing.assert_array_equal(ci, ci2)
    np.testing.assert_array_equal(va, va2)
    deg = np.diff(rp)
    assert deg.max() > 20 * max(1, np.median(deg))  # skewed
    # within a row entries are ordered by (col, weight)
    r = int(np.argmax(deg))
    assert (np.diff(ci[rp[r]:rp[r + 1]]) >= 0).all()


def test_host_gold_equals_oracle_gold_on_synthetic():
    # the in-harness gold (host/inc/spmv_gold.h) is exercised through the app on files;
    # here the oracle's gold is cross-checked against a float64 reference on synthetic data
    rp, ci, va = H.powerlaw(20_000, 300_000, seed=7)
    x = (1 + np.arange(20_000) % 7).astype(np.float32)
    y = O.gold_spmv(rp, ci, va, x)
    ref = np.add.reduceat((x[ci].astype(np.float64) * va), rp[:-1].clip(max=len(ci) - 1))
    ref[np.diff(rp) == 0] = 0
    np.testing.assert_allclose(y, ref, rtol=1e-6)


def _write_mtx(path, header, n, entries):
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {header}\n% generated by the test\n{n} {n} {len(entries)}\n")
        f.write("\n".join(entries) + "\n")


def test_number_formats_match_fscanf_semantics(tmp_path):
    """The hand-rolled tokenizer (exact decimal fast path + strtod fallback) must agree bit-for-bit
    with the oracle's fscanf("%d %d %lg") restatement on every spelling a .mtx may contain."""
    vals = ["1474.779", "-9.017", "+3.5", ".5", "5.", "1e3", "1E-3", "-2.5e+2", "123456789012345678901234567890",
            "0.1234567890123456789", "1e22", "1e23", "9007199254740993", "4.9e-324", "1.7976931348623157e308",
            "0", "-0.0", "007", "3.0000000000000001e5", "2147483647.9", "16", "1e-22", "12345.6789e-2"]
    entries = [f"{i + 1} {(i * 7) % len(vals) + 1} {v}" for i, v in enumerate(vals)]
    p = tmp_path / "formats.mtx"
    _write_mtx(p, "real general", len(vals), entries)
    for trunc in (True, False):
        got = H.mm_load(str(p), truncate=trunc)
        if trunc:
            want = O.mm_load(str(p))
            assert got[:3] == want[:3]
            for a, b in zip(got[3:], want[3:]):
                np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                              b.view(np.uint32) if b.dtype == np.float32 else b)
        else:   # untruncated values == numpy's correctly rounded float32(float64(text))
            with np.errstate(over="ignore"):
                ref = {(i, (i * 7) % len(vals)): np.float32(np.float64(v)) for i, v in enumerate(vals)}
            rows, cols, _, rp, ci, va = got
            for r in range(rows):
                for j in range(rp[r], rp[r + 1]):
                    assert va[j].view(np.uint32) == ref[(ci[j], r)].view(np.uint32), (ci[j], r)


@pytest.mark.parametrize("header,sym", [("real general", False), ("integer symmetric", True), ("pattern symmetric", True)])
def test_parallel_parse_path_matches_oracle(tmp_path, header, sym):
    """> 1 MiB body and >= 50 000 entries takes the sliced (OpenMP) tokenizer."""
    rng = np.random.default_rng(8)
    n, m = 60_000, 120_000
    I, J = rng.integers(1, n + 1, m), rng.integers(1, n + 1, m)
    if "pattern" in header:
        entries = [f"{a} {b}" for a, b in zip(I, J)]
    elif "integer" in header:
        entries = [f"{a} {b} {c}" for a, b, c in zip(I, J, rng.integers(-99, 100, m))]
    else:
        entries = [f"{a}   {b}\t{c:.9f} " for a, b, c in zip(I, J, rng.uniform(-1e4, 1e4, m))]
    p = tmp_path / "big.mtx"
    _write_mtx(p, header, n, entries)
    assert os.path.getsize(p) > (1 << 20)
    for as_int in (False, True):
        got, want = H.mm_load(str(p), elem_is_int=as_int), O.mm_load(str(p), elem_is_int=as_int)
        assert got[:3] == want[:3]
        for a, b in zip(got[3:], want[3:]):
            np.testing.assert_array_equal(a.view(np.int32), b.view(np.int32))


def test_entries_not_one_per_line_fall_back_to_token_scanner(tmp_path):
    # the reference's fscanf loop ignores line structure; so must the loader
    rng = np.random.default_rng(9)
    n, m = 50_000, 60_000
    toks = []
    for a, b, c in zip(rng.integers(1, n + 1, m), rng.integers(1, n + 1, m), rng.integers(1, 99, m)):
        toks += [str(a), str(b), f"{c}.25"]
    p = tmp_path / "wrapped.mtx"
    with open(p, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real general\n{n} {n} {m}\n")
        for k in range(0, len(toks), 7):          # 7 tokens per line: entries straddle lines
            f.write(" ".join(toks[k:k + 7]) + "          \n")
        f.write("\n" * 20000 + " " * (1 << 20) + "\n")   # make the body > 1 MiB
    got, want = H.mm_load(str(p)), O.mm_load(str(p))
    assert got[:3] == want[:3]
    for a, b in zip(got[3:], want[3:]):
        np.testing.assert_array_equal(a.view(np.int32), b.view(np.int32))


def test_cpp_host_mirror_selftest(tmp_path):
    """Run / CSV / size-expression evaluator / SqlStat text / KernelConfig JSON / loader + encode + gold,
    checked in C++ against the behaviour the reference documents (host/test/host_selftest.cpp)."""
    subprocess.check_call(["make", "-s", "-C", HOST, "selftest"], stderr=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "bin", "host_selftest"), str(tmp_path)], capture_output=True, text=True,
                       env={**os.environ, "SH_QUIET_TIMERS": "1"})
    assert r.returncode == 0 and "all passed" in r.stdout, r.stdout[-800:]


def test_binary_csr_cache_roundtrip_and_invalidation(tmp_path, monkeypatch):
    """the rows are cached next to (or, here, away from) the file and only reused while the
    source file's size and mtime are unchanged; a damaged cache falls back to parsing."""
    import shutil
    src = tmp_path / "m.mtx"
    shutil.copyfile(mtx("matrix5"), src)
    cache = tmp_path / "cache"
    cache.mkdir()
    fresh = H.mm_load(str(src))
    monkeypatch.setenv("SH_CSR_CACHE", str(cache))
    first = H.mm_load(str(src))
    files = sorted(os.listdir(cache))
    assert files == ["m.mtx.shcsr.f32"]
    second = H.mm_load(str(src))                       # served from the cache
    for a, b, c in zip(fresh, first, second):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
        np.testing.assert_array_equal(np.asarray(a), np.asarray(c))
    # element type and truncation mode have their own cache files
    H.mm_load(str(src), elem_is_int=True)
    raw = H.mm_load(str(src), truncate=False)
    assert sorted(os.listdir(cache)) == ["m.mtx.shcsr.f32", "m.mtx.shcsr.f32.raw", "m.mtx.shcsr.i32"]
    assert not np.array_equal(raw[5], first[5])        # matrix5 has real values: untruncated rows differ
    # proof that the cache is what is read: doctor a cached value, reload, see it
    p = cache / "m.mtx.shcsr.f32"
    blob = bytearray(p.read_bytes())
    blob[-4:] = np.float32(12345.0).tobytes()
    p.write_bytes(bytes(blob))
    assert H.mm_load(str(src))[5][-1] == np.float32(12345.0)
    # the source file changes -> stale cache ignored and rewritten
    st = os.stat(src)
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    again = H.mm_load(str(src))
    np.testing.assert_array_equal(again[5], fresh[5])
    assert H.mm_load(str(src))[5][-1] == fresh[5][-1]
    # truncated cache file -> parse again
    p.write_bytes(p.read_bytes()[:100])
    np.testing.assert_array_equal(H.mm_load(str(src))[5], fresh[5])
    # PageRank (needs file order) never goes through the cache
    pr = H.mm_load(mtx("matrix3"), normalise=H.NORM_PAGERANK)
    assert pr[0] == 20


REF_EXAMPLE = os.path.join(os.environ.get("SH_REFERENCE", "/root/reference"), "example")


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLE), reason="reference tree not present (GPU box)")
def test_reference_kernel_configs_are_accepted_unmodified():
    """The reference's own example/<algo>/kernel*.json (7 Lift strategies x 5 apps) load into KernelConfig and
    drive executorEncodeMatrix with the reference's size rules: MHeight padding A-6, MWidthC = cl_width/splitSize
    or cols for ragged layouts (inc/kernel_utils.h:57-65); the OpenCL source only serves as the semiring hint."""
    subprocess.check_call(["make", "-s", "-C", HOST, "selftest"], stderr=subprocess.DEVNULL)
    exe = os.path.join(HOST, "bin", "host_selftest")
    want_sr = {"spmv": "plus-times", "pr": "plus-times", "sssp": "min-plus", "bfs": "or-and", "scc": "max-min"}
    for algo, sr in want_sr.items():
        files = sorted(glob.glob(os.path.join(REF_EXAMPLE, algo, "kernel*.json")))
        assert len(files) == 7, algo
        mode = "--encode-int" if algo in ("bfs", "scc") else "--encode"
        r = subprocess.run([exe, mode, mtx("matrix3"), *files], capture_output=True, text=True, timeout=120,
                           env={k: v for k, v in os.environ.items() if k != "SH_QUIET_TIMERS"})
        assert r.returncode == 0, r.stderr[-500:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("ENCODE ")]
        assert len(lines) == 7
        by_name = {}
        for l in lines:
            f = dict(kv.split("=", 1) for kv in l.split()[2:])
            assert f["semiring"] == sr, l
            by_name[f["name"]] = f
        # matrix3: 20 rows, longest row 9 (tests/golden): the reference's size rules
        assert by_name["glb-sdp"]["size_args"] == "20,9,20," and by_name["glb-sdp"]["output"] == "80"
        assert by_name["swrg-slcl-sdp-chunk-128"]["rows"] == "128"                 # 20 + (128 - 20 % 128), quirk A-6
        assert by_name["awrg-alcl-fdp-chunk-rsa-8"]["size_args"] == "24,20,"         # ragged: MWidthC = cols
        assert by_name["awrg-alcl-alcl-edp-split-8"]["size_args"] == "20,2,20,"      # (9 + (8 - 9 % 8)) / 8
