"""Host-side logic of the product (no GPU): the C++ mirror's MatrixMarket loader,
kernel-config/run-file/CLI surface, the in-harness gold, and the generators."""
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
    for k in ["spmv.json", "sssp.json", "bfs.json", "spmv_chunk128.json"]:
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
