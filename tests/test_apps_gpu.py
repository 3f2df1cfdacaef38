"""The C++ host mirror end to end on the GPU: the three apps (reference CLI, Harness<> /
IterativeHarness<> subclasses over the C ABI) on the reference's example matrices."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden, mtx

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "sparseharness_amd", "host")
KERNELS = os.path.join(HOST, "kernels")


def run_app(app, matrix, kernel, env_extra=None, *extra, runfile=None):
    cmd = [os.path.join(HOST, "bin", app), "-m", mtx(matrix), "-f", matrix, "-k", os.path.join(KERNELS, kernel),
           "-r", runfile or os.path.join(KERNELS, "runfile.csv"), "-n", "gpubox", "-e", "exp7", "-i", "3", *extra]
    env = {k: v for k, v in os.environ.items() if k != "SH_QUIET_TIMERS"}
    env.update(env_extra or {})
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)


def test_spmv_app_reports_correct(matrix_name):
    r = run_app("spmv_harness", matrix_name, "spmv.json")
    assert r.returncode == 0, r.stderr[-800:]
    sql = [l for l in r.stdout.splitlines() if l.startswith("INSERT INTO table_name")]
    assert len(sql) == 1
    # 3 raw trials checked against the in-harness gold with exact compare + the median row
    assert sql[0].count('"correct"') == 3 and sql[0].count('"statisticvalue"') == 1 and "badvalues" not in sql[0]
    assert re.search(r'\(\d+(\.\d+)?(e-?\d+)?, "correct", "csr-stream", 1280, 256, "gpubox", ".*", "%s",0,0,"RAW_RESULT", "exp7"\)'
                     % matrix_name, sql[0])
    assert 'PROFILING_DATUM("hipLaunchKernel", "harness"' in r.stdout
    assert "SH_PERF median_ms=" in r.stdout


def test_spmv_app_chunked_config_pads_height():
    # chunkSize 128 pads the height like the reference (quirk A-6); gold still compares the first H rows
    r = run_app("spmv_harness", "matrix4", "spmv_chunk128.json")
    assert r.returncode == 0 and r.stdout.count('"correct"') >= 3
    assert "v_MHeight_2 = 128" in r.stderr   # 111 + (128 - 111 % 128), quirk A-6


@pytest.mark.parametrize("app,kernel,tag", [("sssp_harness", "sssp.json", "sssp"), ("bfs_harness", "bfs.json", "bfs")])
@pytest.mark.parametrize("host_loop", ["0", "1"])
def test_iterative_apps_match_reference(matrix_name, app, kernel, tag, host_loop):
    g = golden(matrix_name)
    r = run_app(app, matrix_name, kernel, {"SH_HOST_LOOP": host_loop}, "-x", "2000")
    assert r.returncode == 0, r.stderr[-800:]
    res = [l for l in r.stdout.splitlines() if l.startswith("SH_RESULT")][0]
    iters, conv = int(g[tag + "_meta"][0]), int(g[tag + "_meta"][1])
    assert f"iterations={iters} converged={conv}" in res
    fin = g[tag + "_final"]
    if tag == "sssp":
        reached = int((fin < np.float32(3.4028235e38)).sum())
        assert f"reached={reached} " in res
        assert float(res.split("distance_sum=")[1]) == pytest.approx(float(fin[fin < 3e38].astype(np.float64).sum()), rel=1e-6)
    else:
        assert res.endswith(f"set={int((fin != 0).sum())}")
    sql = [l for l in r.stdout.splitlines() if l.startswith("INSERT INTO table_name")]
    assert len(sql) == 3                                    # one INSERT per trial
    assert sql[0].count("RAW_RESULT") == iters and "MULTI_ITERATION_SUM" in sql[0] and "MEDIAN_RESULT" in sql[0]


@pytest.mark.parametrize("host_loop", ["0", "1"])
@pytest.mark.parametrize("name", ["matrix2", "matrix3", "matrix4"])   # see tests/test_oracle.py PR_OK
def test_pagerank_app_matches_reference(name, host_loop):
    g = golden(name)
    r = run_app("pr_harness", name, "pr.json", {"SH_HOST_LOOP": host_loop}, "-x", "2000")
    assert r.returncode == 0, r.stderr[-800:]
    res = [l for l in r.stdout.splitlines() if l.startswith("SH_RESULT")][0]
    assert f"iterations={int(g['pr_meta'][0])} converged={int(g['pr_meta'][1])}" in res
    fin = g["pr_final"]
    assert float(res.split("rank_sum=")[1].split()[0]) == pytest.approx(float(fin.astype(np.float64).sum()), rel=1e-6)


@pytest.mark.parametrize("host_loop", ["0", "1"])
def test_scc_app_matches_reference(matrix_name, host_loop):
    g = golden(matrix_name)
    r = run_app("scc_harness", matrix_name, "scc.json", {"SH_HOST_LOOP": host_loop}, "-x", "2000")
    assert r.returncode == 0, r.stderr[-800:]
    res = [l for l in r.stdout.splitlines() if l.startswith("SH_RESULT")][0]
    fin = g["scc_final"]
    assert f"iterations={int(g['scc_meta'][0])} converged={int(g['scc_meta'][1])}" in res
    assert f"labels={len(set(fin.tolist()))} label_sum={int(fin.astype(np.int64).sum())}" in res


def test_experiment_sweep_end_to_end(tmp_path):
    """SURVEY 8f-4 on the GPU: run_all.py over 2 matrices x 2 kernel configs, then the SQL collector."""
    import csv
    import sys
    scripts = os.path.join(ROOT, "scripts")
    data = tmp_path / "data"
    for name in ("matrix3", "matrix4"):
        (data / name).mkdir(parents=True)
        os.symlink(mtx(name), data / name / (name + ".mtx"))
    (data / "datasets.txt").write_text("matrix3\nmatrix4\n")
    kern = tmp_path / "kernels"
    kern.mkdir()
    for k in ("spmv.json", "spmv_chunk128.json"):
        os.symlink(os.path.join(KERNELS, k), kern / k)
    env = {k: v for k, v in os.environ.items() if k != "SH_QUIET_TIMERS"}
    r = subprocess.run([sys.executable, os.path.join(scripts, "run_all.py"), str(data), os.path.join(HOST, "bin", "spmv_harness"),
                        str(kern), os.path.join(KERNELS, "runfile.csv"), "0", str(tmp_path / "res"), "--experiment", "sweep1",
                        "--trials", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-800:]
    r = subprocess.run([sys.executable, os.path.join(scripts, "build_query.py"), str(tmp_path / "res"), "t", "--out",
                        str(tmp_path / "q.sql"), "--csv", str(tmp_path / "q.csv")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "4 INSERT statements" in r.stdout, r.stdout
    rows = list(csv.DictReader(open(tmp_path / "q.csv")))
    assert len(rows) == 4 * 4                                   # 3 raw trials + the median row per run
    assert {r_["matrix"] for r_ in rows} == {"matrix3", "matrix4"} and {r_["experiment_id"] for r_ in rows} == {"sweep1"}
    assert sum(r_["correct"] == "correct" for r_ in rows) == 12 and not any(r_["correct"] == "badvalues" for r_ in rows)


def test_bench_script_emits_the_contract_line():
    """bench.py at a reduced size: one JSON line with the driver's fields, roofline and cpu_baseline objects."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "300000", "--nnz", "6000000", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "GFLOP/s" and d["dtype"] == "f32"
    assert "workload" in d["config"] and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and d["roofline"]["bound"] == "hbm"
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["cores"] == 1
    ac = d["cpu_baseline"]["all_cores"]
    assert ac["cores"] >= 1 and ac["value"] > 0 and ac["same_bits_as_single_thread"]
    assert d["parity"]["mismatches_rel_1e-5"] == 0 and d["value"] > 0
    assert d["frac_raw_values"] is None or 0 < d["frac_raw_values"] < 1


def test_bench_script_two_ranks_rehearsal():
    """The N > 1 path of bench.py (row sharding, barriers, max-over-ranks, rank-0 JSON) with two ranks sharing
    GPU 0 over gloo (SH_BENCH_REHEARSAL=1); the driver runs the real thing, one rank per GPU over RCCL."""
    import json
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SH_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "400000",
                        "--nnz", "8000000", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-1200:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["nnz"] == 8000000
    assert 0 < d["roofline"]["rank_nnz"] < 8000000 and d["parity"]["mismatches_rel_1e-5"] == 0
    assert d["cpu_baseline"] is None and "rehearsal" in d


def test_bench_script_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no launcher around it, WORLD_SIZE unset): the parent spawns the ranks
    before it touches torch / HIP, relays rank 0's JSON line and exits with the children's return code."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SH_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "400000", "--nnz", "8000000",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-1200:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity"]["mismatches_rel_1e-5"] == 0 and "rehearsal" in d
    # and a failing rank fails the parent
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "400000", "--nnz", "8000000"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, SH_LIB="/nonexistent/engine.so"))
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("spec,workload,expect_correct", [("synth:rmat:23", "rmat-23", True),
                                                         ("synth:powerlaw:10000000:200000000", "powerlaw-10M-200M", False)])
def test_headline_sizes_through_the_harness_boundary(spec, workload, expect_correct):
    """BASELINE configs 3 and 5 at full size through the C++ mirror of the reference's app: spmv_harness ->
    Harness<>::benchmark -> executeKernel -> sh_spmv (app/spmv.cpp:52-110), not through the Python binding.
    The median kernel time it reports must agree with what bench.py measures for the same workload.
    R-MAT-23 with x = 1 is exact in float, so every trial must be labelled "correct" by the reference's
    exact compare; the power-law matrix has one 2.4 M-entry row whose sum passes 2^24 (the sequential
    float gold itself is inexact there), so that run may be labelled "badvalues" -- its parity is
    asserted by bench.py's own check, which knows about that row."""
    import json
    import sys
    cmd = [os.path.join(HOST, "bin", "spmv_harness"), "-m", spec, "-f", workload, "-k", os.path.join(KERNELS, "spmv.json"),
           "-r", os.path.join(KERNELS, "runfile.csv"), "-n", "gpubox", "-e", "headline", "-i", "11", "-t", "1000"]
    env = dict(os.environ, SH_QUIET_TIMERS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    perf = [l for l in r.stdout.splitlines() if l.startswith("SH_PERF")]
    sql = [l for l in r.stdout.splitlines() if l.startswith("INSERT INTO table_name")]
    assert len(perf) == 1 and len(sql) == 1 and sql[0].count("RAW_RESULT") == 11
    if expect_correct:
        assert sql[0].count('"correct"') == 11 and "badvalues" not in sql[0]
    app_ms = float(perf[0].split("median_ms=")[1].split()[0])
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline", "--no-ablation"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert b.returncode == 0, b.stderr[-800:]
    d = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][0])
    bench_ms = d["roofline"]["avg_launch_ms"]
    assert d["parity"]["mismatches_rel_1e-5"] == 0
    # one timed launch at a time (events around each launch) against 20 back-to-back launches: allow 15 %
    assert abs(app_ms - bench_ms) <= 0.15 * bench_ms, (app_ms, bench_ms)


def test_bfs_app_at_config4_size_on_the_layout_it_picks_itself():
    """bfs_harness -m synth:rmat:23: BASELINE config 4's graph through the C++ mirror of app/bfs.cpp:94-174.  For a
    matrix of this size the harness uploads the bit-blocked (or,and) layout alone (harness.h) -- 32 row ranges x 16
    column blocks: the default product path for BFS at this size, which the driver-run suite used to see only up to
    700 K rows.  Iteration count and the number of reached vertices against the oracle loop."""
    from oracle import oracle as O
    from sparseharness_amd import hostlib as H
    rp, ci, va = H.rmat(23)
    n = 1 << 23
    x0 = O.initial_vector(O.OR_AND_I32, n)
    want, w_it, w_conv = O.iterate(O.OR_AND_I32, rp, ci, va.astype(np.int32), x0, x0, 1, 0, 1e-4, 200)
    cmd = [os.path.join(HOST, "bin", "bfs_harness"), "-m", "synth:rmat:23", "-f", "rmat-23", "-k", os.path.join(KERNELS, "bfs.json"),
           "-r", os.path.join(KERNELS, "runfile.csv"), "-n", "gpubox", "-e", "config4", "-i", "1", "-t", "1000", "-x", "200"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SH_QUIET_TIMERS="1"), timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    res = [l for l in r.stdout.splitlines() if l.startswith("SH_RESULT")][0]
    assert f"iterations={w_it} converged={int(w_conv)}" in res, res
    assert res.endswith(f"set={int((want != 0).sum())}"), res


@pytest.mark.parametrize("host_loop", ["0", "1"])
def test_bfs_app_on_the_bit_blocked_layout(matrix_name, host_loop):
    """bfs_harness with the matrix uploaded in the bit-blocked (or,and) layout only (what the harness does by itself for
    large matrices; forced here on the reference's example matrices): same launches, same final vector."""
    g = golden(matrix_name)
    r = run_app("bfs_harness", matrix_name, "bfs.json", {"SH_HOST_LOOP": host_loop, "SH_OR_AND_BITS": "2"}, "-x", "2000")
    assert r.returncode == 0, r.stderr[-800:]
    res = [l for l in r.stdout.splitlines() if l.startswith("SH_RESULT")][0]
    assert f"iterations={int(g['bfs_meta'][0])} converged={int(g['bfs_meta'][1])}" in res
    assert res.endswith(f"set={int((g['bfs_final'] != 0).sum())}")
