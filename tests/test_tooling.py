"""the experiment tooling (scripts/run_all.py, analyse.py, build_query.py) on CPU.
The sweep runs the spmv app in --gold_only mode (no GPU); the SQL collector is fed rows in the
format the apps print (host/inc/sql_stat.h).  The GPU end-to-end sweep is in tests/test_apps_gpu.py."""
import csv
import gzip
import os
import subprocess
import sys
import tarfile

from conftest import ROOT

HOST = os.path.join(ROOT, "sparseharness_amd", "host")
SCRIPTS = os.path.join(ROOT, "scripts")


def run(script, *args):
    env = {k: v for k, v in os.environ.items() if k != "SH_QUIET_TIMERS"}   # hostlib sets it for in-process loads
    return subprocess.run([sys.executable, os.path.join(SCRIPTS, script), *map(str, args)], capture_output=True,
                          text=True, timeout=300, cwd=ROOT, env=env)


def test_sweep_driver_and_profile_summary(tmp_path):
    subprocess.check_call(["make", "-s", "-C", HOST, "../libsparseharness_host.so", "bin/spmv_harness"],
                          stderr=subprocess.DEVNULL)
    data = tmp_path / "data"
    for name in ("matrix3", "matrix4"):
        (data / name).mkdir(parents=True)
        os.symlink(os.path.join(ROOT, "tests", "golden", name + ".mtx"), data / name / (name + ".mtx"))
    (data / "datasets.txt").write_text("matrix3\nmatrix4\n")
    kernels = tmp_path / "kernels"
    kernels.mkdir()
    for k in ("spmv.json", "spmv_chunk128.json"):
        os.symlink(os.path.join(HOST, "kernels", k), kernels / k)
    r = run("run_all.py", data, os.path.join(HOST, "bin", "spmv_harness"), kernels,
            os.path.join(HOST, "kernels", "runfile.csv"), 0, tmp_path / "res", "--experiment", "exp42", "--",
            "--gold_only")
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
    root = tmp_path / "res" / "results-exp42"
    status = (root / "runstatus.txt").read_text()
    assert "taskcount: 4" in status and "finished experiments: 4 runs, 0 failed" in status
    res = root / "matrix3" / "result_spmv.txt.gz"
    text = gzip.open(res, "rt").read()
    assert "GOLD rows=20 sum=777" in text and "PROFILING_DATUM(" in text
    # a failing run is reported, not hidden
    (data / "datasets.txt").write_text("matrix3\nmissing\n")
    r = run("run_all.py", data, os.path.join(HOST, "bin", "spmv_harness"), kernels,
            os.path.join(HOST, "kernels", "runfile.csv"), 0, tmp_path / "res2", "--experiment", "e2", "--", "--gold_only")
    assert r.returncode == 1 and "run failed!" in r.stdout

    r = run("analyse.py", res)
    assert r.returncode == 0, r.stderr
    summ = list(csv.reader(open(root / "matrix3" / "profile_summary_spmv.txt")))
    assert summ[0] == ["method", "context", "language", "calls", "minimum", "mean", "maximum", "total"]
    by = {(r_[0], r_[1]): r_ for r_ in summ[1:]}
    assert ("load_from_file", "SparseMatrix") in by and by[("load_from_file", "SparseMatrix")][2] == "C++"
    totals = [float(r_[7]) for r_ in summ[1:]]
    assert totals == sorted(totals, reverse=True)
    n_data = sum(1 for _ in open(root / "matrix3" / "profiling_data_spmv.txt"))
    assert n_data == sum(int(r_[3]) for r_ in summ[1:]) == text.count("PROFILING_DATUM(")
    assert "method" in open(root / "matrix3" / "profile_summary_readable_spmv.txt").readline()


def test_analyse_drops_debug_lines_and_aggregates(tmp_path):
    f = tmp_path / "result_x.txt"
    f.write_text('noise\nPROFILING_DATUM("a", "ctx", 1.5, "C++")\nPROFILING_DATUM("a", "ctx", 0.5, "C++")\n'
                 '[DINFO] PROFILING_DATUM("a", "ctx", 99, "C++")\nPROFILING_DATUM("b", "k", 3, "HIP")\n'
                 ' x PROFILING_DATUM("late", "ctx", 1, "C++")\n')
    assert run("analyse.py", f).returncode == 0
    rows = list(csv.reader(open(tmp_path / "profile_summary_x.txt")))[1:]
    assert rows == [["b", "k", "HIP", "1", "3", "3", "3", "3"], ["a", "ctx", "C++", "2", "0.5", "1", "1.5", "2"]]


SQL = ('INSERT INTO table_name (time, correct, kernel, global, local, host, device, matrix, iteration, trial,statistic, '
       'experiment_id) VALUES (0.0123, "correct", "csr-stream", 1280, 256, "box", "AMD Instinct MI355X (gfx950)", "m3",0,0,'
       '"RAW_RESULT", "e1"),(0.0120, "notchecked", "csr-stream", 1280, 256, "box", "AMD Instinct MI355X (gfx950)", "m3",0,1,'
       '"MEDIAN_RESULT", "e1");')


def test_build_query_from_files_gz_and_archives(tmp_path):
    res = tmp_path / "results-e1"
    (res / "m3").mkdir(parents=True)
    (res / "m3" / "result_a.txt").write_text("Benchmarking run\n" + SQL + "\n")
    with gzip.open(res / "m3" / "result_b.txt.gz", "wt") as f:
        f.write("log prefix " + SQL + "\n")
    inner = tmp_path / "result_c.txt"
    inner.write_text(SQL + "\n")
    with tarfile.open(res / "m3" / "result_c.tar.gz", "w:gz") as t:     # the reference archives results like this
        t.add(inner, arcname="scratch/result_c.txt")
    r = run("build_query.py", res, "spmv_results", "--out", tmp_path / "q.sql", "--csv", tmp_path / "q.csv")
    assert r.returncode == 0 and "3 INSERT statements" in r.stdout and "6 rows" in r.stdout
    q = open(tmp_path / "q.sql").read().splitlines()
    assert len(q) == 3 and all(l.startswith("INSERT INTO spmv_results (time, correct") for l in q)
    rows = list(csv.reader(open(tmp_path / "q.csv")))
    assert rows[0][0] == "time" and len(rows) == 7
    assert rows[1] == ["0.0123", "correct", "csr-stream", "1280", "256", "box", "AMD Instinct MI355X (gfx950)", "m3",
                       "0", "0", "RAW_RESULT", "e1"]
