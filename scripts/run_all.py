#!/usr/bin/env python3
"""Dataset x kernel-config sweep driver (the job of the reference's
scripts/experiments/run_all.sh:53-108, same result layout, no bc/tar/git needed).

  scripts/run_all.py DATASETS APP KERNELFOLDER RUNFILE DEVICE [RESULTS_ROOT]

DATASETS      folder holding datasets.txt (one matrix name per line, matrix at
              <name>/<name>.mtx) -- or, without datasets.txt, every *.mtx below it
APP           one of the harness binaries (sparseharness_amd/host/bin/*_harness)
KERNELFOLDER  folder of kernel-config JSONs (every *.json is run on every matrix)
RUNFILE       launch-geometry CSV (inc/run.h format)
DEVICE        HIP device ordinal

Writes  <RESULTS_ROOT>/results-<exID>/<matrix>/result_<kernel>.txt.gz  (stdout+stderr of the
app: PROFILING_DATUM lines and SQL INSERT rows, what analyse.py / build_query.py read) and
runstatus.txt with progress and the time estimate.  exID = <git hash or 'nogit'>-<timestamp>,
as the reference builds it.  Flags passed to the app are the reference's: -i 5 -t 20 (override
with --trials / --timeout-ms), plus anything after `--`.
"""
import argparse
import glob
import gzip
import os
import socket
import subprocess
import sys
import time


def matrices(folder):
    lst = os.path.join(folder, "datasets.txt")
    if os.path.exists(lst):
        names = [l.strip() for l in open(lst) if l.strip()]
        return [(n, os.path.join(folder, n, n + ".mtx")) for n in names]
    found = sorted(glob.glob(os.path.join(folder, "**", "*.mtx"), recursive=True))
    return [(os.path.splitext(os.path.basename(p))[0], p) for p in found]


def experiment_id():
    try:
        h = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        h = ""
    return f"{h or 'nogit'}-{time.strftime('%Y-%m-%dT%H-%M-%S%z')}"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("datasets"), ap.add_argument("app"), ap.add_argument("kernelfolder")
    ap.add_argument("runfile"), ap.add_argument("device", type=int)
    ap.add_argument("results_root", nargs="?", default="results")
    ap.add_argument("--trials", type=int, default=5)
    ap.add_argument("--timeout-ms", type=int, default=20)
    ap.add_argument("--per-run-limit", type=float, default=3600.0, help="seconds before a run is killed")
    ap.add_argument("--experiment", default=None, help="experiment id (default: git hash + timestamp)")
    a = ap.parse_args(argv)

    ex = a.experiment or experiment_id()
    host = socket.gethostname()
    mats = matrices(a.datasets)
    kernels = sorted(glob.glob(os.path.join(a.kernelfolder, "*.json")))
    if not mats or not kernels:
        sys.exit(f"nothing to do: {len(mats)} matrices, {len(kernels)} kernel configs")
    root = os.path.join(a.results_root, f"results-{ex}")
    os.makedirs(root, exist_ok=True)
    status = open(os.path.join(root, "runstatus.txt"), "w")

    def say(msg):
        print(msg, flush=True)
        status.write(msg + "\n")
        status.flush()

    say(f"Dataset folder: {a.datasets}\nexecutable: {a.app}\nKernelFolder: {a.kernelfolder}\nrunfile: {a.runfile}\n"
        f"Device: {a.device}\nexperiment: {ex}\ntaskcount: {len(mats) * len(kernels)}")
    start, done, failed = time.time(), 0, 0
    total = len(mats) * len(kernels)
    for name, path in mats:
        rdir = os.path.join(root, name)
        os.makedirs(rdir, exist_ok=True)
        for k in kernels:
            kname = os.path.splitext(os.path.basename(k))[0]
            say(f"Processing matrix: {name} - {done}/{total}\nUsing kernel: {kname}")
            cmd = [a.app, "-p", "0", "-d", str(a.device), "-i", str(a.trials), "-m", path, "-f", name, "-k", k,
                   "-r", a.runfile, "-n", host, "-t", str(a.timeout_ms), "-e", ex, *extra]
            t0 = time.time()
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=a.per_run_limit)
                out, rc = r.stdout, r.returncode
            except subprocess.TimeoutExpired as e:
                out, rc = (e.stdout or b"") + b"\nrun_all: killed at the per-run limit\n", -9
            with gzip.open(os.path.join(rdir, f"result_{kname}.txt.gz"), "wb") as f:
                f.write(out)
            if rc != 0:
                failed += 1
                say(f"run failed! (exit code {rc})")
            done += 1
            spent = time.time() - start
            say(f"Run took {time.time() - t0:.1f} seconds, total time of {spent:.1f} seconds; "
                f"estimated total {spent / done * total / 60:.1f} min ({100.0 * done / total:.1f}% done)")
    say(f"finished experiments: {done} runs, {failed} failed, results in {root}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
