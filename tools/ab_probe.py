#!/usr/bin/env python3
"""tools/ab_probe.py -- same-box A/B of engine builds and plan knobs on one workload.

  python tools/ab_probe.py [--workload powerlaw|rmat] [--rows R --nnz Z] [--reps 30] [--rounds 2] \
      --arm name=LIB[,ENV=VAL,...] --arm ...

Every arm is one engine build (LIB: a path under sparseharness_amd/, or "head" for the in-tree library) plus
SH_* environment knobs.  Each (arm, round) runs in a child process of its own (a process loads one engine
build), arms interleaved round by round so that box drift hits all of them alike.  A child times `reps`
(+,x) SpMVs with events through sh_spmv(timed), and compares the result with the first arm's: rows that differ in
bits (integer-valued data: only rows whose float sum passes 2^24 may) and the largest relative difference.  Only the entry points that exist since
round 1 are used, so old builds can be compared.  Prints one line per arm: median of the per-round medians.
Development tool; not the benchmark."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    import numpy as np
    sys.path.insert(0, ROOT)
    os.environ["SH_LIB_PARTIAL"] = "1"   # abi.load(): tolerate entry points an older build lacks
    from sparseharness_amd import hostlib as H
    from sparseharness_amd.engine import PLUS_TIMES_F32, Engine
    if args.workload == "powerlaw":
        rp, ci, va = H.powerlaw(args.rows, args.nnz)
        n = args.rows
    else:
        scale = int(np.log2(args.rows))
        rp, ci, va = H.rmat(scale)
        n = 1 << scale
    if args.real_values:
        va = (va * np.float32(1.0009765625) + np.float32(0.37)).astype(np.float32)
    x = (1 + np.arange(n) % 7).astype(np.float32)
    import time
    with Engine(0) as eng:
        t0 = time.time()
        A = eng.upload_csr(n, n, rp, ci, va)
        up = time.time() - t0
        xv, out = eng.vector(x), eng.alloc(n)
        for _ in range(3):
            eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        ts = sorted(eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out, timed=True) for _ in range(args.reps))
        y = out.download()
        try:
            layout = A.describe()
        except Exception:   # noqa: BLE001 -- builds older than sh_csr_describe
            layout = "?"
        ref_path = os.environ["SH_AB_REF"]
        if os.path.exists(ref_path):
            ref = np.load(ref_path)
            diff = int((ref.view(np.uint32) != y.view(np.uint32)).sum())
            rel = float(np.max(np.abs(ref.astype(np.float64) - y) / np.maximum(1.0, np.abs(ref.astype(np.float64)))))
        else:
            np.save(ref_path, y)
            diff, rel = 0, 0.0
        print(json.dumps({"ms_median": ts[len(ts) // 2] / 1e6, "ms_min": ts[0] / 1e6, "rows_differ": diff, "max_rel": rel,
                          "upload_s": round(up, 3), "layout": layout}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="powerlaw", choices=["powerlaw", "rmat"])
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--nnz", type=int, default=200_000_000)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--real-values", action="store_true")
    ap.add_argument("--arm", action="append", default=[])
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    arms = []
    for spec in args.arm:
        name, rest = spec.split("=", 1)
        parts = rest.split(",")
        lib = parts[0]
        env = dict(kv.split("=", 1) for kv in parts[1:] if kv)
        lib = os.path.join(ROOT, "sparseharness_amd", "libsparseharness_hip.so" if lib == "head" else lib)
        arms.append((name, lib, env))
    res = {name: [] for name, _, _ in arms}
    ref_path = f"/tmp/ab_probe_ref_{os.getpid()}.npy"
    for rnd in range(args.rounds):
        for name, lib, env in arms:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--workload", args.workload, "--rows", str(args.rows),
                   "--nnz", str(args.nnz), "--reps", str(args.reps)] + (["--real-values"] if args.real_values else [])
            e = dict(os.environ, SH_LIB=lib, SH_AB_REF=ref_path, **env)
            p = subprocess.run(cmd, env=e, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not line:
                print(f"{name:24s} FAILED rc={p.returncode} {p.stderr[-300:]}", flush=True)
                continue
            r = json.loads(line[-1])
            res[name].append(r)
            print(f"round {rnd} {name:24s} median {r['ms_median']:.4f} ms  min {r['ms_min']:.4f}  rows_differ {r['rows_differ']} (max rel {r['max_rel']:.1e})"
                  f"  upload {r['upload_s']} s  {r['layout'][:130]}", flush=True)
    print("--- summary (median over rounds of the per-round medians)")
    for name, _, env in arms:
        v = sorted(r["ms_median"] for r in res[name])
        if v:
            worst = max(r["max_rel"] for r in res[name])
            print(f"{name:24s} {v[len(v) // 2]:.4f} ms  (rounds: {' '.join(f'{t:.4f}' for t in v)})  rows_differ {max(r['rows_differ'] for r in res[name])} max_rel {worst:.1e}  {env}")
    if os.path.exists(ref_path):
        os.remove(ref_path)


if __name__ == "__main__":
    main()
