#!/usr/bin/env python3
"""tools/ingest_bench.py [entries=20000000] -- MatrixMarket ingest on this host: writes a synthetic
coordinate file (one entry per line, R-MAT columns), then times the product loader (host/src/sparse_matrix.cpp:
in-memory scanner + OpenMP-sliced tokenizer + counting sort) at 1, 4 and all granted threads, and the oracle's
fscanf restatement of the reference loader once.  Host-only: no GPU."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
entries = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
n = 1 << 22
rng = np.random.default_rng(5)
rows = rng.integers(1, n + 1, entries)
cols = rng.integers(1, n + 1, entries)
vals = rng.integers(1, 17, entries)
path = os.path.join(tempfile.gettempdir(), "sh_ingest.mtx")
t = time.time()
with open(path, "w") as f:
    f.write("%%MatrixMarket matrix coordinate real general\n")
    f.write(f"{n} {n} {entries}\n")
    np.savetxt(f, np.stack([rows, cols, vals], 1), fmt="%d %d %d")
print(f"wrote {os.path.getsize(path) / 1e6:.0f} MB in {time.time() - t:.1f} s", flush=True)
code = ("import sys,time; sys.path.insert(0, %r); from sparseharness_amd import hostlib as H; t=time.time(); "
        "r=H.mm_load(%r); print('%%.3f' %% (time.time()-t), r[0], len(r[4]))") % (ROOT, path)
import bench
granted = bench.usable_host_cores()
for th in sorted({1, 4, granted}):
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, OMP_NUM_THREADS=str(th), SH_QUIET_TIMERS="1", SH_LOG_LEVEL="1"))
    print(f"product loader, {th:3d} threads: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
from oracle import oracle as O
t = time.time()
O.mm_load(path)
print(f"oracle fscanf restatement of the reference loader (1 thread): {time.time() - t:.3f} s")
os.remove(path)
