#!/usr/bin/env python3
"""tools/iter_small.py -- wall time per iteration of the device iteration loop (sh_iterate) on the reference's
own example matrices, where an iteration is launch-bound (a ~10 us kernel): SSSP and BFS to convergence."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402  (checker: iteration counts)
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.engine import Engine  # noqa: E402

eng = Engine(0)
for name in ("matrix", "matrix2"):
    rows, cols, _, rp, ci, va = H.mm_load(os.path.join(ROOT, "tests", "golden", name + ".mtx"))
    for sr, a, b, tag in ((O.MIN_PLUS_F32, 0.0, 0.0, "sssp"), (O.OR_AND_I32, 1, 0, "bfs")):
        dt = O.elem_dtype(sr)
        vals = va.astype(dt)
        x0 = O.initial_vector(sr, rows)
        A = eng.upload_csr(rows, cols, rp, ci, vals)
        xv, yv, sc = eng.vector(x0), eng.vector(x0), eng.alloc(rows)
        walls, its, dev = [], 0, 0
        for rep in range(30):
            xv.upload(x0); yv.upload(x0)
            eng.synchronize()
            t = time.perf_counter()
            its, conv, per, total = eng.iterate(sr, A, xv, yv, sc, a, b, 1e-4, 2000)
            walls.append(time.perf_counter() - t)
            dev = total
        w = sorted(walls)[len(walls) // 2]
        print(f"{name:8s} {tag:5s} rows={rows:6d} nnz={len(ci):7d} launches={its:3d}  wall {w * 1e6 / its:7.2f} us/iteration "
              f"(loop {w * 1e6:8.1f} us), kernels alone {dev / 1e3 / its:6.2f} us/iteration  [{A.describe().split()[0]}]", flush=True)
        for v in (xv, yv, sc):
            v.free()
        A.free()
