#!/usr/bin/env python3
"""BASELINE.json config 4 on one GPU: (min,+) SSSP and (or,and) BFS to convergence on R-MAT
(default scale 23), on-device loop (sh_iterate), checked bit-for-bit against the CPU oracle."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402  (checker only)
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.engine import Engine  # noqa: E402

scale = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 23
rp, ci, va = H.rmat(scale)
n = 1 << scale
out = {"workload": f"rmat-{scale}", "rows": n, "nnz": int(rp[-1])}
with Engine(0) as eng:
    for name, sr, a, b in [("sssp", O.MIN_PLUS_F32, 0.0, 0.0), ("bfs", O.OR_AND_I32, 1, 0)]:
        dt = O.elem_dtype(sr)
        vals = va.astype(dt)
        # (the BFS harness uploads a large (or,and) matrix in the bit-blocked layout only: host/inc/harness.h)
        A = eng.upload_csr(n, n, rp, ci, vals, **({"or_and_bits": 2} if sr == O.OR_AND_I32 and "--no-bits" not in sys.argv else {}))
        x0 = O.initial_vector(sr, n)
        x, y, sc = eng.vector(x0), eng.vector(x0), eng.alloc(n)
        eng.iterate(sr, A, x, y, sc, a, b, 1e-4, 1)          # warm-up trial (first launch loads the code object),
        x.upload(x0)                                         # then reset the inputs as the apps do between trials
        t = time.perf_counter()
        iters, conv, per, total = eng.iterate(sr, A, x, y, sc, a, b, 1e-4, 200)
        wall = time.perf_counter() - t
        got = x.download(dt)
        t = time.perf_counter()
        want, w_it, w_conv = O.iterate(sr, rp, ci, vals, x0, x0, a, b, 1e-4, 200)
        cpu = time.perf_counter() - t
        ok = (iters, conv) == (w_it, w_conv) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        bytes_it = A.algorithmic_bytes(reads_y=(sr == O.MIN_PLUS_F32))
        out[name] = {"plan": A.plan()[0], "layout": A.describe(), "iterations": iters, "converged": conv, "bit_exact_vs_oracle": bool(ok),
                     "device_ms_total": round(total / 1e6, 3), "device_ms_per_iteration": round(total / 1e6 / iters, 4),
                     "device_us_each_iteration": [round(p / 1e3, 1) for p in per],
                     "wall_ms_total_incl_flag_readback": round(wall * 1e3, 3),
                     "algorithmic_GBps": round(bytes_it / (total / iters), 1), "cpu_oracle_seconds_1_thread": round(cpu, 2),
                     "reached": int((got != x0[1]).sum()) if name == "sssp" else int((got != 0).sum())}
        for v in (x, y, sc):
            v.free()
        A.free()
print(json.dumps(out))
