#!/usr/bin/env python3
"""tools/bits_probe.py [scale=23] -- the (or,and) semiring on R-MAT: the x-tiled plan against the bit-blocked layout
(sh_plan_options::or_and_bits) at several frontier densities: device time of one launch (median of 9) and a
bit-for-bit comparison of the two results.  Development tool."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.engine import OR_AND_I32, Engine  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
rp, ci, va = H.rmat(scale)
n = 1 << scale
vals = va.astype(np.int32)
rng = np.random.default_rng(1)
out = {"workload": f"rmat-{scale}", "nnz": int(rp[-1])}
with Engine(0) as eng:
    mats = {"tiled": eng.upload_csr(n, n, rp, ci, vals), "bits": eng.upload_csr(n, n, rp, ci, vals, or_and_bits=2)}
    out["layouts"] = {k: A.describe() for k, A in mats.items()}
    for density in (0.0, 0.001, 0.01, 0.1, 0.5, 1.0):
        x = (rng.random(n) < density).astype(np.int32)
        xv, o = eng.vector(x), eng.alloc(n)
        res, t = {}, {}
        for k, A in mats.items():
            for _ in range(2):
                eng.spmv(OR_AND_I32, A, xv, None, 1, 0, o)
            ts = sorted(eng.spmv(OR_AND_I32, A, xv, None, 1, 0, o, timed=True) for _ in range(9))
            t[k] = round(ts[4] / 1e3, 1)
            res[k] = o.download(np.int32)
        out[f"density_{density}"] = {"us": t, "same_bits": bool(np.array_equal(res["tiled"], res["bits"])), "rows_set": int(res["bits"].sum())}
print(json.dumps(out))
