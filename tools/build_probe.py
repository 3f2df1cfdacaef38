#!/usr/bin/env python3
"""Upload time of a benchmark matrix with the host builder and with the device builder of the tiled layout
(sh_plan_options::build = 1 / 2), and that both matrices give the same SpMV bits.  With SH_LIB pointing at the tools
build (sparseharness_amd/variants/emulate.so) and SH_BUILD_TIMES=1 the phases of both builders are printed too."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.engine import PLUS_TIMES_F32, Engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "powerlaw"
if workload == "powerlaw":
    n = 10_000_000
    rp, ci, va = H.powerlaw(n, 200_000_000)
else:
    n = 1 << 23
    rp, ci, va = H.rmat(23)
x = (1 + np.arange(n) % 7).astype(np.float32)
res = {"workload": workload}
with Engine(0) as eng:
    xv = eng.vector(x)
    outs = {}
    for name, build in (("host", 1), ("device", 2)) * 2:
        t0 = time.perf_counter()
        A = eng.upload_csr(n, n, rp, ci, va, build=build)
        dt = time.perf_counter() - t0
        res.setdefault(name + "_upload_s", []).append(round(dt, 3))
        res[name + "_builder"] = A.builder()
        res[name + "_layout"] = A.describe()
        out = eng.alloc(n)
        eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        ts = sorted(eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out, timed=True) for _ in range(9))
        res.setdefault(name + "_spmv_ms", []).append(round(ts[4] / 1e6, 4))
        outs[name] = out.download()
        A.free()
        out.free()
    res["same_bits"] = bool(np.array_equal(outs["host"].view(np.uint32), outs["device"].view(np.uint32)))
print(json.dumps(res))
