#!/usr/bin/env python3
"""tools/fused_probe.py [rows nnz] -- time the tiled plan on the power-law matrix under the engine env knobs
given as NAME=VAL,NAME=VAL groups in SH_PROBE (semicolon-separated); with an SH_STATS build (SH_LIB) and
SH_STATS_DUMP=1 the engine prints the per-role timeline of every launch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.engine import PLUS_TIMES_F32, Engine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nnz = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
rp, ci, va = H.powerlaw(rows, nnz)
x = (1 + np.arange(rows) % 7).astype(np.float32)
eng = Engine(0)
xv, out = eng.vector(x), eng.alloc(rows)
ref = None
for group in os.environ.get("SH_PROBE", "").split(";"):
    env = dict(kv.split("=") for kv in group.split(",") if kv)
    os.environ.update(env)
    A = eng.upload_csr(rows, rows, rp, ci, va)
    dump = os.environ.pop("SH_STATS_DUMP", None)
    dbg = os.environ.pop("SH_DBG", None)   # role ablations (stats builds): the first launches run complete, so that P is valid
    for _ in range(3):
        eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
    eng.synchronize()
    if dbg:
        os.environ["SH_DBG"] = dbg
    t = []
    for _ in range(10):
        t.append(eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out, timed=True))
    got = out.download()
    if ref is None:
        ref = got
    print(f"{group:50s} median {np.median(t) / 1e3:8.1f} us  min {min(t) / 1e3:8.1f} us  same_bits {np.array_equal(got.view(np.uint32), ref.view(np.uint32))}  {A.describe()}", flush=True)
    if dump:
        os.environ["SH_STATS_DUMP"] = dump
        eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
        eng.synchronize()
        os.environ.pop("SH_STATS_DUMP")
    os.environ.pop("SH_DBG", None)
    A.free()
    for k in env:
        os.environ.pop(k, None)
