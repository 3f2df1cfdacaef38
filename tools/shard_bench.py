#!/usr/bin/env python3
"""Rehearse the per-GPU work of an N-GPU strong-scaling run on ONE GPU: time shard k of N of the
headline matrix (nnz-balanced row ranges, full x), for k in a few positions.  The slowest shard's
step time bounds the N-GPU step time (no collective in single-shot SpMV), so
speedup_bound = t(N=1) / max_k t(shard k).  Development tool; not the benchmark."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparseharness_amd import hostlib as H, partition  # noqa: E402
from sparseharness_amd.engine import PLUS_TIMES_F32, Engine  # noqa: E402

parts_list = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
rp, ci, va = H.powerlaw(10_000_000, 200_000_000)
n = 10_000_000
x_host = (1 + np.arange(n) % 7).astype(np.float32)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream()
eng = Engine(0, stream=stream.cuda_stream)
x_t = torch.from_numpy(x_host).cuda()
x = eng.wrap(x_t.data_ptr(), n)
res = {}
for parts in parts_list:
    bounds = partition.row_bounds(rp, parts, cols=n)
    times = []
    for k in sorted(set([0, parts // 2, parts - 1])):
        r0, r1 = int(bounds[k]), int(bounds[k + 1])
        s_rp, s_ci, s_va = partition.take_rows(rp, ci, va, r0, r1)
        A = eng.upload_csr(r1 - r0, n, s_rp, s_ci, s_va)
        print(parts, k, A.describe(), file=sys.stderr)
        out_t = torch.zeros(r1 - r0, dtype=torch.float32, device="cuda")
        out = eng.wrap(out_t.data_ptr(), r1 - r0)
        for _ in range(5):
            eng.spmv(PLUS_TIMES_F32, A, x, None, 1.0, 0.0, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(30):
            eng.spmv(PLUS_TIMES_F32, A, x, None, 1.0, 0.0, out)
        e1.record(stream)
        torch.cuda.synchronize()
        times.append(round(e0.elapsed_time(e1) / 30, 4))
        A.free()
    res[parts] = {"shard_ms": times, "max_ms": max(times)}
base = res[parts_list[0]]["max_ms"] if parts_list[0] == 1 else None
for parts in parts_list:
    if base:
        res[parts]["speedup_bound"] = round(base / res[parts]["max_ms"], 2)
print(json.dumps(res))
