#!/usr/bin/env python3
"""What the epilogue of an iteration step costs on R-MAT-23: one (min,+) launch without y, with y, and as a step with the
fused convergence test (few rows change / every row changes), timed with events over 20 launches each."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.engine import MIN_PLUS_F32, PLUS_TIMES_F32, Engine  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << scale
rp, ci, va = H.rmat(scale)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream()
eng = Engine(0, stream=stream.cuda_stream)
A = eng.upload_csr(n, n, rp, ci, va)
x_t = torch.rand(n, device="cuda") * 100
out_t = torch.zeros(n, device="cuda")
flag_t = torch.zeros(16, dtype=torch.int32, device="cuda")
x, out = eng.wrap(x_t.data_ptr(), n), eng.wrap(out_t.data_ptr(), n)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


res = {"layout": A.describe()}
res["plus_times_no_y"] = timed(lambda: eng.spmv(PLUS_TIMES_F32, A, x, None, 1.0, 0.0, out))
res["min_plus_no_y_alpha0"] = timed(lambda: eng.spmv(PLUS_TIMES_F32, A, x, None, 1.0, 0.0, out))
res["min_plus_with_y"] = timed(lambda: eng.spmv(MIN_PLUS_F32, A, x, x, 0.0, 0.0, out))
# step with the convergence test: out differs from x in (nearly) every row -> every row raises the flag
res["min_plus_step_all_rows_change"] = timed(lambda: eng.step(MIN_PLUS_F32, A, x, x, 0.0, 0.0, out, 0, 1e-4, flag_t.data_ptr()))
# a fixed point: iterate until nothing changes, then time the step (no row raises the flag)
x0 = np.full(n, np.finfo(np.float32).max, np.float32)
x0[0] = 0
xv, yv, sc = eng.vector(x0), eng.vector(x0), eng.alloc(n)
eng.iterate(MIN_PLUS_F32, A, xv, yv, sc, 0.0, 0.0, 1e-4, 100)
res["min_plus_step_no_row_changes"] = timed(lambda: eng.step(MIN_PLUS_F32, A, xv, xv, 0.0, 0.0, sc, 0, 1e-4, flag_t.data_ptr()))
res["min_plus_with_y_fixed_point"] = timed(lambda: eng.spmv(MIN_PLUS_F32, A, xv, xv, 0.0, 0.0, sc))
print(json.dumps(res))
