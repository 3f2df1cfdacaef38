#!/usr/bin/env python3
"""Row SLABS that share one product array: does P stay in the Infinity Cache when the SpMV is run slab by slab
(phase 1 + phase 2 of slab 0, then of slab 1, ... -- 2 S ordinary launches on one stream), every slab writing and
reading the SAME P buffer, so that dirty P lines are overwritten in the cache instead of travelling to HBM and back?
Needs SH_LIB = a tools build (sh_debug_move_array / sh_debug_set_P).  Usage: slab_probe.py [S ...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SH_PLACEMENT_TRIES", "1")
import torch  # noqa: E402

from sparseharness_amd import abi, hostlib as H, partition  # noqa: E402
from sparseharness_amd.engine import PLUS_TIMES_F32, Engine  # noqa: E402

slabs = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]
n = 10_000_000
rp, ci, va = H.powerlaw(n, 200_000_000)[:3]
x = (1 + np.arange(n) % 7).astype(np.float32)
lib = abi.load()
lib.sh_debug_move_array.restype = C.c_int
lib.sh_debug_move_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
lib.sh_debug_set_P.restype = C.c_int
lib.sh_debug_set_P.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
lib.sh_debug_p_len.restype = C.c_int64
lib.sh_debug_p_len.argtypes = [C.c_void_p]

stream = torch.cuda.current_stream()
with Engine(0, stream=stream.cuda_stream) as eng:
    x_t = torch.from_numpy(x).cuda()
    out_t = torch.zeros(n, dtype=torch.float32, device="cuda")
    xv = eng.wrap(x_t.data_ptr(), n)
    ref = None
    for S in slabs:
        bounds = partition.row_bounds(rp, S, cols=n)
        mats, outs = [], []
        for s in range(S):
            r0, r1 = int(bounds[s]), int(bounds[s + 1])
            s_rp, s_ci, s_va = partition.take_rows(rp, ci, va, r0, r1)
            mats.append(eng.upload_csr(r1 - r0, n, s_rp, s_ci, s_va))
            outs.append(eng.wrap(out_t.data_ptr() + 4 * r0, r1 - r0))

        def step():
            for A, o in zip(mats, outs):
                eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, o)

        def timed(steps=20, warm=3):
            for _ in range(warm):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                step()
            e1.record(stream)
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / steps, 4)

        own = [timed() for _ in range(2)]
        y = out_t.cpu().numpy().copy()
        if ref is None:
            ref = y
        row = {"slabs": S, "layouts": [A.describe().split(" device=")[0].split("tiles=")[1] for A in mats], "own_P_ms": own,
               "same_bits_as_one_matrix": bool(np.array_equal(ref.view(np.uint32), y.view(np.uint32)))}
        if S > 1:
            # the largest slab's product array (a fresh copy of it, the old one stays allocated) serves all
            p_words = [int(lib.sh_debug_p_len(A.h)) for A in mats]
            big = int(np.argmax(p_words))
            assert min(p_words) > 0 and all(w <= p_words[big] for w in p_words)   # (every slab's products fit the shared array)
            addr = C.c_uint64()
            assert lib.sh_debug_move_array(eng.h, mats[big].h, 0, 1, 0, C.byref(addr)) == 0
            for k, A in enumerate(mats):
                if k != big:
                    assert lib.sh_debug_set_P(eng.h, A.h, addr.value) == 0
            out_t.zero_()
            row["shared_P_ms"] = [timed() for _ in range(2)]
            row["shared_same_bits"] = bool(np.array_equal(ref.view(np.uint32), out_t.cpu().numpy().view(np.uint32)))
        print(json.dumps(row), flush=True)
        # (matrices of this S stay allocated: the shared buffer must not be freed under them; 1.35 GB per S)
