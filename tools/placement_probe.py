#!/usr/bin/env python3
"""Does the SpMV time of ONE layout depend on where its arrays happen to be allocated?
  copies   uploads the same matrix K times (all copies stay alive) and times the copies round-robin: times that are
           stable per copy and differ between copies are a placement effect, not run-to-run noise;
  arrays   (needs SH_LIB = the tools build) moves ONE array of one matrix to a fresh allocation, several times per
           array, and times the SpMV after every move: which array's place matters."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparseharness_amd import abi, hostlib as H  # noqa: E402
from sparseharness_amd.engine import PLUS_TIMES_F32, Engine  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "copies"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 10_000_000
rp, ci, va = H.powerlaw(n, 200_000_000)
x = (1 + np.arange(n) % 7).astype(np.float32)


def median_ms(eng, A, xv, out, runs=7):
    eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out)
    ts = sorted(eng.spmv(PLUS_TIMES_F32, A, xv, None, 1.0, 0.0, out, timed=True) for _ in range(runs))
    return round(ts[runs // 2] / 1e6, 4)


with Engine(0) as eng:
    xv, out = eng.vector(x), eng.alloc(n)
    if mode == "copies":
        mats = [eng.upload_csr(n, n, rp, ci, va, build=2) for _ in range(K)]
        times = [[] for _ in mats]
        for rep in range(5):
            for k, A in enumerate(mats):
                times[k].append(median_ms(eng, A, xv, out))
        print(json.dumps({"builder": mats[0].builder()[0], "per_copy_ms": times}))
    elif mode == "lottery":
        # K candidate places for the product array P (all kept allocated), timed round-robin three times: is a candidate's
        # time its own (then picking the best of K at upload is worth the spread), or does it drift with time?
        lib = abi.load()
        lib.sh_debug_move_array.restype = C.c_int
        lib.sh_debug_move_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        lib.sh_debug_set_P.restype = C.c_int
        lib.sh_debug_set_P.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        A = eng.upload_csr(n, n, rp, ci, va, build=2)
        for _ in range(20):
            median_ms(eng, A, xv, out)   # warm
        cands = []
        for k in range(K):
            addr = C.c_uint64()
            assert lib.sh_debug_move_array(eng.h, A.h, 0, 1, 0, C.byref(addr)) == 0
            cands.append(addr.value)
        times = [[] for _ in cands]
        for rep in range(4):
            for k, a in enumerate(cands):
                assert lib.sh_debug_set_P(eng.h, A.h, a) == 0
                times[k].append(median_ms(eng, A, xv, out, runs=15))
        print(json.dumps({"candidates": [hex(a) for a in cands], "per_candidate_ms": times}))
    else:
        lib = abi.load()
        lib.sh_debug_move_array.restype = C.c_int
        lib.sh_debug_move_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        A = eng.upload_csr(n, n, rp, ci, va, build=2)
        res = {"start_ms": [median_ms(eng, A, xv, out) for _ in range(3)]}
        for rnd in range(2):
            for align in (0, 25, 30):
                for which, name in ((0, "P"), (1, "tcol"), (3, "pslot"), (2, "tcode")):
                    rows = []
                    for k in range(K):
                        addr = C.c_uint64()
                        assert lib.sh_debug_move_array(eng.h, A.h, which, 1, align, C.byref(addr)) == 0
                        rows.append((hex(addr.value), median_ms(eng, A, xv, out)))
                    res[f"round{rnd} align=2^{align or 21} {name}"] = rows
        print(json.dumps(res))
