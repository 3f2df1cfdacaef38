#!/usr/bin/env python3
"""Device memory after repeated uploads (device builder + placement trials, bit-blocked layout, host builder) and frees:
the free-memory reading must not drift."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from sparseharness_amd import hostlib as H
from sparseharness_amd.engine import Engine, PLUS_TIMES_F32
n=4_000_000
rp,ci,va=H.powerlaw(n, 80_000_000)
vi=va.astype(np.int32)
with Engine(0) as eng:
    free=[]
    for i in range(8):
        A=eng.upload_csr(n,n,rp,ci,va)
        B=eng.upload_csr(n,n,rp,ci,vi,or_and_bits=2)
        C_=eng.upload_csr(n,n,rp,ci,va,build=1)
        A.free(); B.free(); C_.free()
        eng.synchronize()
        free.append(eng.max_alloc()>>20)
    print("free MiB after each round:", free)
    assert max(free[1:]) - min(free[1:]) < 64, free
