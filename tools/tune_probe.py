import numpy as np, sys
sys.path.insert(0, "/root/repo")
from sparseharness_amd.engine import Engine
rng = np.random.default_rng(3)
n, per_row = 3_000_000, 8
rp = (np.arange(n + 1, dtype=np.int64) * per_row).astype(np.int32)
va = rng.integers(1, 17, n * per_row).astype(np.float32)
band = (np.repeat(np.arange(n, dtype=np.int64), per_row) + rng.integers(-300, 301, n * per_row)).clip(0, n - 1).astype(np.int32)
rand = rng.integers(0, n, n * per_row).astype(np.int32)
with Engine(0) as eng:
    for name, ci in (("banded", band), ("random", rand)):
        A = eng.upload_csr(n, n, rp, ci, va); print(name, A.describe()); A.free()
