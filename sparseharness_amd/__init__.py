"""sparseharness_amd -- MI355X-native CSR SpMV engine behind sparseharness's
Harness / Run / KernelConfig surface.

  csrc/      hand-written HIP (gfx950) kernels + the C ABI of include/sparseharness_hip.h
  host/      C++ mirror of the reference's host interface (Harness<>, SparseMatrix<>,
             Gold<>, KernelConfig<>, Run, SqlStat, the three apps) on top of the C ABI
  abi.py / engine.py / hostlib.py   ctypes bindings used by tests/ and bench.py
  partition.py / distributed.py     row sharding + per-iteration all-gather driver

There is no CPU compute path in this package.
"""
from . import abi  # noqa: F401
from .abi import MAX_MIN_I32, MIN_PLUS_F32, OR_AND_I32, PLUS_TIMES_F32  # noqa: F401
