"""ctypes declarations for the C ABI in include/sparseharness_hip.h.

The shared library is built in-tree by sparseharness_amd/csrc/Makefile
(hipcc --offload-arch=gfx950).  There is no fallback of any kind: if the
library is missing this module raises, and if no HIP device is present
`sh_engine_create` fails with SH_ENODEVICE.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SH_LIB lets kernel-tuning runs point at an alternative in-tree build of the same engine
LIB_PATH = os.environ.get("SH_LIB") or os.path.join(_HERE, "libsparseharness_hip.so")
CSRC = os.path.join(_HERE, "csrc")

SH_OK, SH_EINVAL, SH_ENODEVICE, SH_EHIP, SH_ENOMEM, SH_ESHAPE = 0, -1, -2, -3, -4, -5
PLUS_TIMES_F32, MIN_PLUS_F32, OR_AND_I32, MAX_MIN_I32 = 0, 1, 2, 3


class sh_launch(C.Structure):
    _fields_ = [("global_", C.c_uint64 * 3), ("local", C.c_uint64 * 3)]


class sh_plan_options(C.Structure):
    """Plan options of sh_csr_upload_ex (field comments: include/sparseharness_hip.h)."""
    _fields_ = [("plan", C.c_int32), ("autotune", C.c_int32), ("value_coding", C.c_int32), ("build_threads", C.c_int32),
                ("heavy_per_tile", C.c_int32), ("chunk", C.c_int32), ("xcd_order", C.c_int32), ("fold", C.c_int32), ("or_and_bits", C.c_int32), ("build", C.c_int32), ("placement_tries", C.c_int32)]


class sh_row_pieces(C.Structure):
    """Row pieces of sh_spmv_step_pieces (include/sparseharness_hip.h)."""
    _fields_ = [("n_pieces", C.c_int32), ("piece_rows", C.c_int32), ("element_of_piece", C.c_int64 * 8),
                ("report", C.c_int32), ("reserved", C.c_int32), ("gate", C.c_void_p)]


_vp, _i32, _i64, _u64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_int
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes): one entry per function declared in the header
# (tests/test_abi.py checks this table against include/sparseharness_hip.h).
SIGNATURES = {
    "sh_abi_version": (_int, []),
    "sh_device_count": (_int, []),
    "sh_engine_create": (_int, [_int, _pp]),
    "sh_engine_create_on_stream": (_int, [_int, _vp, _pp]),
    "sh_engine_destroy": (_int, [_vp]),
    "sh_engine_device_name": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "sh_engine_max_alloc": (_int, [_vp, C.POINTER(_u64)]),
    "sh_engine_synchronize": (_int, [_vp]),
    "sh_last_error": (C.c_char_p, [_vp]),
    "sh_csr_upload": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _pp]),
    "sh_plan_options_default": (None, [C.POINTER(sh_plan_options)]),
    "sh_plan_options_from_env": (None, [C.POINTER(sh_plan_options)]),
    "sh_plan_row_work": (_int, [_i64, _i64, _i64, _vp, C.POINTER(sh_plan_options), _vp]),
    "sh_csr_upload_ex": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(sh_plan_options), _pp]),
    "sh_csr_free": (_int, [_vp, _vp]),
    "sh_csr_dims": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "sh_csr_algorithmic_bytes": (_int, [_vp, _int, C.POINTER(_u64)]),
    "sh_csr_plan": (_int, [_vp, C.POINTER(_i32), C.POINTER(_u64)]),
    "sh_csr_describe": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "sh_csr_footprint": (_int, [_vp, C.POINTER(_u64)]),
    "sh_csr_builder": (_int, [_vp, C.POINTER(_i32), C.c_char_p, _i64]),
    "sh_csr_placement": (_int, [_vp, C.POINTER(_i32), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sh_vec_alloc": (_int, [_vp, _i64, _pp]),
    "sh_vec_wrap": (_int, [_vp, _vp, _i64, _pp]),
    "sh_vec_free": (_int, [_vp, _vp]),
    "sh_vec_upload": (_int, [_vp, _vp, _vp, _i64]),
    "sh_vec_download": (_int, [_vp, _vp, _vp, _i64]),
    "sh_vec_fill": (_int, [_vp, _vp, C.c_uint32]),
    "sh_vec_copy": (_int, [_vp, _vp, _vp]),
    "sh_vec_len": (_i64, [_vp]),
    "sh_vec_device_ptr": (_vp, [_vp]),
    "sh_spmv": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(sh_launch), C.POINTER(_u64)]),
    "sh_iterate": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _i32,
                          C.POINTER(sh_launch), C.POINTER(_i32), C.POINTER(_i32),
                          C.POINTER(_u64), C.POINTER(_u64)]),
    "sh_spmv_step": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_double, _vp]),
    "sh_spmv_step_pieces": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(sh_row_pieces), C.c_double, _vp,
                                   C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32))]),
    "sh_csr_piece_state": (_int, [_vp, _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}

_lib = None


def build(force=False):
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", CSRC]
    if force:
        args.insert(1, "-B")
    subprocess.check_call(args)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). "
            "sparseharness_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 with the same SONAME as /opt/rocm's.  Whichever is
    # loaded first serves the whole process, and torch refuses to see any GPU when it ends up on
    # a runtime it was not built with.  So when this binding is used from Python, let torch (if
    # installed) load its runtime first; the engine then shares that one instance, which is also
    # what makes handing torch streams / tensors to the engine valid.  SH_PRELOAD_TORCH=0 skips it.
    if os.environ.get("SH_PRELOAD_TORCH", "1") != "0":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    partial = os.environ.get("SH_LIB") and os.environ.get("SH_LIB_PARTIAL") == "1"   # tools/ab_probe.py: older builds
    for name, (res, args) in SIGNATURES.items():
        if partial and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
