"""The C-ABI library loads and exports exactly what include/sparseharness_hip.h declares.
No compute is called here (no GPU in the authoring container)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from sparseharness_amd import abi

HEADER = os.path.join(ROOT, "include", "sparseharness_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sh_[a-z_0-9]+)\s*\(", text)))


def test_header_cites_reference_interfaces():
    text = open(HEADER).read()
    for cite in ["inc/harness.h:149-195", "inc/harness.h:13-82", "src/sparse_matrix.cpp:122-399",
                 "app/sssp.cpp:97-176", "inc/run.h:9-32"]:
        assert cite in text


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 24
    lib = abi.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(abi.SIGNATURES) == names, "abi.SIGNATURES and the header disagree"
    assert lib.sh_abi_version() == 3


def test_no_cpu_fallback_without_device():
    lib = abi.load()
    if lib.sh_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    rc = lib.sh_engine_create(0, C.byref(h))
    assert rc == abi.SH_ENODEVICE and not h.value
    assert b"no CPU fallback" in lib.sh_last_error(None)
    from sparseharness_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sparseharness_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libsh_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_plan_options_defaults_and_environment(monkeypatch):
    """sh_plan_options: the defaults, and the SH_* environment read by sh_plan_options_from_env (no GPU needed)."""
    lib = abi.load()
    o = abi.sh_plan_options()
    lib.sh_plan_options_default(C.byref(o))
    assert (o.plan, o.autotune, o.value_coding, o.heavy_per_tile, o.xcd_order, o.fold) == (0, 1, 0, 8, 1, 1)
    assert o.chunk == 0
    for k, v in {"SH_PLAN": "tiled", "SH_VALCODE": "off", "SH_AUTOTUNE": "0", "SH_FOLD": "0",
                 "SH_HEAVY_PER_TILE": "4", "SH_BUILD_THREADS": "2", "SH_CHUNK": "4096"}.items():
        monkeypatch.setenv(k, v)
    lib.sh_plan_options_from_env(C.byref(o))
    assert (o.plan, o.value_coding, o.autotune, o.fold, o.heavy_per_tile, o.build_threads, o.chunk) == (2, -1, 0, 0, 4, 2, 4096)
    monkeypatch.setenv("SH_VALCODE", "8")
    monkeypatch.setenv("SH_PLAN", "stream")
    lib.sh_plan_options_from_env(C.byref(o))
    assert (o.plan, o.value_coding) == (1, 8)
    assert (o.build, o.or_and_bits) == (0, 0)            # where the layout is built: the engine decides by size
    for text, want in (("host", 1), ("device", 2), ("gpu", 2), ("auto", 0)):
        monkeypatch.setenv("SH_BUILD", text)
        lib.sh_plan_options_from_env(C.byref(o))
        assert o.build == want, text
    monkeypatch.setenv("SH_OR_AND_BITS", "2")
    lib.sh_plan_options_from_env(C.byref(o))
    assert o.or_and_bits == 2


def test_tiled_kernels_use_no_scratch_and_spill_nothing():
    """The hand-scheduled loaders keep asm-issued loads in flight across compiler-visible code: only sound while
    hipcc neither spills nor uses scratch in those kernels (make asm-check, sparseharness_amd/csrc)."""
    import subprocess
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "sparseharness_amd", "csrc"), "asm-check"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
    assert "0 offenders" in r.stdout and "0 reads of registers with a load in flight" in r.stdout
    # and the prefetch check does bite: a copy of a prefetch register slipped in right behind its load is reported
    import re
    import sys
    lines = open("/tmp/sh_engine_check.s").read().splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith("_ZN2sh18spmv_tiled_phase2s"))
    j = next(i for i in range(k, len(lines)) if lines[i].strip().startswith("global_load_dwordx4") and "#ASMSTART" in lines[i - 1])
    dst = re.search(r"v\[(\d+):", lines[j]).group(1)
    end = next(i for i in range(j, len(lines)) if "#ASMEND" in lines[i])
    doctored = lines[:end + 1] + [f"\tv_mov_b32_e32 v1, v{dst}"] + lines[end + 1:]
    open("/tmp/sh_engine_doctored.s", "w").write("\n".join(doctored))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sparseharness_amd", "csrc", "check_prefetch.py"), "/tmp/sh_engine_doctored.s"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "touches in-flight" in r.stdout, r.stdout[-400:]
