"""The C-ABI library loads and exports every symbol include/pbhip.h declares (no compute calls)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pbhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pbh_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for must in ("pbh_plan_create", "pbh_plan_destroy", "pbh_chirp_generate", "pbh_chirp_upload",
                 "pbh_dedisperse", "pbh_dedisperse_detect", "pbh_fft_c2c", "pbh_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from pulsarbat_amd import _build, _hip
    assert os.path.exists(_build.LIB), "libpbhip.so missing: run __graft_entry__.build()"
    lib = _hip.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in pbhip.h but not exported"
        assert name in _hip.SIGNATURES, f"{name} has no ctypes signature"
    assert b"pbhip" in lib.pbh_version()


def test_argument_validation_without_gpu():
    """Error codes and messages for bad arguments (these paths return before touching HIP)."""
    from pulsarbat_amd import _hip
    lib = _hip.lib()
    h = ctypes.c_void_p()
    assert lib.pbh_plan_create(None, 0, 1024, 1, 1, 0, 0, 1024) == -1
    assert lib.pbh_plan_create(ctypes.byref(h), 0, 1, 1, 1, 0, 0, 1) == -2
    assert b"nsample" in lib.pbh_last_error()
    assert lib.pbh_plan_create(ctypes.byref(h), 0, (1 << 28) + 2, 1, 1, 0, 0, 4) == -2
    assert lib.pbh_plan_create(ctypes.byref(h), 0, 1024, 0, 1, 0, 0, 1024) == -1
    assert lib.pbh_plan_create(ctypes.byref(h), 0, 1024, 1, 1, 7, 0, 1024) == -2
    assert lib.pbh_plan_destroy(None) == 0
    assert lib.pbh_dedisperse(None, None, None, 0, 0) == -1
    assert lib.pbh_fft_c2c(0, None, 0, None, None, 1024, 1, 0, 0, 0) == -1
    assert lib.pbh_fft_c2c(0, None, 5, None, None, 1024, 1, 0, 0, 0) == -2
    assert lib.pbh_plan_create(ctypes.byref(h), 0, 1024, 1, 1, 1, 0, 1024) in (0, -1)  # complex128 plans exist
    if h.value:
        lib.pbh_plan_destroy(h)
    if lib.pbh_device_count() == 0:
        assert lib.pbh_plan_create(ctypes.byref(h), 0, 1024, 1, 1, 0, 0, 1024) == -1
        assert b"not present" in lib.pbh_last_error()


def test_header_is_self_contained_c_and_cpp():
    """include/pbhip.h compiles on its own as C99 and as C++ (what a cgo / JNI / ctypes-generator consumer needs)."""
    import shutil, subprocess
    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pbhip.h")
    for cc, args in (("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        r = subprocess.run([cc, "-Wall", "-Werror", "-fsyntax-only"] + args + [header], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
