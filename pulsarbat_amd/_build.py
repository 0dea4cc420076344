"""In-tree build of libpbhip.so (hipcc, gfx950 only).

``build()`` is what ``__graft_entry__.build()`` calls; it cross-compiles without a GPU.
The shared object is kept next to its sources (``pulsarbat_amd/csrc/libpbhip.so``) so a
repository snapshot carries it to the GPU box.
"""

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpbhip.so")
SOURCES = ["pbhip.hip"]
HEADERS = ["fft_core.hpp", "kernels.hpp", "aux_kernels.hpp", os.path.join("..", "..", "include", "pbhip.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile the HIP library if missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB
    # -fno-slp-vectorize: v_pk_*_f32 has no rate advantage on gfx950 and its even-aligned register
    # pairs inflate VGPR pressure in the fully unrolled butterflies
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
           "-Rpass-analysis=kernel-resource-usage", "-o", LIB + ".tmp"] + SOURCES
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    with open(os.path.join(CSRC, "build.log"), "w") as fh:
        fh.write(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        print("built", LIB)
    return LIB
