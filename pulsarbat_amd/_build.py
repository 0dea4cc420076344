"""In-tree build of libpbhip.so (hipcc, gfx950 only).

``build()`` is what ``__graft_entry__.build()`` calls; it cross-compiles without a GPU.
The shared object is kept next to its sources (``pulsarbat_amd/csrc/libpbhip.so``) so a
repository snapshot carries it to the GPU box.
"""

import hashlib
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpbhip.so")
UNITS = ["pbhip.hip", "pbhip_stream.hip", "pbhip_measure.hip"]   # each compiled twice: float32 and -DPBH_F64 (pbhip_internal.hpp)
SOURCES = UNITS + ["pbhip_api.cpp"]
HEADERS = ["pbh_config.hpp", "pbhip_internal.hpp", "fft_core.hpp", "kernels.hpp", "aux_kernels.hpp", "mixed_kernels.hpp", "fd4_kernels.hpp",
           "bench_kernels.hpp", "host_sched.hpp",
           os.path.join("..", "..", "include", "pbhip.h")]
ARCH = "gfx950"
STAMP = LIB + ".sources"   # SHA-256 of the sources and flags the library was built from (travels with it to the GPU box)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def source_hash():
    """SHA-256 over the sources, headers and build flags: what decides whether the library on disk is the one these
    sources make (mtimes do not survive a repository snapshot; round 3's build() compared mtimes and could leave a box
    running whatever .so the snapshot carried)."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(("flags:" + os.environ.get("PBH_EXTRA_FLAGS", "") + ":" + ARCH).encode())
    return h.hexdigest()


def is_stale():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile the HIP library if missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB
    # -fno-slp-vectorize: v_pk_*_f32 has no rate advantage on gfx950 (tools/micro/pkrate.hip) and its even-aligned register
    # pairs inflate VGPR pressure in the fully unrolled butterflies.
    # the three units are compiled twice each: float32 (pbh32_*) and float64 (-DPBH_F64, pbh64_*); pbhip_api.cpp
    # owns the public pbh_* symbols and dispatches on the plan's dtype.  Seven hipcc processes run side by side.
    common = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]
    common += os.environ.get("PBH_EXTRA_FLAGS", "").split()  # e.g. -DPBH_DIAGNOSTIC for the ablation kernels
    jobs = []
    for unit in UNITS:
        stem = unit[:-4].replace("pbhip", "pbhip%s", 1)   # pbhip32.o, pbhip32_stream.o, ...
        jobs.append((stem % "32" + ".o", common + ["-Rpass-analysis=kernel-resource-usage", "-c", unit]))
        jobs.append((stem % "64" + ".o", common + ["-DPBH_F64", "-Rpass-analysis=kernel-resource-usage", "-c", unit]))
    jobs.append(("pbhip_api.o", common + ["-x", "hip", "-c", "pbhip_api.cpp"]))
    procs = [(obj, subprocess.Popen(cmd + ["-o", obj], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                    text=True)) for obj, cmd in jobs]
    logs = []
    for obj, pr in procs:
        _, err = pr.communicate()
        logs.append(f"==== {obj}\n{err}")
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {obj}:\n" + err[-4000:])
    with open(os.path.join(CSRC, "build.log"), "w") as fh:
        fh.write("\n".join(logs))
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + [o for o, _ in jobs]
    res = subprocess.run(link, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stderr[-4000:])
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    if verbose:
        print("built", LIB)
    return LIB
