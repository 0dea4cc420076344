"""Randomised parity sweep: shapes, dtypes, variants and layouts against the oracle, one process.
usage: python tools/fuzz_parity.py [seconds] [seed]"""
import sys, time, math
import numpy as np
sys.path.insert(0, ".")
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n_ok = n_bad = 0
worst = {"c64": 0.0, "c128": 0.0}
import os
from pulsarbat_amd.transforms.dedispersion import clear_plan_cache
while time.time() - t0 < budget:
    # sometimes force the split column transform (long-block code path) at these small sizes
    q = rng.choice([0, 0, 32, 64])
    if q:
        os.environ["PBH_QMAX"] = str(int(q))
    else:
        os.environ.pop("PBH_QMAX", None)
    clear_plan_cache()
    kind = rng.random()
    if kind < 0.55:
        n = 1 << int(rng.integers(12, 23))
    elif kind < 0.8:
        n = int(rng.integers(2000, 400000))
    else:   # 7-smooth: the mixed-radix column pass (one level by default; both levels when run with PBH_MIXED=2)
        from pulsarbat_amd.utils import _smooth_7
        cand = [v for v in _smooth_7(1 << 21) if v >= 2000 and v & (v - 1)]   # (odd ones too: the plans with mixed-radix rows)
        n = int(rng.choice(cand))
    nchan = int(rng.integers(1, 10))
    npol = int(rng.choice([1, 2]))
    dtype = np.complex64 if rng.random() < 0.7 else np.complex128
    sr = float(rng.choice([1e6, 4e6, 25e6]))
    fc = float(rng.choice([4e8, 1e9, 1.4e9]))
    dm = float(rng.choice([0.0, 1.0, 10.0, 50.0])) * (1e6 / sr) ** 0  # crop scales with sr; keep moderate
    shape = (n, nchan, npol) if npol > 1 else (n, nchan)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    if npol > 1:
        z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
    else:
        z = pb.BasebandSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz)
    try:
        yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    except Exception:
        continue
    if stop - start < 16:
        continue
    mode = rng.choice(["host", "device", "series"])
    variant = str(rng.choice(["auto", "planar5", "direct3"]))
    zz = z if mode == "host" else z.to_device()
    if mode == "series":
        zz = type(z).like(z, zz.data.to_series_major())
    try:
        y = np.asarray(pb.coherent_dedispersion(zz, pb.DM(dm), variant=variant).data)
    except NotImplementedError:
        continue
    # now and then also the detect tail (fused when nscrunch % 64 == 0 on multi-pass planar plans)
    if npol == 2 and rng.random() < 0.3 and stop - start > 4096:
        k = int(rng.choice([1, 48, 64, 256]))
        dmode = str(rng.choice(["I", "linear", "intensity"]))
        got, s0 = pb.dedisperse_detect(zz, pb.DM(dm), mode=dmode, nscrunch=k)
        ref = orc.to_intensity(yr) if dmode == "intensity" else orc.to_stokes(yr, "linear")
        if dmode == "I":
            ref = ref[:, :, 0]
        ref = orc.scrunch(ref, k)
        derr = np.abs(np.asarray(got) - ref).max() / max(np.abs(ref).max(), 1e-30)
        if s0 != start or np.asarray(got).shape != ref.shape or not derr < (1e-4 if dtype == np.complex64 else 1e-7):
            n_bad += 1
            print(f"BAD detect n={n} nchan={nchan} {dmode} k={k} {mode} err={derr:.3e}", flush=True)
    err = np.linalg.norm(y - yr) / max(np.linalg.norm(yr), 1e-30)
    # complex128: the bound is set by last-bit flips of the complex64-rounded chirp (tests/test_gpu_parity.py), which
    # grow with the size of the float64 phase (high DM x wide channels): a few 1e-9
    tol = 1e-5 if dtype == np.complex64 else 1e-8
    key = "c64" if dtype == np.complex64 else "c128"
    worst[key] = max(worst[key], err)
    if y.shape != yr.shape or not err < tol:
        n_bad += 1
        print(f"BAD n={n} nchan={nchan} npol={npol} {key} sr={sr} fc={fc} dm={dm} {mode} {variant} qmax={q} err={err:.3e}", flush=True)
    else:
        n_ok += 1
print(f"cases ok {n_ok}, bad {n_bad}, worst rel err c64 {worst['c64']:.2e}, c128 {worst['c128']:.2e}, {time.time() - t0:.0f} s")
sys.exit(1 if n_bad else 0)
