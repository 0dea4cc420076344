"""Randomised check of the detecting layout pass (dedisperse_detect with nscrunch = 1; test infrastructure): random lengths
(2^k, m 2^k, 7-smooth, one-tile with many series), even and odd series counts from 1 to ~600, every detect mode, host and
device input, against detection of the voltages the ordinary call returns (float64 on the host).
usage: python tests/tools/fuzz_detect_fullres.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    kind = int(rng.integers(0, 4))
    if kind == 0:
        n = 1 << int(rng.integers(12, 19))
    elif kind == 1:
        n = int(rng.choice([3, 5, 7])) << int(rng.integers(12, 16))
    elif kind == 2:
        n = int(rng.choice([20000, 64800, 107520, 400000, 52488, 65625, 294912]))
    else:
        n = 1 << int(rng.integers(10, 15))
    npol = int(rng.choice([1, 2]))
    nchan = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 25, 32, 64, 100, 128, 256, 300]))
    if n * nchan * npol > (1 << 24):
        continue
    mode = str(rng.choice(["intensity", "I", "linear", "circular"])) if npol == 2 else "intensity"
    sr, fc = 1e6, 1.5e9
    dm = float(rng.uniform(0.05, 3.0)) * min(1.0, 8.0 / nchan)
    shape = (n, nchan) + ((2,) if npol == 2 else ())
    x = (rng.standard_normal(shape, dtype=np.float32) + 1j * rng.standard_normal(shape, dtype=np.float32)).astype(np.complex64)
    kw = dict(sample_rate=sr * u.Hz, center_freq=fc * u.Hz)
    z = pb.DualPolarizationSignal(x, pol_type="linear", **kw) if npol == 2 else pb.BasebandSignal(x, **kw)
    dev = bool(rng.integers(0, 2))
    zz = z.to_device() if dev else z
    try:
        y = pb.coherent_dedispersion(zz, pb.DM(dm))
    except ValueError:
        continue
    if len(y) < 1:
        continue
    v = np.asarray(y.data if dev else y.data).reshape(len(y), nchan, npol).astype(np.complex128)
    pw = v.real ** 2 + v.imag ** 2
    if mode == "intensity":
        want, scale = pw, pw.max()
    else:
        ab = np.conj(v[..., 0]) * v[..., 1]
        d = pw[..., 0] - pw[..., 1]
        want = {"I": pw.sum(-1), "linear": np.stack([pw.sum(-1), d, 2 * ab.real, 2 * ab.imag], -1),
                "circular": np.stack([pw.sum(-1), 2 * ab.real, 2 * ab.imag, d], -1)}[mode]
        scale = pw.sum(-1).max()
    got, start = pb.dedisperse_detect(zz, pb.DM(dm), mode=mode, nscrunch=1)
    got = np.asarray(got)
    err = float(np.abs(got.reshape(want.shape) - want).max() / scale)
    cases += 1
    if not err < 2e-5:
        bad += 1
        print(f"FAIL seed {seed} case {cases}: n {n} nchan {nchan} npol {npol} mode {mode} dm {dm:.4g} device {dev} start {start}: {err:.2e}", flush=True)
print(f"fuzz_detect_fullres seed {seed}: {cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
