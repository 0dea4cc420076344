"""Random-shape parity for contrib.stft / istft and time_shift against the numpy oracle.  usage: fuzz_stft.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
t_end = time.time() + budget
n_ok = bad = 0
while time.time() < t_end:
    dtype = np.complex64 if rng.random() < 0.7 else np.complex128
    kind = rng.integers(0, 4)
    if kind == 0:
        n = 1 << int(rng.integers(1, 15))
    elif kind == 1:
        n = 1 << int(rng.integers(14, 19))          # around and beyond one tile
    elif kind == 2:
        n = int(rng.integers(2, 5000))
    else:
        n = int(rng.choice([3, 5, 7])) << 19
    nchan = int(rng.integers(1, 6))
    tail = (nchan,) + ((int(rng.integers(1, 5)),) if rng.random() < 0.7 else ())
    per = int(np.prod(tail))
    nseg = int(rng.integers(1, 6))
    while n * nseg * per > (1 << 23) and nseg > 1:
        nseg -= 1
    if n * nseg * per > (1 << 24):
        continue
    extra = int(rng.integers(0, 5))
    shape = (n * nseg + extra,) + tail
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)
    # the device input is a view into a buffer of NaNs (any read outside the array poisons the result)
    g0 = int(rng.integers(0, 4))
    buf = np.full((32 + g0 + shape[0] + 32,) + tail, np.nan + 1j * np.nan, dtype=dtype)
    buf[32 + g0:32 + g0 + shape[0]] = x
    xdev = pb.DeviceArray.from_host(buf)[32 + g0:32 + g0 + shape[0]]
    z = pb.BasebandSignal(xdev, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
    tol = 4e-6 if dtype == np.complex64 else 1e-11
    s = pb.contrib.stft(z, nperseg=n)
    want = orc.stft(x, n)
    e1 = np.linalg.norm(np.asarray(s) - want) / np.linalg.norm(want)
    f = np.asarray(pb.fft.fft(xdev, axis=0))
    fr = np.fft.fft(x.astype(np.complex128), axis=0)
    e1 = max(e1, np.linalg.norm(f - fr) / np.linalg.norm(fr)) if np.all(np.isfinite(f)) else np.inf
    y = pb.contrib.istft(s, nperseg=n)
    e2 = np.linalg.norm(np.asarray(y) - x[: len(y)]) / np.linalg.norm(x[: len(y)])
    # time shift of the same block (real-valued shift per series or one scalar)
    sh = float(rng.uniform(-30, 30)) if rng.random() < 0.5 else rng.uniform(-30, 30, tail[:1] + (1,) * (len(tail) - 1))
    sig = pb.Signal(xdev, sample_rate=1 * u.kHz)
    e3 = 0.0
    if shape[0] > 64:
        got = np.asarray(pb.time_shift(sig, sh))
        ref, _, _ = orc.time_shift(x, sh)
        e3 = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    # (the complex64-rounded phase ramp of the reference: last-bit differences of a few ramp samples, as for the chirp)
    tol3 = tol if dtype == np.complex64 else 1e-9
    if not (e1 < tol and e2 < tol and e3 < tol3):
        bad += 1
        print(f"BAD n={n} nseg={nseg} tail={tail} {np.dtype(dtype).name} shift={sh}: stft {e1:.2e} istft {e2:.2e} shift {e3:.2e}", flush=True)
    else:
        n_ok += 1
print(f"cases ok {n_ok}, bad {bad}", flush=True)
sys.exit(1 if bad else 0)
