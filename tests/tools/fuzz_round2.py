"""Randomised checks of the round-2 entry points against the oracle, one process:
  * pbh_dedisperse_slices: the result written as a channel slice of a wider array split into row-chunks, guard values
    around every part (any write outside the slice shows)
  * contrib.stft_dedisperse (fused and two-step geometries) vs orc.coherent_dedispersion(orc.stft(x))
  * one-tile blocks with many series, user chirps of every broadcastable shape, incoherent dedispersion (two-pass and
    direct), freq_shift (mixer folded into the first pass)
usage: python tests/tools/fuzz_round2.py [seconds] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import pulsarbat_amd as pb
from pulsarbat_amd import _hip, units as u
from oracle import dedisp_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
count = {}
bad = 0


def rnd(shape, dtype=np.complex64):
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)


def sig(x, sr, fc, **kw):
    if x.ndim == 3 and x.shape[2] == 2:
        return pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear", **kw)
    return pb.BasebandSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, **kw)


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - b) / max(np.linalg.norm(b), 1e-30)


_SMOOTH = None


def smooth_len(lo, hi):
    """An even 7-smooth number in [lo, hi] that is not a power of two: the mixed-radix column passes (one or two levels)
    and, with fewer than five factors of two, the mixed-radix rows."""
    global _SMOOTH
    if _SMOOTH is None:
        from pulsarbat_amd.utils import _smooth_7
        _SMOOTH = [v for v in _smooth_7(1 << 22) if v & (v - 1)]
    return int(rng.choice([v for v in _SMOOTH if lo <= v <= hi]))


def report(kind, ok, msg):
    global bad
    count[kind] = count.get(kind, 0) + 1
    if not ok:
        bad += 1
        print(f"BAD {kind}: {msg}", flush=True)


while time.time() - t0 < budget:
    kind = rng.choice(["slices", "stft", "onetile", "chirp", "incoherent", "freqshift"])
    sr, fc = float(rng.choice([1e6, 8e6])), float(rng.choice([8e8, 1.3e9]))
    if kind == "slices":
        n = 1 << int(rng.integers(12, 19)) if rng.random() < 0.6 else (int(rng.integers(3000, 100000)) if rng.random() < 0.5 else smooth_len(3000, 300000))
        nchan, npol = int(rng.integers(1, 9)), int(rng.choice([1, 2]))
        dtype = np.complex64 if rng.random() < 0.8 else np.complex128
        x = rnd((n, nchan, npol), dtype)
        dm = float(rng.choice([0.5, 3.0]))
        want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
        nout = stop - start
        if nout < 8:
            continue
        total, first = nchan + int(rng.integers(0, 5)), 0
        first = int(rng.integers(0, total - nchan + 1))
        nparts = int(rng.integers(1, 5))
        cuts = sorted(set([0, nout] + [int(c) for c in rng.integers(0, nout + 1, nparts - 1)]))
        rows = cuts
        tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
        guard = 3
        parts = [torch.full((rows[i + 1] - rows[i] + 2 * guard, total, npol), 9.0 - 4.0j, dtype=tdt, device="cuda")
                 for i in range(len(rows) - 1)]
        with _hip.Plan(n, nchan, npol, start, stop, device=0, dtype=dtype) as plan:
            plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, orc.channel_freqs(fc, sr, nchan), fc)
            esz = np.dtype(dtype).itemsize
            plan.dedisperse_slices(pb.DeviceArray.from_host(x), [p.data_ptr() + guard * total * npol * esz for p in parts], rows,
                                   total * npol, first * npol)
            torch.cuda.synchronize()
        got = np.concatenate([p[guard:p.shape[0] - guard].cpu().numpy() for p in parts], axis=0)
        e = relerr(got[:, first:first + nchan], want)
        rest = np.delete(got, np.s_[first:first + nchan], axis=1)
        edges = all(bool((p[:guard] == 9.0 - 4.0j).all() and (p[p.shape[0] - guard:] == 9.0 - 4.0j).all()) for p in parts)
        ok = e < (1e-5 if dtype == np.complex64 else 1e-8) and bool(np.all(rest == np.complex64(9.0 - 4.0j))) and edges
        report(kind, ok, f"n={n} nchan={nchan} npol={npol} {np.dtype(dtype)} total={total} first={first} rows={rows} err={e:.2e} edges={edges}")
    elif kind == "stft":
        m = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048]))
        nseg = 1 << int(rng.integers(13, 18))
        nchan = int(rng.choice([1, 2, 3, 4, 8]))
        pol = bool(rng.integers(0, 2))
        if nseg * m * nchan * (2 if pol else 1) > (1 << 25):
            continue
        shape = (nseg * m, nchan, 2) if pol else (nseg * m, nchan)
        x = rnd(shape)
        dm = float(rng.choice([1.0, 5.0, 20.0]))
        ch = orc.stft(x, m)
        want, start, stop = orc.coherent_dedispersion(ch, dm, sr / m, fc, freq_align="bottom")
        if want.shape[0] < 8:
            continue
        y = pb.contrib.stft_dedisperse(sig(x, sr, fc).to_device(), pb.DM(dm), nperseg=m)
        e = relerr(y, want) if y.shape == want.shape else 1.0
        report(kind, e < 1e-5, f"shape={shape} nperseg={m} dm={dm} err={e:.2e} shapes {y.shape} {want.shape}")
    elif kind == "onetile":
        n = 1 << int(rng.integers(10, 15))
        nchan = int(rng.choice([64, 200, 512, 1000]))
        pol = bool(rng.integers(0, 2))
        shape = (n, nchan, 2) if pol else (n, nchan)
        x = rnd(shape)
        dm = float(rng.choice([0.5, 2.0]))
        want, start, stop = orc.coherent_dedispersion(x, dm, sr / 64, fc)
        if want.shape[0] < 8:
            continue
        z = sig(x, sr / 64, fc)
        y = pb.coherent_dedispersion(z.to_device() if rng.random() < 0.7 else z, pb.DM(dm))
        e = relerr(y, want)
        report(kind, e < 1e-5, f"shape={shape} dm={dm} err={e:.2e}")
    elif kind == "chirp":
        n, nchan = 1 << int(rng.integers(11, 17)), int(rng.integers(1, 6))
        if rng.random() < 0.3:
            n = smooth_len(2000, 150000)
        shape = (n, nchan, 2)
        x = rnd(shape)
        dm = 2.0
        base = orc.chirp_from_signal(dm, shape, sr, fc)
        form = rng.choice(["2d", "3d", "perpol", "shared", "vec"])
        c = {"2d": base[:, :, 0], "3d": base, "shared": base[:, :1], "vec": base[:, 0, 0],
             "perpol": (base * np.exp(2j * np.pi * rng.random((1, nchan, 2)))).astype(np.complex64)}[form]
        want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc, chirp=np.asarray(c))
        if want.shape[0] < 8:
            continue
        z = sig(x, sr, fc)
        dev = rng.random() < 0.5
        y = pb.coherent_dedispersion(z.to_device() if dev else z, pb.DM(dm), chirp=pb.DeviceArray.from_host(c) if dev and rng.random() < 0.5 else c)
        e = relerr(y, want)
        report(kind, e < 1e-5, f"shape={shape} form={form} device={dev} err={e:.2e}")
    elif kind == "incoherent":
        n = int(rng.integers(70000, 300000))
        nchan, inner = int(rng.choice([2, 4, 6, 8, 16])), int(rng.choice([1, 2]))
        shape = (n, nchan, inner) if inner > 1 else (n, nchan)
        x = rnd(shape)
        dm = float(rng.choice([1.0, 8.0]))
        want, crop = orc.incoherent_dedispersion(x, dm, 1e6, 4e8, 1e6)
        if want.shape[0] < 8:
            continue
        z = sig(x, 1e6, 4e8, start_time=pb.Time(56000.0, format="mjd"))
        y = pb.incoherent_dedispersion(z.to_device(), pb.DM(dm))
        report(kind, y.shape == want.shape and np.array_equal(np.asarray(y), want), f"shape={shape} dm={dm}")
    else:
        n = 1 << int(rng.integers(12, 22))
        if rng.random() < 0.3:
            n = smooth_len(4000, 1 << 21)
        nchan = int(rng.choice([1, 2, 4, 8]))
        shape = (n, nchan, 2)
        if n * nchan > (1 << 23):
            continue
        x = rnd(shape)
        per = rng.random() < 0.5
        ft = rng.uniform(-0.3, 0.3, (nchan, 1)) if per else np.array(rng.uniform(-0.3, 0.3))
        want = orc.freq_shift(x, np.broadcast_to(ft, shape[1:]) if per else float(ft))
        z = sig(x, sr, fc)
        y = pb.freq_shift(z.to_device(), (ft * sr) * u.Hz)
        e = relerr(y, want)
        report(kind, e < 2e-5, f"shape={shape} per_channel={per} err={e:.2e}")
print("cases " + ", ".join(f"{k} {v}" for k, v in sorted(count.items())) + f"; bad {bad}; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
