"""Out-of-bounds detector: the input is a view into a larger device buffer whose elements before and after it are
NaN (any read outside the array poisons the result), the output a view into a buffer of sentinels (any write
outside it shows).  Random lengths (2^k, m*2^k, arbitrary, odd), series counts, offsets, both precisions.
usage: fuzz_guard.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from pulsarbat_amd.transforms.dedispersion import _prepare, clear_plan_cache
from oracle import dedisp_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
t_end = time.time() + budget
ok = bad = 0
G = 64   # guard elements on each side
while time.time() < t_end:
    dtype = np.complex64 if rng.random() < 0.7 else np.complex128
    kind = rng.integers(0, 5)
    if kind == 0:
        n = 1 << int(rng.integers(6, 22))
    elif kind == 1:
        n = int(rng.choice([3, 5, 7])) << int(rng.integers(19, 21))
    elif kind == 2:
        n = int(rng.integers(100, 3000000))
    elif kind == 3:
        n = (1 << int(rng.integers(10, 22))) + int(rng.choice([-3, -1, 1, 3]))
    else:
        n = 2 * int(rng.integers(50, 1500000)) + 1
    nchan = int(rng.integers(1, 7))
    tail = (nchan,) + ((int(rng.integers(1, 4)),) if rng.random() < 0.6 else ())
    S = int(np.prod(tail))
    if n * S > (1 << 23):
        continue
    g0 = int(rng.integers(0, 4))          # sample offset of the view inside the buffer (alignment varies)
    rdt = np.float32 if dtype == np.complex64 else np.float64
    buf = np.full((G + g0 + n + G,) + tail, np.nan + 1j * np.nan, dtype=dtype)
    x = (rng.standard_normal((n,) + tail) + 1j * rng.standard_normal((n,) + tail)).astype(dtype)
    buf[G + g0:G + g0 + n] = x
    dbuf = pb.DeviceArray.from_host(buf)
    view = dbuf[G + g0:G + g0 + n]
    sr, fc = 1e6, 1e9
    dm = float(rng.uniform(0.0, 0.3)) * min(1.0, n / 4096)
    z = pb.BasebandSignal(view, sample_rate=sr * u.Hz, center_freq=fc * u.Hz)
    yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    if stop - start < 1:
        continue
    plan, xin, s0, s1 = _prepare(z, pb.DM(dm), None, None, "auto")
    nout = s1 - s0
    mode = "sample"
    if plan.supports_series_major and S > 1 and rng.random() < 0.5:
        # series-major (time fastest) arrays carved out of guarded flat buffers: pitch padded, random odd-ish pads
        mode = "series"
        tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
        def carve(rows, fill):
            pitch = rows + int(rng.integers(0, 40))
            off = G + int(rng.integers(0, 4))
            flat = torch.full((off + S * pitch + G,), fill, dtype=tdt, device="cuda")
            strides = [1]
            acc = pitch
            for d in reversed(tail):
                strides.insert(1, acc)
                acc *= d
            return flat, pb.DeviceArray(flat.as_strided((rows,) + tail, tuple(strides), off)), off, pitch
        fin, xin, _, _ = carve(n, complex(float("nan"), float("nan")))
        xin.tensor.copy_(torch.from_numpy(x).cuda())
        fout, oview, ooff, opitch = carve(nout, 777.0 + 0j)
        plan.dedisperse(xin, out=oview)
        got = oview.tensor.cpu().numpy()
        flat = fout.cpu().numpy()
        mask = np.ones(flat.shape, bool)
        for si in range(S):
            mask[ooff + si * opitch: ooff + si * opitch + nout] = False
        guards_ok = bool(np.all(flat[mask] == 777.0))
    else:
        obuf = pb.DeviceArray.from_host(np.full((G + nout + G,) + tail, 777.0 + 0j, dtype=dtype))
        oview = obuf[G:G + nout]
        plan.dedisperse(xin, out=oview)
        res = np.asarray(obuf)
        got = res[G:G + nout]
        guards_ok = np.all(res[:G] == 777.0) and np.all(res[G + nout:] == 777.0)
    tol = 5e-6 if dtype == np.complex64 else 1e-9
    e = np.linalg.norm(got - yr) / np.linalg.norm(yr) if np.all(np.isfinite(got)) else np.inf
    if not (guards_ok and (s0, s1) == (start, stop) and e < tol):
        bad += 1
        print(f"BAD [{mode}] n={n} tail={tail} g0={g0} {np.dtype(dtype).name} dm={dm:.3f}: err {e:.2e} guards_ok={guards_ok} info={plan.info}", flush=True)
    else:
        ok += 1
    if (ok + bad) % 50 == 0:
        clear_plan_cache()
print(f"cases ok {ok}, bad {bad}", flush=True)
sys.exit(1 if bad else 0)
