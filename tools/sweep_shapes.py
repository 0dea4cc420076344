"""Throughput over a grid of block shapes (looking for cliffs): Gsamples/s per (log2 N, series)."""
import sys, math, json
import numpy as np
sys.path.insert(0, ".")
import torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

def run(log2n, nchan, npol, dm=2.0, band=400e6, center=1.4e9, dtype=np.complex64):
    n = 1 << log2n
    sr = band / nchan
    d = pb.DM(dm)
    top = d.sample_delay((center + band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    bot = d.sample_delay((center - band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
    if stop - start < n // 2:
        return None
    freqs = center + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    tdt = torch.float32 if dtype == np.complex64 else torch.float64
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda", dtype=tdt)))
    plan = _hip.Plan(n, nchan, npol, start, stop, dtype=dtype)
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
    y = DeviceArray.empty((plan.nout, nchan, npol), dtype)
    for _ in range(2):
        plan.dedisperse(x, out=y)
    k = plan.profile(x, y, iters=3)
    tot = sum(ms for _, ms in k)
    plan.close()
    del x, y
    return n * nchan * npol / tot / 1e6, k

if __name__ == "__main__":
    total_log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    dtype = np.complex128 if (len(sys.argv) > 2 and sys.argv[2] == "c128") else np.complex64
    series = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 100, 128, 130, 200, 256, 512, 1000, 2048, 4096]
    print("rows: log2(N); columns: series (nchan x npol); Gsamples/s")
    print("      " + " ".join(f"{s:6d}" for s in series))
    worst = []
    for lg in range(15, 28):
        row = []
        for S in series:
            if lg + math.log2(S) > total_log2 + 0.01 or lg + math.log2(S) < total_log2 - 1.01:
                row.append("     .")
                continue
            nchan, npol = (S // 2, 2) if S % 2 == 0 and S >= 2 else (S, 1)
            try:
                r = run(lg, nchan, npol, dtype=dtype)
            except Exception as e:
                row.append("   ERR")
                worst.append((lg, S, repr(e)[:80]))
                continue
            if r is None:
                row.append("     -")
                continue
            row.append(f"{r[0]:6.1f}")
            if r[0] < 45:
                worst.append((lg, S, round(r[0], 1), {a: round(b, 3) for a, b in r[1]}))
        print(f"2^{lg:2d}  " + " ".join(row), flush=True)
    for w in worst:
        print("slow:", w)
