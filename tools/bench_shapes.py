"""Per-kernel timings for a few other block shapes (same DM/band as the headline)."""
import sys, math, json
import numpy as np
sys.path.insert(0, ".")
import torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

def run(log2n, nchan, npol, dm=56.77, band=400e6, center=1.4e9, nchan_total=None):
    n = 1 << log2n
    nchan_total = nchan_total or nchan
    sr = band / nchan_total
    d = pb.DM(dm)
    top = d.sample_delay((center + band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    bot = d.sample_delay((center - band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
    freqs = (center + sr * (np.arange(nchan_total) + 0.5 - nchan_total / 2))[:nchan]
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
    plan = _hip.Plan(n, nchan, npol, start, stop)
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
    if plan.nout <= 0:
        print(json.dumps({"shape": [n, nchan, npol], "crop": [start, stop], "skipped": "empty valid region at this DM"}), flush=True)
        plan.close()
        return
    y = DeviceArray.empty((plan.nout, nchan, npol), np.complex64)
    for _ in range(2):
        plan.dedisperse(x, out=y)
    k = plan.profile(x, y, iters=5)
    tot = sum(ms for _, ms in k)
    ns = n * nchan * npol
    print(json.dumps({"shape": [n, nchan, npol], "crop": [start, stop], "ms": round(tot, 3),
                      "Gsamples_per_s": round(ns / tot / 1e6, 1), "kernels": {a: round(b, 3) for a, b in k}}), flush=True)
    plan.close()

if __name__ == "__main__":
    run(24, 8, 2)
    run(22, 64, 2, nchan_total=64)
    run(24, 1, 2, nchan_total=8)
    run(24, 2, 2, nchan_total=8)
    run(24, 32, 1, nchan_total=32)
    run(20, 16, 2, nchan_total=16)
    run(26, 2, 2, nchan_total=8)
    run(18, 64, 2, nchan_total=64)
