"""7-smooth block lengths: per-kernel timings of the native mixed-radix plan; run with PBH_MIXED=0 for the padded-convolution
plan the same lengths took before (the environment variable is read once per process).
usage: python tools/bench_smooth.py [n ...]   (default: the lengths DESIGN.md quotes)"""
import sys, math, json, os
import numpy as np
sys.path.insert(0, ".")
import torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
import pulsarbat_amd as pb
from pulsarbat_amd import units as u


def run(n, nchan=8, npol=2, dm=56.77, band=400e6, center=1.4e9):
    sr = band / nchan
    d = pb.DM(dm)
    top = d.sample_delay((center + band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    bot = d.sample_delay((center - band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
    freqs = center + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
    plan = _hip.Plan(n, nchan, npol, start, stop)
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
    y = DeviceArray.empty((plan.nout, nchan, npol), np.complex64)
    for _ in range(2):
        plan.dedisperse(x, out=y)
    k = plan.profile(x, y, iters=5)
    tot = sum(ms for _, ms in k)
    info = plan.info
    print(json.dumps({"PBH_MIXED": os.environ.get("PBH_MIXED", "default (2)"), "shape": [n, nchan, npol], "n1": info["n1"], "n2": info["n2"],
                      "nkernel": info["nkernel"], "ms": round(tot, 3), "Gsamples_per_s": round(n * nchan * npol / tot / 1e6, 1),
                      "kernels": {a: round(b, 3) for a, b in k}}), flush=True)
    plan.close()


if __name__ == "__main__":
    lens = [int(a) for a in sys.argv[1:]] or [10_000_000, 16_000_000, 625 * 16384, 3 ** 4 * 5 ** 3 * 1024, 2 ** 24]
    for n in lens:
        run(n)
