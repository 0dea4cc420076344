"""Dedispersion at lengths that are not powers of two: time per call and error vs the oracle (1 series sample)."""
import sys, math, json, time
import numpy as np
sys.path.insert(0, ".")
import torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc

def run(n, nchan=8, npol=2, dm=56.77, band=400e6, center=1.4e9, check=True):
    sr = band / nchan
    d = pb.DM(dm)
    top = d.sample_delay((center + band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    bot = d.sample_delay((center - band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
    freqs = center + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
    plan = _hip.Plan(n, nchan, npol, start, stop)
    t0 = time.perf_counter()
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
    torch.cuda.synchronize(); tch = time.perf_counter() - t0
    y = DeviceArray.empty((plan.nout, nchan, npol), np.complex64)
    for _ in range(2):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    err = None
    if check:
        c, p = min(3, nchan - 1), min(1, npol - 1)
        xs = x.tensor[:, c, p].cpu().numpy().reshape(-1, 1)
        import scipy.fft
        chirp = orc.transfer_function(dm, n, 1 / sr, freqs[c], center).reshape(-1, 1)
        ref = scipy.fft.ifft(scipy.fft.fft(xs, axis=0) * chirp, axis=0)[start:stop, 0]
        got = y.tensor[:, c, p].cpu().numpy()
        err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(json.dumps({"n": n, "series": nchan * npol, "ms": round(dt * 1e3, 3), "Gsamples_per_s": round(n * nchan * npol / dt / 1e9, 2),
                      "chirp_setup_ms": round(tch * 1e3, 1), "rel_err": err, "kernels": plan.info["nkernel"]}), flush=True)
    plan.close()

if __name__ == "__main__":
    run(10_000_000)
    run(16_000_000)
    run(5_000_000)
    run(1 << 24)
