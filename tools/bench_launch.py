"""Small blocks: wall time per call through the Python host vs the sum of kernel times (launch / host overhead)."""
import sys, time, math
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import _hip, units as u
from pulsarbat_amd.device import DeviceArray

for lg, nchan, npol in ((12, 8, 2), (14, 8, 2), (16, 8, 2), (18, 8, 2), (20, 8, 2)):
    n = 1 << lg
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
    plan = _hip.Plan(n, nchan, npol, 10, n - 10)
    plan.chirp_generate(1e3, 1e-6, 1e9 + 1e6 * np.arange(nchan), 1e9)
    y = DeviceArray.empty((plan.nout, nchan, npol), np.complex64)
    for _ in range(5):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    k = plan.profile(x, y, iters=20)
    ksum = sum(ms for _, ms in k) * 1e3
    print(f"2^{lg} x {nchan} x {npol}: {wall:8.1f} us per call, kernels {ksum:8.1f} us ({len(k)} launches), "
          f"{n * nchan * npol / wall / 1e3:6.2f} Gsamples/s", flush=True)
    plan.close()
