import sys, math, json, numpy as np
sys.path.insert(0, ".")
import torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
n, nchan, npol, dm, band, center = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24, 8, 2, 56.77, 400e6, 1.4e9
sr = band / nchan
d = pb.DM(dm)
top = d.sample_delay((center + band / 2) * u.Hz, center * u.Hz, sr * u.Hz); bot = d.sample_delay((center - band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
freqs = center + sr * (np.arange(nchan) + 0.5 - nchan / 2)
x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda", dtype=torch.float64) * 0.7071))
plan = _hip.Plan(n, nchan, npol, start, stop, dtype=np.complex128)
plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
y = DeviceArray.empty((plan.nout, nchan, npol), np.complex128)
for _ in range(2): plan.dedisperse(x, out=y)
k = plan.profile(x, y, iters=5)
print("c128", n, "x 8 x 2:", round(sum(ms for _, ms in k), 3), "ms", {a: round(b, 3) for a, b in k})
