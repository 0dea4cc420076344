"""configs[4]'s per-GPU share with several scrunch factors (run under rocprofv3 --kernel-trace --stats to see k_detect_reduce)."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np, torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
from bench_configs import crop
n, nchan_tot, nchan, npol, dm, band, center = 1 << 24, 64, 8, 2, 1000.0, 400e6, 1.4e9
sr = band / nchan_tot
start, stop = crop(dm, n, band, center, sr)
freqs = (center + sr * (np.arange(nchan_tot) + 0.5 - nchan_tot / 2))[:nchan]
x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
plan = _hip.Plan(n, nchan, npol, start, stop)
plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
for ns in [int(a) for a in sys.argv[1:]] or [64, 1024, 16384]:
    for _ in range(5):
        plan.dedisperse_detect(x, nscrunch=ns, mode="I")
torch.cuda.synchronize()
