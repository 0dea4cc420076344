"""Per-kernel times of blocks whose series count is not a multiple of four (the five-pass / three-pass fallbacks) next to the headline shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
for log2n, nchan, npol in ((24, 1, 2), (24, 3, 2), (24, 5, 2), (24, 7, 2), (24, 8, 2), (24, 1, 1), (26, 1, 2), (26, 2, 2)):
    n, sr, fc, dm = 1 << log2n, 400e6 / 8, 1.4e9, 56.77
    g = torch.Generator(device="cuda").manual_seed(3)
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), generator=g, device="cuda") * 0.7071))
    freqs = fc + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    crop = n // 8
    with _hip.Plan(n, nchan, npol, 0, n - crop) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, fc)
        y = plan.dedisperse(x)
        ks = plan.profile(x, y, iters=10)
        tot = sum(ms for _, ms in ks)
        print(f"2^{log2n} x {nchan} x {npol}: total {tot:.4f} ms = {n*nchan*npol/tot/1e6:.1f} Gsamples/s  " + " ".join(f"{nm[2:]}={ms:.4f}" for nm, ms in ks), flush=True)
    del x, y
