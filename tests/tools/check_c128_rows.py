import sys
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc
shape, dm = (1 << 20, 2, 2), 40.0
rng = np.random.default_rng(2)
x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5)
yr, start, stop = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
z = pb.DualPolarizationSignal(x, sample_rate=1e6 * u.Hz, center_freq=1e9 * u.Hz, pol_type="linear").to_device()
for variant in ("planar5", "direct3"):
    y = np.asarray(pb.coherent_dedispersion(z, pb.DM(dm), variant=variant))
    e = np.abs(y - yr)
    bad = np.argwhere(e > 1e-6)
    print(variant, f"rel err {np.linalg.norm(y - yr) / np.linalg.norm(yr):.3e}", "bad elements", len(bad),
          "series", sorted(set((int(b[1]), int(b[2])) for b in bad))[:8],
          "t range", (int(bad[:, 0].min()), int(bad[:, 0].max())) if len(bad) else None, flush=True)
    if len(bad):
        n2 = np.unique(bad[:, 0] % 8192)
        print("   distinct n2:", len(n2), "tau = n2 % 512:", np.unique(n2 % 512)[:40], "n2//512:", np.unique(n2 // 512)[:20])
