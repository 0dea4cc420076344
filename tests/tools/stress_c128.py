"""Repeat the complex128 / complex64 planar5 parity case in one process and print any deviation
(looking for order- or timing-dependent results: every repeat must give bit-identical output)."""
import sys
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc

def sig(x):
    return pb.DualPolarizationSignal(x, sample_rate=1e6 * u.Hz, center_freq=1e9 * u.Hz, pol_type="linear")

bad = 0
for dtype, shape, dm in [(np.complex128, (1 << 17, 4, 2), 20.0), (np.complex64, (1 << 18, 4, 2), 20.0),
                         (np.complex128, (1 << 20, 2, 2), 40.0), (np.complex64, (1 << 21, 4, 2), 30.0)]:
    rng = np.random.default_rng(2)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    yr, start, stop = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    z = sig(x).to_device()
    first = {}
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        for variant in ("planar5", "direct3", "planar5"):
            y = np.asarray(pb.coherent_dedispersion(z, pb.DM(dm), variant=variant))
            err = np.linalg.norm(y - yr) / np.linalg.norm(yr)
            if variant not in first:
                first[variant] = y.copy()
                print(dtype.__name__, shape, variant, f"rel err {err:.3e}", flush=True)
            elif not np.array_equal(y, first[variant]):
                bad += 1
                d = np.abs(y - first[variant])
                idx = np.argwhere(d > 0)
                print(f"MISMATCH it={it} {dtype.__name__} {variant}: {len(idx)} elements differ, max {d.max():.3e}, "
                      f"rel err {err:.3e}, first idx {idx[:4].tolist()} last idx {idx[-4:].tolist()}", flush=True)
print("mismatches:", bad)
