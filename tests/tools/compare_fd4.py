import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dedisp_oracle as orc
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
SR, FC = 1e6, 1e9
def run(log2n, nchan, npol, dm, seed, fd4):
    os.environ["PBH_FD4"] = "1" if fd4 else "0"
    n = 1 << log2n
    x = orc.synthetic_block((n, nchan, npol), seed)
    start, stop = orc.crop_bounds(dm, n, nchan, SR, FC, FC)
    freqs = FC + SR * (np.arange(nchan) + 0.5 - nchan / 2)
    with _hip.Plan(n, nchan, npol, start, stop) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / SR, freqs, FC)
        xd = DeviceArray.from_host(x)
        y = plan.dedisperse(xd)
        names = [k for k, _ in plan.profile(xd, y, iters=1)]
        got = np.asarray(plan.dedisperse(xd))
    return got, names, start
for (l, c, p) in [(20, 8, 2), (24, 8, 2)]:
    a, na, start = run(l, c, p, 30.0, 1, True)
    b, nb, _ = run(l, c, p, 30.0, 1, False)
    print(l, c, p, na, nb)
    d = np.abs(a - b).reshape(len(a), -1)
    bad = d > 1e-6
    print("frac bad per series", bad.mean(axis=0))
    rows = np.nonzero(bad.any(axis=1))[0] + start
    print("n bad rows", len(rows), "first", rows[:20])
    if len(rows):
        n2 = rows % 16384; n1 = rows // 16384
        print("n2 hist (mod 64):", np.bincount(n2 % 64, minlength=64))
        print("n2//64 unique count", len(np.unique(n2 // 64)), "n1 unique", len(np.unique(n1)))
