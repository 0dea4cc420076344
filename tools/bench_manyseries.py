#!/usr/bin/env python3
"""Blocks with more than 16 series through the four-pass schedule and through the five passes (PBH_FD4=0, read at every call):
per-kernel times and the distance between the two results.  usage: tools/bench_manyseries.py [log2n nchan npol] ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(log2n, nchan, npol):
    import torch
    from pulsarbat_amd import _hip
    from pulsarbat_amd.device import DeviceArray
    n, sr, fc, dm = 1 << log2n, 400e6 / nchan, 1.4e9, 56.77
    g = torch.Generator(device="cuda").manual_seed(3)
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), generator=g, device="cuda") * 0.7071))
    freqs = fc + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    delay = 4.148808e3 * dm * abs((fc / 1e6 - 200.0) ** -2 - (fc / 1e6 + 200.0) ** -2)
    crop = min(int(delay * sr) + 1, n // 2)
    res = {}
    with _hip.Plan(n, nchan, npol, 0, n - crop) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, fc)
        for fd4 in ("1", "0"):
            os.environ["PBH_FD4"] = fd4
            y = plan.dedisperse(x)
            ks = plan.profile(x, y, iters=10)
            y = plan.dedisperse(x)
            res[fd4] = (y._t[:: 257].clone(), ks)
            print(f"2^{log2n} x {nchan} x {npol} PBH_FD4={fd4}: total {sum(ms for _, ms in ks):.4f} ms  " + " ".join(f"{nm[2:]}={ms:.4f}" for nm, ms in ks), flush=True)
            del y
        os.environ.pop("PBH_FD4", None)
    a, b = res["1"][0], res["0"][0]
    d = float((a - b).abs().pow(2).sum().sqrt() / b.abs().pow(2).sum().sqrt())
    print(f"   relative L2 distance of the two results: {d:.2e}", flush=True)
    assert d < 5e-7, d


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:]] or [24, 16, 2, 22, 32, 2, 24, 10, 2]
    for i in range(0, len(args), 3):
        run(*args[i:i + 3])
