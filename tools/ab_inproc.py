#!/usr/bin/env python3
"""A/B of diagnostic switches INSIDE one process: the same plan, the same buffers (so the same allocation classes), the switch
changed between profile runs (the library reads its diagnostic switches at every launch).  Needs a -DPBH_DIAGNOSTIC build:
    PBH_EXTRA_FLAGS=-DPBH_DIAGNOSTIC python -c "from pulsarbat_amd import _build; _build.build(force=True)"
    gpurun -- 'PBH_EXTRA_FLAGS=-DPBH_DIAGNOSTIC tools/gpu_job.sh TAG 300 py:240:tools/ab_inproc.py,PBH_FD4_SP=0,PBH_FD4_SP=20'
usage: tools/ab_inproc.py [--log2n 24] [--nchan 8] [--rounds 3] CONFIG [CONFIG ...]     CONFIG: NAME=VAL[+NAME=VAL...] or -"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = sys.argv[1:]
    log2n, nchan, rounds = 24, 8, 3
    while args and args[0].startswith("--"):
        if args[0] == "--log2n": log2n = int(args[1])
        elif args[0] == "--nchan": nchan = int(args[1])
        elif args[0] == "--rounds": rounds = int(args[1])
        else: sys.exit("unknown option " + args[0])
        args = args[2:]
    from pulsarbat_amd import _hip
    from pulsarbat_amd.device import DeviceArray
    n, npol, sr, fc, dm = 1 << log2n, 2, 400e6 / nchan, 1.4e9, 56.77
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((n, nchan, npol), dtype=np.float32) + 1j * rng.standard_normal((n, nchan, npol), dtype=np.float32)).astype(np.complex64)
    freqs = fc + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    delay = 4.148808e3 * dm * abs((fc / 1e6 - 200.0) ** -2 - (fc / 1e6 + 200.0) ** -2)
    crop = min(int(delay * sr) + 1, n // 2)
    names_all = sorted({kv.split("=", 1)[0] for c in args if c != "-" for kv in c.split("+")})
    for mode in ("copy", "rmw"):
        ms = _hip.stream_bench(1 << 31, 10, 0, mode)
        print(f"yardstick {mode}: {(2 << 31) / ms * 1e-6:.0f} GB/s", flush=True)
    os.environ.setdefault("PBH_TRACE_ALLOC", "1")
    with _hip.Plan(n, nchan, npol, 0, n - crop) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, fc)
        xd = DeviceArray.from_host(x)
        y = plan.dedisperse(xd)
        ref = None
        res = {c: [] for c in args}
        for r in range(rounds):
            for c in args:
                for k in names_all:
                    os.environ.pop(k, None)
                if c != "-":
                    for kv in c.split("+"):
                        k, v = kv.split("=", 1)
                        os.environ[k] = v
                ks = plan.profile(xd, y, iters=20)
                out = np.asarray(y)[:: 4099]
                if ref is None: ref = out
                dev = float(np.abs(out - ref).max() / np.abs(ref).max())
                tot = sum(ms for _, ms in ks)
                res[c].append(tot)
                print(f"{c:30s} total {tot:.4f}  " + " ".join(f"{nm[2:]}={ms:.4f}" for nm, ms in ks) + f"  dev {dev:.1e}", flush=True)
        for c, v in res.items():
            print(f"== {c:30s} median {sorted(v)[len(v) // 2]:.4f}  min {min(v):.4f}")


if __name__ == "__main__":
    main()
