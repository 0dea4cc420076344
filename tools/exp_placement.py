#!/usr/bin/env python3
"""Round 4: where do the per-process differences of the headline step come from?

Round 3 saw 3.905-4.003 ms for the same binary in different processes of ONE box, the layout passes moving with it
(0.76 <-> 0.83 ms).  This script re-creates the buffers of the headline block many times inside one process, with a
spacer allocation of varying size in front so that every trial lands elsewhere, and prints the addresses next to the
per-kernel times (run with PBH_TRACE_ALLOC=1 to get the library's own allocations on stderr).

usage: python tools/exp_placement.py [trials] [--pad BYTES ...]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pulsarbat_amd import _hip  # noqa: E402
from pulsarbat_amd.device import DeviceArray  # noqa: E402

N, NCHAN, NPOL = 1 << 24, 8, 2
START, STOP = 1408404, 14607231
SR, FC, DM = 50e6, 1.4e9, 56.77


def one(trial, spacer_bytes, src, order):
    sp = torch.empty(spacer_bytes, dtype=torch.uint8, device="cuda") if spacer_bytes else None
    bufs = {}

    def mk_x():
        bufs["x"] = torch.empty((N, NCHAN, NPOL), dtype=torch.complex64, device="cuda")
        bufs["x"].copy_(src)

    def mk_y():
        bufs["y"] = torch.empty((STOP - START, NCHAN, NPOL), dtype=torch.complex64, device="cuda")

    def mk_plan():
        bufs["plan"] = _hip.Plan(N, NCHAN, NPOL, START, STOP)

    steps = {"x": mk_x, "y": mk_y, "p": mk_plan}
    for ch in order:
        steps[ch]()
    plan, x, y = bufs["plan"], DeviceArray(bufs["x"]), DeviceArray(bufs["y"])
    freqs = FC + SR * (np.arange(NCHAN) + 0.5 - NCHAN / 2)
    plan.chirp_generate(DM / 2.41e-4 * 1e12, 1 / SR, freqs, FC)
    plan.profile(x, y, iters=2)
    kern = plan.profile(x, y, iters=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    rec = {"trial": trial, "order": order, "spacer_MiB": spacer_bytes >> 20, "x": hex(bufs["x"].data_ptr()),
           "y": hex(bufs["y"].data_ptr()), "ms_step": round(ms, 4), "kernel_ms": {k: round(v, 4) for k, v in kern},
           "kernel_total": round(sum(v for _, v in kern), 4)}
    print(json.dumps(rec), flush=True)
    plan.close()
    del plan, x, y, bufs, sp
    torch.cuda.empty_cache()


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    g = torch.Generator(device="cuda").manual_seed(1)
    src = torch.view_as_complex(torch.randn((N, NCHAN, NPOL, 2), generator=g, device="cuda") * 0.7071)
    rng = np.random.default_rng(4)
    orders = ["xyp", "pxy", "xpy"]
    for t in range(trials):
        spacer = 0 if t < 3 else int(rng.integers(0, 2048)) * (2 << 20) + int(rng.integers(0, 32)) * 65536
        one(t, spacer, src, orders[t % 3])


if __name__ == "__main__":
    main()
