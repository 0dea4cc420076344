#!/usr/bin/env python3
"""Depth-first schedule experiment: config-2 block, PBH_DEPTH (read once per process) from the environment.
Prints ms/step, per-kernel HIP-event times and a checksum of the output (the schedules must agree bit for bit)."""
import hashlib
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pulsarbat_amd import _hip  # noqa: E402
from pulsarbat_amd.device import DeviceArray  # noqa: E402
import pulsarbat_amd as pb  # noqa: E402
from pulsarbat_amd import units as u  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nsample, nchan, npol = 1 << log2n, 8, 2
sr, fc, band = 400e6 / nchan, 1.4e9, 400e6
dm = pb.DM(56.77 if log2n >= 22 else 5.0)
d_top = dm.sample_delay((fc + band / 2) * u.Hz, fc * u.Hz, sr * u.Hz)
d_bot = dm.sample_delay((fc - band / 2) * u.Hz, fc * u.Hz, sr * u.Hz)
start = math.ceil(-min(0, d_top, d_bot))
stop = nsample - math.ceil(max(0, d_top, d_bot))
freqs = fc + sr * (np.arange(nchan) + 0.5 - nchan / 2)
gen = torch.Generator(device="cuda")
gen.manual_seed(1234)
x = torch.randn((nsample, nchan, npol, 2), generator=gen, device="cuda", dtype=torch.float32) * 2 ** -0.5
x = DeviceArray(torch.view_as_complex(x))
plan = _hip.Plan(nsample, nchan, npol, start, stop, device=0)
y = DeviceArray.empty((plan.nout, nchan, npol), np.complex64, device=0)
plan.chirp_generate(float(dm.value) / 2.41e-4 * 1e12, 1.0 / sr, freqs, fc)
for _ in range(3):
    plan.dedisperse(x, out=y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    plan.dedisperse(x, out=y)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
kern = plan.profile(x, y, iters=5)
h = hashlib.sha256(np.asarray(y.tensor[::97].cpu()).tobytes()).hexdigest()[:16]
print("PBH_DEPTH=%s log2n=%d: %.3f ms/step  %s  sum %.3f  sha %s" % (
    os.environ.get("PBH_DEPTH", "0"), log2n, ms, " ".join("%s=%.3f" % (k, v) for k, v in kern),
    sum(v for _, v in kern), h), flush=True)
