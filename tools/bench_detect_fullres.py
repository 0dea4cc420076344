"""dedisperse + detect at FULL time resolution (nscrunch = 1: to_intensity / to_stokes of the dedispersed voltages) at the
headline geometry, with the last layout pass detecting (default) and as dedisperse -> stored voltages -> k_detect
(PBH_DETECT_REINT=0), a child process per setting."""
import json, os, subprocess, sys

CHILD = r'''
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
sys.path.insert(0, "tools")
from bench_configs import crop
n, nchan, npol, dm, band, center = 1 << LOG2N, NCHAN, 2, 56.77, 400e6, 1.4e9
sr = band / nchan
start, stop = crop(dm, n, band, center, sr)
freqs = center + sr * (np.arange(nchan) + 0.5 - nchan / 2)
x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
plan = _hip.Plan(n, nchan, npol, start, stop)
plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
res = {}
for mode in ("intensity", "I", "linear"):
    oe = {"intensity": (nchan, npol), "I": (nchan,), "linear": (nchan, 4)}[mode]
    out = DeviceArray.empty((stop - start,) + oe, np.float32)
    for _ in range(3):
        plan.dedisperse_detect(x, nscrunch=1, mode=mode, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        plan.dedisperse_detect(x, nscrunch=1, mode=mode, out=out)
    torch.cuda.synchronize()
    res[mode] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
y = DeviceArray.empty((stop - start, nchan, npol), np.complex64)
for _ in range(3):
    plan.dedisperse(x, out=y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    plan.dedisperse(x, out=y)
torch.cuda.synchronize()
res["voltages only"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
print(json.dumps(res))
'''
# usage: python tools/bench_detect_fullres.py [log2n=24] [nchan=8]    (17 1024: a channelised block, two-axis layout tiles)
CHILD = CHILD.replace("LOG2N", sys.argv[1] if len(sys.argv) > 1 else "24").replace("NCHAN", sys.argv[2] if len(sys.argv) > 2 else "8")
for flag in ("1", "0"):
    env = dict(os.environ, PBH_DETECT_REINT=flag)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(f"PBH_DETECT_REINT={flag}: ms per step {line[-1] if line else r.stderr[-800:]}", flush=True)
