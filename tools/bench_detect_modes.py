"""dedisperse + detect + 1024x scrunch at configs[4]'s per-GPU geometry for every detect mode, with the detection inside the
inverse column pass (default) and as a read pass of its own (PBH_DETECT_COLQ=0), a child process per setting."""
import json, os, subprocess, sys, time

CHILD = r'''
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
sys.path.insert(0, "tools")
from bench_configs import crop
n, nchan_tot, nchan, npol, dm, band, center = 1 << LOG2N, NCHAN_TOT, 8, 2, DMV, 400e6, 1.4e9
sr = band / nchan_tot
start, stop = crop(dm, n, band, center, sr)
freqs = (center + sr * (np.arange(nchan_tot) + 0.5 - nchan_tot / 2))[:nchan]
x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
plan = _hip.Plan(n, nchan, npol, start, stop)
plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
res = {}
for mode in ("intensity", "I", "linear", "circular"):
    for _ in range(3):
        out = plan.dedisperse_detect(x, nscrunch=1024, mode=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = plan.dedisperse_detect(x, nscrunch=1024, mode=mode)
    torch.cuda.synchronize()
    res[mode] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
print(json.dumps(res))
'''
# usage: python tools/bench_detect_modes.py [log2n=24] [nchan_total=64] [dm=1000]   (22 8 56.77: configs[3]'s chunk geometry)
log2n = sys.argv[1] if len(sys.argv) > 1 else "24"
nchan_tot = sys.argv[2] if len(sys.argv) > 2 else "64"
dmv = sys.argv[3] if len(sys.argv) > 3 else "1000.0"
CHILD = CHILD.replace("LOG2N", log2n).replace("NCHAN_TOT", nchan_tot).replace("DMV", dmv)
print(f"2^{log2n} samples x 8 of {nchan_tot} channels x 2 pol, DM {dmv}, 1024x scrunch")
for flag in ("1", "0"):
    env = dict(os.environ, PBH_DETECT_COLQ=flag)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(f"PBH_DETECT_COLQ={flag}: ms per step {line[-1] if line else r.stderr[-600:]}", flush=True)
