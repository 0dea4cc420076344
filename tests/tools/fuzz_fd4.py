"""Randomised check of the four-pass schedule (csrc/fd4_kernels.hpp) with guards: the input is a view into a larger device
buffer of NaNs (any read outside the block poisons the result), the output a view into a buffer of sentinels (any write outside
it shows).  Quad-series blocks S in {4, 8, ..., 64}, 2^20 ... 2^24 samples (column tiles of 64 ... 1024 rows), random crops,
channel / pol splits and view offsets; every case against the five-pass schedule on the same input (PBH_FD4=0; equal to the
last bits) and, up to 2^21 samples, against the oracle.
usage: fuzz_fd4.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from oracle import dedisp_oracle as orc
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
t_end = time.time() + budget
ok = bad = vs_oracle = 0
G = 48
SR, FC = 1e6, 1e9
print(f"seed {seed}", flush=True)
while time.time() < t_end:
    log2n = int(rng.integers(20, 25))
    S = int(rng.choice([4, 8, 12, 16, 16, 20, 24, 28, 32, 36, 40, 48, 64]))
    n = 1 << log2n
    if n * S > (1 << 27):
        continue
    splits = [(c, S // c) for c in range(1, S + 1) if S % c == 0 and S // c in (1, 2, 4)]
    nchan, npol = splits[int(rng.integers(0, len(splits)))]
    dm = float(rng.uniform(0.0, 60.0)) * (n / (1 << 24)) * (8.0 / max(nchan, 1)) + float(rng.uniform(0, 0.2))
    start, stop = orc.crop_bounds(dm, n, nchan, SR, FC, FC)
    if stop - start < 64:
        continue
    g0 = int(rng.integers(0, 4))
    tail = (nchan, npol)
    buf = torch.full((G + g0 + n + G,) + tail, float("nan"), dtype=torch.complex64, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.view_as_complex(torch.randn((n,) + tail + (2,), generator=gen, device="cuda") * 0.7071)
    buf[G + g0:G + g0 + n] = x
    xin = DeviceArray(buf[G + g0:G + g0 + n])
    nout = stop - start
    freqs = FC + SR * (np.arange(nchan) + 0.5 - nchan / 2)

    def run(fd4):
        if fd4:
            os.environ.pop("PBH_FD4", None)
        else:
            os.environ["PBH_FD4"] = "0"
        obuf = torch.full((G + nout + G,) + tail, 777.0 + 0j, dtype=torch.complex64, device="cuda")
        with _hip.Plan(n, nchan, npol, start, stop) as plan:
            plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / SR, freqs, FC)
            oview = DeviceArray(obuf[G:G + nout])
            plan.dedisperse(xin, out=oview)
            names = [k for k, _ in plan.profile(xin, oview, iters=1)]
            plan.dedisperse(xin, out=oview)
        torch.cuda.synchronize()
        guards = bool(torch.all(obuf[:G] == 777.0).item() and torch.all(obuf[G + nout:] == 777.0).item())
        return obuf[G:G + nout].clone(), names, guards

    y4, names4, g4 = run(True)
    y5, names5, g5 = run(False)
    os.environ.pop("PBH_FD4", None)
    fin = bool(torch.all(torch.isfinite(torch.view_as_real(y4))).item())
    d = (torch.linalg.vector_norm(y4 - y5) / torch.linalg.vector_norm(y5)).item() if fin else float("inf")
    e = -1.0
    if log2n <= 21 and fin:
        yr, s0, s1 = orc.coherent_dedispersion(x.cpu().numpy(), dm, SR, FC, workers=8)
        e = float(np.linalg.norm(y4.cpu().numpy() - yr) / np.linalg.norm(yr)) if (s0, s1) == (start, stop) else float("inf")
        vs_oracle += 1
    good = g4 and g5 and fin and names4[0] == "k_col_fwd" and len(names4) == 4 and names5[0] == "k_deinterleave" and d < 5e-7 and e < 1e-5
    if good:
        ok += 1
    else:
        bad += 1
        print(f"BAD 2^{log2n} x {nchan} x {npol} g0={g0} dm={dm:.3f} crop=({start},{stop}): guards {g4}/{g5} finite {fin} "
              f"d(4,5)={d:.2e} oracle={e:.2e} kernels {names4}", flush=True)
    if (ok + bad) % 10 == 0:
        print(f"... {ok} ok, {bad} bad ({vs_oracle} also against the oracle)", flush=True)
    del buf, x, xin, y4, y5
    torch.cuda.empty_cache()
print(f"fuzz_fd4 seed {seed}: {ok} ok, {bad} bad, {vs_oracle} against the oracle")
sys.exit(1 if bad else 0)
