"""time_shift / dedispersion error vs the oracle at lengths that are not native (padded-convolution path)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc
rng = np.random.default_rng(1)
for n in (2621441, 2621443, 1000003, 300001, 100003, 30011, 10007):
    x = (rng.standard_normal((n, 1, 1)) + 1j * rng.standard_normal((n, 1, 1))).astype(np.complex64)
    sig = pb.Signal(pb.DeviceArray.from_host(x), sample_rate=1 * u.kHz)
    for sh in (-21.31, 7.0):
        got = np.asarray(pb.time_shift(sig, sh))
        ref, _, _ = orc.time_shift(x, sh)
        e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        print("time_shift", n, sh, f"{e:.2e}", "BAD" if e > 4e-6 else "", flush=True)
    z = pb.BasebandSignal(pb.DeviceArray.from_host(x[:, :, 0]), sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
    y = np.asarray(pb.coherent_dedispersion(z, pb.DM(3.0)))
    yr, _, _ = orc.coherent_dedispersion(x[:, :, 0], 3.0, 1e6, 1e9)
    e = np.linalg.norm(y - yr) / np.linalg.norm(yr)
    print("dedispersion", n, f"{e:.2e}", "BAD" if e > 4e-6 else "", flush=True)
