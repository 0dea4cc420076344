import sys
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc
rng = np.random.default_rng(6)
for n, tail in (((5 << 19) + 1, (1, 1)), ((5 << 19) + 1, (2, 1)), (300001, (1, 1)), (300001, (2, 2)), (40001, (1,)), (1 << 20, (1,)), ((1 << 20) + 1, (1,))):
    shape = (n,) + tail
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    for sh in (29.092832789142875, -28.65244659147736, 29.0, 0.5):
        sig = pb.Signal(pb.DeviceArray.from_host(x), sample_rate=1 * u.kHz)
        got = np.asarray(pb.time_shift(sig, sh)).reshape(n, -1)
        ref = orc.time_shift(x, sh)[0].reshape(n, -1)
        d = np.abs(got - ref).max(axis=1)
        e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        worst = np.argsort(d)[-2:]
        print(n, tail, sh, f"{e:.2e}", "BAD" if e > 4e-6 else "", worst.tolist(), [round(float(v), 4) for v in d[worst]], flush=True)
