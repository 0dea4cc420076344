import sys, numpy as np
sys.path.insert(0, ".")
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc
ok = True
for shape in [(1 << 19, 2, 2), (1 << 20, 4, 2), (1 << 21, 2, 2), (1 << 17, 8, 2), (1 << 22, 2, 2)]:
    x = orc.synthetic_block(shape, 3)
    z = pb.DualPolarizationSignal(x, sample_rate=1e6 * u.Hz, center_freq=1e9 * u.Hz, pol_type="linear")
    y = np.asarray(pb.coherent_dedispersion(z, pb.DM(20.0), variant="block3"))
    yr, a, b = orc.coherent_dedispersion(x, 20.0, 1e6, 1e9)
    e = np.linalg.norm(y - yr) / np.linalg.norm(yr)
    print(shape, y.shape == yr.shape, e, flush=True)
    ok &= (y.shape == yr.shape) and e < 1e-5
print("ALL OK" if ok else "FAIL")
