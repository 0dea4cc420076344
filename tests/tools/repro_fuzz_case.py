"""Re-run one case of fuzz_parity.py: repro_fuzz_case.py n nchan npol dtype sr fc dm mode variant qmax [repeat]"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
n, nchan, npol = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dtype = np.complex64 if sys.argv[4] == "c64" else np.complex128
sr, fc, dm = float(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])
mode, variant, q = sys.argv[8], sys.argv[9], int(sys.argv[10])
rep = int(sys.argv[11]) if len(sys.argv) > 11 else 3
if q:
    os.environ["PBH_QMAX"] = str(q)
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc
rng = np.random.default_rng(5)
shape = (n, nchan, npol) if npol > 1 else (n, nchan)
x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
z = (pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear") if npol > 1
     else pb.BasebandSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz))
yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
zz = z if mode == "host" else z.to_device()
if mode == "series":
    zz = type(z).like(z, zz.data.to_series_major())
for r in range(rep):
    y = np.asarray(pb.coherent_dedispersion(zz, pb.DM(dm), variant=variant).data)
    err = np.linalg.norm(y.reshape(len(y), -1) - yr.reshape(len(yr), -1), axis=0) / np.linalg.norm(yr.reshape(len(yr), -1), axis=0)
    print(f"run {r}: crop [{start},{stop}) per-series rel err max {err.max():.3e} ; bad series {np.nonzero(err > 1e-5)[0].tolist()} ; nan {np.isnan(y).any()}", flush=True)
from pulsarbat_amd.transforms.dedispersion import _plan_for, _crop_bounds
plan, _ = _plan_for(z, pb.DM(dm), z.center_freq, (start, stop), variant=variant)
print(plan.info)
