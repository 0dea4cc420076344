"""First-light GPU check: HIP path vs oracle on a sweep of sizes (debug aid, prints errors)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pulsarbat_amd as pb
from pulsarbat_amd import units as u, _hip
from oracle import dedisp_oracle as orc

def relerr(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def check(shape, dm, sr, fc, variant="auto", ref=None):
    x = orc.synthetic_block(shape, 1234)
    cls = pb.DualPolarizationSignal if (len(shape) == 3 and shape[2] == 2) else pb.BasebandSignal
    kw = dict(sample_rate=sr * u.Hz, center_freq=fc * u.Hz)
    if cls is pb.DualPolarizationSignal:
        kw["pol_type"] = "linear"
    z = cls(x, **kw)
    t0 = time.time()
    y = pb.coherent_dedispersion(z, pb.DM(dm), variant=variant, ref_freq=None if ref is None else ref * u.Hz)
    t1 = time.time()
    yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc, ref_freq_hz=ref)
    e = relerr(np.asarray(y), yr) if len(yr) else 0.0
    ok = (np.asarray(y).shape == yr.shape) and e < 1e-5
    print(f"{'OK ' if ok else 'BAD'} shape={shape} dm={dm} variant={variant} out={np.asarray(y).shape} "
          f"crop=({start},{stop}) relerr={e:.3e} gpu_call={t1-t0:.3f}s", flush=True)
    return ok

if __name__ == "__main__":
    print(_hip.lib().pbh_version(), "devices:", _hip.lib().pbh_device_count(), flush=True)
    ok = True
    # chirp kernel vs oracle
    for N in (4096, 1 << 16):
        c = pb.DM(56.77).chirp_function(N, (1 / 50e6) * u.s, 1225e6 * u.Hz, 1.4e9 * u.Hz)
        cr = orc.transfer_function(56.77, N, 1 / 50e6, 1225e6, 1.4e9)
        print("chirp", N, "max abs err", float(np.abs(c - cr).max()), flush=True)
    for n in (32, 64, 1024, 4096, 8192, 16384):
        ok &= check((n, 4, 2), 10.0 if n >= 4096 else 0.01, 1e6, 1e9)
    for n in (15, 16, 18, 19, 20):
        for variant in ("direct3", "planar5"):
            ok &= check((1 << n, 4, 2), 10.0, 1e6, 1e9, variant=variant)
    ok &= check((1 << 20, 1, 1), 0.0, 400e6, 1.4e9)
    ok &= check((1 << 18, 3, 2), 5.0, 1e6, 1e9)
    ok &= check((1 << 18, 5), 5.0, 1e6, 1e9)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
