"""Throughput of the "next" rows on device data (2^24 x 8 x 2 complex64 unless noted): time_shift, freq_shift,
incoherent_dedispersion, to_stokes, real_to_complex."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

n, nchan, npol = 1 << 24, 8, 2
x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
z = pb.DualPolarizationSignal(x, sample_rate=50 * u.MHz, center_freq=1.4 * u.GHz, pol_type="linear")
tot = n * nchan * npol
for name, fn in (("time_shift 1.5 samples", lambda: pb.time_shift(z, 1.5)),
                 ("time_shift 7 samples (integer)", lambda: pb.time_shift(z, 7)),
                 ("freq_shift 1 MHz", lambda: pb.freq_shift(z, 1 * u.MHz)),
                 ("incoherent_dedispersion DM 56.77", lambda: pb.incoherent_dedispersion(z, pb.DM(56.77))),
                 ("to_stokes", lambda: z.to_stokes()),
                 ("to_intensity", lambda: z.to_intensity()),
                 ("to_circular", lambda: z.to_circular()),
                 ("coherent_dedispersion DM 56.77", lambda: pb.coherent_dedispersion(z, pb.DM(56.77)))):
    ms = timed(fn)
    print(f"{name:36s} {ms:8.3f} ms  {tot / ms / 1e6:7.1f} Gsamples/s", flush=True)
r = pb.DeviceArray(torch.randn((1 << 25, 8), device="cuda"))
ms = timed(lambda: pb.utils.real_to_complex(r, axis=0))
print(f"{'real_to_complex (2^25 x 8 float32)':36s} {ms:8.3f} ms  {(1 << 25) * 8 / ms / 1e6:7.1f} G real samples/s", flush=True)
