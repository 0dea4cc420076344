"""stft -> coherent_dedispersion: the fused call (pbh_stft_dedisperse) against the two calls, device-resident data."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import units as u


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n, nchan, npol = 1 << 24, 8, 2
x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
z = pb.DualPolarizationSignal(x, sample_rate=50 * u.MHz, center_freq=1.4 * u.GHz, pol_type="linear")
tot = n * nchan * npol
for nperseg in (32, 64, 128, 256, 512, 1024):
    dm = pb.DM(56.77)
    a = timed(lambda: pb.contrib.stft(z, nperseg=nperseg))
    b = timed(lambda: pb.coherent_dedispersion(pb.contrib.stft(z, nperseg=nperseg), dm))
    c = timed(lambda: pb.contrib.stft_dedisperse(z, dm, nperseg=nperseg))
    print(f"nperseg {nperseg:5d}: stft {a:6.3f} ms | stft + dedispersion {b:6.3f} ms | fused {c:6.3f} ms "
          f"({tot / c / 1e6:5.1f} Gsamples/s)", flush=True)

# the way back: coherent_dedispersion -> istft, fused (pbh_dedisperse_istft) against the two calls
print("--- coherent_dedispersion -> istft")
for nperseg in (32, 64, 128, 256, 512, 1024):
    dm = pb.DM(56.77)
    zc = pb.contrib.stft(z, nperseg=nperseg)
    a = timed(lambda: pb.coherent_dedispersion(zc, dm))
    b = timed(lambda: pb.contrib.istft(pb.coherent_dedispersion(zc, dm), nperseg=nperseg))
    c = timed(lambda: pb.contrib.dedisperse_istft(zc, dm, nperseg=nperseg))
    print(f"nperseg {nperseg:5d}: dedispersion {a:6.3f} ms | dedispersion + istft {b:6.3f} ms | fused {c:6.3f} ms "
          f"({tot / c / 1e6:5.1f} Gsamples/s)", flush=True)
    del zc
