"""contrib.stft / istft throughput on device arrays: (2^26, nchan, npol) complex64, by nperseg."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import _hip

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for nchan, npol in ((1, 2), (4, 2), (1, 1)):
    n = (1 << 27) // (nchan * npol)
    x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
    for nper in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 1 << 16):
        ms = timed(lambda: _hip.stft(x, nper))
        y = _hip.stft(x, nper)
        ms2 = timed(lambda: _hip.stft(y, nper, inverse=True))
        print(f"nchan {nchan} npol {npol} nperseg {nper:6d}: stft {ms:.3f} ms {2 * x.nbytes / ms / 1e6:6.0f} GB/s   istft {ms2:.3f} ms {2 * x.nbytes / ms2 / 1e6:6.0f} GB/s", flush=True)
        del y
