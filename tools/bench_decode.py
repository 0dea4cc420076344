"""Reader-side decode throughput (pbh_decode): a GUPPI-like 8-bit block [chan][time][pol][re,im] of the config-2
shape, device-resident raw bytes -> complex64, and the same from host memory (PCIe-inclusive) beside the upload of
the already-unpacked complex64 array that the reference's host-side readers would need."""
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

n, nchan, npol = 1 << 24, 8, 2
rng = np.random.default_rng(0)
raw = rng.integers(0, 256, n * nchan * npol * 2, dtype=np.uint8)
draw = pb.DeviceArray.from_host(raw)
lays = {"guppi [chan][time][pol]": dict(stride_t=npol, stride_c=n * npol, stride_p=1),
        "dada  [time][chan][pol]": dict(stride_t=nchan * npol, stride_c=npol, stride_p=1)}
for name, st in lays.items():
    lay = dict(nbits=8, ncomp=2, code=0, blk_samples=n, blk_stride=raw.size, hdr_bytes=0, elem0=0, **st)
    for sm in (True, False):
        ms = timed(lambda: _hip.decode(draw, lay, 0, n, nchan, npol, series_major=sm))
        tot = n * nchan * npol
        print(f"{name} -> {'series-major' if sm else 'sample-major'} complex64, device raw: {ms:.3f} ms "
              f"{tot / ms / 1e6:.1f} Gsamples/s  {tot * 10 / ms / 1e6:.0f} GB/s (2 B in + 8 B out)", flush=True)
lay = dict(nbits=8, ncomp=2, code=0, blk_samples=n, blk_stride=raw.size, hdr_bytes=0, elem0=0, **lays["guppi [chan][time][pol]"])
ms = timed(lambda: _hip.decode(raw, lay, 0, n, nchan, npol, series_major=True), reps=3)
print(f"host raw int8 -> device complex64 (PCIe-inclusive): {ms:.1f} ms  {n * nchan * npol / ms / 1e6:.2f} Gsamples/s", flush=True)
z = np.zeros((n, nchan, npol), np.complex64)
ms = timed(lambda: pb.DeviceArray.from_host(z), reps=3)
print(f"host complex64 -> device (what a host-side unpack has to upload): {ms:.1f} ms  {n * nchan * npol / ms / 1e6:.2f} Gsamples/s", flush=True)
