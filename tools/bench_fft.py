"""Time pb.fft.fft on device arrays (n, batch).  PBH_NATIVE_FFT=0 selects the Bluestein ring for comparison."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
import torch

def run(n, batch, dtype=np.complex64, reps=5):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((n, batch), dtype=np.float32) + 1j * rng.standard_normal((n, batch), dtype=np.float32)).astype(dtype)
    d = pb.DeviceArray.from_host(x)
    for _ in range(3):   # (the first calls of a new size pay for the plan and for torch's allocator growing its pool)
        y = pb.fft.fft(d, axis=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        y = pb.fft.fft(d, axis=0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"n={n} batch={batch} {np.dtype(dtype).name}: {ms:.2f} ms  {n * batch / ms / 1e6:.1f} Gsamples/s", flush=True)

if __name__ == "__main__":
    for n, b in ((1 << 24, 16), (1 << 20, 64), (1 << 26, 4), (3 << 22, 16), (1 << 16, 1024)):
        run(n, b)
    run(1 << 22, 16, np.complex128)
