import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
x = pb.DeviceArray(torch.view_as_complex(torch.randn((1 << 24, 8, 2, 2), device="cuda")))
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("to_series_major", timed(lambda: x.to_series_major()), "ms")
xs = x.to_series_major()
print("contiguous (back)", timed(lambda: xs.contiguous()), "ms")
