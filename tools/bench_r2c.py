"""utils.real_to_complex on device float32 data: the half-length transform (pbh_real_to_complex) against the full-length route."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd.utils import real_to_complex


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for shape in ((1 << 25, 8), (1 << 25, 16), (1 << 22, 64), (1 << 25, 1)):
    x = pb.DeviceArray(torch.randn(shape, device="cuda"))
    os.environ["PBH_R2C_HALF"] = "1"
    a = timed(lambda: real_to_complex(x))
    os.environ["PBH_R2C_HALF"] = "0"
    b = timed(lambda: real_to_complex(x))
    n = np.prod(shape)
    print(f"{shape}: half-length {a:6.3f} ms ({n / a / 1e6:6.1f} G real samples/s) | full-length {b:6.3f} ms", flush=True)
    del x
    pb.transforms.dedispersion.clear_plan_cache()
