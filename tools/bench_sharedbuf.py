"""Writes into / reads out of a SharedBuffer (hipMemCreate + file descriptor) against an ordinary allocation: fill, clone,
and the dedispersion's pitched last pass (pbh_dedisperse_slices) into each."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import _hip
from pulsarbat_amd.node import SharedBuffer, NodeBuffer


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n, nchan, npol, total = 1 << 24, 8, 2, 16
start, stop = 1408404, 14607231
plan = _hip.Plan(n, nchan, npol, start, stop)
freqs = 1.4e9 + 25e6 * (np.arange(total) + 0.5 - total / 2)
plan.chirp_generate(56.77 / 2.41e-4 * 1e12, 1 / 25e6, freqs[:nchan], 1.4e9)
x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
shape = (plan.nout, total, npol)
for name, cls in (("hipMalloc (NodeBuffer)", NodeBuffer), ("hipMemCreate (SharedBuffer)", SharedBuffer)):
    shp = shape if cls is SharedBuffer else (plan.nout // 4, total, npol)   # NodeBuffer: <= 2040 MiB
    b = cls(shp, np.complex64, 0)
    t = b.array.tensor
    nb = t.numel() * 8 / 1e9
    a = timed(lambda: t.fill_(1.0))
    c = timed(lambda: t.clone())
    if cls is SharedBuffer:
        d = timed(lambda: plan.dedisperse_slices(x, [b.ptr], [0, plan.nout], total * npol, 0))
    else:
        q = plan.nout // 4
        bs = [cls(shp, np.complex64, 0) for _ in range(3)] + [b]
        d = timed(lambda: plan.dedisperse_slices(x, [k.ptr for k in bs], [0, q, 2 * q, 3 * q, plan.nout - (plan.nout - 4 * q)], total * npol, 0)) if False else float("nan")
    print(f"{name:30s} {nb:5.2f} GB: fill {a:6.3f} ms ({nb / a * 1e3:6.0f} GB/s) | clone {c:6.3f} ms ({2 * nb / c * 1e3:6.0f} GB/s) | dedisperse into it {d:6.3f} ms", flush=True)
y = pb.DeviceArray.empty((plan.nout, nchan, npol), np.complex64)
print(f"dedisperse into a compact torch array: {timed(lambda: plan.dedisperse(x, out=y)):6.3f} ms")
full = torch.empty(shape, dtype=torch.complex64, device="cuda")
print(f"dedisperse_slice into a torch (hipMalloc) full-band array: {timed(lambda: plan.dedisperse_slice(x, full.data_ptr(), total * npol, 0)):6.3f} ms")

# the same physical allocation through a mapping made from its DESCRIPTOR (what a peer rank has) -- here in the same process
import os
from pulsarbat_amd.node import SharedPeer, _Cai
b = SharedBuffer(shape, np.complex64, 0)
peer = SharedPeer(b.fd, b.nbytes, 0)
tp = torch.as_tensor(_Cai(peer.ptr, shape, np.complex64, peer), device="cuda:0")
nb = tp.numel() * 8 / 1e9
a = timed(lambda: tp.fill_(1.0))
c = timed(lambda: tp.clone())
d = timed(lambda: plan.dedisperse_slices(x, [peer.ptr], [0, plan.nout], total * npol, 0))
print(f"{'imported mapping (SharedPeer)':30s} {nb:5.2f} GB: fill {a:6.3f} ms ({nb / a * 1e3:6.0f} GB/s) | clone {c:6.3f} ms ({2 * nb / c * 1e3:6.0f} GB/s) | dedisperse into it {d:6.3f} ms", flush=True)
