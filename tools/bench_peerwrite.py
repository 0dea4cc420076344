"""Cross-PROCESS writes on one GPU: a child process maps the parent's buffer -- hipIpc handle (PeerBuffer) or file descriptor
(SharedPeer) -- and times a fill and a pitched dedispersion into it."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import multiprocessing as mp
from multiprocessing import reduction


def child(kind, token, nbytes, shape, q):
    import torch
    import pulsarbat_amd as pb
    from pulsarbat_amd import _hip
    from pulsarbat_amd.node import PeerBuffer, SharedPeer, _Cai
    import os
    if kind == "ipc":
        peer = PeerBuffer(token, 0)
    else:
        fd = token.detach(); peer = SharedPeer(fd, nbytes, 0); os.close(fd)
    t = torch.as_tensor(_Cai(peer.ptr, shape, np.complex64, peer), device="cuda:0")

    def timed(fn, reps=5):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    n, nchan, npol, total = 1 << 22, 8, 2, 16
    start, stop = 352101, 3651808
    plan = _hip.Plan(n, nchan, npol, start, stop)
    freqs = 1.4e9 + 25e6 * (np.arange(total) + 0.5 - total / 2)
    plan.chirp_generate(56.77 / 2.41e-4 * 1e12, 1 / 25e6, freqs[:nchan], 1.4e9)
    x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
    y = pb.DeviceArray.empty((plan.nout, nchan, npol), np.complex64)
    nb = t.numel() * 8 / 1e9
    a = timed(lambda: t.fill_(1.0))
    d = timed(lambda: plan.dedisperse_slices(x, [peer.ptr], [0, plan.nout], total * npol, 0))
    e = timed(lambda: plan.dedisperse(x, out=y))
    q.put(f"{kind}: fill {a:6.3f} ms ({nb / a * 1e3:6.0f} GB/s) | dedisperse into the peer's block {d:6.3f} ms | into a local array {e:6.3f} ms")
    del t; peer.close()


if __name__ == "__main__":
    import torch
    from pulsarbat_amd.node import NodeBuffer, SharedBuffer
    n, total, npol = 1 << 22, 16, 2
    nout = 3651808 - 352101
    shape = (nout, total, npol)
    ctx = mp.get_context("spawn")
    for kind in ("ipc", "fd"):
        buf = (NodeBuffer if kind == "ipc" else SharedBuffer)(shape, np.complex64, 0)
        token = buf.handle() if kind == "ipc" else reduction.DupFd(buf.fd)
        q = ctx.Queue()
        p = ctx.Process(target=child, args=(kind, token, buf.nbytes, shape, q))
        p.start()
        print(q.get(timeout=200), flush=True)
        p.join(60)
        buf.close()
