"""The N>1 path on CPU: world_size-2 gloo processes shard channels, run the per-shard transform and gather.
There is no GPU here, so the workers replace the plan factory (`transforms.dedispersion._plan_for`) by one whose
plan runs the oracle -- the product entry point itself has no test hook.  The result must equal the oracle on the
full block, which pins the shard bookkeeping: channel partition, per-shard channel frequencies, full-band crop,
start_time, the gather order and the root-only gather.  The real plan runs the same entry point in
tests/test_gpu_sharded.py."""

import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import pulsarbat_amd as pb
from pulsarbat_amd import shard, units as u
from oracle import dedisp_oracle as orc

SHAPE, DM, SR, FC = (4096, 6, 2), 8.0, 1e6, 1e9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OraclePlan:
    """Stands in for _hip.Plan on the CPU box: same call, the oracle's arithmetic."""

    def __init__(self, freqs_hz, ref_hz, start, stop):
        self.freqs_hz, self.ref_hz, self.start, self.stop = freqs_hz, ref_hz, start, stop
        self.nout = stop - start

    def dedisperse(self, x):
        import scipy.fft
        chirp = np.stack([orc.transfer_function(DM, x.shape[0], 1 / SR, f, self.ref_hz) for f in self.freqs_hz], axis=1)
        y = scipy.fft.ifft(scipy.fft.fft(x, axis=0) * chirp[:, :, None], axis=0)
        return np.ascontiguousarray(y[self.start:self.stop])


def _oracle_plan_for(z, dm, ref_freq, crop, chirp=None, variant="auto", per_pol=False, dtype=None, device=None):
    assert chirp is None and not per_pol
    freqs = np.asarray(u.to_value(z.channel_freqs, u.Hz), dtype=np.float64)
    return _OraclePlan(freqs, u.to_value(ref_freq, u.Hz), crop[0], crop[1]), False


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from pulsarbat_amd.transforms import dedispersion as dd
    dd._plan_for = _oracle_plan_for          # this process only
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = orc.synthetic_block(SHAPE, 5)
    z = pb.DualPolarizationSignal(x, sample_rate=SR * u.Hz, center_freq=FC * u.Hz, pol_type="linear",
                                  start_time=pb.Time(56000.0, format="mjd"))
    zl = shard.shard_signal(z, world, rank)
    kw = dict(band_min=z.min_freq, band_max=z.max_freq, ref_freq=z.center_freq, device=0)
    full = shard.coherent_dedispersion_sharded(zl, pb.DM(DM), gather=True, **kw)
    local = shard.coherent_dedispersion_sharded(zl, pb.DM(DM), gather=False, **kw)
    at_root = shard.coherent_dedispersion_sharded(zl, pb.DM(DM), gather="root", root=1, **kw)
    assert (at_root is None) == (rank != 1)
    if at_root is not None:
        assert np.array_equal(np.asarray(at_root), np.asarray(full)) and type(at_root) is type(full)
    q.put((rank, np.asarray(full), full.channel_freqs.to_value(u.Hz),
           (full.start_time - z.start_time).to_value(u.s), np.asarray(local).shape))
    dist.barrier()
    dist.destroy_process_group()


def test_channel_slice():
    for nchan, world in ((8, 2), (64, 8), (6, 4), (5, 2), (3, 3)):
        got = []
        for r in range(world):
            sl = shard.channel_slice(nchan, world, r)
            got.extend(range(sl.start, sl.stop))
        assert got == list(range(nchan))
    with pytest.raises(ValueError):
        shard.channel_slice(8, 2, 2)


def test_sharded_equals_full_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = orc.synthetic_block(SHAPE, 5)
    want, start, stop = orc.coherent_dedispersion(x, DM, SR, FC)
    fwant = orc.channel_freqs(FC, SR, SHAPE[1])
    for rank, full, freqs, dt, lshape in results:
        assert full.shape == want.shape
        assert np.allclose(full, want, rtol=0, atol=1e-6)
        assert np.allclose(freqs, fwant)
        assert abs(dt - start / SR) < 1e-12
        assert lshape == (stop - start, SHAPE[1] // world, SHAPE[2])


def _scatter_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1)
    counts = [3, 2]                                                          # ragged channel blocks
    full = (rng.standard_normal((256, 5, 2)) + 1j * rng.standard_normal((256, 5, 2))).astype(np.complex64)
    shared = full[:, :1]                                                     # one row for every channel: a broadcast
    got = shard._scatter_chirp(full if rank == 1 else None, 1, counts, rank, None, 0)
    got1 = shard._scatter_chirp(shared if rank == 0 else None, 0, counts, rank, None, 0)
    vec = shard._scatter_chirp(full[:, 0, 0] if rank == 0 else None, 0, counts, rank, None, 0)   # (N,) -> (N, 1)
    q.put((rank, got, got1, vec))
    dist.barrier()
    dist.destroy_process_group()


def test_user_chirp_scatter_world2():
    """X1 on the host channel: rank 1 holds the full-band chirp, every rank receives its (ragged) channel block; a chirp
    with one channel row is broadcast whole (reference dedispersion.py:121-124, sharded)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(1)
    full = (rng.standard_normal((256, 5, 2)) + 1j * rng.standard_normal((256, 5, 2))).astype(np.complex64)
    for rank, got, got1, vec in results:
        lo, hi = (0, 3) if rank == 0 else (3, 5)
        assert got.shape == (256, hi - lo, 2) and np.array_equal(got, full[:, lo:hi])
        assert np.array_equal(got1, full[:, :1])
        assert vec.shape == (256, 1) and np.array_equal(vec[:, 0], full[:, 0, 0])
