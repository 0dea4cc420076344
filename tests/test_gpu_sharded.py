"""GPU tests of the channel-sharded path with the REAL plan and of BASELINE.json's
configs at full size against the oracle.

* configs[1] (2^24 x 8 x 2, DM 56.77): ALL 16 series against the oracle.
* configs[2]: one rank's share of the 64-channel block (8 of 64 channels x 6.25 MHz, full-band crop
  [176051, 16505967)) through coherent_dedispersion_sharded on a one-rank process group.
* configs[4]: the same share at DM 1000 with the fused Stokes-I + 1024x scrunch tail -> (8689, 8).
* the sharded HIP branch at world = 1 (two shards run one after the other on the one GPU) and at world = 2
  (two processes on cuda:0, gloo for the host channel, real pbh_node_* IPC mappings between the processes).
Reference: Dask chunking over the non-time axes (pulsarbat/core.py:332-345), Signal.compute() (core.py:298-309),
crop from the full band's edges (dedispersion.py:127-133), user chirps (dedispersion.py:121-125).
"""

import os
import socket

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import shard, units as u
from oracle import dedisp_oracle as orc
from tests._detect_check import assert_detect_close

pytestmark = pytest.mark.gpu

RTOL_L2 = 1e-5      # relative L2 per series (BASELINE.json north_star)
NCPU = os.cpu_count() or 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def one_rank_group():
    """A one-rank process group (gloo carries the host-side objects; the data never leaves the GPU)."""
    import torch.distributed as dist
    if dist.is_initialized():
        yield None
        return
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield None
    dist.destroy_process_group()


def oracle_shard(x, dm, sr, freqs, ref, start, stop):
    """The reference expression (dedispersion.py:125) on a shard: per-channel chirps for the shard's
    frequencies, crop from the FULL band."""
    import scipy.fft
    n = x.shape[0]
    chirp = np.stack([orc.transfer_function(dm, n, 1 / sr, f, ref) for f in freqs], axis=1)
    chirp = chirp.reshape(chirp.shape + (1,) * (x.ndim - 2))
    y = scipy.fft.ifft(scipy.fft.fft(x, axis=0, workers=NCPU) * chirp, axis=0, workers=NCPU)
    return y[start:stop]


def per_series_l2(got, ref):
    got = np.asarray(got).reshape(ref.shape[0], -1)
    ref = ref.reshape(ref.shape[0], -1)
    return np.linalg.norm(got - ref, axis=0) / np.linalg.norm(ref, axis=0)


def device_block(shape, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = torch.randn(tuple(shape) + (2,), generator=g, device="cuda", dtype=torch.float32) * 2 ** -0.5
    return torch.view_as_complex(t)


def test_config1_full_size_all_series():
    """BASELINE configs[1] at full size: every one of the 16 series against the oracle (a wrong channel -> chirp
    row mapping at N1 = 1024 shows here and nowhere else)."""
    n, nchan, npol, sr, fc, dm = 1 << 24, 8, 2, 50e6, 1.4e9, 56.77
    xt = device_block((n, nchan, npol), 20260002)
    z = pb.DualPolarizationSignal(pb.DeviceArray(xt), sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
    y = pb.coherent_dedispersion(z, pb.DM(dm))
    assert y.shape == (14607231 - 1408404, nchan, npol)
    x = xt.cpu().numpy()
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc, workers=NCPU)
    assert (start, stop) == (1408404, 14607231)
    err = per_series_l2(y, want)
    assert err.shape == (16,) and err.max() < RTOL_L2, f"per-series relative L2: {err}"


@pytest.mark.parametrize("n", [625 << 14, 10_000_000, 10_935_000, 13_671_875])
def test_7smooth_lengths_full_size_all_series(n):
    """configs[1]'s block at the lengths the reference's fast_len would crop it to (utils.py:68-130): 625 * 2^14 (one
    mixed-radix column level), 10^7 (two levels), 2^3 3^7 5^4 and the odd 5^9 7 (mixed-radix rows too) -- every one of the
    16 series against the oracle, same band and DM as the headline."""
    nchan, npol, sr, fc, dm = 8, 2, 50e6, 1.4e9, 56.77
    xt = device_block((n, nchan, npol), 20260003 + n % 1000)
    z = pb.DualPolarizationSignal(pb.DeviceArray(xt), sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
    y = pb.coherent_dedispersion(z, pb.DM(dm))
    x = xt.cpu().numpy()
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc, workers=NCPU)
    assert y.shape == want.shape == (stop - start, nchan, npol)
    err = per_series_l2(y, want)
    assert err.shape == (16,) and err.max() < RTOL_L2, f"per-series relative L2: {err}"


@pytest.mark.parametrize("rank", [0, 3, 5, 7])
def test_config2_rank_share_full_size(one_rank_group, rank):
    """BASELINE configs[2]: 64 channels of 6.25 MHz over 8 GPUs; this is rank `rank`'s share (8 channels x 2 pol x
    2^24) through the sharded entry point with the real plan, full-band crop [176051, 16505967)."""
    n, nchan_total, npol, fc, dm, world = 1 << 24, 64, 2, 1.4e9, 56.77, 8
    sr = 400e6 / nchan_total
    sl = shard.channel_slice(nchan_total, world, rank)
    freqs_all = orc.channel_freqs(fc, sr, nchan_total)
    freqs = freqs_all[sl]
    xt = device_block((n, sl.stop - sl.start, npol), 20260003 + rank)
    zl = pb.DualPolarizationSignal(pb.DeviceArray(xt), sample_rate=sr * u.Hz, center_freq=float(freqs.mean()) * u.Hz,
                                   pol_type="linear", start_time=pb.Time(56000.0, format="mjd"))
    assert np.allclose(zl.channel_freqs.to_value(u.Hz), freqs)
    lo, hi = (fc - 200e6) * u.Hz, (fc + 200e6) * u.Hz
    y = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), band_min=lo, band_max=hi, ref_freq=fc * u.Hz)
    assert y.shape == (16505967 - 176051, 8, 2)
    assert abs((y.start_time - zl.start_time).to_value(u.s) - 176051 / sr) < 1e-12
    want = oracle_shard(xt.cpu().numpy(), dm, sr, freqs, fc, 176051, 16505967)
    err = per_series_l2(y, want)
    assert err.max() < RTOL_L2, f"per-series relative L2: {err}"


@pytest.mark.parametrize("rank", [2, 7])
def test_config4_rank_share_full_size(one_rank_group, rank):
    """BASELINE configs[4]: the configs[2] geometry at DM 1000 with the fused Stokes-I detect + 1024x time scrunch
    (crop [3101118, 11999198), 8 898 080 valid samples -> (8689, nchan); one rank's 8 channels)."""
    n, nchan_total, npol, fc, dm, world = 1 << 24, 64, 2, 1.4e9, 1000.0, 8
    sr = 400e6 / nchan_total
    sl = shard.channel_slice(nchan_total, world, rank)
    freqs = orc.channel_freqs(fc, sr, nchan_total)[sl]
    xt = device_block((n, 8, npol), 20260005 + rank)
    zl = pb.DualPolarizationSignal(pb.DeviceArray(xt), sample_rate=sr * u.Hz, center_freq=float(freqs.mean()) * u.Hz,
                                   pol_type="linear")
    lo, hi = (fc - 200e6) * u.Hz, (fc + 200e6) * u.Hz
    got, start = shard.dedisperse_detect_sharded(zl, pb.DM(dm), band_min=lo, band_max=hi, ref_freq=fc * u.Hz,
                                                 mode="I", nscrunch=1024, gather=True)
    assert start == 3101118 and got.shape == (8689, 8) and got.dtype == np.float32
    yr = oracle_shard(xt.cpu().numpy(), dm, sr, freqs, fc, 3101118, 11999198)
    assert_detect_close(got, yr, "I", 1024)   # float64 sums of the oracle's voltages, 1e-5 of every output's own Stokes I


def _small_case():
    shape, dm, sr, fc = (1 << 18, 8, 2), 30.0, 4e6, 1.2e9
    return shape, dm, sr, fc


def test_sharded_hip_two_shards_one_after_the_other(one_rank_group):
    """The sharded HIP branch (shard.py, real plan) on device data: the two shards of a 2-way split are run one after
    the other on the one GPU; their concatenation must equal the oracle on the full block.  gather=True / "root" on
    the one-rank group go through ChannelGather (pbh_node_alloc + pbh_dedisperse_slice)."""
    shape, dm, sr, fc = _small_case()
    x = orc.synthetic_block(shape, 31)
    z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear",
                                  start_time=pb.Time(56000.0, format="mjd")).to_device()
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    parts = []
    for r in range(2):
        zl = shard.shard_signal(z, 2, r)
        for g in (False, True, "root"):
            y = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), band_min=z.min_freq, band_max=z.max_freq,
                                                    ref_freq=z.center_freq, gather=g)
            assert isinstance(y.data, pb.DeviceArray) and y.shape == (stop - start, 4, 2)
            assert abs((y.start_time - z.start_time).to_value(u.s) - start / sr) < 1e-12
            sl = shard.channel_slice(8, 2, r)
            assert per_series_l2(y, want[:, sl]).max() < RTOL_L2
        parts.append(np.asarray(y))
    assert per_series_l2(np.concatenate(parts, axis=1), want).max() < RTOL_L2
    # host-resident shard: computed on the process's current device, result back on the host
    zl = shard.shard_signal(pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz,
                                                      pol_type="linear"), 2, 1)
    y = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), band_min=z.min_freq, band_max=z.max_freq,
                                            ref_freq=z.center_freq)
    assert isinstance(y.data, np.ndarray)
    assert per_series_l2(y, want[:, 4:]).max() < RTOL_L2


def _world2_worker(rank, world, port, q, nchan):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # both ranks share the one GPU of the test box (the mappings are real IPC all the same); on a multi-GPU node
    # tools/first_8gpu.sh sets PBH_TEST_ONE_DEVICE_PER_RANK=1: rank r computes on device r, the peer writes cross xGMI
    dev = rank if os.environ.get("PBH_TEST_ONE_DEVICE_PER_RANK") == "1" and torch.cuda.device_count() > rank else 0
    torch.cuda.set_device(dev)
    if nchan == 8:
        os.environ["PBH_GATHER_CHUNK_BYTES"] = str(3 << 20)   # the destination blocks become a dozen row-chunks
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape, dm, sr, fc = _small_case()
        shape = (shape[0], nchan, 2)
        x = orc.synthetic_block(shape, 41)
        z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear",
                                      start_time=pb.Time(56000.0, format="mjd"))
        zl = shard.shard_signal(z, world, rank).to_device()
        kw = dict(band_min=z.min_freq, band_max=z.max_freq, ref_freq=z.center_freq)
        full = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), gather=True, **kw)
        root = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), gather="root", root=1, **kw)
        # user chirp held by rank 1 only, scattered by channel (the oracle's chirp: results must agree with `full`)
        chirp = orc.chirp_from_signal(dm, shape, sr, fc) if rank == 1 else None
        viac = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), gather=True, chirp=chirp, chirp_src=1, **kw)
        # a stream of blocks through the CACHED gather (set up once, double-buffered chunks), issued back to back with no
        # synchronisation in between: every result must be its own block's, none torn by the next run's peer writes
        ngather = len(shard._GATHERS)
        outs = []
        for k in range(5):
            zk = type(zl).like(zl, pb.DeviceArray(zl.data.tensor * complex(k + 2, -k)))
            outs.append(shard.coherent_dedispersion_sharded(zk, pb.DM(dm), gather=True, **kw))
        assert len(shard._GATHERS) == ngather
        ref = np.asarray(full)
        stream_err = max(float(np.abs(np.asarray(o) - ref * complex(k + 2, -k)).max() / np.abs(ref).max())
                         for k, o in enumerate(outs))
        # one rank fails inside a run (its plan does not fit the gather): BOTH ranks get GatherError, nobody hangs,
        # and close() still pairs up
        from pulsarbat_amd import _hip
        from pulsarbat_amd.node import ChannelGather, GatherError
        from pulsarbat_amd.transforms.dedispersion import _plan_for
        start, stop = shard._full_band_crop(pb.DM(dm), len(zl), zl.sample_rate, z.min_freq, z.max_freq, z.center_freq)
        plan, _ = _plan_for(zl, pb.DM(dm), z.center_freq, (start, stop), device=dev)
        g = ChannelGather(plan.nout, zl.nchan, 2, np.complex64, dev, mode="all")
        failed = ""
        try:
            if rank == 0:
                with _hip.Plan(len(zl), zl.nchan, 2, start, stop - 32, device=dev) as bad:
                    g.run(bad, zl.data.contiguous())
            else:
                g.run(plan, zl.data.contiguous())
        except GatherError as exc:
            failed = str(exc)
        again = g.run(plan, zl.data.contiguous())        # the gather is still usable afterwards
        g.close()
        shard.release_gathers()
        q.put((rank, ref, full.channel_freqs.to_value(u.Hz),
               None if root is None else np.asarray(root), np.asarray(viac), stream_err, failed, np.asarray(again)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nchan", [8, 6, 5])
def test_sharded_hip_world2_on_one_gpu(nchan):
    """Two processes, each a rank with its own plan, on the one GPU: the full-band block of every destination rank is
    written by both ranks' pipelines (own slice locally, the peer's through a pbh_node_import mapping).  8 channels:
    the last kernel writes the pitched rows itself; 6: compact result + placing pass; 5: ragged shards (3 + 2)."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world2_worker, args=(r, world, port, q, nchan)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    shape, dm, sr, fc = _small_case()
    shape = (shape[0], nchan, 2)
    x = orc.synthetic_block(shape, 41)
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    for rank, full, freqs, root, viac, stream_err, failed, again in results:
        assert full.shape == want.shape
        assert per_series_l2(full, want).max() < RTOL_L2
        assert np.allclose(freqs, orc.channel_freqs(fc, sr, nchan))
        assert (root is None) == (rank != 1)
        if root is not None:
            assert per_series_l2(root, want).max() < RTOL_L2
        assert per_series_l2(viac, want).max() < RTOL_L2
        assert stream_err < 1e-5, f"rank {rank}: a result of the back-to-back stream differs ({stream_err:.2e})"
        assert "gather run failed" in failed and (("rank 0" in failed) == (rank == 0))
        assert np.array_equal(again, full)


def _shared_child(dupfd, nbytes, q):
    import os
    import torch
    from pulsarbat_amd.node import SharedPeer, _Cai
    fd = dupfd.detach()                      # the descriptor arrives over a Unix socket (multiprocessing's resource sharer)
    peer = SharedPeer(fd, nbytes, 0)
    os.close(fd)
    t = torch.as_tensor(_Cai(peer.ptr, (nbytes // 4,), np.float32, peer), device="cuda:0")
    seen = float(t[::4097].double().sum().item())
    t.mul_(2.0)
    torch.cuda.synchronize()
    q.put(seen)
    del t
    peer.close()


def test_shared_buffer_beyond_two_gib():
    """A destination block of the gather is one contiguous `SharedBuffer` (hipMemCreate + a POSIX file descriptor): 2.5 GiB
    mapped by a second process, which reads the owner's values and writes its own -- a size at which hipIpcOpenMemHandle
    never returns on this stack (tools/ipc_probe.py), hence the <= 1-GiB row-chunks of the older form."""
    import multiprocessing as mp
    from multiprocessing import reduction
    import torch
    from pulsarbat_amd.node import SharedBuffer
    n = (5 << 29) // 4
    buf = SharedBuffer((n,), np.float32, 0)
    t = buf.array.tensor
    t.copy_(torch.arange(n, device="cuda", dtype=torch.float32) % 1000)
    torch.cuda.synchronize()
    want = float(t[::4097].double().sum().item())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_shared_child, args=(reduction.DupFd(buf.fd), buf.nbytes, q))
    p.start()
    seen = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0 and seen == want
    torch.cuda.synchronize()
    assert float(t[::4097].double().sum().item()) == 2 * want      # the peer's writes are visible to the owner
    del t
    buf.close()


def _rccl_world1_worker(port, q):
    """Every RCCL line of the sharded path, once, on a one-rank "nccl" group (the first GPU call of this process is the
    process group's own)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.cuda.set_device(0)
        shape, dm, sr, fc = _small_case()
        x = orc.synthetic_block(shape, 43)
        z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear",
                                      start_time=pb.Time(56000.0, format="mjd"))
        zl = shard.shard_signal(z, 1, 0).to_device()
        kw = dict(band_min=z.min_freq, band_max=z.max_freq, ref_freq=z.center_freq)
        out = {"backend": dist.get_backend()}
        # X1 over RCCL: a DEVICE full-band chirp on rank 0, scattered by channel (dist.scatter of device tensors), then the
        # gather by peer writes with its status all-reduces and object collectives over nccl
        chirp = pb.DeviceArray.from_host(orc.chirp_from_signal(dm, shape, sr, fc))
        y = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), chirp=chirp, chirp_src=0, gather="all", **kw)
        out["scatter_all"] = np.asarray(y)
        y = shard.coherent_dedispersion_sharded(zl, pb.DM(dm), gather="root", **kw)
        out["root"] = np.asarray(y)
        # a chirp with ONE channel row: the broadcast branch (one-channel band so that the row is the right chirp)
        x1 = np.ascontiguousarray(x[:, :1])
        z1 = pb.DualPolarizationSignal(x1, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear").to_device()
        kw1 = dict(band_min=z1.min_freq, band_max=z1.max_freq, ref_freq=z1.center_freq)
        row = pb.DeviceArray.from_host(orc.chirp_from_signal(dm, x1.shape, sr, fc).reshape(shape[0], 1))
        y = shard.coherent_dedispersion_sharded(z1, pb.DM(dm), chirp=row, chirp_src=0, gather=True, **kw1)
        out["bcast"] = np.asarray(y)
        # configs[4]'s exchange: the detected output gathered with a device all_gather
        d, start = shard.dedisperse_detect_sharded(zl, pb.DM(dm), mode="I", nscrunch=64, gather=True, **kw)
        out["detect"], out["start"] = np.asarray(d), start
        shard.release_gathers()
        dist.barrier()
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_lines_at_world1():
    """`shard._scatter_chirp`'s device branch (RCCL scatter, and the single-row broadcast), `ChannelGather`'s collectives
    and `dedisperse_detect_sharded`'s device all-gather on an "nccl" process group of one rank, all against the oracle:
    the first multi-GPU run must not be the first time these lines execute (core.py:298-309, 332-345 are the reference's
    compute()/chunking they replace)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and out["backend"] == "nccl"
    shape, dm, sr, fc = _small_case()
    x = orc.synthetic_block(shape, 43)
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    assert per_series_l2(out["scatter_all"], want).max() < RTOL_L2
    assert per_series_l2(out["root"], want).max() < RTOL_L2
    want1 = orc.coherent_dedispersion(np.ascontiguousarray(x[:, :1]), dm, sr, fc)[0]
    assert per_series_l2(out["bcast"], want1).max() < RTOL_L2
    wd = orc.scrunch(orc.to_stokes(want, "linear")[:, :, 0], 64)
    assert out["start"] == start and out["detect"].shape == wd.shape
    assert_detect_close(out["detect"], want, "I", 64)


@pytest.mark.parametrize("shape,dtype,total,first", [
    ((1 << 16, 8, 2), np.complex64, 24, 8),     # row transposes write the pitched rows themselves
    ((1 << 16, 4, 2), np.complex64, 7, 3),      # 8 series at element offset 6, pitch 14
    ((1 << 16, 8, 1), np.complex64, 11, 3),     # odd offset and pitch: no 16-byte vectors, generic kernel
    ((1 << 16, 3, 2), np.complex64, 5, 2),      # two-axis layout tiles: compact result + placing pass
    ((4096, 2, 2), np.complex64, 6, 4),         # single-tile plan
    ((1 << 15, 2, 2), np.complex128, 4, 1),     # float64 build
    ((3000, 2, 2), np.complex64, 3, 1),         # arbitrary length (convolution plan)
])
def test_dedisperse_slice_geometries(shape, dtype, total, first):
    """pbh_dedisperse_slice: the (nout, nchan, npol) result lands in channels [first, first + nchan) of a wider
    (nout, total, npol) array, the rest of which is left untouched."""
    import torch
    from pulsarbat_amd import _hip
    n, nchan, npol = shape
    dm, sr, fc = 5.0, 1e6, 1e9
    x = orc.synthetic_block(shape, 77).astype(dtype)
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    freqs = orc.channel_freqs(fc, sr, nchan)
    with _hip.Plan(n, nchan, npol, start, stop, device=0, dtype=dtype) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, fc)
        tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
        full = torch.full((stop - start, total, npol), -7.0 + 3.0j, dtype=tdt, device="cuda")
        plan.dedisperse_slice(pb.DeviceArray.from_host(x), full.data_ptr(), total * npol, first * npol)
        torch.cuda.synchronize()
    got = full.cpu().numpy()
    tol = RTOL_L2 if dtype == np.complex64 else 1e-9
    assert per_series_l2(got[:, first:first + nchan], want).max() < tol
    rest = np.delete(got, np.s_[first:first + nchan], axis=1)
    assert np.all(rest == np.complex64(-7.0 + 3.0j))


class TestUserChirpGenerality:
    """dedispersion.py:121-125: the chirp is multiplied in unchecked, so any array that broadcasts against z.data
    works in the reference; mirrored from tests/test_dedispersion.py:141-164 (2-D and 3-D precomputed chirps)."""

    @pytest.mark.parametrize("dm", [10, 20, 50])
    def test_precomputed_chirp_2d_3d(self, dm):
        shape, fcen, sr = (8192, 4, 2), 1e9, 1e6
        x = orc.synthetic_block(shape, 3)
        z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fcen * u.Hz, pol_type="linear")
        DM = pb.DM(dm)
        y1 = pb.coherent_dedispersion(z, DM)
        chirp = DM.chirp_from_signal(z)
        assert chirp.shape == (8192, 4, 1)
        for c in (chirp, np.asarray(chirp)[:, :, 0]):
            y2 = pb.coherent_dedispersion(z, DM, chirp=c)
            assert np.allclose(np.asarray(y1), np.asarray(y2), atol=2e-6)

    def test_per_pol_and_shared_chirps(self):
        shape, sr, fc, dm = (1 << 15, 3, 2), 1e6, 1e9, 12.0
        x = orc.synthetic_block(shape, 9)
        z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
        rng = np.random.default_rng(5)
        base = orc.chirp_from_signal(dm, shape, sr, fc)                      # (N, 3, 1)
        perpol = (base * np.exp(2j * np.pi * rng.random((1, 3, 2)))).astype(np.complex64)   # differs between pols
        shared = base[:, 1:2]                                                # (N, 1, 1): one row for every channel
        vec = base[:, 0, 0]                                                   # (N,): axes are appended on the right
        for c in (perpol, shared, vec, np.complex64(0.5 - 0.25j)):
            want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc, chirp=np.asarray(c) if np.ndim(c) else
                                                          np.full((1, 1, 1), c))
            for zz in (z, z.to_device()):
                y = pb.coherent_dedispersion(zz, pb.DM(dm), chirp=c)
                assert y.shape == want.shape and y.dtype == np.complex64
                assert per_series_l2(y, want).max() < RTOL_L2
        got, s0 = pb.dedisperse_detect(z, pb.DM(dm), chirp=perpol, mode="I", nscrunch=64)
        yr, _, _ = orc.coherent_dedispersion(x, dm, sr, fc, chirp=perpol)
        assert_detect_close(got, yr, "I", 64)
        with pytest.raises(ValueError):
            pb.coherent_dedispersion(z, pb.DM(dm), chirp=np.ones((shape[0], 2), np.complex64))

    def test_complex128_chirp_keeps_its_precision(self):
        """complex128 data x complex128 chirp stays complex128 end to end; complex64 data x complex128 chirp is
        complex128 in numpy (the reference's product), so it is here."""
        import scipy.fft
        shape, sr, fc, dm = (1 << 14, 2, 2), 1e6, 1e9, 3.0
        rng = np.random.default_rng(11)
        x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5
        n = shape[0]
        f = orc.channel_freqs(fc, sr, 2)
        ph = np.stack([orc.phase_cycles(dm, n, 1 / sr, fi, fc, np.arange(n)) for fi in f], axis=1)
        chirp = np.exp(-2j * np.pi * ph)                                     # float64 precision, NOT rounded to c64
        start, stop = orc.crop_bounds(dm, n, 2, sr, fc, fc)
        want = scipy.fft.ifft(scipy.fft.fft(x, axis=0) * chirp[:, :, None], axis=0)[start:stop]
        z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
        y = pb.coherent_dedispersion(z, pb.DM(dm), chirp=chirp)
        assert y.dtype == np.complex128 and per_series_l2(y, want).max() < 1e-12
        z64 = pb.DualPolarizationSignal(x.astype(np.complex64), sample_rate=sr * u.Hz, center_freq=fc * u.Hz,
                                        pol_type="linear")
        y64 = pb.coherent_dedispersion(z64, pb.DM(dm), chirp=chirp)
        assert y64.dtype == np.complex128 and per_series_l2(y64, want).max() < 1e-6


def test_config3_full_size_stream():
    """BASELINE configs[3] at full size: 2^28 samples x 8 chan x 2 pol streamed from host memory in 2^22-sample chunks
    (overlap-save, hop 615 915, 430 chunks, double-buffered hipMemcpyAsync); three of the chunks against the oracle, the
    chunk layout against the concatenate-of-reference-calls recipe (transforms.py:59-148)."""
    from pulsarbat_amd import _hip
    import psutil
    if psutil.virtual_memory().available < 110 * (1 << 30):
        pytest.skip("needs ~70 GB of host memory for the 2^28-sample input and output (34 GB each)")
    total, n, nchan, npol, dm, band, fc = 1 << 28, 1 << 22, 8, 2, 56.77, 400e6, 1.4e9
    sr = band / nchan
    start, stop = orc.crop_bounds(dm, n, nchan, sr, fc, fc)
    assert (start, stop) == (1408404, 2024319)
    hop = stop - start
    nchunk = (total - n) // hop + 1
    assert nchunk == 430
    # 34 GB of input without 34 GB worth of random numbers: one random 2^24-sample block, repeated with a different
    # complex factor per repeat (the hop is not commensurate with the block, so every chunk sees different data)
    base = orc.synthetic_block((1 << 24, nchan, npol), 20260004)
    x = np.empty((total, nchan, npol), np.complex64)
    for k in range(total >> 24):
        np.multiply(base, np.complex64(np.exp(0.37j * k) * (1 + 0.01 * k)), out=x[k << 24:(k + 1) << 24])
    freqs = orc.channel_freqs(fc, sr, nchan)
    with _hip.Plan(n, nchan, npol, start, stop, device=0) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, fc)
        y, ms = plan.dedisperse_stream(x)
    assert y.shape == (nchunk * hop, nchan, npol) and ms > 0
    for k in (0, 215, 429):
        want = oracle_shard(x[k * hop:k * hop + n], dm, sr, freqs, fc, start, stop)
        err = per_series_l2(y[k * hop:(k + 1) * hop], want)
        assert err.max() < RTOL_L2, f"chunk {k}: per-series relative L2 {err}"
    print(f"configs[3] full size: {ms:.0f} ms, {nchunk * n * nchan * npol / ms / 1e3:.0f} Msamples/s in, "
          f"{y.shape[0] * nchan * npol / ms / 1e3:.0f} Msamples/s valid, H2D {nchunk * n * nchan * npol * 8 / ms / 1e6:.1f} GB/s")
