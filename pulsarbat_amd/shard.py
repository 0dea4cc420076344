"""Channel sharding of the hot path across the GPUs of one node (one process per GPU).

Every (channel, polarisation) series is an independent FFT -> chirp -> IFFT, the chirp
depends only on the channel frequency (reference dedispersion.py:70-73) and the crop only on
the edges of the FULL band (dedispersion.py:127-131).  That is the decomposition the reference
gets from Dask chunks over the non-time axes (core.py:332-345); here a chunk is a rank's
contiguous block of channels (both polarisations stay together so the chirp is shared and
Stokes detection stays local).  There is no data-path collective: ranks generate their own
chirps from scalars.  Two optional exchange steps exist:

* ``chirp=`` with ``chirp_src=r``: rank ``r`` holds a user-supplied full-band chirp and the ranks
  receive their channel blocks of it (one scatter over RCCL / gloo; a chirp shared by all channels is
  one broadcast) -- the sharded form of dedispersion.py:121-124.
* ``gather=``: the analogue of ``Signal.compute()`` on a chunked dask array (core.py:298-309).  For
  device data the ranks' pipelines write their channel slices directly into the destination ranks'
  full-band blocks over xGMI (``pulsarbat_amd.node.ChannelGather``: ``pbh_node_*`` +
  ``pbh_dedisperse_slice``); host data is gathered with a gloo all-gather.
"""

import math
import threading

import numpy as np

from . import units as u
from .core import BasebandSignal

__all__ = ["channel_slice", "shard_signal", "coherent_dedispersion_sharded", "dedisperse_detect_sharded", "release_gathers"]

# ChannelGather objects by (nout, npol, dtype, device, group, mode, root), next to the plan cache (transforms.dedispersion._PLANS):
# a stream of blocks through coherent_dedispersion_sharded(gather=...) sets the destination buffers and their peer mappings up
# ONCE.  Construction and release are collective, so a rank may use its cached gather only if EVERY rank has one for the
# call: the ranks agree on that with one small all-reduce per call (a rank-local key can hit on one rank and miss on another --
# a thread identity reused by one process only, a channel count that changed on some ranks only -- and the missing rank's
# set-up collectives would then meet the others' run).  The key holds nothing that differs between ranks except the device.
_GATHERS = {}
_GATHERS_LOCK = threading.Lock()
_GATHER_CACHE_SIZE = 4


def _all_hit(group, dev, hit):
    """Collective: True if every rank of the group found a usable cached gather (one MIN all-reduce of a flag)."""
    import torch
    import torch.distributed as dist
    cdev = torch.device("cuda", int(dev)) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if hit else 0], dtype=torch.int32, device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def _gather_for(plan, nchan, npol, dtype, dev, group, mode, root):
    from .node import ChannelGather
    key = (plan.nout, int(npol), np.dtype(dtype).str, int(dev), id(group) if group is not None else None, mode, int(root))
    with _GATHERS_LOCK:
        g = _GATHERS.get(key)
    usable = g is not None and g.nchan_local == int(nchan) and not getattr(g, "_closed", False)
    if not _all_hit(group, dev, usable):
        # some rank has to build: everybody builds (and everybody first closes what it had for the key, in step)
        if g is not None:
            with _GATHERS_LOCK:
                _GATHERS.pop(key, None)
            g.close()
        g = ChannelGather(plan.nout, nchan, npol, dtype, dev, group=group, mode=mode, root=root)
        stale = []
        with _GATHERS_LOCK:
            _GATHERS[key] = g
            mine = [k for k in _GATHERS if k[4] == key[4]]
            while len(mine) > _GATHER_CACHE_SIZE:      # oldest first: every rank built them in the same order
                stale.append(_GATHERS.pop(mine.pop(0)))
        for old in stale:
            old.close()
    return key, g


def release_gathers(group=None):
    """Collective: close the cached gathers of ``group`` (all groups when None): destination buffers, peer mappings.  Call it
    on every rank of the group, e.g. before ``destroy_process_group``; the memory is otherwise held for re-use by later calls."""
    gid = id(group) if group is not None else None
    with _GATHERS_LOCK:
        stale = [_GATHERS.pop(k) for k in [k for k in _GATHERS if group is None or k[4] == gid]]   # insertion order: the same on every rank
    for g in stale:
        g.close()


def channel_slice(nchan, world, rank):
    """Contiguous block partition of ``nchan`` channels over ``world`` ranks (ragged allowed)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(nchan, world)
    lo = rank * base + min(rank, extra)
    return slice(lo, lo + base + (1 if rank < extra else 0))


def shard_signal(z, world, rank):
    """This rank's channels of ``z`` as a signal of the same type.  Frequency bookkeeping is the
    reference's own freq-axis slicing (core.py:479-498): center_freq becomes the shard's centre."""
    sl = channel_slice(z.nchan, world, rank)
    if sl.stop <= sl.start:
        raise ValueError(f"rank {rank} of {world} gets no channels (nchan = {z.nchan})")
    return z[:, sl]


def _full_band_crop(dm, nsample, sample_rate, band_min, band_max, ref_freq):
    """start/stop of dedispersion.py:127-131 evaluated on the edges of the full band."""
    top = float(dm.sample_delay(band_max, ref_freq, sample_rate))
    bot = float(dm.sample_delay(band_min, ref_freq, sample_rate))
    return math.ceil(-min(0, top, bot)), nsample - math.ceil(max(0, top, bot))


def _rank_device(z_local, device):
    """The GPU this rank computes on: the data's own device, else ``device=``, else the process's current
    device (one process per GPU: ``torch.cuda.set_device(LOCAL_RANK)``) -- never a fixed GPU 0."""
    from .device import DeviceArray
    if isinstance(z_local.data, DeviceArray):
        return z_local.data.device_index
    if device is not None:
        return int(device)
    import torch
    return torch.cuda.current_device()


def _scatter_chirp(chirp, src, counts, rank, group, device):
    """Rank ``src`` holds a full-band chirp (channels on axis 1); every rank gets its channel block.
    A chirp with a single channel row (shared by all channels) is broadcast whole.  The payload travels as a
    real view over RCCL (device tensors) or gloo (host tensors); shapes go first as a small object."""
    import torch
    import torch.distributed as dist
    from .device import DeviceArray
    backend = dist.get_backend(group)
    on_gpu = backend == "nccl"
    meta = [None]
    if rank == src:
        if on_gpu and not isinstance(chirp, DeviceArray):
            chirp = DeviceArray.from_host(np.asarray(chirp), device=device)   # through pbh_transfer, not a torch copy
        if not on_gpu and isinstance(chirp, DeviceArray):
            chirp = chirp.get()
        c = chirp.tensor if isinstance(chirp, DeviceArray) else torch.from_numpy(np.ascontiguousarray(chirp))
        if c.dim() < 2:
            c = c.reshape(tuple(c.shape) + (1,) * (2 - c.dim()))
        meta = [(tuple(c.shape), str(c.dtype).replace("torch.", ""))]
    dist.broadcast_object_list(meta, src=_global_rank(group, src), group=group)
    shape, dname = meta[0]
    tdtype = getattr(torch, dname)
    dev = torch.device("cuda", device) if on_gpu else torch.device("cpu")
    world = len(counts)
    if shape[1] == 1:   # one row for every channel: a plain broadcast of the chirp
        buf = c.contiguous() if rank == src else torch.empty(shape, dtype=tdtype, device=dev)
        view = torch.view_as_real(buf) if buf.is_complex() else buf
        dist.broadcast(view, src=_global_rank(group, src), group=group)
        mine = buf
    else:
        if shape[1] != sum(counts):
            raise ValueError(f"the full-band chirp has {shape[1]} channels, the ranks hold {sum(counts)}")
        cmax = max(counts)
        pshape = (shape[0], cmax) + tuple(shape[2:])
        recv = torch.empty(pshape, dtype=tdtype, device=dev)
        pieces = None
        if rank == src:
            pieces, lo = [], 0
            for n in counts:
                piece = torch.zeros(pshape, dtype=tdtype, device=dev)
                piece[:, :n] = c[:, lo:lo + n]
                pieces.append(torch.view_as_real(piece) if piece.is_complex() else piece)
                lo += n
        dist.scatter(torch.view_as_real(recv) if recv.is_complex() else recv, pieces, src=_global_rank(group, src),
                     group=group)
        mine = recv[:, :counts[rank]].contiguous()
    return DeviceArray(mine) if mine.is_cuda else mine.numpy()


def _global_rank(group, r):
    import torch.distributed as dist
    return r if group is None else dist.get_global_rank(group, r)


def coherent_dedispersion_sharded(z_local, DM, /, *, band_min, band_max, ref_freq, chirp=None, chirp_src=None,
                                  group=None, gather=False, root=0, device=None, variant="auto"):
    """Dedisperse this rank's channel shard consistently with the full-band call.

    z_local   this rank's BasebandSignal shard (``shard_signal``), host or device resident
    band_min / band_max   ``min_freq`` / ``max_freq`` of the FULL signal (crop, dedispersion.py:127-128)
    ref_freq  reference frequency of the full call (the full signal's ``center_freq`` by default there)
    chirp     user-supplied chirp (dedispersion.py:121-124), unchecked as in the reference: this rank's block
              (anything that broadcasts against ``z_local.data``), or -- with ``chirp_src=r`` -- the FULL-band
              chirp on rank ``r`` only (``None`` elsewhere), scattered by channel
    gather    ``False``: the shard result stays where it is (``persist()`` semantics);
              ``True`` / ``"all"``: every rank gets the full-band signal; ``"root"``: only rank ``root`` does
              (the others return ``None``: their slice was written into the root's block)
    device    GPU for host-resident shards (default: the process's current device)

    With ``gather`` the call is collective; for device data the gather's chunks and mappings are cached per geometry
    (``release_gathers``).
    """
    if not isinstance(z_local, BasebandSignal):
        raise TypeError("Signal must be a BasebandSignal object.")
    if gather not in (False, True, "all", "root"):
        raise ValueError("gather must be False, True / 'all' or 'root'")
    start, stop = _full_band_crop(DM, len(z_local), z_local.sample_rate, band_min, band_max, ref_freq)
    kw = {}
    if z_local.start_time is not None:
        kw["start_time"] = z_local.start_time + start / z_local.sample_rate

    from .device import DeviceArray
    from .transforms.dedispersion import _broadcast_chirp, _geometry, _plan_for
    nsample, nchan, npol = _geometry(z_local)
    on_device = isinstance(z_local.data, DeviceArray)
    dev = _rank_device(z_local, device)
    if chirp_src is not None:
        import torch.distributed as dist
        counts = [None] * dist.get_world_size(group)
        dist.all_gather_object(counts, int(nchan), group=group)
        chirp = _scatter_chirp(chirp, int(chirp_src), [int(c) for c in counts], dist.get_rank(group), group, dev)
    rows, per_pol, dtype = (None, False, np.dtype(z_local.dtype)) if chirp is None else _broadcast_chirp(chirp, z_local)
    pchan, ppol = (nchan * npol, 1) if per_pol else (nchan, npol)
    data = z_local.data if dtype == np.dtype(z_local.dtype) else z_local.data.astype(dtype)
    x = data.contiguous() if on_device else np.ascontiguousarray(data)
    center = (band_min + band_max) / 2
    # the per-thread plan cache of coherent_dedispersion, keyed on this shard's geometry, the FULL band's crop and the
    # shard's channel frequencies: a stream of blocks re-uses plan and chirp
    plan, _ = _plan_for(z_local, DM, ref_freq, (start, stop), chirp=rows, variant=variant, per_pol=per_pol, dtype=dtype,
                        device=dev)
    if gather and on_device:
        mode = "root" if gather == "root" else "all"
        key, g = _gather_for(plan, pchan, ppol, dtype, dev, group, mode, root)
        try:
            full = g.run(plan, x)
        except Exception:
            # a failed run raises on every rank (node.GatherError): all of them drop the gather, in step
            with _GATHERS_LOCK:
                _GATHERS.pop(key, None)
            g.close()
            raise
        if full is None:   # a non-root rank of a root gather: its slice was written into the root's block
            return None
        n_total = g.nchan_total // (npol if per_pol else 1)
        full = DeviceArray(full.tensor.reshape((plan.nout, n_total) + tuple(z_local.shape[2:])))
        return type(z_local).like(z_local, full, center_freq=center, freq_align="center", **kw)
    y = plan.dedisperse(x)   # device data: asynchronous on the current stream, like every other device transform
    out = type(z_local).like(z_local, y, **kw)
    if not gather:
        return out
    return _gather_channels(out, band_min, band_max, group, root if gather == "root" else None)


def dedisperse_detect_sharded(z_local, DM, /, *, band_min, band_max, ref_freq, mode="I", nscrunch=1, group=None,
                              gather=False, device=None, variant="auto"):
    """``dedisperse_detect`` (dedispersion + detection + ``nscrunch``-fold time sum, BASELINE configs[4]) on this
    rank's channel shard with the FULL band's crop.  Returns ``(array, start)``: float32 ``(nout // nscrunch,
    nchan_local[, npol | 4])``, or with ``gather=True`` the full-band ``(..., nchan_total, ...)`` array on every
    rank -- the detected output is ~1/nscrunch of the voltages (2.2 MB at configs[4]), so this gather is one
    small all-gather over RCCL (device data) or gloo (host data)."""
    if not isinstance(z_local, BasebandSignal):
        raise TypeError("Signal must be a BasebandSignal object.")
    from .device import DeviceArray
    from .transforms.dedispersion import _plan_for
    start, stop = _full_band_crop(DM, len(z_local), z_local.sample_rate, band_min, band_max, ref_freq)
    on_device = isinstance(z_local.data, DeviceArray)
    dev = _rank_device(z_local, device)
    x = z_local.data.contiguous() if on_device else np.ascontiguousarray(z_local.data)
    plan, _ = _plan_for(z_local, DM, ref_freq, (start, stop), variant=variant, device=dev)
    y = plan.dedisperse_detect(x, nscrunch=nscrunch, mode=mode)
    if not gather:
        return y, start
    import torch
    import torch.distributed as dist
    t = y.tensor if on_device else torch.from_numpy(y)
    if on_device and dist.get_backend(group) != "nccl":
        t = torch.from_numpy(y.get())   # a host-side group carries host tensors
    world = dist.get_world_size(group)
    counts = [None] * world
    dist.all_gather_object(counts, int(t.shape[1]), group=group)
    cmax = max(counts)
    mine = torch.zeros((t.shape[0], cmax) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
    mine[:, :t.shape[1]] = t
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    full = torch.cat([b[:, :c] for b, c in zip(bufs, counts)], dim=1).contiguous()
    if on_device:
        return (DeviceArray(full) if full.is_cuda else DeviceArray.from_host(full.numpy(), device=dev)), start
    return full.numpy(), start


def _gather_channels(shard, band_min, band_max, group, root=None):
    """Gather of host-resident shards along axis 1 over the group's own backend: to every rank (``root=None``) or to rank
    ``root`` only (the others return None).  Shards may be ragged in nchan, so sizes are exchanged first."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(shard.data))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [None] * world
    dist.all_gather_object(sizes, int(t.shape[1]), group=group)
    sizes = [int(n) for n in sizes]
    # channel-major contiguous pieces so every rank contributes one flat buffer; ragged shards (nchan % world != 0) are
    # padded to the largest one -- gloo's all_gather / gather want equal sizes -- and trimmed after the collective
    piece = torch.view_as_real(t.transpose(0, 1).contiguous())
    nmax = max(sizes)
    mine = piece
    if piece.shape[0] != nmax:
        mine = torch.zeros((nmax,) + tuple(piece.shape[1:]), dtype=piece.dtype)
        mine[:piece.shape[0]] = piece
    bufs = None
    if root is None or rank == root:
        bufs = [torch.empty_like(mine) for _ in sizes]
    if root is None:
        dist.all_gather(bufs, mine, group=group)
    else:
        dist.gather(mine, bufs if rank == root else None, dst=_global_rank(group, root), group=group)
        if rank != root:
            return None
    bufs = [b[:n] for b, n in zip(bufs, sizes)]
    full = torch.view_as_complex(torch.cat(bufs, dim=0)).transpose(0, 1).contiguous()
    center = (band_min + band_max) / 2
    return type(shard).like(shard, full.numpy(), center_freq=center, freq_align="center")
