"""Channel sharding of the hot path across the GPUs of one node (one process per GPU).

Every (channel, polarisation) series is an independent FFT -> chirp -> IFFT, the chirp
depends only on the channel frequency (reference dedispersion.py:70-73) and the crop only on
the edges of the FULL band (dedispersion.py:127-131).  That is the decomposition the reference
gets from Dask chunks over the non-time axes (core.py:332-345); here a chunk is a rank's
contiguous block of channels (both polarisations stay together so the chirp is shared and
Stokes detection stays local).  There is no data-path collective: ranks generate their own
chirps from scalars.  ``gather=True`` adds the one real exchange step, an all-gather of the
output along the channel axis (RCCL over xGMI for device data, gloo for host data) -- the
analogue of ``Signal.compute()`` on a chunked dask array.
"""

import math

import numpy as np

from . import units as u
from .core import BasebandSignal

__all__ = ["channel_slice", "shard_signal", "coherent_dedispersion_sharded"]


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


def coherent_dedispersion_sharded(z_local, DM, /, *, band_min, band_max, ref_freq, group=None,
                                  gather=False, variant="auto", _transform=None):
    """Dedisperse this rank's channel shard consistently with the full-band call.

    z_local   this rank's BasebandSignal shard (``shard_signal``), host or device resident
    band_min / band_max   ``min_freq`` / ``max_freq`` of the FULL signal (crop, dedispersion.py:127-128)
    ref_freq  reference frequency of the full call (the full signal's ``center_freq`` by default there)
    gather    all-gather the shards along the channel axis and return the full-band signal on every
              rank; otherwise the shard result stays where it is (``persist()`` semantics)
    _transform  test hook: ``f(x, start, stop, chan_freqs_hz, ref_hz) -> y`` replacing the HIP plan
                (used by the CPU gloo tests with the oracle; never set by product code)
    """
    if not isinstance(z_local, BasebandSignal):
        raise TypeError("Signal must be a BasebandSignal object.")
    start, stop = _full_band_crop(DM, len(z_local), z_local.sample_rate, band_min, band_max, ref_freq)
    freqs = np.asarray(u.to_value(z_local.channel_freqs, u.Hz), dtype=np.float64)
    ref_hz = u.to_value(ref_freq, u.Hz)

    if _transform is not None:
        y = _transform(np.asarray(z_local.data), start, stop, freqs, ref_hz)
    else:
        from . import _hip
        from .device import DeviceArray
        from .transforms.dedispersion import _geometry
        nsample, nchan, npol = _geometry(z_local)
        on_device = isinstance(z_local.data, DeviceArray)
        dev = z_local.data.device_index if on_device else 0
        with _hip.Plan(nsample, nchan, npol, start, stop, device=dev, variant=variant, dtype=z_local.dtype) as plan:
            plan.chirp_generate(DM._coeff_s_mhz2 * 1e12, u.to_value(z_local.dt, u.s), freqs, ref_hz)
            x = z_local.data.contiguous() if on_device else np.ascontiguousarray(z_local.data)
            y = plan.dedisperse(x)
            if on_device:
                import torch
                torch.cuda.synchronize(dev)

    kw = {}
    if z_local.start_time is not None:
        kw["start_time"] = z_local.start_time + start / z_local.sample_rate
    out = type(z_local).like(z_local, y, **kw)
    if not gather:
        return out
    return _all_gather_channels(out, band_min, band_max, group)


def _all_gather_channels(shard, band_min, band_max, group):
    """All-gather along axis 1.  Shards may be ragged in nchan, so sizes are exchanged first."""
    import torch
    import torch.distributed as dist
    from .device import DeviceArray

    on_device = isinstance(shard.data, DeviceArray)
    t = shard.data.tensor if on_device else torch.from_numpy(np.ascontiguousarray(shard.data))
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.shape[1]], dtype=torch.int64, device=t.device), group=group)
    # channel-major contiguous pieces so every rank contributes one flat buffer
    mine = torch.view_as_real(t.transpose(0, 1).contiguous())
    bufs = [torch.empty((int(s.item()),) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
            for s in sizes]
    dist.all_gather(bufs, mine, group=group)
    full = torch.view_as_complex(torch.cat(bufs, dim=0)).transpose(0, 1).contiguous()
    data = DeviceArray(full) if on_device else full.numpy()
    center = (band_min + band_max) / 2
    return type(shard).like(shard, data, center_freq=center, freq_align="center")
