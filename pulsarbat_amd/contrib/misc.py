"""``contrib.stft`` / ``contrib.istft``: the critically sampled channeliser
(reference pulsarbat/contrib/misc.py:17-93).

Same call surface and return conventions as the reference (``NotImplemented`` for
anything but ``window="boxcar", noverlap=0, nfft=None``; ``ValueError`` for a
non-baseband signal).  The arithmetic -- per-segment FFT over time, ``fftshift``,
``/ nperseg`` and the re-layout to ``(nseg, nchan*nperseg, ...)`` -- is one HIP
kernel for power-of-two ``nperseg`` up to a tile (``pbh_stft``), the Bluestein ring
otherwise.  Host (numpy) signals round-trip through the device; DeviceArray
signals stay in HBM.
"""

import numpy as np

from .. import _hip
from ..core import BasebandSignal
from ..device import DeviceArray

__all__ = ["stft", "istft", "stft_dedisperse", "dedisperse_istft"]


def _data(z):
    return z.data if isinstance(z.data, DeviceArray) else np.ascontiguousarray(z.data)


def stft(z, /, window="boxcar", nperseg=256, noverlap=0, nfft=None):
    """Short-time Fourier transform of a baseband signal (misc.py:17-55): rectangular window,
    no overlap -- each block of ``nperseg`` samples becomes ``nperseg`` sub-channels."""
    if window != "boxcar" or noverlap != 0 or nfft is not None:
        return NotImplemented
    if not isinstance(z, BasebandSignal):
        raise ValueError("z must be a BasebandSignal.")
    nperseg = int(nperseg)
    z = z[: len(z) - len(z) % nperseg, :]
    x = _hip.stft(_data(z), nperseg, inverse=False)
    falign = "center" if nperseg % 2 else "bottom"
    return type(z).like(z, x, sample_rate=z.sample_rate / nperseg, freq_align=falign)


def istft(z, /, window="boxcar", nperseg=256, noverlap=0, nfft=None):
    """Inverse of :func:`stft` (misc.py:58-93)."""
    if window != "boxcar" or noverlap != 0 or nfft is not None:
        return NotImplemented
    if not isinstance(z, BasebandSignal):
        raise ValueError("z must be a BasebandSignal.")
    nperseg = int(nperseg)
    x = _hip.stft(_data(z), nperseg, inverse=True)
    return type(z).like(z, x, sample_rate=z.sample_rate * nperseg, freq_align="center")


def stft_dedisperse(z, DM, /, *, nperseg=256, ref_freq=None, variant="auto"):
    """``coherent_dedispersion(stft(z, nperseg=nperseg), DM, ref_freq=ref_freq)`` -- the channelise-then-dedisperse
    pipeline (misc.py:41-55 followed by transforms/dedispersion.py:118-133) -- as ONE call of the HIP library
    (``pbh_stft_dedisperse``) for device-resident signals: the channeliser writes its output series-major into the
    dedispersion's work buffer, so the channelised block is neither stored in the reference layout nor de-interleaved
    again.  Same result, metadata included, as the two calls; signals on the host, or geometries the fused kernel does
    not cover, run the two steps one after the other."""
    import math
    from ..transforms.dedispersion import _plan_for, coherent_dedispersion
    if not isinstance(z, BasebandSignal):
        raise ValueError("z must be a BasebandSignal.")
    nperseg = int(nperseg)
    if not isinstance(z.data, DeviceArray) or z.data.dtype != np.complex64 or nperseg < 2:
        return coherent_dedispersion(stft(z, nperseg=nperseg), DM, ref_freq=ref_freq, variant=variant)
    z = z[: len(z) - len(z) % nperseg, :]
    nseg = len(z) // nperseg
    # the channelised signal's metadata (sample rate, channel grid, start time) through the container's own rules, on a
    # one-sample placeholder: the arrays themselves never exist in the reference layout
    falign = "center" if nperseg % 2 else "bottom"
    shape1 = (1, z.shape[1] * nperseg) + tuple(z.shape[2:])
    meta = type(z).like(z, np.zeros(shape1, dtype=z.dtype), sample_rate=z.sample_rate / nperseg, freq_align=falign)
    ref = meta.center_freq if ref_freq is None else ref_freq
    top = float(DM.sample_delay(meta.max_freq, ref, meta.sample_rate))
    bot = float(DM.sample_delay(meta.min_freq, ref, meta.sample_rate))
    start, stop = math.ceil(-min(0, top, bot)), nseg - math.ceil(max(0, top, bot))

    class _Geom:   # what _plan_for reads of a signal: geometry, channel grid, sampling, dtype, data (for the device)
        shape = (nseg,) + shape1[1:]
        ndim = len(shape1)
        dtype = z.dtype
        data = z.data
        channel_freqs = meta.channel_freqs
        dt = meta.dt

        def __len__(self):
            return nseg

    plan, _ = _plan_for(_Geom(), DM, ref, (start, stop), variant=variant)
    y = plan.stft_dedisperse(z.data, nperseg)
    kw = {}
    if z.start_time is not None:
        kw["start_time"] = z.start_time + start / meta.sample_rate
    return type(z).like(z, y, sample_rate=meta.sample_rate, freq_align=falign, **kw)


def dedisperse_istft(z, DM, /, *, nperseg=256, ref_freq=None, chirp=None, variant="auto"):
    """``istft(coherent_dedispersion(z, DM, ref_freq=ref_freq, chirp=chirp), nperseg=nperseg)`` -- the way back from a
    channelised block (transforms/dedispersion.py:118-133 followed by misc.py:58-93) -- as ONE call of the HIP library
    (``pbh_dedisperse_istft``) for device-resident complex64 signals: the dedispersion's last pass leaves its cropped
    result series-major and the synthesis filterbank reads that, so the channelised result is neither re-interleaved into
    the reference layout nor read back.  Same result, metadata included, as the two calls; signals on the host, user
    chirps and other dtypes run the two steps one after the other."""
    from ..transforms.dedispersion import _crop_bounds, _plan_for, coherent_dedispersion
    if not isinstance(z, BasebandSignal):
        raise ValueError("z must be a BasebandSignal.")
    nperseg = int(nperseg)
    if (not isinstance(z.data, DeviceArray) or z.data.dtype != np.complex64 or chirp is not None or nperseg < 2
            or z.shape[1] % nperseg):
        return istft(coherent_dedispersion(z, DM, ref_freq=ref_freq, chirp=chirp, variant=variant), nperseg=nperseg)
    ref = z.center_freq if ref_freq is None else ref_freq
    start, stop = _crop_bounds(z, DM, ref)
    plan, _ = _plan_for(z, DM, ref, (start, stop), variant=variant)
    y = plan.dedisperse_istft(z.data, nperseg)
    kw = {}
    if z.start_time is not None:
        kw["start_time"] = z.start_time + start / z.sample_rate
    return type(z).like(z, y, sample_rate=z.sample_rate * nperseg, freq_align="center", **kw)
