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

__all__ = ["stft", "istft"]


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
