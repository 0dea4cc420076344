"""``pb.fft``: FFT functions dispatched on the array type (reference pulsarbat/fft.py:1-48).

Attribute access builds, per name, a ``functools.singledispatch`` function whose
default is ``scipy.fft.<name>`` (reference fft.py:36-38) carrying scipy's
``__name__``/``__qualname__``/``__doc__`` (fft.py:45-47).  Where the reference
registers ``dask.array.Array`` (fft.py:40-43), this build registers
:class:`~pulsarbat_amd.device.DeviceArray`: ``fft``/``ifft`` of device-resident
complex64 data run on the HIP kernels through ``pbh_fft_c2c``.  The other twelve
names are not on the hot path and have no device implementation.
"""

from functools import singledispatch

import scipy.fft

from .device import DeviceArray

_FFT_FUNCS = [
    "fft", "fft2", "fftn", "ifft", "ifft2", "ifftn", "rfft", "rfft2", "rfftn",
    "irfft", "irfft2", "irfftn", "hfft", "ihfft",
]


def __dir__():
    return sorted(_FFT_FUNCS)


def _device_c2c(name):
    inverse = name == "ifft"

    def run(x, n=None, axis=-1, norm=None, overwrite_x=False, workers=None, *, plan=None):
        from . import _hip
        if n is not None or norm not in (None, "backward"):
            raise NotImplementedError("device FFT supports n=None, norm=None only")
        if axis % x.ndim != 0:
            raise NotImplementedError("device FFT runs along axis 0 (the time axis) only")
        return _hip.fft_c2c(x.contiguous(), inverse=inverse)

    return run


def __getattr__(name):
    if name not in _FFT_FUNCS:
        raise AttributeError(f"module {__name__} has no attribute {name}")

    _fft_func = getattr(scipy.fft, name)

    @singledispatch
    def func(*args, **kwargs):
        return _fft_func(*args, **kwargs)

    if name in ("fft", "ifft"):
        func.register(DeviceArray)(_device_c2c(name))
    else:
        @func.register(DeviceArray)
        def _(*args, **kwargs):
            raise NotImplementedError(f"pb.fft.{name} has no device implementation (only fft/ifft do)")

    func.__qualname__ = _fft_func.__qualname__
    func.__name__ = _fft_func.__name__
    func.__doc__ = _fft_func.__doc__
    return func
