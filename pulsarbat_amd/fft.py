"""``pb.fft``: FFT functions dispatched on the array type (reference pulsarbat/fft.py:1-48).

Attribute access builds, per name, a ``functools.singledispatch`` function whose
default is ``scipy.fft.<name>`` (reference fft.py:36-38) carrying scipy's
``__name__``/``__qualname__``/``__doc__`` (fft.py:45-47).  Where the reference
registers ``dask.array.Array`` (fft.py:40-43), this build registers
:class:`~pulsarbat_amd.device.DeviceArray`: all fourteen names run on device-resident data with
scipy's ``n`` / ``s`` / ``axis`` / ``axes`` / ``norm`` semantics, each built from ``pbh_fft_c2c`` (the HIP
transform along one axis at a time; e.g. the channeliser's ``pb.fft.fft(x, axis=2, n=nfft)``,
contrib/misc.py:47); the real and Hermitian forms slice or extend the complex transform.
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


def _norm_scale(norm, n, inverse):
    """Factor applied on top of scipy's norm=None convention (forward unscaled, inverse 1/n)."""
    if norm in (None, "backward"):
        return 1.0
    if norm == "ortho":
        return n ** 0.5 if inverse else n ** -0.5
    if norm == "forward":
        return float(n) if inverse else 1.0 / n
    raise ValueError(f'Invalid norm value {norm!r}, should be "backward", "ortho" or "forward"')


def _c2c_1d(x, n, axis, norm, inverse):
    """1-D complex transform of a DeviceArray along any axis with scipy's n (zero padding / truncation) and norm:
    the axis is brought to the front (a view), padded or cut, made contiguous and run through pbh_fft_c2c."""
    import torch
    from . import _hip
    t = x.tensor
    if not t.is_complex():
        t = t.to(torch.complex64 if t.dtype == torch.float32 else torch.complex128)
    axis = axis % t.dim()
    t = t.movedim(axis, 0)
    m = t.shape[0]
    n = m if n is None else int(n)
    if n < 1:
        raise ValueError(f"invalid number of data points ({n}) specified")
    if n < m:
        t = t[:n]
    elif n > m:
        t = torch.cat([t, torch.zeros((n - m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)], dim=0)
    y = _hip.fft_c2c(DeviceArray(t.contiguous()), inverse=inverse).tensor
    sc = _norm_scale(norm, n, inverse)
    if sc != 1.0:
        y = y * sc
    return DeviceArray(y.movedim(0, axis))


def _axes_and_sizes(x, s, axes, default_last):
    nd = x.ndim
    if axes is None:
        axes = list(range(nd))[-default_last:] if (s is None and default_last) else list(range(nd - (len(s) if s is not None else nd), nd))
    axes = [a % nd for a in (axes if hasattr(axes, "__len__") else [axes])]
    if s is None:
        s = [None] * len(axes)
    if len(s) != len(axes):
        raise ValueError("when given, axes and shape arguments have to be of the same length")
    return axes, list(s)


def _device_impl(name):
    """Device implementations of the fourteen names: every transform is built from pbh_fft_c2c along one axis at a
    time; the real and Hermitian forms slice / extend the complex one (scipy.fft semantics for n, s, axis, axes, norm)."""
    import torch
    inverse = name.startswith("i") and name != "ihfft"

    def fft1(x, n=None, axis=-1, norm=None, overwrite_x=False, workers=None, *, plan=None):
        return _c2c_1d(x, n, axis, norm, inverse)

    def fftnd(default_last):
        def run(x, s=None, axes=None, norm=None, overwrite_x=False, workers=None, *, plan=None):
            ax, sz = _axes_and_sizes(x, s, axes, default_last)
            for a, n in zip(ax, sz):
                x = _c2c_1d(x, n, a, norm, inverse)
            return x
        return run

    def rfft(x, n=None, axis=-1, norm=None, overwrite_x=False, workers=None, *, plan=None):
        if x.tensor.is_complex():
            raise TypeError("x must be a real sequence")
        m = x.shape[axis % x.ndim] if n is None else int(n)
        y = _c2c_1d(x, n, axis, norm, False).tensor
        return DeviceArray(y.narrow(axis % y.dim(), 0, m // 2 + 1))

    def _hermitian_full(t, axis, n):
        """Full length-n spectrum from its first n//2 + 1 bins along `axis` (input cut / zero-padded to that many)."""
        half = n // 2 + 1
        m = t.shape[axis]
        if m > half:
            t = t.narrow(axis, 0, half)
        elif m < half:
            pad = list(t.shape)
            pad[axis] = half - m
            t = torch.cat([t, torch.zeros(pad, dtype=t.dtype, device=t.device)], dim=axis)
        head = t.narrow(axis, 0, 1)
        t = torch.cat([torch.complex(head.real, torch.zeros_like(head.real)), t.narrow(axis, 1, half - 1)], dim=axis)
        ntail = n - half
        tail = t.narrow(axis, 1, ntail).flip(axis).conj() if ntail > 0 else t.narrow(axis, 0, 0)
        if n % 2 == 0 and half >= 2:   # the Nyquist bin of an even length is real
            nyq = t.narrow(axis, half - 1, 1)
            t = torch.cat([t.narrow(axis, 0, half - 1), torch.complex(nyq.real, torch.zeros_like(nyq.real))], dim=axis)
        return torch.cat([t, tail.resolve_conj()], dim=axis)

    def irfft(x, n=None, axis=-1, norm=None, overwrite_x=False, workers=None, *, plan=None):
        t = x.tensor
        if not t.is_complex():
            t = t.to(torch.complex64 if t.dtype == torch.float32 else torch.complex128)
        axis = axis % t.dim()
        n = 2 * (t.shape[axis] - 1) if n is None else int(n)
        if n < 1:
            raise ValueError(f"Invalid number of data points ({n}) specified")
        y = _c2c_1d(DeviceArray(_hermitian_full(t, axis, n)), None, axis, norm, True).tensor
        return DeviceArray(y.real.contiguous())

    def hfft(x, n=None, axis=-1, norm=None, overwrite_x=False, workers=None, *, plan=None):
        # hfft(x, n) = irfft(conj(x), n) * n  (norm=None)
        t = x.tensor
        axis_ = axis % t.dim()
        n_ = 2 * (t.shape[axis_] - 1) if n is None else int(n)
        y = irfft(DeviceArray(t.conj().resolve_conj()), n_, axis, None).tensor * float(n_)
        return DeviceArray(y * _norm_scale(norm, n_, False))

    def ihfft(x, n=None, axis=-1, norm=None, overwrite_x=False, workers=None, *, plan=None):
        # ihfft(x, n) = conj(rfft(x, n)) / n  (norm=None)
        m = x.shape[axis % x.ndim] if n is None else int(n)
        y = rfft(x, n, axis, None).tensor.conj().resolve_conj() / float(m)
        return DeviceArray(y * _norm_scale(norm, m, True))

    def rfftnd(default_last):
        def run(x, s=None, axes=None, norm=None, overwrite_x=False, workers=None, *, plan=None):
            ax, sz = _axes_and_sizes(x, s, axes, default_last)
            x = rfft(x, sz[-1], ax[-1], norm)
            for a, n in zip(ax[:-1], sz[:-1]):
                x = _c2c_1d(x, n, a, norm, False)
            return x
        return run

    def irfftnd(default_last):
        def run(x, s=None, axes=None, norm=None, overwrite_x=False, workers=None, *, plan=None):
            ax, sz = _axes_and_sizes(x, s, axes, default_last)
            for a, n in zip(ax[:-1], sz[:-1]):
                x = _c2c_1d(x, n, a, norm, True)
            return irfft(x, sz[-1], ax[-1], norm)
        return run

    table = {"fft": fft1, "ifft": fft1, "fft2": fftnd(2), "ifft2": fftnd(2), "fftn": fftnd(0), "ifftn": fftnd(0),
             "rfft": rfft, "irfft": irfft, "hfft": hfft, "ihfft": ihfft, "rfft2": rfftnd(2), "rfftn": rfftnd(0),
             "irfft2": irfftnd(2), "irfftn": irfftnd(0)}
    return table[name]


def __getattr__(name):
    if name not in _FFT_FUNCS:
        raise AttributeError(f"module {__name__} has no attribute {name}")

    _fft_func = getattr(scipy.fft, name)

    @singledispatch
    def func(*args, **kwargs):
        return _fft_func(*args, **kwargs)

    func.register(DeviceArray)(_device_impl(name))

    func.__qualname__ = _fft_func.__qualname__
    func.__name__ = _fft_func.__name__
    func.__doc__ = _fft_func.__doc__
    return func
