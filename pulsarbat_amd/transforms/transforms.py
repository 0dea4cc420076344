"""``time_shift`` / ``freq_shift`` (reference pulsarbat/transforms/transforms.py:211-361; SURVEY.md 8f
rank 2).  Same FFT * H * IFFT skeleton as coherent dedispersion -- H is a phase ramp (time_shift) or
a band mask after a mixer (freq_shift) -- so both run on the plan pipeline of libpbhip.so with the
transfer function generated on the device (``pbh_chirp_special``), plus two elementwise kernels
(``pbh_mix``, ``pbh_zero_edges``).  Argument handling, broadcasting rules, zero fill, crop and
``start_time`` bookkeeping follow the reference line by line.  Any length (other than powers of two through the Bluestein plan).
"""

import math

import numpy as np

from .. import _hip
from .. import units as u
from ..core import BasebandSignal
from ..device import DeviceArray

__all__ = ["signal_transform", "concatenate", "time_shift", "freq_shift", "snippet", "fast_len"]


def _per_series(arr, z):
    """Broadcast ``arr`` (already given trailing length-1 axes) over ``z.sample_shape`` -> flat (S,)."""
    try:
        full = np.broadcast_to(arr, z.sample_shape)
    except ValueError:
        raise ValueError(f"shift of shape {np.shape(arr)} does not broadcast to the sample shape {z.sample_shape}")
    return np.ascontiguousarray(full, dtype=np.float64).reshape(-1)


def _to_device_complex(z):
    """(DeviceArray (N, S) complex, was_device, was_real, original dtype)."""
    data = z.data
    on_dev = isinstance(data, DeviceArray)
    dt = np.dtype(data.dtype)
    real = dt.kind != "c"
    cdt = np.complex64 if dt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.complex128
    if on_dev:
        d = data.contiguous()
        if real or d.dtype != cdt:
            d = d.astype(cdt)
    else:
        d = DeviceArray.from_host(np.ascontiguousarray(data).astype(cdt, copy=False))
    return d, on_dev, real, dt


def _from_device(y, on_dev, real, dt, shape):
    if on_dev:
        t = y.tensor.reshape(shape)
        return DeviceArray(t.real.contiguous() if real else t)
    a = y.get().reshape(shape)
    return np.ascontiguousarray(a.real).astype(dt, copy=False) if real else a


def time_shift(z, /, shift, crop=False):
    """Shift signal data in time by a phase gradient in the frequency domain (transforms.py:211-293).

    ``shift`` is in samples (number / array broadcasting over the sample shape) or a time Quantity.
    Samples that wrapped around are zero-filled; ``crop=True`` removes them instead and advances
    ``start_time`` by ``max(0, shift.max()) * dt``.
    """
    if hasattr(shift, "unit") and hasattr(shift, "to_value"):
        shift = (shift * z.sample_rate).to_value(u.one)
    shift = np.array(shift, dtype=np.float64)
    if shift.ndim >= z.ndim:
        raise ValueError(f"shift has too many dimensions. Expected <= {z.ndim - 1} dimensions, "
                         f"got {shift.ndim} dimensions!")
    if np.allclose(shift, 0):
        return z
    if shift.ndim > 0:
        shift = shift[(slice(None),) * shift.ndim + (None,) * (z.ndim - shift.ndim - 1)]
    sh = _per_series(shift, z)

    start = max([0] + [int(math.ceil(a)) for a in sh if a >= 0])
    stop = min([0] + [int(math.floor(a)) for a in sh if a < 0])
    N = len(z)
    x, on_dev, real, dt = _to_device_complex(z)
    S = sh.size
    lo, hi = (start, N + stop) if crop else (0, N)
    same = bool(np.all(sh == sh[0]))   # one shift for every series: a single shared phase ramp
    plan = _hip.filter_plan(N, S, lo, hi, x.device_index, x.dtype, shared=same)
    plan.chirp_special(sh[:1] if same else sh, 0)
    y = plan.dedisperse(DeviceArray(x.tensor.reshape((N, 1, S) if same else (N, S, 1))))
    y = DeviceArray(y.tensor.reshape(hi - lo if hi > lo else 0, S))
    if not crop:
        _hip.zero_edges(y, sh)
    out = _from_device(y, on_dev, real, dt, (y.shape[0],) + tuple(z.sample_shape))
    kw = {}
    if crop and z.start_time is not None:
        kw["start_time"] = z.start_time + start / z.sample_rate
    return type(z).like(z, out, **kw)


def freq_shift(z, /, shift):
    """Shift a baseband signal in frequency by mixing with a sinusoid; the out-of-band part is zeroed
    (transforms.py:296-361)."""
    if not isinstance(z, BasebandSignal):
        raise TypeError("Signal must be a BasebandSignal object.")
    try:
        hz = np.asarray(u.to_value(shift.to(u.Hz) if hasattr(shift, "to") else shift, u.Hz), dtype=np.float64)
        if not hasattr(shift, "to"):
            raise TypeError
    except Exception:
        raise ValueError("shift must be a Quantity with units of frequency.")
    if hz.ndim == 0:
        hz = hz[None]
    if hz.ndim >= z.ndim:
        raise ValueError(f"shift has too many dimensions. Expected <= {z.ndim - 1} dimensions, "
                         f"got {hz.ndim} dimensions!")
    hz = hz[(slice(None),) * hz.ndim + (None,) * (z.ndim - hz.ndim - 1)]
    ft = _per_series(hz * u.to_value(z.dt, u.s), z)

    N = len(z)
    x, on_dev, real, dt = _to_device_complex(z)
    S = ft.size
    same = bool(np.all(ft == ft[0]))   # one shift for every series: a single shared band mask
    plan = _hip.filter_plan(N, S, 0, N, x.device_index, x.dtype, shared=same)
    plan.chirp_special((ft[:1] if same else ft) * N, 1)
    # the mixer exp(2 pi i ft n) is applied inside the transform's first pass (no copy, no mixing pass of its own)
    y = plan.dedisperse_mix(DeviceArray(x.tensor.reshape((N, 1, S) if same else (N, S, 1))), ft)
    out = _from_device(DeviceArray(y.tensor.reshape(N, S)), on_dev, False, dt, (N,) + tuple(z.sample_shape))
    return type(z).like(z, out)


def fast_len(z, /):
    """Crop a signal to the largest 7-smooth length <= len(z) (transforms.py:364-382)."""
    from ..utils import prev_fast_len
    return z[: prev_fast_len(len(z))]


def snippet(z, /, t, n):
    """Extract ``n`` samples of ``z`` starting at ``t`` (transforms.py:151-208).

    ``t`` is a number of samples (int or float), a time Quantity relative to the start of the
    signal, or a Time.  A start that is not a whole number of samples is reached by the HIP
    ``time_shift`` (phase gradient in the Fourier domain, cropped), as in the reference.
    """
    import operator
    from ..time import Time
    if (n := operator.index(n)) < 0:
        raise ValueError("n must be a non-negative integer.")
    if isinstance(t, Time):
        if z.start_time is None:
            raise ValueError("t is a Time object, but signal has no start time.")
        t = (t - z.start_time).to(u.s)
    if hasattr(t, "unit") and hasattr(t, "to_value"):
        t = (t * z.sample_rate).to_value(u.one)
    if np.ndim(t) != 0:
        raise ValueError("t must be a scalar.")
    if (t < 0) or (len(z) < t + n):
        raise ValueError("Requested snippet goes out of bounds.")
    if (i := int(t)) < t:
        shift = i - t
        new_start = None if z.start_time is None else z.start_time - shift * z.dt
        shifted = time_shift(z, shift, crop=True).data
        z = type(z).like(z, shifted, start_time=new_start)
    return z[i:i + n]


def signal_transform(func):
    """Turn an array function ``func(x, **kwargs) -> array`` into a signal transform
    (transforms.py:21-56, without the dask branch: device arrays are passed to ``func`` as they are).

    The wrapper takes a Signal in place of the array and returns ``signal_type.like(signal, result,
    **signal_kwargs)``; ``signal_type`` defaults to the input's type.
    """
    import functools
    from ..core import Signal

    @functools.wraps(func)
    def wrapper(x, *args, signal_type=None, signal_kwargs=None, dask_kwargs=None, **kwargs):
        cls = type(x) if signal_type is None else signal_type
        if not (isinstance(cls, type) and issubclass(cls, Signal)):
            raise TypeError("Signal type must be a subclass of pulsarbat.Signal!")
        return cls.like(x, func(x.data, **kwargs), **(signal_kwargs or {}))

    return wrapper


def concatenate(signals, /, axis=0):
    """Join contiguous signals along time (``axis`` 0 / "time") or another axis (1 / "freq" joins
    frequency-contiguous RadioSignals) -- transforms.py:59-148.  This is what defines the expected
    result of the streaming driver: ``concatenate([coherent_dedispersion(chunk_k, ...)])``.

    Checks, as in the reference: one common type and sample rate; along time the start times must
    follow one another; across another axis they must agree; RadioSignals need a common channel
    bandwidth and, along frequency, adjacent channel grids.  numpy data is joined with numpy,
    device data on the device.
    """
    from ..core import Signal, RadioSignal
    from ..time import Time
    signals = list(signals)
    if not signals:
        raise ValueError("Need at least one signal to concatenate.")
    first = signals[0]
    if not isinstance(first, Signal):
        raise TypeError("Signals must be pulsarbat.Signal objects.")
    if any(type(s) is not type(first) for s in signals):
        raise TypeError("All signals must have same type!")
    sr = first.sample_rate
    if not all(u.isclose(sr, s.sample_rate) for s in signals):
        raise ValueError("Signals must have the same sample_rate!")

    along_time = axis in (0, "time")
    start, offset = None, 0
    for s in signals:
        if s.start_time is not None:
            expected_first = s.start_time - (offset / sr) if along_time else s.start_time
            if start is None:
                start = expected_first
            elif not Time.isclose(start, expected_first):
                raise ValueError("Signals not contiguous in time." if along_time
                                 else "Signals have different start_time.")
        if along_time:
            offset += len(s)
    kw = {"start_time": start}

    if isinstance(first, RadioSignal):
        cbw = first.chan_bw
        if not all(u.isclose(cbw, s.chan_bw) for s in signals):
            raise ValueError("RadioSignals must have the same chan_bw!")
        if axis in (1, "freq"):
            for lo, hi in zip(signals, signals[1:]):
                if not u.isclose(hi.channel_freqs[0] - lo.channel_freqs[-1], cbw):
                    raise ValueError("Signals not contiguous in frequency.")
            f_lo, f_hi = first.channel_freqs[0], signals[-1].channel_freqs[-1]
        else:
            ref = u.to_value(first.channel_freqs, u.Hz)
            for s in signals:
                other = u.to_value(s.channel_freqs, u.Hz)
                if np.shape(other) != np.shape(ref) or not np.allclose(other, ref):
                    raise ValueError("Signals have different frequency channels.")
            f_lo, f_hi = first.channel_freqs[0], first.channel_freqs[-1]
        kw["center_freq"] = (f_lo + f_hi) / 2
        kw["freq_align"] = "center"
    elif axis == "freq":
        raise TypeError("Signals must be pb.RadioSignal objects when axis is 'freq'.")

    ax = 0 if along_time else (1 if axis == "freq" else int(axis))
    if all(isinstance(s.data, DeviceArray) for s in signals):
        import torch
        data = DeviceArray(torch.cat([s.data.tensor for s in signals], dim=ax))
    else:
        data = np.concatenate([np.asarray(s.data) for s in signals], axis=ax)
    return type(first).like(first, data, **kw)
