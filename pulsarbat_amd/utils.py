"""Handy utilities (reference pulsarbat/utils.py): ``real_to_complex`` on the HIP pipeline
(SURVEY.md 8f rank 4), ``next_fast_len`` / ``prev_fast_len`` as plain host arithmetic."""

from functools import lru_cache

import numpy as np

from . import _hip
from .device import DeviceArray

__all__ = ["real_to_complex", "next_fast_len", "prev_fast_len", "next_native_len", "prev_native_len"]


def real_to_complex(z, axis=0):
    """Complex baseband representation of a real baseband signal (utils.py:15-65): analytic signal
    via the FFT (Hilbert weights), shift by -B/2, decimate by 2.  ``float32 -> complex64``,
    ``float64 -> complex128``.  numpy in -> numpy out, DeviceArray in -> DeviceArray out; the
    arithmetic is the plan pipeline (FFT * h * IFFT) plus one decimating kernel.
    """
    on_dev = isinstance(z, DeviceArray)
    if not on_dev:
        z = np.asarray(z)
    if np.dtype(z.dtype).kind == "c":
        raise ValueError("Input must be real-valued.")
    out_dtype = np.complex64 if z.dtype == np.float32 else np.complex128
    N = z.shape[axis]
    if N == 0:
        return z.astype(out_dtype)
    if N == 1:  # fft of one sample is itself, h = [1], no decimation
        return z.astype(out_dtype)
    # time on axis 0, everything else flattened into series (plumbing: views / one copy)
    if on_dev:
        t = z.tensor.movedim(axis, 0)
        lead = tuple(t.shape[1:])
        # float32 data whose half length is a power of two beyond one tile: a HALF-length complex transform (the real series,
        # time fastest, is the complex series x[2m] + i x[2m+1]) instead of two full-length ones on a complex copy
        half = _hip.real_to_complex_half(DeviceArray(t.reshape(N, -1).contiguous())) if z.dtype == np.float32 else None
        if half is not None:
            return DeviceArray(half.tensor.reshape((N // 2,) + lead).movedim(0, axis).contiguous())
        x = DeviceArray(t.reshape(N, -1).contiguous()).astype(out_dtype)
    else:
        a = np.moveaxis(z, axis, 0)
        lead = a.shape[1:]
        x = DeviceArray.from_host(np.ascontiguousarray(a.reshape(N, -1)).astype(out_dtype))
    S = x.shape[1]
    plan = _hip.filter_plan(N, S, 0, N, x.device_index, out_dtype, shared=True)   # the same Hilbert weights for all
    plan.chirp_special(np.zeros(1), 2)
    y = plan.dedisperse(DeviceArray(x.tensor.reshape(N, 1, S)))
    out = _hip.decimate2(DeviceArray(y.tensor.reshape(N, S)))
    nout = out.shape[0]
    if on_dev:
        return DeviceArray(out.tensor.reshape((nout,) + lead).movedim(0, axis).contiguous())
    return np.ascontiguousarray(np.moveaxis(out.get().reshape((nout,) + tuple(lead)), 0, axis))


def _smooth_7(limit):
    """All 7-smooth numbers <= limit, ascending."""
    vals = {1}
    for p in (2, 3, 5, 7):
        new = set()
        for v in vals:
            while v * p <= limit:
                v *= p
                new.add(v)
        vals |= new
    return sorted(vals)


@lru_cache(maxsize=None)
def next_fast_len(target):
    """Smallest 7-smooth number >= target (utils.py:68-97)."""
    target = int(target)
    if target < 1:
        raise ValueError("target must be a positive integer")
    limit = 1
    while limit < target:
        limit *= 2
    return next(v for v in _smooth_7(limit) if v >= target)


@lru_cache(maxsize=None)
def prev_fast_len(target):
    """Largest 7-smooth number <= target (utils.py:100-130)."""
    target = int(target)
    if target < 1:
        raise ValueError("target must be a positive integer")
    return _smooth_7(target)[-1]


def _native_lens(limit):
    """Lengths the HIP pipeline transforms without a convolution detour ("Lengths"), ascending, <= limit:
    powers of two from 32; m * 2^k with m in (3, 5, 7) and 2^19 <= 2^k <= 2^24 (complex64 tile sizes); and the 7-smooth
    lengths q * 2^k whose odd-and-beyond part q <= 1024 fits one mixed-radix column pass (5 <= k <= 14, or k = 14 with an
    even q): about 70 % of the power-of-two rate, against 50 % for a length that needs the padded convolution."""
    vals = set()
    v = 32
    while v <= limit and v <= 1 << 28:
        vals.add(v)
        v *= 2
    for m in (3, 5, 7):
        for k in range(19, 25):
            if m << k <= limit:
                vals.add(m << k)
    for q in _smooth_7(1024):
        if q < 2:
            continue
        for k in (range(5, 15) if q % 2 else (14,)):
            if q << k <= min(limit, 1 << 27):
                vals.add(q << k)
    return sorted(vals)


def next_native_len(target):
    """Smallest length >= target that runs natively on the device (the analogue of ``next_fast_len`` for this
    build: 7-smooth lengths in general go through one padded convolution, about 2x the native cost)."""
    target = int(target)
    if target < 1:
        raise ValueError("target must be a positive integer")
    limit = 32
    while limit < target:
        limit *= 2
    return next(v for v in _native_lens(limit) if v >= target)


def prev_native_len(target):
    """Largest natively transformed length <= target (at least 32)."""
    target = int(target)
    if target < 32:
        raise ValueError("target must be at least 32")
    return _native_lens(target)[-1]
