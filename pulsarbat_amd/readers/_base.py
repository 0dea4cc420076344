"""Reader base class: the public contract of the reference's ``pulsarbat.readers.BaseReader``
(pulsarbat/readers/_base.py:44-105 states it: a stream of ``shape[0]`` samples of ``shape[1:]`` / ``dtype`` at
``sample_rate`` starting at ``start_time``; ``read(offset, n)`` returns a ``signal_type``), written around two small
pieces of this build's own: a sample clock (offset <-> time arithmetic) and a stream description.  There is no dask
branch: reads return numpy or device arrays, and the reference's ``use_dask`` / ``chunks`` keywords are accepted and
ignored so that call sites written for the reference keep working."""

import operator

import numpy as np

from .. import units as u
from ..time import Time
from ..core import Signal, _positive_frequency

__all__ = ["BaseReader", "OutOfBoundsError"]


class OutOfBoundsError(EOFError):
    """A position before the first or after the last sample of the stream was asked for."""


class _SampleClock:
    """Sample index <-> time for a uniformly sampled stream: ``rate`` samples per second from ``epoch`` (a Time, or
    None for a stream without absolute times), ``count`` samples long."""

    def __init__(self, rate, epoch, count):
        self.rate, self.epoch, self.count = rate, epoch, count

    def elapsed(self, index, unit=u.s):
        return (index / self.rate).to(unit)

    def instant(self, index):
        return None if self.epoch is None else self.epoch + index / self.rate

    def index_of(self, when):
        """Nearest sample index of an absolute Time or of a duration since the first sample."""
        since = when - self.epoch if isinstance(when, Time) else when
        k = int(round(float(u.to_value(since * self.rate, u.one))))
        if not 0 <= k <= self.count:
            raise OutOfBoundsError("Given time is out of bounds!")
        return k

    def covers(self, when):
        """True inside [first sample, end of stream); the end is excluded even when rounding lands exactly on it."""
        if self.epoch is None:
            return False
        first, end = self.epoch, self.instant(self.count)
        if when.isclose(end) and not when.isclose(first):
            return False
        return bool(first <= when < end)


class BaseReader:
    """A stream of samples that can be read in pieces.

    Subclasses implement ``_read_array(offset, n, /)`` (or override ``read``).  Constructor arguments as in the
    reference: ``shape`` (whole stream), ``dtype``, ``signal_type`` (a Signal subclass, default Signal),
    ``sample_rate`` (frequency Quantity), ``start_time`` (Time or None); any further keyword is kept both as an
    attribute of the reader and as an argument for ``signal_type``."""

    def __init__(self, /, *, shape, dtype, signal_type=Signal, sample_rate, start_time=None, **signal_kwargs):
        if not isinstance(signal_type, type) or not issubclass(signal_type, Signal):
            raise ValueError("Bad signal_type. Must be Signal or subclass.")
        dims = tuple(operator.index(d) for d in shape)
        if len(dims) == 0:
            raise ValueError("Invalid shape.")
        self._signal_type, self._signal_kwargs = signal_type, dict(signal_kwargs)
        self.__dict__.update(self._signal_kwargs)           # center_freq, pol_type, ... readable off the reader
        self._shape, self._dtype = dims, np.dtype(dtype)
        self.sample_rate = sample_rate
        self.start_time = start_time
        self._self_check()

    def _self_check(self):
        """A zero-length read exposes a subclass whose output disagrees with what it declared, at construction."""
        probe = self.read(0, 0)
        if tuple(probe.shape) != (0,) + self.sample_shape:
            raise ValueError("Provided shape does not match output shape!")
        if probe.dtype != self._dtype:
            raise ValueError("Provided dtype does not match output dtype!")

    # ---- what the stream is --------------------------------------------------------------------------------
    @property
    def shape(self):
        """Shape of the whole stream, time first."""
        return self._shape

    @property
    def sample_shape(self):
        """Shape of one time sample."""
        return self._shape[1:]

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def dtype(self):
        return self._dtype

    def __len__(self):
        return self._shape[0]

    @property
    def sample_rate(self):
        return self._sample_rate

    @sample_rate.setter
    def sample_rate(self, value):
        self._sample_rate = _positive_frequency("sample_rate", value)

    @property
    def start_time(self):
        return self._start_time

    @start_time.setter
    def start_time(self, value):
        if value is None:
            self._start_time = None
            return
        try:
            stamp = Time(value, format="isot", precision=9)
            scalar = stamp.isscalar
        except Exception:
            scalar = False
        if not scalar:
            raise ValueError("Invalid start_time. Must be a scalar astropy Time object.")
        self._start_time = stamp

    @property
    def _clock(self):
        return _SampleClock(self._sample_rate, self._start_time, self._shape[0])

    @property
    def stop_time(self):
        """Time just after the last sample (None without a start_time)."""
        return self._clock.instant(len(self))

    @property
    def dt(self):
        return self._clock.elapsed(1)

    @property
    def time_length(self):
        return self._clock.elapsed(len(self))

    # ---- time <-> offset -----------------------------------------------------------------------------------
    def time_at(self, offset, /, unit=None):
        """Time of sample ``offset``: with ``unit`` the duration since the start, otherwise the absolute Time."""
        return self._clock.instant(offset) if unit is None else self._clock.elapsed(offset, unit)

    def offset_at(self, t, /):
        """Nearest sample offset to ``t`` (a Time, or a time Quantity counted from the start)."""
        return self._clock.index_of(t)

    def contains(self, t, /):
        """Whether ``t`` (one Time, or a sequence of them) lies within the stream."""
        clock = self._clock
        if isinstance(t, (list, tuple, np.ndarray)):
            return np.fromiter((clock.covers(x) for x in t), dtype=bool, count=len(t))
        return clock.covers(t)

    __contains__ = contains

    # ---- reading -------------------------------------------------------------------------------------------
    def _read_array(self, offset, n, /):
        return NotImplemented

    def read(self, offset, n, /, **kwargs):
        """``n`` samples from sample ``offset``, as a ``signal_type`` that starts at ``time_at(offset)``."""
        for ignored in ("use_dask", "chunks"):
            kwargs.pop(ignored, None)
        first, count = operator.index(offset), operator.index(n)
        if first < 0:
            raise ValueError("offset must be a non-negative int.")
        if count < 0:
            raise ValueError("n must be a non-negative int.")
        if first + count > len(self):
            raise OutOfBoundsError("Cannot read beyond end of stream")
        data = self._read_array(first, count, **kwargs)
        return self._signal_type(data, sample_rate=self._sample_rate, start_time=self.time_at(first),
                                 **self._signal_kwargs)

    # ---- presentation --------------------------------------------------------------------------------------
    def _container(self):
        return f"{self._signal_type.__name__}<shape={self.shape}, dtype={self.dtype}>"

    def _attr_repr(self):
        began = "N/A" if self._start_time is None else self._start_time.isot
        return "".join(f"{label}: {value}\n" for label, value in
                       (("Start time", began), ("Sample rate", self.sample_rate), ("Time length", self.time_length)))

    def __str__(self):
        title = f"{type(self).__name__} @ {hex(id(self))}"
        return f"{title}\n{'-' * len(title)}\nData Container: {self._container()}\n{self._attr_repr()}".strip()

    def __repr__(self):
        kind = self._signal_type.__name__
        return f"{type(self).__name__}<{kind}(shape={self.shape}, dtype={self.dtype})> @ {hex(id(self))}"

    def __dir__(self):
        return sorted({*object.__dir__(self), *self._signal_kwargs})
