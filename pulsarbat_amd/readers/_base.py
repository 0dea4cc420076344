"""Reader base class (reference pulsarbat/readers/_base.py:38-360): sample bookkeeping (offset <-> time),
bounds checks, and ``read(offset, n)`` wrapping whatever ``_read_array`` returns in the reader's signal type.
The dask branch of the reference (``use_dask=True`` / ``dask_read``) has no counterpart: reads return numpy or
device arrays, and the keyword is accepted and ignored so that reference call sites keep working."""

import operator

import numpy as np

from .. import units as u
from ..time import Time
from ..core import Signal, _positive_frequency

__all__ = ["BaseReader", "OutOfBoundsError"]


class OutOfBoundsError(EOFError):
    """A position outside the stream was asked for (_base.py:38-41)."""


class BaseReader:
    """``shape`` / ``dtype`` / ``sample_rate`` / ``start_time`` of a stream of samples and ``read(offset, n)``
    (_base.py:44-105).  Subclasses supply ``_read_array(offset, n)``; extra keyword arguments become both
    attributes of the reader and arguments of ``signal_type``."""

    def __init__(self, /, *, shape, dtype, signal_type=Signal, sample_rate, start_time=None, **signal_kwargs):
        if not (isinstance(signal_type, type) and issubclass(signal_type, Signal)):
            raise ValueError("Bad signal_type. Must be Signal or subclass.")
        self._signal_type = signal_type
        self._signal_kwargs = signal_kwargs
        for name, value in signal_kwargs.items():
            setattr(self, name, value)
        self._dtype = np.dtype(dtype)
        self._shape = tuple(operator.index(a) for a in shape)
        if not self._shape:
            raise ValueError("Invalid shape.")
        self.sample_rate = sample_rate
        self.start_time = start_time
        # an empty read now surfaces a reader whose output disagrees with its declared shape / dtype
        z = self.read(0, 0)
        if z.shape != (0,) + self.sample_shape:
            raise ValueError("Provided shape does not match output shape!")
        if z.dtype != self.dtype:
            raise ValueError("Provided dtype does not match output dtype!")

    # ---- description ---------------------------------------------------------------------------------
    def _attr_repr(self):
        st = "N/A" if self.start_time is None else self.start_time.isot
        return f"Start time: {st}\nSample rate: {self.sample_rate}\nTime length: {self.time_length}\n"

    def __str__(self):
        head = f"{type(self).__name__} @ {hex(id(self))}"
        body = f"Data Container: {self._signal_type.__name__}<shape={self.shape}, dtype={self.dtype}>\n"
        return (f"{head}\n{'-' * len(head)}\n{body}{self._attr_repr()}").strip()

    def __repr__(self):
        return (f"{type(self).__name__}<{self._signal_type.__name__}(shape={self.shape}, dtype={self.dtype})>"
                f" @ {hex(id(self))}")

    def __dir__(self):
        return sorted(set(object.__dir__(self)) | set(self._signal_kwargs))

    def __len__(self):
        return self._shape[0]

    shape = property(lambda self: self._shape, doc="Shape of the whole stream.")
    sample_shape = property(lambda self: self._shape[1:], doc="Shape of one sample.")
    ndim = property(lambda self: len(self._shape))
    dtype = property(lambda self: self._dtype)

    # ---- time axis -----------------------------------------------------------------------------------
    @property
    def sample_rate(self):
        return self._sample_rate

    @sample_rate.setter
    def sample_rate(self, sample_rate):
        self._sample_rate = _positive_frequency("sample_rate", sample_rate)

    @property
    def start_time(self):
        return self._start_time

    @start_time.setter
    def start_time(self, start_time):
        try:
            t = None if start_time is None else Time(start_time, format="isot", precision=9)
            assert t is None or t.isscalar
        except Exception:
            raise ValueError("Invalid start_time. Must be a scalar astropy Time object.")
        self._start_time = t

    @property
    def stop_time(self):
        return self.time_at(len(self))

    @property
    def dt(self):
        return (1 / self.sample_rate).to(u.s)

    @property
    def time_length(self):
        return (len(self) / self.sample_rate).to(u.s)

    def contains(self, t, /):
        """Whether time(s) fall in [start, stop), the stop edge excluded up to rounding (_base.py:210-217)."""
        many = isinstance(t, (list, tuple, np.ndarray))
        if self.start_time is None:
            return np.zeros(len(t), bool) if many else False
        t0, t1 = self.start_time, self.stop_time

        def one(x):
            edge = (not x.isclose(t1)) or x.isclose(t0)
            return bool(edge and t0 <= x < t1)
        return np.array([one(x) for x in t]) if many else one(t)

    def __contains__(self, t):
        return self.contains(t)

    def offset_at(self, t, /):
        """Nearest sample offset of an absolute Time or of a time Quantity relative to the start (_base.py:223-247)."""
        if isinstance(t, Time):
            t = t - self.start_time
        offset = int(round(float(u.to_value(t * self.sample_rate, u.one))))
        if offset < 0 or offset > len(self):
            raise OutOfBoundsError("Given time is out of bounds!")
        return offset

    def time_at(self, offset, /, unit=None):
        """Time of a sample offset: a Quantity in ``unit`` from the start, else the absolute Time (_base.py:249-273)."""
        if unit is not None:
            return (offset / self.sample_rate).to(unit)
        if self.start_time is None:
            return None
        return self.start_time + (offset / self.sample_rate)

    # ---- reading -------------------------------------------------------------------------------------
    def _read_array(self, offset, n, /):
        return NotImplemented

    def read(self, offset, n, /, **kwargs):
        """``n`` samples from ``offset`` as a ``signal_type`` whose ``start_time`` is ``time_at(offset)``
        (_base.py:298-333)."""
        kwargs.pop("use_dask", None)
        kwargs.pop("chunks", None)
        if (offset := operator.index(offset)) < 0:
            raise ValueError("offset must be a non-negative int.")
        if (n := operator.index(n)) < 0:
            raise ValueError("n must be a non-negative int.")
        if offset + n > len(self):
            raise OutOfBoundsError("Cannot read beyond end of stream")
        return self._signal_type(self._read_array(offset, n, **kwargs), sample_rate=self.sample_rate,
                                 start_time=self.time_at(offset), **self._signal_kwargs)
