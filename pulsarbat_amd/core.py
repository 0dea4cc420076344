"""Signal containers: the API surface the hot path lives behind.

Same class names, constructor signatures, validation errors and slicing /
``like()`` semantics as the reference containers (pulsarbat/core.py:12-19), so
``pb.coherent_dedispersion`` is a drop-in.  Only what the hot path touches is
here; dask helpers (core.py:298-345) are reduced to ``compute``/``persist`` with
device <-> host meaning (chunk <-> GPU shard, ``persist()`` <->
results left resident on the device, ``compute()`` <-> copy to host).

``Signal.data`` is duck-typed exactly as in the reference (core.py:59-97):
anything with ``ndim/shape/dtype/astype/__getitem__/__len__`` works, which is
the seam where :class:`pulsarbat_amd.device.DeviceArray` lives.
"""

import inspect
import operator
import pprint

import numpy as np

from . import units as u
from .time import Time

__all__ = [
    "Signal",
    "RadioSignal",
    "IntensitySignal",
    "FullStokesSignal",
    "BasebandSignal",
    "DualPolarizationSignal",
]


class InvalidSignalError(ValueError):
    """Raised for data that cannot form the requested signal (core.py:22-25)."""


def _is_device(x):
    from .device import DeviceArray
    return isinstance(x, DeviceArray)


def _positive_frequency(name, value, positive=True):
    """Validation shared by sample_rate / center_freq / chan_bw setters
    (core.py:237-248, 510-521, 533-544): must be a scalar frequency quantity."""
    try:
        hz = u.to_value(value.to(u.Hz) if isinstance(value, u.Quantity) else value, u.Hz)
        if not hasattr(value, "to"):
            raise TypeError
        ok = np.ndim(hz) == 0 and (hz > 0 or not positive)
    except Exception:
        ok = False
    if not ok:
        what = "a positive scalar" if positive else "a scalar"
        raise ValueError(f"Invalid {name}. Must be {what} Quantity with units of Hz or equivalent.")
    return value


class Signal(np.lib.mixins.NDArrayOperatorsMixin):
    """Base signal: samples ``z`` with time on axis 0, plus ``sample_rate``,
    optional ``start_time`` and ``meta`` (core.py:28-97)."""

    _req_dtype = ()
    _req_shape = (None,)
    _axes_labels = {"time": 0}

    def __init__(self, z, /, *, sample_rate, start_time=None, meta=None):
        need = len(self._req_shape)
        if z.ndim < need:
            raise InvalidSignalError(
                f"Expected signal with at least {need} dimension(s), "
                f"got signal with {z.ndim} dimension(s) instead.")
        for got, want in zip(z.shape[:need], self._req_shape):
            if want is not None and got != want:
                raise InvalidSignalError(
                    f"Signal has invalid shape. Expected {self._req_shape}, got {z.shape} instead.")
        if int(np.prod(z.shape[1:])) == 0:
            raise InvalidSignalError("Sample shape must have non-zero size!")

        self._data = self._coerce_dtype(z)
        self.sample_rate = sample_rate
        self.start_time = start_time
        self.meta = meta

    def _coerce_dtype(self, z):
        """dtype rule of core.py:78-92: keep if allowed, else safe-cast to the first
        required dtype, else InvalidSignalError."""
        if not self._req_dtype or z.dtype in self._req_dtype:
            return z
        try:
            return z.astype(self._req_dtype[0], casting="safe")
        except TypeError:
            raise InvalidSignalError(f"Invalid dtype. Expected {self._req_dtype}, got {z.dtype}.")

    # --- numpy protocol (core.py:99-122, 152-153) -----------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        if method != "__call__" or ufunc is np.matmul:
            return NotImplemented
        unwrap = lambda a: a.data if isinstance(a, Signal) else a
        outs = (None,) * ufunc.nout if out is None else out
        res = ufunc(*map(unwrap, inputs), out=tuple(map(unwrap, outs)), **kwargs)
        if res is NotImplemented:
            return NotImplemented
        res = (res,) if ufunc.nout == 1 else res
        wrapped = tuple(type(self).like(self, r) if o is None else o for r, o in zip(res, outs))
        return wrapped[0] if len(wrapped) == 1 else wrapped

    def __array__(self, dtype=None, copy=None):
        a = np.asanyarray(self.data)
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return len(self.data)

    # --- printing: the text layout is the reference's (core.py:124-147), the assembly is ours --------------
    def _attr_lines(self):
        """(label, value) pairs shown under the header; subclasses extend the list."""
        began = self.start_time.isot if self.start_time is not None else "N/A"
        return [("Sample rate", self.sample_rate), ("Time length", self.time_length), ("Start time", began)]

    def _attr_repr(self):
        return "".join(f"{label}: {value}\n" for label, value in self._attr_lines())

    def __str__(self):
        title = f"{type(self).__name__} @ {hex(id(self))}"
        holder = type(self.data)
        parts = [title, "-" * len(title),
                 f"Data Container: {holder.__module__}.{holder.__name__}<shape={self.shape}, dtype={self.dtype}>",
                 self._attr_repr().rstrip("\n")]
        if self.meta is not None:
            parts += ["", "Meta", "----", pprint.pformat(self.meta, sort_dicts=False, depth=2)]
        return "\n".join(parts).strip()

    def __repr__(self):
        return f"pulsarbat.{type(self).__name__}<shape={self.shape}, dtype={self.dtype}> @ {hex(id(self))}"

    # --- slicing: what a time slice does to the metadata (core.py:155-176) ---------------------------------
    def _time_slice(self, index):
        first, _, step = index.indices(self.shape[0])
        assert step > 0, "Time axis slicing does not support negative step"
        changes = {}
        if step != 1:                       # every step-th sample: a lower sample rate
            changes["sample_rate"] = self.sample_rate / step
        if self.start_time is not None:     # the first kept sample sets the new start
            changes["start_time"] = self.start_time + first / self.sample_rate
        return changes

    _sliceable_axes = 1

    def _slice_kwargs(self, index):
        return self._time_slice(index[0])

    def __getitem__(self, index):
        index = index if isinstance(index, tuple) else (index,)
        n = self._sliceable_axes
        if any(not isinstance(a, slice) for a in index[:n]):
            names = "time axis" if n == 1 else "time and frequency axes"
            raise IndexError(f"Only supports slicing on {names}.")
        return type(self).like(self, self.data[index], **self._slice_kwargs(index))

    def get_axis(self, axis):
        """Axis number of an integer (negative counts from the end) or of an axis label."""
        if isinstance(axis, str) or not hasattr(axis, "__index__"):
            number = self.axes_labels.get(axis)
        else:
            number = operator.index(axis)
        if number is None or number >= self.ndim or number < -self.ndim:
            raise ValueError("Invalid axis.")
        return number

    # --- attributes ---------------------------------------------------------------------
    @property
    def axes_labels(self):
        return self._axes_labels

    @property
    def meta(self):
        return self._meta

    @meta.setter
    def meta(self, value):
        if value is not None:
            try:
                value = dict(value)
            except (TypeError, ValueError):
                raise ValueError("meta must be a dict.") from None
        self._meta = value

    @property
    def data(self):
        return self._data

    @property
    def shape(self):
        return self.data.shape

    @property
    def sample_shape(self):
        return self.shape[1:]

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def sample_rate(self):
        return self._sample_rate

    @sample_rate.setter
    def sample_rate(self, sample_rate):
        self._sample_rate = _positive_frequency("sample_rate", sample_rate)

    @property
    def dt(self):
        """Sample spacing ``1 / sample_rate`` in seconds (core.py:250-253)."""
        return (1 / self.sample_rate).to(u.s)

    @property
    def time_length(self):
        return (len(self) / self.sample_rate).to(u.s)

    @property
    def start_time(self):
        return self._start_time

    @start_time.setter
    def start_time(self, start_time):
        try:
            if start_time is None:
                t = None
            elif hasattr(start_time, "jd1") and hasattr(start_time, "jd2"):
                t = start_time  # astropy Time supplied by a caller that has astropy
                assert t.isscalar
            else:
                t = Time(start_time, format="isot", precision=9)
        except Exception:
            raise ValueError("Invalid start_time. Must be a scalar astropy Time object.")
        self._start_time = t

    @property
    def stop_time(self):
        if self.start_time is None:
            return None
        return self.start_time + self.time_length

    def contains(self, t, /):
        if self.start_time is None:
            return False
        return bool(self.start_time <= t < self.stop_time)

    def __contains__(self, t):
        return self.contains(t)

    # --- host/device residency (reference: dask compute/persist, core.py:298-322) -----------
    def compute(self, **kwargs):
        """Signal with data materialised on the host (device -> numpy copy)."""
        return type(self).like(self, np.asarray(self.data))

    def persist(self, **kwargs):
        """Signal with data left where it is (device-resident stays resident)."""
        return type(self).like(self, self.data)

    def to_device(self, device=None, series_major=False):
        """Signal whose data lives in HBM as a :class:`DeviceArray`.

        ``series_major=True`` stores it with time as the fastest axis (same shape, other strides): the
        layout the column passes of ``coherent_dedispersion`` work in, so device-resident pipelines that
        keep their arrays this way skip the two layout passes."""
        from .device import DeviceArray
        d = DeviceArray.from_host(self.data, device=device)
        if series_major and d.ndim >= 2:
            d = d.to_series_major()
        return type(self).like(self, d)

    @classmethod
    def like(cls, obj, z=None, /, **kwargs):
        """Build a ``cls`` signal taking every constructor argument not given in
        ``kwargs`` from the same-named attribute of ``obj`` (core.py:347-379)."""
        for name, p in inspect.signature(cls).parameters.items():
            if p.kind is p.POSITIONAL_ONLY or name in kwargs:
                continue
            if hasattr(obj, name):
                kwargs[name] = getattr(obj, name)
            elif p.default is p.empty:
                raise ValueError(f"Missing required keyword argument: {name}")
        return cls(obj.data if z is None else z, **kwargs)


class RadioSignal(Signal):
    """Heterodyned signal ``(nsample, nchan, ...)`` with ``center_freq``, ``chan_bw``
    and ``freq_align`` (core.py:382-574).  Channel i sits at
    ``center_freq + chan_bw * (i + a - nchan/2)``, a = 0 / 0.5 / 1 for
    bottom / center / top alignment."""

    _req_shape = (None, None)
    _axes_labels = {"time": 0, "freq": 1}
    _sliceable_axes = 2

    def __init__(self, z, /, *, sample_rate, start_time=None, center_freq, chan_bw,
                 freq_align="center", meta=None):
        super().__init__(z, sample_rate=sample_rate, start_time=start_time, meta=meta)
        self.center_freq = center_freq
        self.chan_bw = chan_bw
        self.freq_align = freq_align

    def _attr_repr(self):
        return (super()._attr_repr()
                + f"Channel Bandwidth: {self.chan_bw}\n"
                + f"Total Bandwidth: {self.bandwidth}\n"
                + f"Center Frequency: {self.center_freq}\n")

    def _freq_slice(self, index):
        sl = slice(*index.indices(self.shape[1]))
        assert sl.step == 1, "Does not support slice step for frequency axis"
        assert sl.stop > sl.start, "Empty frequency slice!"
        f = self.channel_freqs[sl]
        return {"center_freq": (f[0] + f[-1]) / 2, "freq_align": "center"}

    def _slice_kwargs(self, index):
        kw = self._time_slice(index[0])
        if len(index) > 1:
            kw.update(self._freq_slice(index[1]))
        return kw

    @property
    def nchan(self):
        return self.shape[self.get_axis("freq")]

    @property
    def center_freq(self):
        return self._center_freq

    @center_freq.setter
    def center_freq(self, center_freq):
        self._center_freq = _positive_frequency("center_freq", center_freq, positive=False)

    @property
    def chan_bw(self):
        return self._chan_bw

    @chan_bw.setter
    def chan_bw(self, chan_bw):
        self._chan_bw = _positive_frequency("chan_bw", chan_bw)

    @property
    def bandwidth(self):
        return self.chan_bw * self.nchan

    @property
    def max_freq(self):
        return self.center_freq + self.bandwidth / 2

    @property
    def min_freq(self):
        return self.center_freq - self.bandwidth / 2

    @property
    def freq_align(self):
        return self._freq_align

    @freq_align.setter
    def freq_align(self, freq_align):
        if freq_align not in {"bottom", "center", "top"}:
            raise ValueError("Invalid freq_align. Expected: {'bottom', 'center', 'top'}")
        self._freq_align = "center" if self.nchan % 2 else freq_align

    @property
    def channel_freqs(self):
        offset = {"bottom": 0, "center": 0.5, "top": 1}[self.freq_align]
        ids = np.arange(self.nchan) + offset - self.nchan / 2
        return self.center_freq + self.chan_bw * ids


class IntensitySignal(RadioSignal):
    """Real-valued intensities (core.py:577-611)."""

    _req_dtype = (np.float64, np.float32)


class FullStokesSignal(IntensitySignal):
    """``(nsample, nchan, 4, ...)`` Stokes [I, Q, U, V], PSR/IEEE convention
    (core.py:614-701)."""

    _req_shape = (None, None, 4)
    _axes_labels = {"time": 0, "freq": 1, "pol": 2}
    _stokes_ids = {"I": 0, "Q": 1, "U": 2, "V": 3}

    def __getitem__(self, key):
        if not isinstance(key, str):
            return super().__getitem__(key)
        if key not in self._stokes_ids:
            raise KeyError("Invalid key. Should be in {'I', 'Q', 'U', 'V'}.")
        x = np.take(np.asarray(self.data), self._stokes_ids[key], axis=self.get_axis("pol"))
        return IntensitySignal.like(self, x)

    stokesI = property(lambda self: self["I"])
    stokesQ = property(lambda self: self["Q"])
    stokesU = property(lambda self: self["U"])
    stokesV = property(lambda self: self["V"])


class BasebandSignal(RadioSignal):
    """Complex baseband (Nyquist-sampled analytic) signal: ``chan_bw == sample_rate``
    (core.py:704-774)."""

    _req_dtype = (np.complex128, np.complex64)

    def __init__(self, z, /, *, sample_rate, start_time=None, center_freq,
                 freq_align="center", meta=None):
        super().__init__(z, sample_rate=sample_rate, start_time=start_time,
                         center_freq=center_freq, chan_bw=sample_rate,
                         freq_align=freq_align, meta=meta)

    def to_intensity(self):
        """``re^2 + im^2`` per element (core.py:766-774).  Device-resident data is
        detected by the HIP kernel; host arrays keep numpy container semantics."""
        if _is_device(self.data):
            from . import _hip
            z = _hip.detect(self.data, mode="intensity")
        else:
            z = self.data.real ** 2 + self.data.imag ** 2
        return IntensitySignal.like(self, z)


class DualPolarizationSignal(BasebandSignal):
    """``(nsample, nchan, 2, ...)`` baseband with ``pol_type`` 'linear' [X, Y] or
    'circular' [L, R]; L = X - iY, R = X + iY (core.py:777-966)."""

    _req_shape = (None, None, 2)
    _axes_labels = {"time": 0, "freq": 1, "pol": 2}

    def __init__(self, z, /, *, sample_rate, start_time=None, center_freq,
                 freq_align="center", pol_type, meta=None):
        super().__init__(z, sample_rate=sample_rate, start_time=start_time,
                         center_freq=center_freq, freq_align=freq_align, meta=meta)
        self.pol_type = pol_type

    def _attr_repr(self):
        basis = {"linear": "[X, Y]", "circular": "[L, R]"}[self.pol_type]
        return super()._attr_repr() + f"Polarization Type: {self.pol_type} {basis}\n"

    @property
    def pol_type(self):
        return self._pol_type

    @pol_type.setter
    def pol_type(self, pol_type):
        if pol_type not in {"linear", "circular"}:
            raise ValueError("pol_type must be in {'linear', 'circular'}")
        self._pol_type = pol_type

    def _pols(self):
        ax = self.get_axis("pol")
        d = np.asarray(self.data)
        return ax, np.take(d, 0, axis=ax), np.take(d, 1, axis=ax)

    def to_linear(self):
        """core.py:882-904: X = (L + R)/sqrt2, Y = i(L - R)/sqrt2."""
        z = self.data
        if self.pol_type == "circular":
            if _is_device(z) and self.ndim == 3:
                from . import _hip
                z = _hip.pol_basis(z, to_circular=False)
            else:
                ax, L, R = self._pols()
                z = np.stack([L + R, 1j * (L - R)], axis=ax) / np.sqrt(2)
        return type(self).like(self, z, pol_type="linear")

    def to_circular(self):
        """core.py:906-928: L = (X - iY)/sqrt2, R = (X + iY)/sqrt2."""
        z = self.data
        if self.pol_type == "linear":
            if _is_device(z) and self.ndim == 3:
                from . import _hip
                z = _hip.pol_basis(z, to_circular=True)
            else:
                ax, X, Y = self._pols()
                z = np.stack([X - 1j * Y, X + 1j * Y], axis=ax) / np.sqrt(2)
        return type(self).like(self, z, pol_type="circular")

    def to_stokes(self):
        """IQUV (core.py:930-966).  linear: I=XX+YY, Q=XX-YY, U=2Re(X*Y), V=2Im(X*Y);
        circular: I=LL+RR, Q=2Re(L*R), U=2Im(L*R), V=LL-RR."""
        if _is_device(self.data):
            from . import _hip
            return FullStokesSignal.like(self, _hip.detect(self.data, mode=self.pol_type))
        ax, A, B = self._pols()
        AA = A.real ** 2 + A.imag ** 2
        BB = B.real ** 2 + B.imag ** 2
        AB = A.conj() * B
        if self.pol_type == "linear":
            comps = [AA + BB, AA - BB, 2 * AB.real, 2 * AB.imag]
        else:
            comps = [AA + BB, 2 * AB.real, 2 * AB.imag, AA - BB]
        return FullStokesSignal.like(self, np.stack(comps, axis=ax))
