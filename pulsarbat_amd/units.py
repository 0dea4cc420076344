"""Minimal physical-units layer for the host API.

The reference attaches ``astropy.units`` Quantities to every signal attribute
(``sample_rate``, ``center_freq``, ``chan_bw``: pulsarbat/core.py:237-248,
510-544) and validates them with ``.to(u.Hz)``.  astropy is not installed on the
build or GPU boxes and cannot be, so this module provides the small subset the
hot path needs, with the same spelling: ``1 * u.MHz``, ``q.to(u.Hz)``,
``q.to_value(u.s)``, ``u.isclose``.  astropy Quantities are also accepted
wherever a quantity is expected: every consumer goes through :func:`to_value`,
which falls back to the astropy string-unit API (``q.to_value("Hz")``).

Dimensions tracked: time, length, angle (cycle/rad).  Nothing else is needed
for frequencies, delays and dispersion measures.
"""

import math
import numbers

import numpy as np

__all__ = [
    "Unit", "Quantity", "to_value", "isclose", "one", "dimensionless_unscaled",
    "s", "ms", "us", "ns", "Hz", "kHz", "MHz", "GHz", "m", "cm", "pc",
    "rad", "cycle",
]


class UnitConversionError(ValueError):
    pass


class Unit:
    """A scale factor to SI plus integer exponents for (time, length, angle)."""

    __slots__ = ("scale", "dims", "name")
    __array_priority__ = 20000

    def __init__(self, scale, dims, name=None):
        self.scale = float(scale)
        self.dims = tuple(dims)
        self.name = name

    # --- algebra between units -------------------------------------------
    def _combine(self, other, sign):
        dims = tuple(a + sign * b for a, b in zip(self.dims, other.dims))
        scale = self.scale * other.scale if sign > 0 else self.scale / other.scale
        if other.dims == (0, 0, 0) and other.scale == 1.0:
            return Unit(scale, dims, self.name)
        if sign > 0 and self.dims == (0, 0, 0) and self.scale == 1.0:
            return Unit(scale, dims, other.name)
        op = " " if sign > 0 else " / "
        return Unit(scale, dims, f"{self}{op}{other}")

    def __mul__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, +1)
        if isinstance(other, Quantity):
            return Quantity(other.value, self * other.unit)
        return Quantity(other, self)

    def __rmul__(self, other):
        return Quantity(other, self)

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, -1)
        if isinstance(other, Quantity):
            return Quantity(1.0 / other.value, self / other.unit)
        return Quantity(1.0 / np.asarray(other, dtype=float), self)

    def __rtruediv__(self, other):
        return Quantity(other, Unit(1.0 / self.scale, tuple(-d for d in self.dims),
                                    f"1 / ({self})"))

    def __pow__(self, p):
        return Unit(self.scale ** p, tuple(d * p for d in self.dims), f"({self})^{p}")

    def __eq__(self, other):
        return (isinstance(other, Unit) and self.dims == other.dims
                and math.isclose(self.scale, other.scale, rel_tol=1e-15))

    def __hash__(self):
        return hash(self.dims)

    def __repr__(self):
        return self.name if self.name is not None else f"Unit({self.scale:g}, {self.dims})"

    def is_equivalent(self, other):
        return self.dims == _as_unit(other).dims


_T, _L, _A = (1, 0, 0), (0, 1, 0), (0, 0, 1)
one = dimensionless_unscaled = Unit(1.0, (0, 0, 0), "")
s = Unit(1.0, _T, "s")
ms = Unit(1e-3, _T, "ms")
us = Unit(1e-6, _T, "us")
ns = Unit(1e-9, _T, "ns")
Hz = Unit(1.0, (-1, 0, 0), "Hz")
kHz = Unit(1e3, (-1, 0, 0), "kHz")
MHz = Unit(1e6, (-1, 0, 0), "MHz")
GHz = Unit(1e9, (-1, 0, 0), "GHz")
m = Unit(1.0, _L, "m")
cm = Unit(1e-2, _L, "cm")
pc = Unit(3.0856775814913673e16, _L, "pc")
rad = Unit(1.0, _A, "rad")
cycle = Unit(2.0 * math.pi, _A, "cycle")

_BY_NAME = {u.name: u for u in (s, ms, us, ns, Hz, kHz, MHz, GHz, m, cm, pc, rad, cycle)}
_BY_NAME[""] = _BY_NAME["one"] = one


def _as_unit(u):
    if isinstance(u, Unit):
        return u
    if isinstance(u, str):
        try:
            return _BY_NAME[u]
        except KeyError:
            raise UnitConversionError(f"unknown unit {u!r}")
    raise TypeError(f"not a unit: {u!r}")


class Quantity:
    """``value`` (float or ndarray) times a :class:`Unit`."""

    __slots__ = ("value", "unit")
    __array_priority__ = 10000
    __array_ufunc__ = None  # make ndarray defer to our reflected operators

    def __init__(self, value, unit=one):
        if isinstance(value, Quantity):
            value = value.to_value(unit)
        if isinstance(value, (numbers.Real, np.generic)):
            self.value = float(value)
        else:
            self.value = np.asarray(value, dtype=float)
        self.unit = _as_unit(unit)

    # --- conversion --------------------------------------------------------
    def to(self, unit):
        unit = _as_unit(unit)
        if unit.dims != self.unit.dims:
            raise UnitConversionError(f"'{self.unit}' and '{unit}' are not convertible")
        if unit.scale == self.unit.scale:
            return Quantity(self.value, unit)
        return Quantity(self.value * (self.unit.scale / unit.scale), unit)

    def to_value(self, unit=None):
        return self.value if unit is None else self.to(unit).value

    @property
    def si(self):
        return self.value * self.unit.scale

    @property
    def isscalar(self):
        return np.ndim(self.value) == 0

    @property
    def shape(self):
        return np.shape(self.value)

    def __len__(self):
        return len(self.value)

    def __iter__(self):
        for v in self.value:
            yield Quantity(v, self.unit)

    def __getitem__(self, i):
        return Quantity(self.value[i], self.unit)

    def round(self):
        return Quantity(np.round(self.value), self.unit)

    # --- arithmetic -----------------------------------------------------------
    @staticmethod
    def _coerce(other):
        if isinstance(other, Quantity):
            return other
        if isinstance(other, Unit):
            return Quantity(1.0, other)
        if hasattr(other, "unit") and hasattr(other, "to_value"):  # astropy Quantity
            return Quantity(other.to_value(other.unit), _from_foreign_unit(other))
        return Quantity(other, one)

    def _same(self, other):
        other = self._coerce(other)
        if other.unit.dims != self.unit.dims:
            # numbers (incl. inf / 0) are allowed against dimensionless only
            raise UnitConversionError(
                f"can only combine quantities of equal dimensions "
                f"('{self.unit}' vs '{other.unit}')")
        return other.to_value(self.unit)

    def __add__(self, other):
        return Quantity(self.value + self._same(other), self.unit)

    __radd__ = __add__

    def __sub__(self, other):
        return Quantity(self.value - self._same(other), self.unit)

    def __rsub__(self, other):
        return Quantity(self._same(other) - self.value, self.unit)

    def __neg__(self):
        return type(self)(-self.value, self.unit)

    def __pos__(self):
        return self

    def __abs__(self):
        return Quantity(abs(self.value), self.unit)

    def __mul__(self, other):
        o = self._coerce(other)
        return Quantity(self.value * o.value, self.unit * o.unit)

    __rmul__ = __mul__

    def __truediv__(self, other):
        o = self._coerce(other)
        return Quantity(self.value / o.value, self.unit / o.unit)

    def __rtruediv__(self, other):
        o = self._coerce(other)
        with np.errstate(divide="ignore"):
            return Quantity(o.value / self.value, o.unit / self.unit)

    def __pow__(self, p):
        return Quantity(self.value ** p, self.unit ** p)

    # --- comparisons ---------------------------------------------------------
    def _cmp(self, other, op):
        return op(self.value, self._same(other))

    def __eq__(self, other):
        try:
            return self._cmp(other, np.equal)
        except (UnitConversionError, TypeError):
            return False

    def __ne__(self, other):
        r = self.__eq__(other)
        return np.logical_not(r)

    def __lt__(self, other):
        return self._cmp(other, np.less)

    def __le__(self, other):
        return self._cmp(other, np.less_equal)

    def __gt__(self, other):
        return self._cmp(other, np.greater)

    def __ge__(self, other):
        return self._cmp(other, np.greater_equal)

    def __hash__(self):
        return hash((self.si if self.isscalar else None, self.unit.dims))

    def __float__(self):
        if self.unit.dims != (0, 0, 0):
            raise TypeError("only dimensionless quantities convert to float")
        return float(self.value * self.unit.scale)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.value, dtype=dtype)

    def __repr__(self):
        return f"<Quantity {self.value} {self.unit}>"

    def __str__(self):
        return f"{self.value} {self.unit}"

    def __reduce__(self):
        return (Quantity, (self.value, Unit(self.unit.scale, self.unit.dims, self.unit.name)))


def _from_foreign_unit(q):
    """Map an astropy quantity's unit onto ours by probing the few base units we know."""
    for name in ("Hz", "s", "rad", "m", ""):
        try:
            scale = (1.0 * q.unit).to_value(name or "")
        except Exception:
            continue
        base = _BY_NAME[name]
        return Unit(scale * base.scale, base.dims, str(q.unit))
    raise UnitConversionError(f"unsupported foreign unit {q.unit!r}")


def to_value(q, unit):
    """Value of ``q`` in ``unit`` (a :class:`Unit` or its name).

    Accepts our Quantity, an astropy Quantity (string-unit API), or a bare
    number when ``unit`` is dimensionless / the number is inf (the reference
    passes ``numpy.inf`` as a frequency: tests/test_dedispersion.py:17-21).
    """
    unit = _as_unit(unit)
    if isinstance(q, Quantity):
        return q.to_value(unit)
    if hasattr(q, "to_value") and hasattr(q, "unit"):
        return q.to_value(unit.name)
    arr = np.asarray(q, dtype=float)
    if unit.dims == (0, 0, 0) or np.all(np.isinf(arr)):
        return float(arr) if arr.ndim == 0 else arr
    raise UnitConversionError(f"expected a quantity convertible to '{unit}', got {q!r}")


def isclose(a, b, rtol=1e-5, atol=None):
    """``astropy.units.isclose`` for our quantities."""
    a = Quantity._coerce(a)
    bv = a._same(b)
    at = 0.0 if atol is None else a._same(atol)
    return np.isclose(a.value, bv, rtol=rtol, atol=at)
