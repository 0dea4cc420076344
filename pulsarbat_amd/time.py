"""Minimal absolute-time type for ``Signal.start_time``.

The reference stores ``start_time`` as ``astropy.time.Time(..., format="isot",
precision=9)`` (pulsarbat/core.py:265-276) and only ever (a) adds a duration to
it when a signal is sliced in time (core.py:162-163), (b) subtracts two of them
(tests/test_dedispersion.py:67, 94) and (c) prints ``.isot`` (core.py:125).
astropy is not installable here, so this class covers exactly that, keeping an
integer day count plus float64 seconds-of-day (resolution ~1e-11 s, finer than
the reference's nanosecond print precision).  An astropy ``Time`` passed by a
caller who has astropy is kept as is (duck-typed on ``jd1``/``jd2``).
"""

import datetime as _dt
import math

import numpy as np

from . import units as u

__all__ = ["Time"]

_MJD_UNIX_EPOCH = 40587  # MJD of 1970-01-01


class Time:
    __slots__ = ("_day", "_sec", "precision")

    def __init__(self, val, val2=None, *, format=None, precision=9, scale="utc"):
        self.precision = precision
        if isinstance(val, Time):
            self._day, self._sec = val._day, val._sec
            return
        if isinstance(val, str):
            format = "isot"
        if format in (None, "mjd") and isinstance(val, (int, float, np.floating, np.integer)):
            day = math.floor(val)
            sec = (float(val) - day) * 86400.0
            if val2 is not None:
                sec += float(val2) * 86400.0
        elif format == "isot":
            txt = str(val).rstrip("Z")
            whole, _, fracdigits = txt.partition(".")
            d = _dt.datetime.fromisoformat(whole)
            frac = float("0." + fracdigits) if fracdigits else 0.0
            day = d.date().toordinal() - _dt.date(1970, 1, 1).toordinal() + _MJD_UNIX_EPOCH
            sec = d.hour * 3600 + d.minute * 60 + d.second + frac
        elif format == "unix":
            day = _MJD_UNIX_EPOCH + math.floor(val / 86400.0)
            sec = float(val) - (day - _MJD_UNIX_EPOCH) * 86400.0
        else:
            raise ValueError(f"unsupported Time input {val!r} (format={format!r})")
        self._day, self._sec = self._normalise(int(day), float(sec))

    @staticmethod
    def _normalise(day, sec):
        carry = math.floor(sec / 86400.0)
        return day + carry, sec - carry * 86400.0

    @classmethod
    def _from_parts(cls, day, sec, precision=9):
        t = cls.__new__(cls)
        t._day, t._sec = cls._normalise(day, sec)
        t.precision = precision
        return t

    @classmethod
    def now(cls):
        return cls(_dt.datetime.now(_dt.timezone.utc).timestamp(), format="unix")

    # --- properties ------------------------------------------------------------
    @property
    def isscalar(self):
        return True

    @property
    def shape(self):
        return ()

    @property
    def mjd(self):
        return self._day + self._sec / 86400.0

    @property
    def isot(self):
        # round the seconds of the day to the printed precision FIRST, as an integer count of 10^-precision seconds, so
        # that 59.9999999996 s carries into the minute (hour, day) instead of printing ":59" + ".000000000"
        unit = 10 ** self.precision
        ticks = int(round(self._sec * unit))
        day = self._day
        if ticks >= 86400 * unit:
            ticks -= 86400 * unit
            day += 1
        whole, frac = divmod(ticks, unit)
        date = _dt.date.fromordinal(day - _MJD_UNIX_EPOCH + _dt.date(1970, 1, 1).toordinal())
        h, rem = divmod(whole, 3600)
        mi, s = divmod(rem, 60)
        fs = f".{frac:0{self.precision}d}" if self.precision else ""
        return f"{date.isoformat()}T{h:02d}:{mi:02d}:{s:02d}{fs}"

    # --- arithmetic ------------------------------------------------------------
    def __add__(self, dt):
        return Time._from_parts(self._day, self._sec + u.to_value(dt, u.s), self.precision)

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, Time):
            return u.Quantity((self._day - other._day) * 86400.0 + (self._sec - other._sec), u.s)
        return Time._from_parts(self._day, self._sec - u.to_value(other, u.s), self.precision)

    def _key(self):
        return (self._day, self._sec)

    def __eq__(self, other):
        return isinstance(other, Time) and self._key() == other._key()

    def __lt__(self, other):
        return self._key() < other._key()

    def __le__(self, other):
        return self._key() <= other._key()

    def __gt__(self, other):
        return self._key() > other._key()

    def __ge__(self, other):
        return self._key() >= other._key()

    def __hash__(self):
        return hash(self._key())

    def isclose(self, other, atol=None):
        tol = 1e-9 if atol is None else u.to_value(atol, u.s)
        return abs((self - other).value) <= tol

    def __repr__(self):
        return f"<Time {self.isot}>"

    def __reduce__(self):
        return (Time._from_parts, (self._day, self._sec, self.precision))
