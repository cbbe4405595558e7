"""Minimal units and time helpers (astropy is not available on the GPU box).

The reference passes `astropy.units.Quantity` and `astropy.time.Time`
objects around (baseband_tasks/base.py:104-107).  Here rates and frequencies
are plain floats in Hz, durations floats in seconds, and absolute times are
`Time` objects below.  Astropy objects are still accepted at the API
boundary (duck-typed through ``to_value`` / ``isot``) so code written for the
reference keeps working where astropy is installed.

Use as ``from baseband_tasks_amd import units as u``; ``16 * u.MHz``.
"""
import datetime as _dt
import math
import numbers

import numpy as np

Hz = 1.0
kHz = 1.0e3
MHz = 1.0e6
GHz = 1.0e9
s = 1.0
ms = 1.0e-3
us = 1.0e-6
ns = 1.0e-9
one = 1.0
cycle = 1.0


def _strip(value, unit_name):
    """Float(s) in SI for floats or astropy quantities."""
    if hasattr(value, 'to_value'):
        return value.to_value(unit_name)
    return value


def to_hz(value):
    """Rate or frequency in Hz as float / ndarray (None passes through)."""
    if value is None:
        return None
    value = _strip(value, 'Hz')
    if isinstance(value, numbers.Real):
        return float(value)
    return np.asarray(value, dtype=float)


def to_seconds(value):
    value = _strip(value, 's')
    if isinstance(value, numbers.Real):
        return float(value)
    return np.asarray(value, dtype=float)


_EPOCH = _dt.datetime(1970, 1, 1)


class Time:
    """Absolute UTC time as (integer seconds since 1970-01-01, fraction).

    Supports ``t + seconds``, ``t - seconds``, ``t2 - t1 -> seconds`` and
    comparisons; enough for `start_time`, `time`, `stop_time` and
    `seek(Time)` of the stream interface (base.py:286-310, 331-341).
    Leap seconds are ignored (as in ``datetime``).
    """
    __slots__ = ('sec', 'frac')

    def __init__(self, value, frac=0.0):
        if isinstance(value, Time):
            sec, fr = value.sec, value.frac + frac
        elif isinstance(value, str):
            sec, fr = self._parse(value)
            fr += frac
        elif hasattr(value, 'isot'):        # astropy Time
            sec, fr = self._parse(str(value.isot))
            fr += frac
        elif isinstance(value, _dt.datetime):
            delta = value - _EPOCH
            sec = delta.days * 86400 + delta.seconds
            fr = delta.microseconds * 1e-6 + frac
        else:
            sec, fr = int(value), float(frac)
        carry = math.floor(fr)
        self.sec = int(sec) + int(carry)
        self.frac = float(fr - carry)

    @staticmethod
    def _parse(text):
        text = text.strip().replace(' ', 'T')
        if '.' in text:
            main, digits = text.split('.')
            fr = float('0.' + digits) if digits else 0.0
        else:
            main, fr = text, 0.0
        fmt = '%Y-%m-%dT%H:%M:%S' if 'T' in main else '%Y-%m-%d'
        delta = _dt.datetime.strptime(main, fmt) - _EPOCH
        return delta.days * 86400 + delta.seconds, fr

    @property
    def isot(self):
        base = _EPOCH + _dt.timedelta(seconds=self.sec)
        digits = '%.9f' % self.frac
        if digits.startswith('1'):          # rounding spilled over
            base += _dt.timedelta(seconds=1)
            digits = '0.000000000'
        return base.strftime('%Y-%m-%dT%H:%M:%S') + digits[1:]

    @property
    def unix(self):
        return self.sec + self.frac

    def jd1_jd2(self):
        """Julian date as two doubles whose sum is exact to the float: whole days (as
        ``x.5``) and the fraction of the day -- the pair astropy's `Time` stores."""
        days, rest = divmod(self.sec, 86400)
        return 2440587.5 + days, (rest + self.frac) / 86400.

    @classmethod
    def from_jd(cls, jd1, jd2=0.0):
        days = math.floor(jd1 - 2440587.5)
        seconds = ((jd1 - 2440587.5) - days + jd2) * 86400.
        whole = math.floor(seconds + 0.5e-9)
        return cls(days * 86400 + int(whole), max(seconds - whole, 0.0))

    def __add__(self, seconds):
        seconds = to_seconds(seconds)
        whole = math.floor(seconds)
        return Time(self.sec + int(whole), self.frac + (seconds - whole))

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, Time) or hasattr(other, 'isot'):
            other = Time(other)
            return (self.sec - other.sec) + (self.frac - other.frac)
        return self.__add__(-to_seconds(other))

    def _key(self):
        return (self.sec, self.frac)

    def __eq__(self, other):
        return isinstance(other, Time) and self._key() == other._key()

    def __lt__(self, other):
        return self._key() < Time(other)._key()

    def __le__(self, other):
        return self._key() <= Time(other)._key()

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return "Time('%s')" % self.isot

    __str__ = __repr__


def is_time(value):
    return isinstance(value, Time) or hasattr(value, 'isot')
