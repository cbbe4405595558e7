"""Dispersion measure arithmetic (float64, host side).

Same quantities as the reference's `DispersionMeasure`
(baseband_tasks/dm.py:7-120) without astropy: the value is in pc / cm^3,
frequencies are given in Hz, delays are returned in seconds and phases in
cycles.  Internally frequencies are converted to MHz so the arithmetic
follows the reference expression term by term.
"""
import numpy as np

from . import units as u

__all__ = ['DispersionMeasure']


class DispersionMeasure(float):
    """Electron column density in pc / cm^3 with dispersion helpers.

    The constant relating DM to delay is fixed to the Tempo value,
    1 / 2.41e-4 s MHz^2 cm^3 / pc (reference dm.py:36-38).
    """
    #: s MHz^2 cm^3 / pc
    dispersion_delay_constant = 1. / 2.41e-4

    def __new__(cls, dm):
        if hasattr(dm, 'to_value'):
            dm = dm.to_value('pc / cm3')
        return super().__new__(cls, dm)

    def __neg__(self):
        return DispersionMeasure(-float(self))

    def __repr__(self):
        return f"DispersionMeasure({float(self)!r} pc / cm3)"

    @staticmethod
    def _mhz(freq):
        return None if freq is None else np.asanyarray(u.to_hz(freq), dtype=float) / 1e6

    def time_delay(self, freq, ref_freq=None):
        """Delay in seconds of ``freq`` relative to ``ref_freq`` (infinite
        frequency if None):  D * DM * (1/f^2 - 1/f_ref^2)   (dm.py:42-76)."""
        f, fr = self._mhz(freq), self._mhz(ref_freq)
        d = self.dispersion_delay_constant * float(self)
        ref_inv2 = 0. if fr is None else 1. / fr ** 2
        return d * (1. / f ** 2 - ref_inv2)

    def phase_delay(self, freq, ref_freq=None):
        """Phase of the dispersion transfer function in cycles:
        D * DM * f * (1/f_ref - 1/f)^2   (dm.py:78-105)."""
        f, fr = self._mhz(freq), self._mhz(ref_freq)
        d = self.dispersion_delay_constant * float(self)
        ref_inv = 0. if fr is None else 1. / fr
        return d * f * (ref_inv - 1. / f) ** 2 * 1e6

    def phase_factor(self, freq, ref_freq=None):
        """exp(2 pi i phase_delay)   (dm.py:107-120)."""
        return np.exp(self.phase_delay(freq, ref_freq) * (2j * np.pi))
