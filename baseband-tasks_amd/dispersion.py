"""Coherent (de)dispersion on the GPU.

`Disperse` / `Dedisperse` keep the constructor and stream surface of the
reference (baseband_tasks/dispersion.py:16-190); the per-frame arithmetic
``ifft(fft(x) * phase_factor)[pad_start:pad_start + spf]``
(dispersion.py:135-139) runs in libbbt_hip.so.
"""
import os

import numpy as np

from . import hip
from . import units as u
from .base import getattr_if_none, _stream_rate, _stream_start
from .dm import DispersionMeasure
from .overlap_save import SpectralMultiplyTask
from .sampling import ShiftSamples

__all__ = ['Disperse', 'Dedisperse', 'DisperseSamples', 'DedisperseSamples']


class Disperse(SpectralMultiplyTask):
    """Coherently disperse a time stream.

    Parameters
    ----------
    ih : stream
        Input, time along the first axis, complex64.
    dm : float or `DispersionMeasure`
        In pc / cm^3; negative values dedisperse.
    reference_frequency : float or array, optional
        Frequency (Hz) to which the data are dispersed; default the mean
        band centre.
    samples_per_frame : int, optional
        Output samples per frame.  Default: the block that keeps padding
        below 25 % of it, rounded up to a length the FFT engine likes.
    frequency, sideband : optional
        Override / provide the stream's metadata (frequency in Hz).
    """

    def __init__(self, ih, dm, *, reference_frequency=None, samples_per_frame=None,
                 frequency=None, sideband=None):
        dm = DispersionMeasure(dm)
        frequency = u.to_hz(getattr_if_none(ih, 'frequency', frequency))
        sideband = np.asanyarray(getattr_if_none(ih, 'sideband', sideband))
        rate = _stream_rate(ih)
        # band edges (dispersion.py:54-61)
        half = rate / 2.
        if np.dtype(ih.dtype).kind == 'c':
            f_lo, f_hi = frequency - half, frequency + half
        else:
            f_lo = frequency + np.minimum(sideband, 0.) * half
            f_hi = frequency + np.maximum(sideband, 0.) * half
        if reference_frequency is None:
            reference_frequency = np.mean(f_lo + f_hi) / 2.
        else:
            reference_frequency = u.to_hz(reference_frequency)
        # extreme delays across the band -> padding (dispersion.py:66-74)
        pf_lo, pf_hi, pf_ref = self._padding_band(f_lo, f_hi, reference_frequency, half)
        d_lo = dm.time_delay(pf_lo, pf_ref)
        d_hi = dm.time_delay(pf_hi, pf_ref)
        d_max = max(np.max(d_lo), np.max(d_hi))
        d_min = min(np.min(d_lo), np.min(d_hi))
        pad_start = int(np.ceil(d_max * rate))
        pad_end = int(np.ceil(-d_min * rate))
        # reference frequency outside the band: part of the delay is a plain
        # shift of the time stamps (dispersion.py:78-93)
        if pad_start < 0:
            assert pad_end > 0
            sample_offset = pad_start
            pad_end += pad_start
            pad_start = 0
        elif pad_end < 0:
            sample_offset = -pad_end
            pad_start += pad_end
            pad_end = 0
        else:
            sample_offset = 0
        start_time = _stream_start(ih) + sample_offset / rate
        super().__init__(ih, pad_start, pad_end, samples_per_frame=samples_per_frame,
                         frequency=frequency, sideband=sideband, start_time=start_time)
        self._dm = dm
        self.reference_frequency = reference_frequency
        self._sample_offset = sample_offset
        self._keep_from = self._pad_start
        self._pad_slice = slice(self._pad_start, self._pad_start + self.samples_per_frame)
        self._phase_factor = None

    def _padding_band(self, f_lo, f_hi, reference_frequency, half_rate):
        """Band edges and reference frequency that set the padding: this
        stream's own (dispersion.py:66-74).  `sharding.SubbandDedisperse`
        substitutes those of the whole band a shard was cut from."""
        return f_lo, f_hi, reference_frequency

    @property
    def phase_factor(self):
        """exp(2 pi i phase) per FFT bin, float64 arithmetic cast to complex64
        (dispersion.py:115-129); shape ``(N,) +`` broadcastable sample shape
        (``N // 2 + 1`` frequencies for a real stream)."""
        if self._phase_factor is None:
            n = self._ih_samples_per_frame
            if self._real:       # non-negative frequencies only, as for rfft (fourier/base.py:150-153)
                fft_freq = np.fft.rfftfreq(n, d=1. / self.sample_rate)
            else:
                fft_freq = np.fft.fftfreq(n, d=1. / self.sample_rate)
            fft_freq = fft_freq.reshape(fft_freq.shape + (1,) * len(self.sample_shape))
            frequency = self.frequency + fft_freq * self.sideband
            phase = self._dm.phase_delay(frequency, self.reference_frequency)
            phase = phase * self.sideband
            if self._sample_offset != 0:
                phase = phase + (self._sample_offset / self.sample_rate) * fft_freq
            self._phase_factor = np.exp(phase * (2j * np.pi)).astype(np.complex64)
        return self._phase_factor

    def _spectral_response(self):
        return self.phase_factor

    #: Make the plan's chirp on the GPU (float64, `hip.chirp`) instead of uploading `phase_factor`:
    #: the host takes 4.4 s for the 8 x 2^24 points of config 4's share, the GPU 7 ms with the plan's tables.
    #: Complex streams only; `phase_factor` itself stays what the reference's attribute is.
    DEVICE_CHIRP = os.environ.get('BBT_DEVICE_CHIRP', '1') != '0'

    def _response_columns(self):
        if not self.DEVICE_CHIRP or self._real or self._phase_factor is not None:
            return super()._response_columns()
        ndim = len(self.sample_shape)
        freq, side, ref = (np.asanyarray(a, dtype=float) for a in
                           (self.frequency, self.sideband, self.reference_frequency))
        try:
            bshape = np.broadcast_shapes((1,) * ndim, freq.shape, side.shape, ref.shape)
        except ValueError:
            return super()._response_columns()
        if len(bshape) != ndim:
            return super()._response_columns()
        cols = [np.broadcast_to(a, bshape).ravel() for a in (freq, side, ref)]
        d_dm = self._dm.dispersion_delay_constant * float(self._dm)
        try:
            columns = hip.chirp(self._ih_samples_per_frame, cols[0], cols[1], cols[2], self.sample_rate, d_dm,
                                self._sample_offset / self.sample_rate)
        except hip.HipError:
            # (geometries `bbt_chirp` refuses -- more than 65535 distinct columns, a reference
            # frequency that is not positive: the host attribute handles every case the reference does)
            return super()._response_columns()
        ncol = cols[0].shape[0]
        index = np.broadcast_to(np.arange(ncol).reshape(bshape), self.sample_shape).ravel().astype(np.int32)
        if self._n_stream_even != self._n_stream:
            index = np.concatenate([index, index[-1:]])
        return columns, index

    @property
    def dm(self):
        return self._dm

    def close(self):
        super().close()
        self._phase_factor = None


class Dedisperse(Disperse):
    """Coherently dedisperse a time stream (reference dispersion.py:149-190);
    parameters as for `Disperse`, with ``dm`` the DM to remove."""

    def __init__(self, ih, dm, *, reference_frequency=None, samples_per_frame=None,
                 frequency=None, sideband=None):
        super().__init__(ih, -DispersionMeasure(dm), reference_frequency=reference_frequency,
                         samples_per_frame=samples_per_frame, frequency=frequency,
                         sideband=sideband)

    @property
    def dm(self):
        return -self._dm


class DisperseSamples(ShiftSamples):
    """Incoherent dispersion: shift every stream by the dispersive delay at its
    mid-channel frequency, rounded to whole samples (no in-channel smearing;
    reference dispersion.py:193-250).  Parameters as for `Disperse`."""

    def __init__(self, ih, dm, *, reference_frequency=None, samples_per_frame=None,
                 frequency=None, sideband=None):
        if frequency is not None or sideband is not None:
            from .base import SetAttribute
            ih = SetAttribute(ih, frequency=frequency, sideband=sideband)
        frequency = ih.frequency
        if np.dtype(ih.dtype).kind != 'c':
            # mid-channel frequency of a real stream
            frequency = frequency + ih.sideband * _stream_rate(ih) / 2.
        if reference_frequency is None:
            reference_frequency = np.mean(frequency)
        else:
            reference_frequency = u.to_hz(reference_frequency)
        dm = DispersionMeasure(dm)
        delay = dm.time_delay(frequency, reference_frequency)          # seconds
        super().__init__(ih, delay * _stream_rate(ih), samples_per_frame=samples_per_frame)
        self.reference_frequency = reference_frequency
        self._dm = dm

    @property
    def dm(self):
        return self._dm


class DedisperseSamples(DisperseSamples):
    """Incoherent dedispersion (reference dispersion.py:253-298)."""

    def __init__(self, ih, dm, *, reference_frequency=None, samples_per_frame=None,
                 frequency=None, sideband=None):
        super().__init__(ih, -DispersionMeasure(dm), reference_frequency=reference_frequency,
                         samples_per_frame=samples_per_frame, frequency=frequency,
                         sideband=sideband)

    @property
    def dm(self):
        return -self._dm
