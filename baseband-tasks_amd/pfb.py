"""Polyphase filter bank on the GPU (reference baseband_tasks/pfb.py:14-154)."""
import numpy as np

from . import hip
from .base import PaddedTaskBase, getattr_if_none, _stream_rate
from .channelize import _RowFFTTask, _check_n, _prod
from .device_task import fetch_device

__all__ = ['sinc_hamming', 'PolyphaseFilterBank', 'PolyphaseFilterBankSamples']


def sinc_hamming(n_tap, n_sample, sinc_scale=1.):
    """sinc(n_tap * s * (k/N - 1/2)) * hamming(N), N = n_tap * n_sample,
    shaped ``(n_tap, n_sample)`` (reference pfb.py:14-45)."""
    n = n_tap * n_sample
    x = n_tap * sinc_scale * np.linspace(-0.5, 0.5, n, endpoint=False)
    return (np.sinc(x) * np.hamming(n)).reshape(n_tap, n_sample)


class _NoHostTask(PaddedTaskBase):
    def task(self, data):
        raise NotImplementedError("the polyphase filter is evaluated on the GPU inside "
                                  "PolyphaseFilterBank; its padded stream has no host path.")


class PolyphaseFilterBank(_RowFFTTask):
    """Channelize with a polyphase filter: spectrum ``i`` is the FFT over ``c``
    of ``sum_t x[(i + t) n + c] * response[t, c]`` (the definition in
    reference pfb.py:91-100; the reference's Fourier-domain class,
    pfb.py:103-154, computes the same thing).

    Geometry follows the reference: an inner padded stream (``.padded``) with
    ``(n_tap - 1) n / 2`` samples of padding on each side, channelized with
    ``padded.samples_per_frame // n`` spectra per frame; the time stamp of
    spectrum 0 is therefore ``(n_tap - 1) n / 2`` input samples after the
    start of ``ih``.

    Parameters
    ----------
    ih : stream (complex64)
    response : array (n_tap, n)
    samples_per_frame : int, optional
        Spectra per frame.
    frequency, sideband : optional overrides of the stream metadata.
    """

    def __init__(self, ih, response, samples_per_frame=None, frequency=None, sideband=None):
        response = np.asanyarray(response)
        n_tap, n = response.shape
        _check_n(n, minimum=256)
        if np.dtype(ih.dtype) not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated filter bank handles complex64 and float32 streams; "
                            f"got {ih.dtype}.")
        self._real = np.dtype(ih.dtype).kind == 'f'
        n_out = n // 2 + 1 if self._real else n
        pad = (n_tap - 1) * n
        assert pad % 2 == 0
        if samples_per_frame is not None:
            samples_per_frame = samples_per_frame * n
        self.padded = _NoHostTask(ih, pad_start=pad // 2, pad_end=pad // 2,
                                  samples_per_frame=samples_per_frame)
        if self.padded._ih_samples_per_frame % n:
            raise ValueError("the input block of the polyphase filter "
                             f"({self.padded._ih_samples_per_frame} samples) must be a "
                             f"multiple of n={n}; pass samples_per_frame.")
        self._response = response
        self._source = ih
        rate = _stream_rate(ih)
        frequency = getattr_if_none(ih, 'frequency', frequency, required=False)
        sideband = getattr_if_none(ih, 'sideband', sideband, required=False)
        if frequency is not None:
            fft_freq = (np.fft.rfftfreq if self._real else np.fft.fftfreq)(n, d=1. / rate)
            frequency = frequency + fft_freq.reshape((n_out,) + (1,) * (ih.ndim - 1)) * sideband
        self._setup_streams(n, _prod(ih.shape[1:]))
        self._reshape = (self.padded._ih_samples_per_frame // n, n) + tuple(ih.shape[1:])
        super().__init__(self.padded, shape=(-1, n_out) + tuple(ih.shape[1:]),
                         sample_rate=rate / n,
                         samples_per_frame=self.padded.samples_per_frame // n,
                         frequency=frequency, sideband=sideband, dtype=np.complex64)

    def _get_plan(self):
        if self._plan is None:
            self._plan = hip.PfbPlan(self._response, self._n_stream_even)
        return self._plan

    def _compute_frames(self, first, last, out):
        start, stop = self._frame_span(first, last)
        n_spectra = stop - start
        n, n_tap = self._n, self._response.shape[0]
        x = fetch_device(self._source, start * n, (n_spectra + n_tap - 1) * n)
        x = x.reshape((n_spectra + n_tap - 1) * n, self._n_stream)
        s, se = self._n_stream, self._n_stream_even
        if self._real:
            x = hip.real_to_complex(x)
            final, out = out, hip.DeviceArray((n_spectra * n, s), np.complex64)
        flat = out.reshape(n_spectra * n, s)
        if se != s:
            x = hip.pad_streams_to_even(x, s)
            tmp = hip.DeviceArray((n_spectra * n, se), np.complex64)
            self._get_plan().execute(x, tmp, n_spectra)
            hip.strip_stream_pad(tmp, n_spectra * n, s, flat)
        else:
            self._get_plan().execute(x, flat, n_spectra)
        if self._real:
            hip.keep_half_spectrum(flat, n, s, final)

    def ppf(self, data):
        raise NotImplementedError("the filter and the FFT are one GPU kernel here; "
                                  "use read().")

    def close(self):
        super().close()
        self.__dict__.pop('_source', None)


#: The GPU evaluates the time-domain definition directly, so both reference
#: classes map to the same task.
PolyphaseFilterBankSamples = PolyphaseFilterBank
