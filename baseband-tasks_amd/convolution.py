"""Overlap-save convolution in the Fourier domain on the GPU (reference
baseband_tasks/convolution.py:65-127)."""
import numpy as np

from .base import check_broadcast_to
from .overlap_save import SpectralMultiplyTask

__all__ = ['Convolve']


class Convolve(SpectralMultiplyTask):
    """Convolve a time stream with ``response`` (first axis = taps; 1-D
    responses apply to every stream, otherwise trailing axes broadcast to
    the sample shape).  ``offset`` moves the sample the result is attributed
    to (reference convolution.py:75-93)."""

    def __init__(self, ih, response, *, offset=0, samples_per_frame=None):
        response = np.asanyarray(response)
        if response.ndim == 1 and ih.ndim > 1:
            response = response.reshape(response.shape[:1] + (1,) * (ih.ndim - 1))
        else:
            check_broadcast_to(response, response.shape[:1] + tuple(ih.shape[1:]))
        self._response = response
        pad = response.shape[0] - 1
        super().__init__(ih, pad - offset, offset, samples_per_frame=samples_per_frame)
        self._keep_from = self._pad_start + self._pad_end
        self._ft_response_cache = None

    @property
    def _ft_response(self):
        """FFT of the zero-extended response (convolution.py:108-114),
        evaluated in float64 and cast to complex64."""
        if self._ft_response_cache is None:
            n = self._ih_samples_per_frame
            long_response = np.zeros((n,) + self._response.shape[1:], np.complex128)
            long_response[:self._response.shape[0]] = self._response
            self._ft_response_cache = np.fft.fft(long_response, axis=0).astype(np.complex64)
        return self._ft_response_cache

    def _spectral_response(self):
        return self._ft_response

    def close(self):
        super().close()
        self._ft_response_cache = None
