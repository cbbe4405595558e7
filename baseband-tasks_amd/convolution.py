"""Overlap-save convolution in the Fourier domain on the GPU (reference
baseband_tasks/convolution.py:65-127)."""
import os

import numpy as np

from . import hip
from .base import check_broadcast_to
from .device_task import fetch_device
from .overlap_save import SpectralMultiplyTask

__all__ = ['Convolve', 'ConvolveSamples']


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

    def _repr_item(self, key, default, value=None):
        # the 'offset' argument (= the padding at the end), not the sample pointer
        if key == 'offset' and value is None:
            value = self._pad_end
        return super()._repr_item(key, default, value)

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

    # -- short responses: directly in the time domain ------------------------------
    #: Largest number of taps convolved directly (real / complex responses); the
    #: direct kernel costs time in proportion to the taps, the Fourier-domain
    #: plan three passes over the stream whatever the response.  Both give the
    #: same linear convolution (the reference's result is block independent).
    #: Measured on MI355X, 2 streams: 129 real taps run at 68 Gsamples/s directly
    #: against 55 through the Fourier-domain plan; break-even near 165 real taps.
    FIR_MAX_TAPS = 160
    FIR_MAX_TAPS_COMPLEX = 80
    _fir = None
    _fir_paired = False

    def _time_response(self):
        """Response in the time domain, ``(n_tap,) + b`` with ``b`` broadcastable
        to the sample shape."""
        return self._response

    def _use_fir(self):
        if self._fir is None:
            resp = np.asarray(self._time_response())
            is_real = not np.iscomplexobj(resp) or not np.any(resp.imag)
            limit = self.FIR_MAX_TAPS if is_real else self.FIR_MAX_TAPS_COMPLEX
            if resp.shape[0] > limit or (self._real and not is_real):
                self._fir = False
            else:
                full = np.broadcast_to(resp, resp.shape[:1] + tuple(self.sample_shape))
                full = full.reshape(resp.shape[0], self._n_stream).astype(np.complex64)
                # two real streams with the same (real) taps are one complex stream
                s = self._n_stream
                self._fir_paired = bool(self._real and self.PAIR_REAL_STREAMS and s % 2 == 0
                                        and np.array_equal(full[:, 0::2], full[:, 1::2]))
                if self._fir_paired:
                    full = full[:, 0::2]
                if full.shape[1] % 2 and full.shape[1] > 1:      # (one stream runs as it is)
                    full = np.concatenate([full, np.zeros_like(full[:, :1])], axis=1)
                self._fir = hip.FirPlan(np.ascontiguousarray(full))
        return self._fir is not False

    # -- medium responses: Fourier domain on short blocks, one kernel ----------------
    #: The linear convolution does not depend on the block length, so a response that
    #: fits a transform one workgroup holds in LDS (bbt_osm plans of <= 4096 samples:
    #: transform, multiply and inverse in ONE kernel, one read and one write of the
    #: stream) need not go through frame-sized blocks (three passes) nor through the
    #: direct filter (time in proportion to the taps).  SHORT_BLOCK: block length, or
    #: 'auto' = the power of two in [1024, 4096] that keeps the padding at or below an
    #: eighth of the block, or 0 to disable the route.  Measured on MI355X (config 5:
    #: 129 real taps, 8 streams, then Dedisperse): direct filter 6.6, blocks of 4096
    #: 6.9, 2048 7.3, 1024 8.0 G complete samples/s; the resampler alone 15 -> 21 (2048).
    SHORT_BLOCK = os.environ.get('BBT_SHORT_BLOCK', 'auto')
    SHORT_BLOCK_MIN_TAPS = 48
    _short = None

    def _short_block_length(self):
        n, pad = self.SHORT_BLOCK, self._response.shape[0] - 1
        if n == 'auto':
            n = 1024
            while n < 8 * pad:
                n *= 2
        n = int(n)
        return n if 256 <= n <= 4096 and 2 * pad <= n else 0

    def _short_blocks(self):
        """The inner task that convolves on short blocks, or None."""
        if self._short is None:
            n, taps = self._short_block_length(), self._response.shape[0]
            ok = (n and taps >= self.SHORT_BLOCK_MIN_TAPS and self._ih_samples_per_frame > n
                  and self.ih.shape[0] >= n)
            if ok:
                inner = Convolve(self.ih, self._time_response(), offset=self._pad_end,
                                 samples_per_frame=n - (taps - 1))
                inner.FIR_MAX_TAPS = inner.FIR_MAX_TAPS_COMPLEX = inner.SHORT_BLOCK = 0
                ok = inner._ih_samples_per_frame == n
            self._short = inner if ok else False
        return self._short or None

    def _input_span(self, first, last):
        start, stop = self._frame_span(first, last)
        pad = self._pad_start + self._pad_end
        short = self._short_blocks()
        if short is not None:
            return (self.ih,) + short._span_blocks(start, stop - start)[:2]
        if self._use_fir():
            return self.ih, start, stop - start + pad
        return super()._input_span(first, last)

    def read_planar(self, start, count):
        """Samples [start, start + count) of this (short-block, complex, even-S) convolution as
        S / 2 arrays of two-stream samples, shape (S / 2, count, 2): what a downstream
        overlap-save plan reads best (`SpectralMultiplyTask._planar_input`).  Not cached."""
        out = hip.DeviceArray((self._n_stream // 2, count, 2), np.complex64)
        self._short_blocks()._compute_span(start, count, out, planar=True)
        return out

    def _compute_frames(self, first, last, out):
        start, stop = self._frame_span(first, last)
        short = self._short_blocks()
        if short is not None:
            return short._compute_span(start, stop - start, out)
        if not self._use_fir():
            return super()._compute_frames(first, last, out)
        n_out, pad = stop - start, self._pad_start + self._pad_end
        x = fetch_device(self.ih, start, n_out + pad)
        s, final = self._n_stream, None
        if self._real and self._fir_paired:
            s //= 2                    # (n, S) float32 == (n, S/2) complex64, byte for byte
            x = hip.DeviceArray((n_out + pad, s), np.complex64, ptr=x.ptr, owner=x)
            out = hip.DeviceArray((n_out, s), np.complex64, ptr=out.ptr, owner=out)
        elif self._real:
            x = hip.real_to_complex(x.reshape(n_out + pad, s))
            final, out = out, hip.DeviceArray((n_out, s), np.complex64)
        se = s if s == 1 else s + s % 2
        if se != s:
            x = hip.pad_streams_to_even(x, s)
            target = hip.DeviceArray((n_out, se), np.complex64)
        else:
            target = out
        self._fir.execute(x, target, n_out)
        if se != s:
            hip.strip_stream_pad(target, n_out, s, out)
        if final is not None:
            hip.real_part(out, final)

    def close(self):
        if self._short:
            self._short._plan_close()
        self._short = None
        super().close()
        self._ft_response_cache = None
        if self._fir:
            self._fir.close()
        self._fir = None

    def _plan_close(self):
        """Release the plan of an inner short-block task (its stream stays open:
        it is the outer task's)."""
        if self._plan is not None:
            self._plan.close()
            self._plan = None


class ConvolveSamples(Convolve):
    """Convolve a time stream with a response in the time domain (reference
    convolution.py:23-62: `numpy.convolve` per stream, 'valid' part).  Same
    arguments and result as `Convolve`; here it always takes the direct filter
    kernel (`bbt_fir_execute`) up to 1024 taps -- beyond that the
    Fourier-domain plan, which gives the same linear convolution."""
    FIR_MAX_TAPS = 1024
    FIR_MAX_TAPS_COMPLEX = 1024
    SHORT_BLOCK = 0
