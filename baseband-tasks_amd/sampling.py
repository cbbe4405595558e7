"""Fractional-sample shifting / resampling as a Fourier-domain convolution on
the GPU (reference baseband_tasks/sampling.py:63-312)."""
import numpy as np

from . import units as u
from . import hip
from .base import PaddedTaskBase, TaskBase, check_broadcast_to, _stream_rate, _stream_start
from .device_task import DeviceTaskMixin, fetch_device
from .convolution import Convolve
from .units import Time

__all__ = ['to_sample', 'seek_float', 'ShiftAndResample', 'Resample', 'TimeDelay',
           'ShiftSamples']


def to_sample(ih, offset):
    """Offset in (float) samples: numbers are samples, astropy time
    quantities are converted with the sample rate (sampling.py:17-20)."""
    if hasattr(offset, 'to_value'):
        return offset.to_value('s') * _stream_rate(ih)
    return np.asanyarray(offset, dtype=float) if np.ndim(offset) else float(offset)


def seek_float(ih, offset, whence=0):
    """Like ``ih.seek`` without rounding; may differ per stream
    (sampling.py:23-60)."""
    if u.is_time(offset):
        offset = (Time(offset) - _stream_start(ih)) * _stream_rate(ih)
        whence = 0
    offset = to_sample(ih, offset)
    check_broadcast_to(offset, ih.shape[1:])
    if whence == 0 or whence == 'start':
        return offset
    elif whence == 1 or whence == 'current':
        return ih.offset + offset
    elif whence == 2 or whence == 'end':
        return ih.shape[0] + offset
    raise ValueError("invalid 'whence'; should be 0 or 'start', 1 or "
                     "'current', or 2 or 'end'.")


class ShiftAndResample(Convolve):
    """Shift a stream in time by ``shift`` samples (may differ per stream) and
    optionally resample it so that a sample falls on ``offset``; ``lo`` adds the
    phase rotation of a mixed-down signal.  Response: windowed sinc of
    ``2 * pad + 1`` taps (reference sampling.py:63-227)."""

    def __init__(self, ih, shift, offset=None, whence='start', *, lo=None, pad=64,
                 samples_per_frame=None):
        self._shift = to_sample(ih, shift)
        shift_mean = np.mean(self._shift)
        if offset is None:
            d_time = shift_mean
            self._offset = None
        else:
            self._offset = seek_float(ih, offset, whence)
            d_time = self._offset + np.around(shift_mean - self._offset)
        sample_shift = np.array(self._shift - d_time, ndmin=ih.ndim - 1, dtype=float)
        response = self._windowed_sinc(pad, sample_shift)
        if samples_per_frame is None:
            samples_per_frame = max(ih.samples_per_frame, pad * 14)
        super().__init__(ih, response, offset=pad - int(round(float(sample_shift.min()))),
                         samples_per_frame=samples_per_frame)
        self._lo = None if lo is None else u.to_hz(lo)
        self._start_time = self._start_time + float(d_time) / self.sample_rate

    @staticmethod
    def _windowed_sinc(pad, sample_shift):
        """sinc(x) cos^2(pi x / (2 pad + 2)) on x = -pad..pad minus the
        fractional shift, per stream (sampling.py:177-193)."""
        i_max = int(round(float(sample_shift.max())))
        i_min = int(round(float(sample_shift.min())))
        n_result = 2 * pad + 1 + i_max - i_min
        result = np.zeros((n_result,) + sample_shift.shape)
        flat = result.reshape(n_result, -1)
        for k, shift in enumerate(sample_shift.ravel()):
            i_shift = int(round(float(shift)))
            x = np.arange(-pad, pad + 1) - (shift - i_shift)
            flat[i_shift - i_min:i_shift - i_max + n_result, k] = (
                np.sinc(x) * np.cos(np.pi * x / (2 * pad + 2)) ** 2)
        return result

    @property
    def _ft_response(self):
        base = super()._ft_response
        if self._lo is None:
            return base
        # phase rotation -shift/fs * lo * sideband cycles (sampling.py:211-220)
        phase = self._shift / self.sample_rate * self._lo * self.sideband
        return (base * np.exp(-2j * np.pi * phase)).astype(np.complex64)


    def _time_response(self):
        if self._lo is None:
            return self._response
        phase = self._shift / self.sample_rate * self._lo * self.sideband
        return self._response * np.exp(-2j * np.pi * phase)

    def _repr_item(self, key, default, value=None):
        # the 'offset' argument, not the sample pointer of the same name
        if key == 'offset':
            value = self._offset
        return super()._repr_item(key, default, value)


class Resample(ShiftAndResample):
    """Resample so that a sample falls exactly on ``offset`` and leave the
    sample pointer there (reference sampling.py:230-312)."""

    def __init__(self, ih, offset, whence='start', *, pad=64, samples_per_frame=None):
        super().__init__(ih, shift=0., offset=offset, whence=whence, pad=pad,
                         samples_per_frame=samples_per_frame)
        self.seek(_stream_start(ih) + float(self._offset) / _stream_rate(ih))


class TimeDelay(DeviceTaskMixin, TaskBase):
    """Delay a complex stream: the delay is added to the time stamps and, if
    the signal was mixed with local oscillator ``lo`` (Hz) before sampling, the
    phases are rotated by ``-delay * lo * sideband`` cycles (no resampling;
    reference sampling.py:315-377).  ``lo=None`` means no rotation."""

    def __init__(self, ih, delay, *, lo, frequency=None, sideband=None):
        assert np.dtype(ih.dtype).kind == 'c', "Time delay only works on complex data."
        if np.dtype(ih.dtype) != np.complex64:
            raise TypeError(f"the accelerated TimeDelay handles complex64 streams; got {ih.dtype}.")
        self._delay = to_sample(ih, delay)
        self._lo = None if lo is None else u.to_hz(lo)
        seconds = self._delay / _stream_rate(ih)
        super().__init__(ih, frequency=frequency, sideband=sideband)
        self._start_time = self._start_time + float(seconds)
        if self._lo is None:
            self._phase_factor = None
        else:
            cycles = seconds * self._lo * self.sideband
            self._phase_factor = np.exp(-2j * np.pi * cycles).astype(np.complex64)
        self._factor_dev = None

    def _compute_frames(self, first, last, out):
        start, stop = self._frame_span(first, last)
        x = fetch_device(self.ih, start, stop - start)
        if self._phase_factor is None:
            out.copy_from_device(x)
            return
        n_elem = 1
        for d in self.sample_shape:
            n_elem *= d
        hip.scale_streams(x, out, stop - start, n_elem, self._factor())

    def _factor(self):
        if self._factor_dev is None:
            self._factor_dev = hip.DeviceArray.from_host(np.ascontiguousarray(
                np.broadcast_to(self._phase_factor, self.sample_shape)).ravel())
        return self._factor_dev

    def task(self, data):
        data = np.ascontiguousarray(data, dtype=np.complex64)
        if self._phase_factor is None:
            return data
        out = hip.DeviceArray(data.shape, np.complex64)
        hip.scale_streams(hip.DeviceArray.from_host(data), out, data.shape[0],
                          data.size // max(data.shape[0], 1), self._factor())
        return out.to_host()


class ShiftSamples(DeviceTaskMixin, PaddedTaskBase):
    """Shift streams by integer numbers of samples (no resampling): positive
    shifts delay a stream.  ``shift`` (samples, rounded; or an astropy time
    quantity) broadcasts to the sample shape; the output starts
    ``shift.max()`` samples after the input and is ``ptp(shift)`` samples
    shorter (reference sampling.py:380-425).  float32 or complex64 streams."""

    def __init__(self, ih, shift, *, samples_per_frame=None):
        shift = self._shift = np.round(to_sample(ih, shift)).astype(int)
        check_broadcast_to(shift, ih.shape[1:])
        if np.dtype(ih.dtype).itemsize not in (4, 8):
            raise TypeError(f"the accelerated ShiftSamples handles 4- and 8-byte samples; got {ih.dtype}.")
        start_time = _stream_start(ih) + int(shift.max()) / _stream_rate(ih)
        super().__init__(ih, pad_start=0, pad_end=int(np.ptp(shift)),
                         samples_per_frame=samples_per_frame, start_time=start_time)
        # element e of a complete output sample comes from input sample i + offsets[e]
        self._offsets = np.broadcast_to(shift.max() - shift, self.sample_shape).ravel()
        self._plan = None

    def _compute_frames(self, first, last, out):
        if self._plan is None:
            self._plan = hip.ShiftPlan(self._offsets, self.dtype.itemsize)
        start, stop = self._frame_span(first, last)
        x = fetch_device(self.ih, start, stop - start + self._pad_end)
        self._plan.execute(x, out, stop - start)

    def task(self, data):
        n_out = data.shape[0] - self._pad_end
        if self._plan is None:
            self._plan = hip.ShiftPlan(self._offsets, self.dtype.itemsize)
        out = hip.DeviceArray((n_out,) + tuple(self.sample_shape), self.dtype)
        self._plan.execute(hip.DeviceArray.from_host(np.ascontiguousarray(data, dtype=self.dtype)),
                           out, n_out)
        return out.to_host()

    def close(self):
        super().close()
        self._drop_cache()
        if self._plan is not None:
            self._plan.close()
            self._plan = None
