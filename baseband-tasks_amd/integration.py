"""Stepwise integration on the GPU (reference baseband_tasks/integration.py:
52-303, the integer-``step`` case)."""
import operator

import numpy as np

from . import hip
from .base import BaseTaskBase, _stream_rate, _stream_start
from .device_task import DeviceTaskMixin, fetch_device
from .functions import _DetectTask

__all__ = ['Integrate']


def _prod(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class Integrate(DeviceTaskMixin, BaseTaskBase):
    """Integrate a stream over ``step`` consecutive samples.

    Parameters
    ----------
    ih : stream
        float32 (e.g. the output of `Square` / `Power`) or complex64.
    step : int, optional
        Input samples per output sample; default: everything from ``start``.
        (Integration in time or pulse-phase units, i.e. non-integer steps and
        the ``phase`` callable of the reference, is outside the accelerated
        path.)
    start : int or `~baseband_tasks_amd.units.Time`
        Offset (or time, rounded to the nearest sample) of the first sample.
    average : bool
        True: averages.  False: like the reference, `read` returns a structured
        array with the sums in ``'data'`` and the number of samples summed in
        ``'count'`` (the task's ``dtype`` is that structured type; the frames in
        HBM hold the sums).
    samples_per_frame : int
        Output samples per frame (framing only).
    dtype : optional, must equal the input dtype.

    When ``ih`` is a GPU `Square` or `Power`, detection and integration run as
    one kernel on the undetected stream (the detected stream is never stored).
    """
    max_frames_per_call = 1 << 16

    def __init__(self, ih, step=None, phase=None, *, start=0, average=True,
                 samples_per_frame=1, dtype=None):
        if phase is not None:
            raise NotImplementedError("integration in pulse phase is outside the accelerated path.")
        ih_start = ih.seek(start)
        ih_n = ih.shape[0] - ih_start
        if ih_start < 0 or ih_n < 0:
            raise ValueError("'start' is not within the underlying stream.")
        if step is None:
            step = ih_n
        try:
            if isinstance(step, float) and step.is_integer():
                step = int(step)                # (3.0 samples is 3 samples)
            step = operator.index(step)
        except TypeError:
            raise NotImplementedError("integration over time intervals (non-integer step) is "
                                      "outside the accelerated path.") from None
        in_dtype = np.dtype(ih.dtype)
        if in_dtype not in (np.dtype(np.float32), np.dtype(np.complex64)):
            raise TypeError(f"the accelerated Integrate handles float32/complex64; got {in_dtype}.")
        if dtype is not None and np.dtype(dtype) != in_dtype:
            raise TypeError("the accelerated Integrate keeps the input dtype.")
        n_out = int(ih_n / step + 0.5 / step)
        assert n_out >= 1, "time per frame larger than total time in stream"
        rate = _stream_rate(ih)
        self._start = start
        self._step, self._ih_start = step, ih_start
        self.average = bool(average)
        self._sum_dtype = in_dtype
        out_dtype = in_dtype if average else np.dtype([('data', in_dtype), ('count', int)])
        super().__init__(ih, shape=(n_out,) + tuple(ih.shape[1:]), sample_rate=rate / step,
                         samples_per_frame=samples_per_frame,
                         start_time=_stream_start(ih) + ih_start / rate, dtype=out_dtype)

    @property
    def _device_dtype(self):
        return self._sum_dtype

    def read(self, count=None, out=None):
        if self.average:
            return super().read(count, out)
        if out is not None:
            raise NotImplementedError("average=False: read() makes its own structured output.")
        count = self._prepare_read(count, None)
        sums = super().read(count, np.empty((count,) + tuple(self.sample_shape), self._sum_dtype))
        result = np.empty(sums.shape, self.dtype)
        result['data'] = sums
        result['count'] = self._step
        return result

    def _repr_item(self, key, default, value=None):
        # 'step' as given (None = everything), 'start' as given
        if key == 'step' and self._step == self.ih.shape[0] - self._ih_start and self.shape[0] == 1:
            return None
        return super()._repr_item(key, default, value)

    def _compute_frames(self, first, last, out):
        a, b = self._frame_span(first, last)
        n_out, step = b - a, self._step
        src = self.ih
        if isinstance(src, _DetectTask) and not src.closed:
            # a channelizer on top of an overlap-save task detects and sums in
            # that task's last pass: neither stream is ever stored
            fused = getattr(src.ih, '_compute_detected', None)
            if (fused is not None and not src._real and getattr(src, '_inner', 1) == 1
                    and not getattr(src.ih, 'closed', False)
                    and fused(self._ih_start + a * step, n_out, step, src._mode, self.average, out)):
                return
            x = fetch_device(src.ih, self._ih_start + a * step, n_out * step)
            src._detect(x, n_out, step, out, self.average)
            return
        x = fetch_device(src, self._ih_start + a * step, n_out * step)
        n_float = _prod(self.sample_shape) * (2 if self._sum_dtype.kind == 'c' else 1)
        hip.detect_integrate(x, out, n_out, step, n_float, 2, self.average)

    def close(self):
        super().close()
        self._drop_cache()
