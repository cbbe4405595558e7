"""Detection on the GPU: `Square` and `Power` (reference
baseband_tasks/functions.py:19-143)."""
import numpy as np

from . import hip
from .base import TaskBase, simplify_shape
from .device_task import DeviceTaskMixin, fetch_device

__all__ = ['Square', 'Power']


def _prod(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class _DetectTask(DeviceTaskMixin, TaskBase):
    _mode = 0
    _real = False

    def _detect(self, x, n_out, step, out, average=True):
        """x: input complete samples on the device; fills ``out``."""
        n_elem = _prod(self.ih.shape[1:])
        if self._real:           # x^2, then (if asked) a plain sum
            if step == 1:
                hip.square_real(x, out)
            else:
                tmp = hip.DeviceArray((n_out * step, n_elem), np.float32)
                hip.square_real(x, tmp)
                hip.detect_integrate(tmp, out, n_out, step, n_elem, 2, average)
            return
        inner = getattr(self, '_inner', 1)
        if self._mode == 1 and inner > 1:        # polarization axis not last
            hip.detect_power_axis(x, out, n_out, step, n_elem // (2 * inner), inner, average)
            return
        hip.detect_integrate(x, out, n_out, step, n_elem, self._mode, average)

    def _compute_frames(self, first, last, out):
        start, stop = self._frame_span(first, last)
        # a channelizer on top of an overlap-save task detects in that task's last pass: the
        # spectra are never stored (as `Integrate` of this task does, with sums)
        fused = getattr(self.ih, '_compute_detected', None)
        if (fused is not None and not self._real and getattr(self, '_inner', 1) == 1
                and not getattr(self.ih, 'closed', False)
                and fused(start, stop - start, 1, self._mode, False, out)):
            return
        x = fetch_device(self.ih, start, stop - start)
        self._detect(x, stop - start, 1, out)

    def task(self, data):
        data = np.ascontiguousarray(data, dtype=np.float32 if self._real else np.complex64)
        out = hip.DeviceArray((data.shape[0],) + tuple(self.sample_shape), self.dtype)
        self._detect(hip.DeviceArray.from_host(data), data.shape[0], 1, out)
        return out.to_host()


class Square(_DetectTask):
    """Intensities by squaring: ``re^2 + im^2`` per stream (no cross terms;
    see `Power`).  ``polarization`` defaults to the doubled input labels
    (reference functions.py:19-56)."""
    _mode = 0

    def __init__(self, ih, polarization=None):
        if np.dtype(ih.dtype) not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated Square handles complex64 and float32 streams; got "
                            f"{ih.dtype}.")
        self._real = np.dtype(ih.dtype).kind == 'f'
        if polarization is None and getattr(ih, 'polarization', None) is not None:
            polarization = np.char.add(ih.polarization, ih.polarization)
        super().__init__(ih, dtype=np.float32, polarization=polarization)


class Power(_DetectTask):
    """Powers and cross terms of two polarizations X, Y along the
    polarization axis (2 -> 4): ``|X|^2, |Y|^2, Re(X Y*), Im(X Y*)``, labelled
    XX, YY, XY, YX (reference functions.py:59-143).  Any sample axis may be
    the polarization axis; the last one is the fast layout."""
    _mode = 1

    def __init__(self, ih, polarization=None):
        self._polarization_given = polarization
        if polarization is None:
            pol = ih.polarization
            if pol.size != 2:
                raise ValueError("stream should have exactly 2 polarizations. "
                                 "Reshape appropriately.")
            polarization = np.char.add(pol[[0, 1, 0, 1]], pol[[0, 1, 1, 0]])
        else:
            polarization = simplify_shape(np.asanyarray(polarization))
            if not (polarization.size == 4 == len(np.unique(polarization))
                    and 4 in polarization.shape):
                raise ValueError('output polarizations should have 4 unique '
                                 'elements along one axis.')
        self._axis = ih.ndim - polarization.ndim + polarization.shape.index(4)
        if ih.shape[self._axis] != 2:
            raise ValueError(f"input shape should be 2 along polarization axis"
                             f" ({self._axis}), not {ih.shape[self._axis]}.")
        if np.dtype(ih.dtype).kind != 'c':
            raise ValueError("Power only works on a complex timestream.")
        if np.dtype(ih.dtype) != np.complex64:
            raise TypeError(f"the accelerated Power handles complex64 streams; got {ih.dtype}.")
        self._inner = _prod(ih.shape[self._axis + 1:])       # 1: (X, Y) adjacent, the fast layout
        shape = ih.shape[:self._axis] + (4,) + ih.shape[self._axis + 1:]
        super().__init__(ih, shape=shape, polarization=polarization, dtype=np.float32)

    def _repr_item(self, key, default, value=None):
        # 'polarization' as given: labels taken from the stream are not an argument
        if key == 'polarization':
            if self._polarization_given is None:
                return None
            value = np.asanyarray(self._polarization_given)
        return super()._repr_item(key, default, value)
