"""Synthetic sources (reference baseband_tasks/generators.py) plus
`DeviceStream`, a source whose samples already live in HBM."""
import numpy as np

from .base import Base
from .hip import DeviceArray, as_device_array

__all__ = ['StreamGenerator', 'EmptyStreamGenerator', 'Noise', 'NoiseGenerator', 'HostStream',
           'DeviceStream']


class StreamGenerator(Base):
    """Frames produced by ``function(stream)``; the stream pointer is at the
    start of the frame when the function is called and it must return
    ``samples_per_frame`` samples (reference generators.py:16-90)."""

    def __init__(self, function, shape, start_time, sample_rate, samples_per_frame=1,
                 dtype=np.complex64, **kwargs):
        super().__init__(shape=shape, start_time=start_time, sample_rate=sample_rate,
                         samples_per_frame=samples_per_frame, dtype=dtype, **kwargs)
        self._function = function

    def _read_frame(self, frame_index):
        return self._function(self)


class EmptyStreamGenerator(Base):
    """Uninitialised frames, to be filled by a `Task` (reference generators.py:93-151)."""

    def _read_frame(self, frame_index):
        return np.empty((self.samples_per_frame,) + self.sample_shape, self.dtype)


class Noise:
    """Reproducible Gaussian noise frames: the Philox counter is re-seeded per
    frame with the frame's sample offset (reference generators.py:154-190), so
    any frame can be regenerated bit-for-bit in any order."""

    def __init__(self, seed=None):
        self.seed = seed
        self._bit_generator = np.random.Philox(seed)
        self._rng = np.random.Generator(self._bit_generator)
        self._state0 = self._bit_generator.state

    def __call__(self, sh):
        state = self._state0
        state['state']['counter'][1] = sh.tell()
        self._bit_generator.state = state
        shape = (sh.samples_per_frame,) + tuple(sh.sample_shape)
        if sh.complex_data:
            shape = shape[:-1] + (shape[-1] * 2,)
        numbers = self._rng.normal(size=shape)
        if sh.complex_data:
            numbers = numbers.view(np.complex128)
        return numbers.astype(sh.dtype, copy=False)


class NoiseGenerator(StreamGenerator):
    """Stream of unit-variance (per component) normal noise; choose
    ``samples_per_frame`` large (reference generators.py:193-245)."""

    def __init__(self, shape, start_time, sample_rate, samples_per_frame,
                 dtype=np.complex64, seed=None, **kwargs):
        super().__init__(function=Noise(seed), shape=shape, start_time=start_time,
                         sample_rate=sample_rate, samples_per_frame=samples_per_frame,
                         dtype=dtype, **kwargs)


class HostStream(Base):
    """A stream whose samples sit in a NumPy array (or memory map) on the host:
    the host-side counterpart of `DeviceStream`, for data that does not come
    from a file reader.  ``read(out=...)`` is one copy; a device task on top
    uploads straight from the array when its memory is page-locked (``pin``:
    lock it in place with hipHostRegister -- on by default for C-contiguous
    arrays; arrays from `host_pipeline.pinned_empty` are pinned already), run
    m + 1 going up while run m is transformed (host_pipeline.py)."""

    def __init__(self, data, start_time, sample_rate, samples_per_frame=None, *, pin=True, **kwargs):
        self._data = data if isinstance(data, np.ndarray) else np.asarray(data)
        if samples_per_frame is None:
            samples_per_frame = min(self._data.shape[0], 1 << 20)
        self._pinned = None
        self._pin = bool(pin)
        super().__init__(shape=self._data.shape, start_time=start_time, sample_rate=sample_rate,
                         samples_per_frame=samples_per_frame, dtype=self._data.dtype, **kwargs)

    def host_view(self, start, count):
        """The samples [start, start + count) as a view of the array if it is
        C-contiguous and page-locked, else None."""
        if self._pinned is None:
            from . import host_pipeline
            self._pinned = bool(self._data.flags.c_contiguous and
                                (host_pipeline.is_pinned(self._data) or
                                 (self._pin and host_pipeline.pin_array(self._data))))
        return self._data[start:start + count] if self._pinned else None

    def read(self, count=None, out=None):
        count = self._prepare_read(count, out)
        piece = self._data[self.offset:self.offset + count]
        self.offset += count
        if out is None:
            return piece.copy()
        out[...] = piece
        return out

    def _read_frame(self, frame_index):
        start = frame_index * self.samples_per_frame
        return self._data[start:min(start + self.samples_per_frame, self.shape[0])].copy()

    def close(self):
        super().close()
        self._data = None


class DeviceStream(Base):
    """A stream resident in HBM.

    ``data`` is a `hip.DeviceArray`, a torch tensor on the GPU, or a host
    array / stream of this package (which is uploaded once).  ``read_device``
    returns zero-copy views, so a task chain on top never touches the host.
    """
    _produces_on_device = True
    #: any range of the stream is there for the taking: a task that reads straight from it need
    #: not bound how much it asks for at once (`DeviceTaskMixin.read_device`)
    _resident = True

    def __init__(self, data, start_time, sample_rate, samples_per_frame=None, **kwargs):
        if isinstance(data, np.ndarray):
            data = DeviceArray.from_host(data)
        elif isinstance(data, Base):
            source = data
            old = source.tell()
            source.seek(0)
            dev = DeviceArray(source.shape, source.dtype)
            step = max(source.samples_per_frame, 1 << 20)
            for start in range(0, source.shape[0], step):
                n = min(step, source.shape[0] - start)
                dev[start:start + n].copy_from_host(source.read(n))
            source.seek(old)
            if samples_per_frame is None:
                samples_per_frame = source.samples_per_frame
            for key in ('frequency', 'sideband', 'polarization'):
                if key not in kwargs and getattr(source, key, None) is not None:
                    kwargs[key] = getattr(source, key)
            data = dev
        self._data = as_device_array(data)
        if samples_per_frame is None:
            samples_per_frame = min(self._data.shape[0], 1 << 20)
        super().__init__(shape=self._data.shape, start_time=start_time,
                         sample_rate=sample_rate, samples_per_frame=samples_per_frame,
                         dtype=self._data.dtype, **kwargs)

    def read_device(self, count=None):
        count = self._prepare_read(count, None)
        view = self._data[self.offset:self.offset + count]
        self.offset += count
        return view

    def read(self, count=None, out=None):
        count = self._prepare_read(count, out)
        view = self._data[self.offset:self.offset + count]
        self.offset += count
        return view.to_host(out)

    def _read_frame(self, frame_index):
        start = frame_index * self.samples_per_frame
        return self._data[start:min(start + self.samples_per_frame, self.shape[0])].to_host()
