"""Stream-reader runtime: the file-handle-like surface every task exposes.

Mirrors the public behaviour of the reference's runtime
(baseband_tasks/base.py: `Base` 87-497, `BaseTaskBase` 499-611, `TaskBase`
613-707, `PaddedTaskBase` 709-795, `Task` 798-889, `SetAttribute` 892-951):
``shape, sample_shape, size, ndim, dtype, complex_data, sample_rate,
samples_per_frame, start_time / time / stop_time, offset, seek, tell, read,
close, closed``, context-manager use, ``__array__``, time slicing, and the
``frequency / sideband / polarization`` metadata with its broadcasting rules.

Differences that are deliberate: rates are floats in Hz and times are
`units.Time` (no astropy here); and every stream also has ``read_device``
which returns the samples as a `hip.DeviceArray` in HBM (host-only sources
upload; GPU tasks hand over their device frames without a round trip).
"""
import inspect
import operator
import types
import warnings

import numpy as np

from . import units as u
from .units import Time

__all__ = ['SinglePrecision', 'Base', 'BaseTaskBase', 'TaskBase', 'PaddedTaskBase', 'Task',
           'SetAttribute', 'META_ATTRIBUTES']

META_ATTRIBUTES = ('frequency', 'sideband', 'polarization')


# ---------------------------------------------------------------------------
# metadata helpers (base.py:24-53 in the reference)
def check_broadcast_to(value, sample_shape):
    """np.broadcast_to with a clearer error (value must fit the sample shape)."""
    try:
        return np.broadcast_to(value, sample_shape, subok=True)
    except ValueError as exc:
        exc.args += ("value cannot be broadcast to sample shape",)
        raise


def simplify_shape(value):
    """Collapse axes along which all entries are equal; drop leading unit axes."""
    value = np.asanyarray(value)
    for axis in range(value.ndim):
        first = value[(slice(None),) * axis + (slice(0, 1),)]
        if value.strides[axis] == 0 or np.all(value == first):
            value = first
    lead = 0
    while lead < value.ndim and value.shape[lead] == 1:
        lead += 1
    return value.reshape(value.shape[lead:]).copy()


def getattr_if_none(ih, attr, value=None, *, required=True, **kwargs):
    """`value`, else kwargs[attr], else ih.attr; TypeError if still missing."""
    if value is None:
        value = kwargs.get(attr, None)
        if value is None:
            value = getattr(ih, attr, None)
    if required and value is None:
        raise TypeError(f"{attr!r} should either be defined by the "
                        "underlying stream or passed in.")
    return value


def _normalise_meta(attr, value):
    if attr == 'frequency':
        return u.to_hz(value)
    if attr == 'sideband':
        return np.where(np.asanyarray(value) > 0, np.int8(1), np.int8(-1))
    return np.asanyarray(value)


def _stream_rate(ih):
    return u.to_hz(ih.sample_rate)


def _stream_start(ih):
    return Time(ih.start_time)


# ---------------------------------------------------------------------------
class Base:
    """Common machinery of sources and tasks.

    Subclasses provide ``_read_frame(frame_index) -> ndarray`` holding
    ``samples_per_frame`` complete samples (fewer only for a final frame).

    Parameters
    ----------
    shape : tuple
        ``(n_complete_samples,) + sample_shape``.
    start_time : `~baseband_tasks_amd.units.Time` (or ISO string)
    sample_rate : float
        Complete samples per second, in Hz.
    samples_per_frame : int
    dtype : numpy dtype
    frequency, sideband, polarization : optional metadata, broadcastable to
        the sample shape (frequency in Hz; ``frequency`` and ``sideband`` must
        be given together).
    """
    offset = 0
    closed = False
    _frame_index = None
    _frame = None

    def __init__(self, shape, start_time, sample_rate, *, samples_per_frame=1,
                 dtype=np.complex64, **kwargs):
        self._shape = tuple(shape)
        self._start_time = Time(start_time)
        self._sample_rate = u.to_hz(sample_rate)
        self._samples_per_frame = operator.index(samples_per_frame)
        self._dtype = np.dtype(dtype)
        unknown = set(kwargs) - set(META_ATTRIBUTES)
        if unknown:
            raise TypeError("__init__() got unexpected keyword argument "
                            f"{sorted(unknown)[0]!r}")
        given = {k: v for k, v in kwargs.items() if v is not None}
        if ('frequency' in given) != ('sideband' in given):
            raise ValueError('frequency and sideband should both be passed in.')
        self.meta = dict(getattr(self, 'meta', {}) or {})
        attributes = dict(self.meta.get('__attributes__', {}))
        for attr, value in given.items():
            attributes[attr] = self._check_shape(_normalise_meta(attr, value))
        if attributes:
            self.meta['__attributes__'] = attributes

    # -- metadata -----------------------------------------------------------
    def __getattr__(self, attr):
        if attr in META_ATTRIBUTES:
            value = self.__dict__.get('meta', {}).get('__attributes__', {}).get(attr)
            if value is None:
                raise AttributeError(f"{attr} not set.")
            return value
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {attr!r}")

    def __dir__(self):
        return sorted(set(META_ATTRIBUTES).union(super().__dir__()))

    def _check_shape(self, value):
        return simplify_shape(check_broadcast_to(value, self.sample_shape))

    # -- description ----------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def sample_shape(self):
        return self._shape[1:]

    @property
    def samples_per_frame(self):
        return self._samples_per_frame

    @property
    def size(self):
        n = 1
        for d in self._shape:
            n *= d
        return n

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def dtype(self):
        return self._dtype

    @property
    def complex_data(self):
        return self._dtype.kind == 'c'

    @property
    def sample_rate(self):
        return self._sample_rate

    @property
    def start_time(self):
        return self._tell_time(0)

    @property
    def time(self):
        return self._tell_time(self.offset)

    @property
    def stop_time(self):
        return self._tell_time(self._shape[0])

    def _tell_time(self, offset):
        return self._start_time + offset / self._sample_rate

    # -- repr: the constructor call that would rebuild the object (reference base.py:174-233) --
    def _constructor_parameters(self):
        """Parameters of this class's constructor and, as long as a constructor
        forwards ``**kwargs``, of its bases' constructors (first mention wins)."""
        found = {}
        for cls in type(self).__mro__:
            try:
                params = inspect.signature(cls).parameters
            except (TypeError, ValueError):
                break
            for key, par in params.items():
                found.setdefault(key, par)
            if cls is Base or not any(par.kind is par.VAR_KEYWORD for par in params.values()):
                break
        return found

    def _repr_item(self, key, default, value=None):
        """``key=value`` for a constructor argument whose value on the instance
        (attribute ``key`` or ``_key``) is set and differs from ``default``; else None."""
        if value is None:
            value = getattr(self, key, None)
            if value is None:
                value = getattr(self, '_' + key, None)
            if value is None:
                return None
        if default is not inspect.Parameter.empty:
            try:
                if np.all(value == default):
                    return None
            except Exception:
                pass
        if isinstance(value, Time):
            value = value.isot
        elif isinstance(value, np.ndarray) and value.size > 16:
            value = np.array2string(value, threshold=6)
        return f"{key}={value}".replace('\n', ',')

    def __repr__(self):
        name = type(self).__name__
        params = self._constructor_parameters()
        items = [self._repr_item(key, par.default) for key, par in params.items()
                 if par.kind not in (par.VAR_KEYWORD, par.VAR_POSITIONAL)]
        items += [self._repr_item(key, None) for key in self.meta.get('__attributes__', {}) if key not in params]
        sep = ",\n " + " " * len(name)
        return f"{name}({sep.join(item for item in items if item)})"

    # -- pointer ---------------------------------------------------------------
    def seek(self, offset, whence=0):
        """Move the sample pointer: integer samples, seconds-offset given as a
        quantity-like float is NOT guessed -- pass a `Time` for absolute times
        (rounded to the nearest sample), an int for samples."""
        if u.is_time(offset):
            offset = int(round((Time(offset) - self.start_time) * self.sample_rate))
            whence = 0
        elif hasattr(offset, 'to_value'):       # astropy time quantity
            offset = int(round(offset.to_value('s') * self.sample_rate))
        else:
            offset = operator.index(offset)
        if whence == 0 or whence == 'start':
            self.offset = offset
        elif whence == 1 or whence == 'current':
            self.offset += offset
        elif whence == 2 or whence == 'end':
            self.offset = self.shape[0] + offset
        else:
            raise ValueError("invalid 'whence'; should be 0 or 'start', 1 or "
                             "'current', or 2 or 'end'.")
        return self.offset

    def tell(self, unit=None):
        """Offset in samples (default), 'time' for the absolute time, or a
        float number of seconds scaled by ``unit`` (e.g. ``u.ms``)."""
        if unit is None:
            return self.offset
        if isinstance(unit, str) and unit == 'time':
            return self._tell_time(self.offset)
        return self.offset / self.sample_rate / float(unit)

    # -- reading -----------------------------------------------------------------
    def _prepare_read(self, count, out):
        if self.closed:
            raise ValueError("I/O operation on closed stream.")
        samples_left = self.shape[0] - self.offset
        if out is None:
            if count is None or count < 0:
                count = max(0, samples_left)
        else:
            assert out.shape[1:] == self.sample_shape, (
                "'out' must have trailing shape {}".format(self.sample_shape))
            count = out.shape[0]
        if count > samples_left:
            raise EOFError("cannot read from beyond end of input.")
        if self.offset < 0:
            raise OSError("cannot read from before the start of the stream "
                          f"(offset {self.offset}).")
        return count

    def read(self, count=None, out=None):
        """Read ``count`` complete samples from the pointer (all that is left
        if None) into a new array or into ``out``."""
        count = self._prepare_read(count, out)
        if out is None:
            out = np.empty((count,) + self.sample_shape, dtype=self.dtype)
        start = self.offset
        done = 0
        while done < count:
            frame, skip = self._get_frame(start + done)
            n = min(count - done, len(frame) - skip)
            out[done:done + n] = frame[skip:skip + n]
            done += n
            self.offset = start + done
        return out

    def _get_frame(self, offset):
        """(frame, index of ``offset`` inside it); one frame is cached."""
        index, skip = divmod(offset, self.samples_per_frame)
        if index != self._frame_index:
            self.offset = index * self.samples_per_frame
            self._frame = self._read_frame(index)
            self._frame_index = index
        return self._frame, skip

    def read_device(self, count=None):
        """As `read`, but returns a `hip.DeviceArray` in HBM.  For host-only
        streams this uploads; the returned array is owned by the caller."""
        from .hip import DeviceArray
        return DeviceArray.from_host(self.read(count))

    # -- conveniences --------------------------------------------------------------
    def __getitem__(self, item):
        if isinstance(item, tuple) and len(item) == 1:
            item = item[0]
        if not isinstance(item, slice):
            raise NotImplementedError(
                "only slices along the time axis are supported here "
                "(shaping.GetItem/GetSlice are outside the accelerated path).")
        return _TimeSlice(self, item)

    def __array__(self, dtype=None, copy=None):
        old = self.tell()
        try:
            self.seek(0)
            return np.array(self.read(), dtype=dtype)
        finally:
            self.seek(old)

    def __array_ufunc__(self, *args, **kwargs):
        return NotImplemented

    def __array_function__(self, *args, **kwargs):
        return NotImplemented

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def close(self):
        self.closed = True
        self._frame = None
        self._frame_index = None


# ---------------------------------------------------------------------------
class BaseTaskBase(Base):
    """A stream defined on top of another stream ``ih``; parameters default to
    those of ``ih`` (reference base.py:499-611)."""

    def __init__(self, ih, *, ih_samples_per_frame=None, start_time=None,
                 shape=None, sample_rate=None, samples_per_frame=None,
                 dtype=None, **kwargs):
        self.ih = ih
        if ih_samples_per_frame is None:
            ih_samples_per_frame = ih.samples_per_frame
        self._ih_samples_per_frame = operator.index(ih_samples_per_frame)
        if shape is None:
            shape = ih.shape
        if start_time is None:
            start_time = _stream_start(ih)
        if sample_rate is None:
            sample_rate = _stream_rate(ih)
        if dtype is None:
            dtype = ih.dtype
        if samples_per_frame is None:
            samples_per_frame = ih_samples_per_frame
        # inherit metadata, explicit keywords win, Nones are dropped
        self.meta = {k: (dict(v) if isinstance(v, dict) else v)
                     for k, v in (getattr(ih, 'meta', {}) or {}).items()}
        inherited = self.meta.pop('__attributes__', {})
        for attr in META_ATTRIBUTES:
            value = kwargs.pop(attr, None)
            if value is None:
                value = inherited.get(attr)
                if value is None:
                    value = getattr(ih, attr, None)   # foreign stream readers
            if value is not None:
                kwargs[attr] = value
        super().__init__(shape=shape, start_time=start_time, sample_rate=sample_rate,
                         samples_per_frame=samples_per_frame, dtype=dtype, **kwargs)

    def _repr_item(self, key, default, value=None):
        """Arguments left at None default to what the underlying stream has."""
        if key == 'ih':
            return 'ih'
        if default is None:
            if key == 'samples_per_frame':
                default = getattr(self, '_ih_samples_per_frame', None)
            elif key == 'ih_samples_per_frame':
                default = getattr(self.ih, 'samples_per_frame', None)
            else:
                default = getattr(self.ih, key, None)
        return super()._repr_item(key, default, value)

    def __repr__(self):
        head = super().__repr__()
        if head.count('\n') == 1:
            head = ' '.join(part.strip() for part in head.split('\n'))
        inner = repr(self.ih) if not self.closed else '(closed)'
        return head + "\nih: " + "\n    ".join(inner.split('\n'))

    def close(self):
        """Drop the reference to the underlying stream (it is not closed)."""
        super().close()
        self.__dict__.pop('ih', None)


class TaskBase(BaseTaskBase):
    """Frame-by-frame task: ``_read_frame`` reads ``ih_samples_per_frame``
    input samples and returns ``self.task(data)`` (reference base.py:613-707).
    """

    def __init__(self, ih, *, ih_samples_per_frame=None, shape=None,
                 sample_rate=None, samples_per_frame=None, **kwargs):
        ih_rate = _stream_rate(ih)
        if sample_rate is None:
            sample_rate = ih_rate
            ratio = 1.
        else:
            sample_rate = u.to_hz(sample_rate)
            ratio = ih_rate / sample_rate
            # rates here are floats in Hz: (fs / 7) * 7 is not fs to the last bit, so snap
            # a ratio within rounding of an integer (or of 1 / integer) onto it
            nearest = float(round(ratio)) if ratio >= 1. else 1. / max(round(1. / ratio), 1)
            if abs(nearest - ratio) <= 1e-12 * ratio:
                ratio = nearest
        if samples_per_frame is None:
            if ih_samples_per_frame is None:
                ih_samples_per_frame = ih.samples_per_frame
            spf = ih_samples_per_frame / ratio
            assert spf % 1 == 0, "inferred samples per frame must be integer"
            samples_per_frame = int(spf)
        elif ih_samples_per_frame is None:
            ih_spf = samples_per_frame * ratio
            assert ih_spf % 1 == 0, "inferred input samples per frame must be integer"
            ih_samples_per_frame = int(ih_spf)
        assert ih_samples_per_frame <= ih.shape[0], (
            "time per frame larger than total time in stream")
        if shape is None or shape[0] == -1:
            n = (ih.shape[0] // ih_samples_per_frame) * samples_per_frame
            shape = (n,) + tuple(ih.shape[1:] if shape is None else shape[1:])
        super().__init__(ih=ih, ih_samples_per_frame=ih_samples_per_frame, shape=shape,
                         sample_rate=sample_rate, samples_per_frame=samples_per_frame,
                         **kwargs)
        alignment = max(1, int(ratio))
        self._ih_stop = (self.ih.shape[0] // alignment) * alignment

    def _seek_frame(self, frame_index):
        return self.ih.seek(frame_index * self._ih_samples_per_frame)

    def _read_frame(self, frame_index):
        start = self._seek_frame(frame_index)
        stop = min(start + self._ih_samples_per_frame, self._ih_stop)
        return self.task(self.ih.read(stop - start))


class PaddedTaskBase(TaskBase):
    """Overlap-save segmentation: each output frame of ``samples_per_frame``
    samples needs ``pad_start`` extra input samples before and ``pad_end``
    after it (reference base.py:709-795).

    ``samples_per_frame`` defaults to the size that keeps the padding below
    25 % of the input block; ``next_fast_len`` (from the FFT engine) may
    enlarge the input block.
    """

    def __init__(self, ih, pad_start=0, pad_end=0, *, samples_per_frame=None,
                 next_fast_len=None, **kwargs):
        self._pad_start = operator.index(pad_start)
        self._pad_end = operator.index(pad_end)
        if self._pad_start < 0 or self._pad_end < 0:
            raise ValueError("padding values must be 0 or positive.")
        pad = self._pad_start + self._pad_end
        if samples_per_frame is None:
            ih_spf = max(ih.samples_per_frame, 4 * pad)
        else:
            ih_spf = samples_per_frame + pad
        if next_fast_len:
            ih_spf = next_fast_len(ih_spf)
        samples_per_frame = ih_spf - pad
        if pad > samples_per_frame:
            warnings.warn("task will be inefficient; for {} samples per frame, "
                          "more ({}) will be added for padding."
                          .format(samples_per_frame, pad))
        start_time = kwargs.pop('start_time', None)
        if start_time is None:
            start_time = _stream_start(ih)
        kwargs['start_time'] = Time(start_time) + self._pad_start / _stream_rate(ih)
        self._frame_offset = 0
        super().__init__(ih, ih_samples_per_frame=ih_spf,
                         shape=(ih.shape[0] - pad,) + tuple(ih.shape[1:]),
                         samples_per_frame=samples_per_frame, **kwargs)

    def _block_start(self, frame_index):
        """(input start, output samples to skip) for a frame; the last frame
        is re-aligned to end at the end of the input."""
        wanted = frame_index * self.samples_per_frame
        last_start = self.ih.shape[0] - self._ih_samples_per_frame
        if wanted > last_start:
            return last_start, wanted - last_start
        return wanted, 0

    def _seek_frame(self, frame_index):
        start, self._frame_offset = self._block_start(frame_index)
        return self.ih.seek(start)

    def _get_frame(self, offset):
        frame, skip = super()._get_frame(offset)
        return frame, skip + self._frame_offset


class Task(TaskBase):
    """Apply a user callable to each frame (function ``f(data)`` or method-like
    ``f(task, data)``; reference base.py:798-889)."""

    def __init__(self, ih, task, method=None, **kwargs):
        if method is None:
            try:
                spec = inspect.getfullargspec(task)
                narg = len(spec.args) - len(spec.defaults or ())
                if inspect.ismethod(task):
                    narg -= 1
                assert 1 <= narg <= 2
                method = narg == 2
            except Exception as exc:
                exc.args += ("cannot determine whether ``task`` is a "
                             "function or method. Pass in ``method``.",)
                raise
        self.task = types.MethodType(task, self) if method else task
        super().__init__(ih, **kwargs)


class SinglePrecision(TaskBase):
    """float64 / complex128 stream -> float32 / complex64, frame by frame on the
    host.  The kernels of this package compute in single precision (like the
    reference on its usual float32 / complex64 `baseband` data); the reference
    also accepts double-precision streams (its PFB tests use them), so a chain
    written for such a stream becomes ``Dedisperse(SinglePrecision(fh), dm)``.
    Streams that are already single precision pass through unchanged."""

    def __init__(self, ih, **kwargs):
        kind = np.dtype(ih.dtype).kind
        if kind not in 'fc':
            raise TypeError(f"cannot convert a stream of {ih.dtype} to single precision.")
        super().__init__(ih, dtype=np.complex64 if kind == 'c' else np.float32, **kwargs)

    def task(self, data):
        return np.ascontiguousarray(data, dtype=self.dtype)


class SetAttribute(TaskBase):
    """Pass-through that sets/overrides start_time, sample_rate or metadata
    (reference base.py:892-951)."""

    def __init__(self, ih, *, start_time=None, sample_rate=None, **kwargs):
        super().__init__(ih, start_time=start_time, sample_rate=sample_rate, **kwargs)
        self._passthrough = not set(kwargs).difference(META_ATTRIBUTES)

    def read(self, count=None, out=None):
        if not self._passthrough:
            return super().read(count, out)
        count = self._prepare_read(count, out)
        self.ih.seek(self.offset)
        result = self.ih.read(count, out) if out is not None else self.ih.read(count)
        self.offset += count
        return result

    @property
    def _produces_on_device(self):
        # (a pass-through: on the device if the stream below is)
        return self._passthrough and bool(getattr(self.ih, '_produces_on_device', False))

    def read_device(self, count=None):
        if not self._passthrough or not hasattr(self.ih, 'read_device'):
            return super().read_device(count)
        count = self._prepare_read(count, None)
        self.ih.seek(self.offset)
        result = self.ih.read_device(count)
        self.offset += count
        return result

    def task(self, data):
        return data


class _TimeSlice(Base):
    """``stream[start:stop]``: a window on the time axis."""

    def __init__(self, ih, item):
        start, stop, step = item.indices(ih.shape[0])
        if step != 1:
            raise NotImplementedError("strided time slices are outside the accelerated path.")
        stop = max(start, stop)
        self.ih = ih
        self._first = start
        self.meta = getattr(ih, 'meta', {})
        super().__init__((stop - start,) + tuple(ih.shape[1:]),
                         _stream_start(ih) + start / _stream_rate(ih), _stream_rate(ih),
                         samples_per_frame=ih.samples_per_frame, dtype=ih.dtype)

    def read(self, count=None, out=None):
        count = self._prepare_read(count, out)
        self.ih.seek(self._first + self.offset)
        result = self.ih.read(count, out) if out is not None else self.ih.read(count)
        self.offset += count
        return result

    @property
    def _produces_on_device(self):
        return bool(getattr(self.ih, '_produces_on_device', False))

    def read_device(self, count=None):
        count = self._prepare_read(count, None)
        self.ih.seek(self._first + self.offset)
        result = self.ih.read_device(count)
        self.offset += count
        return result
