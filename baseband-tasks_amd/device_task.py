"""Frame machinery shared by the GPU tasks.

A GPU task computes whole output frames in HBM, many per C-ABI call, and
keeps the last computed run of frames as its frame cache (the reference keeps
one host frame: baseband_tasks/base.py:459-465).  ``read`` copies to the host
only what the caller asked for; ``read_device`` returns a zero-copy view of
the cache (valid until the next read on the same task).
"""
import numpy as np

from . import hip
from .hip import DeviceArray

__all__ = ['DeviceTaskMixin', 'fetch_device']


def fetch_device(ih, start, count):
    """Samples [start, start+count) of stream ``ih`` as a DeviceArray.

    Streams of this package hand over device memory directly; any other
    stream reader (e.g. a `baseband` file handle) is read on the host and
    uploaded.
    """
    ih.seek(start)
    if hasattr(ih, 'read_device'):
        return ih.read_device(count)
    return DeviceArray.from_host(ih.read(count))


class DeviceTaskMixin:
    """Mix into a `TaskBase` subclass; the subclass provides

    ``_compute_frames(first, last, out)``: fill DeviceArray ``out`` (flat
    ``(n_samples,) + sample_shape``) with output frames ``first..last-1``.
    """
    #: upper bound on frames computed by one call (bounds device memory)
    max_frames_per_call = 32

    @property
    def _device_dtype(self):
        """dtype of the frames in HBM (a task may present another one to its readers)."""
        return self.dtype

    _cache = None          # DeviceArray holding frames [_cache_first, _cache_last)
    _cache_first = 0
    _cache_last = 0
    _cache_buffer = None   # reusable allocation behind _cache

    def _n_frames(self):
        spf = self.samples_per_frame
        return -(-self.shape[0] // spf)

    def _frame_span(self, first, last):
        """Output samples covered by frames [first, last)."""
        spf = self.samples_per_frame
        return first * spf, min(last * spf, self.shape[0])

    def _out_buffer(self, n_samples):
        row = 1
        for d in self.sample_shape:
            row *= d
        need = n_samples * row
        buf = self._cache_buffer
        if buf is None or buf.size < need:
            # drop the old one first so peak memory is one buffer
            self._cache = self._cache_buffer = None
            buf = self._cache_buffer = DeviceArray((need,), self._device_dtype)
        return buf[:need].reshape((n_samples,) + tuple(self.sample_shape))

    def _ensure_frames(self, first, last):
        """Make frames [first, last) resident; returns (cache, first sample)."""
        if self._cache is not None and self._cache_first <= first and last <= self._cache_last:
            return self._cache, self._cache_first * self.samples_per_frame
        start, stop = self._frame_span(first, last)
        reuse = None
        if (self._cache is not None and self._cache_last - 1 == first
                and last > first + 1):
            # sequential reading: keep the frame straddled by the previous
            # request instead of recomputing it
            s0, s1 = self._frame_span(first, first + 1)
            c0 = self._cache_first * self.samples_per_frame
            reuse = DeviceArray((s1 - s0,) + tuple(self.sample_shape), self._device_dtype)
            reuse.copy_from_device(self._cache[s0 - c0:s1 - c0])
        out = self._out_buffer(stop - start)
        if reuse is not None:
            out[:len(reuse)].copy_from_device(reuse)
            self._compute_frames(first + 1, last, out[len(reuse):])
        else:
            self._compute_frames(first, last, out)
        self._cache, self._cache_first, self._cache_last = out, first, last
        return out, start

    def read_device(self, count=None):
        """Like ``read`` but the samples stay in HBM.  The result is a view of
        this task's frame cache: consume it before reading from the task
        again."""
        count = self._prepare_read(count, None)
        if count == 0:
            return DeviceArray((0,) + tuple(self.sample_shape), self._device_dtype)
        spf = self.samples_per_frame
        first = self.offset // spf
        last = (self.offset + count - 1) // spf + 1
        if last - first > self.max_frames_per_call:
            # too much for one cache: assemble piecewise into a fresh array
            out = DeviceArray((count,) + tuple(self.sample_shape), self._device_dtype)
            done = 0
            while done < count:
                pos = self.offset
                f0 = pos // spf
                f1 = min(last, f0 + self.max_frames_per_call)
                cache, c0 = self._ensure_frames(f0, f1)
                n = min(count - done, self._frame_span(f0, f1)[1] - pos)
                out[done:done + n].copy_from_device(cache[pos - c0:pos - c0 + n])
                done += n
                self.offset = pos + n
            return out
        cache, c0 = self._ensure_frames(first, last)
        view = cache[self.offset - c0:self.offset - c0 + count]
        self.offset += count
        return view

    def read(self, count=None, out=None):
        count = self._prepare_read(count, out)
        if out is None:
            out = np.empty((count,) + tuple(self.sample_shape), dtype=self._device_dtype)
        spf = self.samples_per_frame
        done = 0
        while done < count:
            pos = self.offset
            f0 = pos // spf
            f1 = min((pos + (count - done) - 1) // spf + 1, f0 + self.max_frames_per_call)
            cache, c0 = self._ensure_frames(f0, f1)
            n = min(count - done, self._frame_span(f0, f1)[1] - pos)
            piece = cache[pos - c0:pos - c0 + n]
            target = out[done:done + n]
            if isinstance(target, np.ndarray) and target.flags.c_contiguous \
                    and target.dtype == self._device_dtype:
                piece.to_host(target)
            else:
                out[done:done + n] = piece.to_host()
            done += n
            self.offset = pos + n
        return out

    def _read_frame(self, frame_index):
        cache, c0 = self._ensure_frames(frame_index, frame_index + 1)
        s0, s1 = self._frame_span(frame_index, frame_index + 1)
        return cache[s0 - c0:s1 - c0].to_host()

    def _get_frame(self, offset):
        index, skip = divmod(offset, self.samples_per_frame)
        if index != self._frame_index:
            self._frame = self._read_frame(index)
            self._frame_index = index
        return self._frame, skip

    def _drop_cache(self):
        self._cache = self._cache_buffer = None
        self._cache_first = self._cache_last = 0

    def invalidate_cache(self):
        """Forget cached frames (they will be recomputed) but keep the device
        buffer, so repeated passes do not reallocate."""
        self._cache = None
        self._cache_first = self._cache_last = 0
        self._frame = self._frame_index = None

    def synchronize(self):
        hip.synchronize()
