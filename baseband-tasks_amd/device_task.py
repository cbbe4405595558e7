"""Frame machinery shared by the GPU tasks.

A GPU task computes whole output frames in HBM, many per C-ABI call, and
keeps the last computed run of frames as its frame cache (the reference keeps
one host frame: baseband_tasks/base.py:459-465).  ``read`` copies to the host
only what the caller asked for; ``read_device`` returns a zero-copy view of
the cache (valid until the next read on the same task).
"""
import numpy as np

from . import hip
from . import host_pipeline
from .hip import DeviceArray

__all__ = ['DeviceTaskMixin', 'fetch_device']


def fetch_device(ih, start, count):
    """Samples [start, start+count) of stream ``ih`` as a DeviceArray.

    Streams of this package hand over device memory directly; any other
    stream reader (e.g. a `baseband` file handle) is read on the host and
    uploaded -- through the stream's `host_pipeline.HostUploader`, which may
    have the range on its way already (`prefetch_device`).
    """
    if produces_on_device(ih):
        ih.seek(start)
        return ih.read_device(count)
    up = host_pipeline.uploader_for(ih) if count > 0 else None
    if up is not None:
        return up.fetch(start, count)
    ih.seek(start)
    return DeviceArray.from_host(ih.read(count))


def produces_on_device(ih):
    """Does ``ih.read_device`` hand over samples that are (made) in HBM -- a device task, a
    `DeviceStream`, an unpacking reader -- rather than upload what ``ih.read`` returns (the
    fallback every `Base` stream has)?"""
    return bool(getattr(ih, '_produces_on_device', False))


def host_request(ih, start, count):
    """Follow a chain of device tasks down to the host stream that a fetch of
    samples [start, start+count) of ``ih`` would read: (host stream, start,
    count), or None if the chain ends on the device or a task on the way
    cannot say what it needs (`_input_span`)."""
    for _ in range(16):
        if not produces_on_device(ih):
            return ih, start, count
        span = getattr(ih, '_input_span_of_samples', None)
        nxt = span(start, count) if span is not None else None
        if nxt is None:
            return None
        ih, start, count = nxt
    return None


def prefetch_device(ih, start, count):
    """Start the upload of what `fetch_device(ih, start, count)` will need, if
    that comes from a host stream (no-op otherwise)."""
    try:
        req = host_request(ih, start, count)
    except Exception:
        return
    if req is None:
        return
    up = host_pipeline.uploader_for(req[0])
    if up is not None:
        up.prefetch(req[1], req[2])


class DeviceTaskMixin:
    """Mix into a `TaskBase` subclass; the subclass provides

    ``_compute_frames(first, last, out)``: fill DeviceArray ``out`` (flat
    ``(n_samples,) + sample_shape``) with output frames ``first..last-1``.
    """
    _produces_on_device = True
    _max_frames_per_call = None

    @property
    def max_frames_per_call(self):
        """Upper bound on the frames computed by one call (bounds device memory): as many as make
        512 MiB of output, 32 at least -- a call is a handful of kernel launches whatever it holds
        (32 frames of 8192 samples are 4 MiB), a plan with lanes wants a dozen blocks of 2^20 samples
        per call to keep them busy, and every task of a chain holds two such buffers of 288 GB.
        Assignable."""
        if self._max_frames_per_call is not None:
            return self._max_frames_per_call
        try:
            row = np.dtype(self._device_dtype).itemsize
            for d in self.sample_shape:
                row *= d
            frame = max(int(self.samples_per_frame) * row, 1)
        except Exception:
            return 32
        return max(32, min((1 << 29) // frame, 1 << 18))

    @max_frames_per_call.setter
    def max_frames_per_call(self, value):
        self._max_frames_per_call = None if value is None else int(value)

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
        if buf is not None and buf.pending:
            # A deferred plan call (hip.DEFER_JOIN) may still be writing the buffer of the previous
            # run -- or its reader may not have come for it yet.  Taking it again would order this
            # run after that one (the drain the deferral avoids), so runs alternate between two
            # buffers: the other one was written two calls back, its event has long fired.
            self._cache_buffer, self._cache_buffer_b = self._cache_buffer_b, buf
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
        if last - first > self.max_frames_per_call + 2:
            # too much for one cache: assemble piecewise into a fresh array.  (+ 2: a consumer that
            # sizes its own runs by the same bound asks for as many samples as this task's bound holds,
            # and they straddle a frame at either end -- that request is still ONE run here, not a
            # piecewise read whose pieces the one-range-ahead uploader cannot follow: a host stream
            # under Channelize(Dedisperse(...)) read at 1.0 instead of 2.3 Gsamples/s.)  `max_frames_per_call`
            # bounds what a run asks of the upstream task (its cache); when the frames are
            # computed straight into the result from a stream that is resident in HBM anyway,
            # nothing needs bounding and all whole frames go in one run
            out = DeviceArray((count,) + tuple(self.sample_shape), self._device_dtype)
            per = self.max_frames_per_call
            try:
                span = self._input_span(first, first + 1)
            except Exception:
                span = None
            if span is not None and getattr(span[0], '_resident', False):
                per = last - first
            done = 0
            while done < count:
                pos = self.offset
                f0 = pos // spf
                left = count - done
                direct = pos == f0 * spf                  # (a run that starts inside a frame: that frame, via the cache)
                f1 = f0 + 1
                if direct:
                    f1 = min(last, f0 + per)
                    if self._frame_span(f0, f1)[1] - pos > left:
                        f1 = f0 + left // spf             # whole frames only; the rest of the request is a partial one
                    if f1 == f0:
                        direct, f1 = False, f0 + 1
                s0, s1 = self._frame_span(f0, f1)
                n = min(left, s1 - pos)
                if direct:
                    # whole frames: computed straight into their place -- no cache, no copy; the
                    # runs fill disjoint slices of the fresh array, so deferred plan calls need not
                    # wait for each other (`DeviceArray.fresh`) and the lanes run on from run to run
                    piece = out[done:done + n]
                    piece.fresh = True
                    self._compute_frames(f0, f1, piece)
                else:
                    cache, c0 = self._ensure_frames(f0, f1)
                    out[done:done + n].copy_from_device(cache[pos - c0:pos - c0 + n])
                done += n
                self.offset = pos + n
            return out
        cache, c0 = self._ensure_frames(first, last)
        view = cache[self.offset - c0:self.offset - c0 + count]
        self.offset += count
        return view

    def _input_span(self, first, last):
        """(upstream, start, count): the one `fetch_device` call that computing
        frames [first, last) makes, or None if the task cannot tell in advance.
        Lets ``read`` start the upload of the next run of frames while this one
        is computed."""
        return None

    def _input_span_of_samples(self, start, count):
        """`_input_span` for the frames a ``read_device(count)`` at ``start`` would compute."""
        spf = self.samples_per_frame
        first, last = start // spf, (start + count - 1) // spf + 1
        if self._cache is not None and self._cache_first <= first and last <= self._cache_last:
            return None
        if last - first > self.max_frames_per_call + 2:
            return None
        return self._input_span(first, last)

    #: Frames per run of the pipelined host path of ``read`` (upload, transforms and
    #: download of consecutive runs overlap); None: `max_frames_per_call`.
    host_frames_per_run = None

    def _read_pipelined(self, count, out):
        """``read`` with the three stages of consecutive runs of frames overlapping
        (host_pipeline.py): while run m is computed on the current stream, the input
        of run m + 1 is read and uploaded by the host stream's worker and the result
        of run m - 1 goes down on a stream of its own, into page-locked memory."""
        spf = self.samples_per_frame
        per = self.host_frames_per_run
        if not per:
            # runs of about 128 MiB of output: long enough to move at the bus's rate, short enough
            # that a read of a gigabyte is many runs whose stages overlap (32 frames of 2^20 samples
            # are half a gigabyte: two runs, nothing to overlap)
            row = np.dtype(self._device_dtype).itemsize
            for d in self.sample_shape:
                row *= d
            per = max(1, min(self.max_frames_per_call, -(-(1 << 27) // max(spf * row, 1))))
        runs, pos, done = [], self.offset, 0
        while done < count:
            f0 = pos // spf
            f1 = min((pos + (count - done) - 1) // spf + 1, f0 + per)
            n = min(count - done, self._frame_span(f0, f1)[1] - pos)
            runs.append((f0, f1, pos, done, n))
            pos += n
            done += n
        down = host_pipeline.copy_streams()[1]
        # two result buffers in turn: run m + 2 is computed into the buffer run m came from, once
        # that run has gone down
        buffers = [self._cache_buffer, self._cache_buffer_b]
        for buf in buffers:
            # A buffer may still be owed a deferred call from before this read (a `read_device`
            # result nobody touched, or one only a downstream deferred plan read): `_out_buffer`
            # would then swap the two behind this loop's back and two consecutive runs would share
            # one buffer, the second computing into it while the first is still going down.
            # Settled here, and again by `piece.ptr` after every run, no swap happens inside.
            if buf is not None:
                buf.ptr
        gone = [None, None]
        computed = host_pipeline.StreamEvent()
        for i, (f0, f1, pos, done, n) in enumerate(runs):
            if i + 1 < len(runs):
                nxt = self._input_span(runs[i + 1][0], runs[i + 1][1])
                if nxt is not None:
                    prefetch_device(*nxt)
            b = i % 2
            self._cache_buffer = buffers[b]
            if gone[b] is not None:
                host_pipeline.current_stream_wait(gone[b])
            other = buffers[1 - b]
            if (gone[1 - b] is not None and other is not None and buffers[b] is not None
                    and other.owner is buffers[b].owner):
                # (both turns on one allocation -- cannot happen after the settling above; if it
                # ever does, the run waits for the other turn's download as well)
                host_pipeline.current_stream_wait(gone[1 - b])
            cache, c0 = self._ensure_frames(f0, f1)
            buffers[b] = self._cache_buffer
            piece = cache[pos - c0:pos - c0 + n]
            target = out[done:done + n]
            src = piece.ptr                  # (orders the current stream after a deferred producer)
            computed.record()
            down.wait(computed)
            hip.check(hip.lib().bbt_memcpy_d2h(target.ctypes.data, src, piece.nbytes, down.handle))
            gone[b] = host_pipeline.StreamEvent().record(down)
            self.offset = pos + n
        self._cache_buffer_b = buffers[(len(runs)) % 2]      # (the other one stays the cache's)
        down.synchronize()
        return out

    _cache_buffer_b = None

    def read(self, count=None, out=None):
        count = self._prepare_read(count, out)
        # (page-locked results and the three-stream pipeline from 1 MiB on: below that the
        # pinned block -- 2 MiB at least -- and the extra streams cost more than they hide)
        row = np.dtype(self._device_dtype).itemsize
        for d in self.sample_shape:
            row *= d
        pipelined = host_pipeline.ENABLED and count * row >= (1 << 20)
        if out is None:
            empty = host_pipeline.pinned_empty if pipelined else np.empty
            out = empty((count,) + tuple(self.sample_shape), dtype=self._device_dtype)
        if (pipelined and isinstance(out, np.ndarray) and out.flags.c_contiguous
                and out.dtype == self._device_dtype and host_pipeline.is_pinned(out)):
            return self._read_pipelined(count, out)
        spf = self.samples_per_frame
        done = 0
        while done < count:
            pos = self.offset
            f0 = pos // spf
            f1 = min((pos + (count - done) - 1) // spf + 1, f0 + self.max_frames_per_call)
            cache, c0 = self._ensure_frames(f0, f1)
            n = min(count - done, self._frame_span(f0, f1)[1] - pos)
            piece = cache[pos - c0:pos - c0 + n]
            # (``out`` may be anything with a shape and slice assignment -- the HDF5 writer,
            # Integrate's accumulator: reference base.py:416-433 -- not only an array)
            target = out[done:done + n] if isinstance(out, np.ndarray) else None
            if target is not None and target.flags.c_contiguous and target.dtype == self._device_dtype:
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
        self._cache = self._cache_buffer = self._cache_buffer_b = None
        self._cache_first = self._cache_last = 0

    def invalidate_cache(self):
        """Forget cached frames (they will be recomputed) but keep the device
        buffer, so repeated passes do not reallocate."""
        self._cache = None
        self._cache_first = self._cache_last = 0
        self._frame = self._frame_index = None

    def synchronize(self):
        hip.synchronize()
