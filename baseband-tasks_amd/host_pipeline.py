"""The host path of ``read()``: NumPy in, NumPy out, PCIe both ways at once.

The reference's ``Base.read`` (base.py:389-438) returns host arrays and pulls
its input frame by frame from a host reader (base.py:699-706).  Kept as it is
on a GPU that means: read a run of input blocks, copy them up, compute, copy
the result down, and only then touch the next run -- the bus idle while the
GPU computes and the GPU idle while the bus moves data, both through pageable
memory (0.25-0.30 Gsamples/s for the metric pipeline, round 2).  Here the
three stages of consecutive runs overlap:

* `HostUploader` (one per host stream that feeds a device task): a worker
  thread reads run m + 1 from the stream -- straight into page-locked memory,
  or not at all when the stream's samples already sit in page-locked memory
  (`HostStream`) -- and queues its upload on a stream of its own, while the
  caller's stream computes run m;
* `DeviceTaskMixin.read` queues the download of run m - 1 on a third stream,
  into a page-locked result array (`pinned_empty`: what ``read`` returns is an
  ordinary ndarray whose memory happens to be pinned and goes back to a small
  pool when the array is garbage collected).

Events order the three streams (`bbt_stream_wait_event`); nothing here touches
the kernels.  ``BBT_HOST_PIPELINE=0`` switches back to the synchronous copies.
"""
import ctypes as C
import os
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import hip

__all__ = ['ENABLED', 'pinned_empty', 'is_pinned', 'Stream', 'StreamEvent', 'HostUploader', 'uploader_for']

ENABLED = os.environ.get('BBT_HOST_PIPELINE', '1') != '0'


# --------------------------------------------------------------------------- page-locked host arrays
class _PinnedPool:
    """Page-locked blocks by size; blocks of garbage-collected arrays come back
    here instead of going through hipHostFree / hipHostMalloc again (both take
    milliseconds per GB)."""

    def __init__(self):
        self.lock = threading.Lock()
        self.free = {}                      # nbytes -> [ptr, ...]
        self.cached = 0
        self.ranges = {}                    # ptr -> nbytes of every live or cached block
        # idle page-locked bytes kept at most: BBT_PINNED_POOL_GB, default 48 GB shared between the
        # ranks of a node (one process per GPU: LOCAL_WORLD_SIZE), and never more than a quarter
        # of the host's memory
        limit = os.environ.get('BBT_PINNED_POOL_GB')
        if limit is None:
            limit = 48. / max(int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1), 1)
            try:
                ram = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES')
                limit = min(limit, ram / 4 / 2**30)
            except (ValueError, OSError):
                pass
        self.limit = int(float(limit) * 2**30)

    def take(self, nbytes):
        with self.lock:
            stack = self.free.get(nbytes)
            if stack:
                self.cached -= nbytes
                return stack.pop()
        ptr = C.c_void_p()
        hip.check(hip.lib().bbt_host_alloc(C.byref(ptr), nbytes))
        with self.lock:
            self.ranges[ptr.value] = nbytes
        return ptr.value

    def give(self, ptr, nbytes):
        with self.lock:
            if self.cached + nbytes <= self.limit:
                self.free.setdefault(nbytes, []).append(ptr)
                self.cached += nbytes
                return
            self.ranges.pop(ptr, None)
        try:
            hip.lib().bbt_host_free(ptr)
        except Exception:
            pass

    def trim(self):
        with self.lock:
            blocks = [p for stack in self.free.values() for p in stack]
            self.free.clear()
            self.cached = 0
            for p in blocks:
                self.ranges.pop(p, None)
        for p in blocks:
            hip.lib().bbt_host_free(p)

    def covers(self, address, nbytes):
        with self.lock:
            for ptr, size in self.ranges.items():
                if ptr <= address and address + nbytes <= ptr + size:
                    return True
        return False


_pool = _PinnedPool()
_registered = {}            # address -> nbytes of ranges page-locked in place (pin_array)


class _PinnedBlock:
    def __init__(self, nbytes):
        self.nbytes = nbytes
        self.ptr = _pool.take(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                _pool.give(self.ptr, self.nbytes)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.complex64):
    """``np.empty(shape, dtype)`` in page-locked memory (sizes rounded up to 2 MiB so that
    repeated reads of about the same size find their block in the pool)."""
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if np.ndim(shape) else (shape,)))
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n == 0:
        return np.empty(shape, dtype)
    try:
        block = _PinnedBlock(-(-n // (2 << 20)) * (2 << 20))
    except hip.HipError:
        # (a memlock limit, or more than the host can lock: an ordinary array -- `read` then
        # takes the synchronous path, as `is_pinned` says no)
        return np.empty(shape, dtype)
    buf = (C.c_ubyte * n).from_address(block.ptr)
    buf._bbt_block = block                       # the array's base keeps the block alive
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def pin_array(array):
    """Page-lock the memory of ``array`` in place (hipHostRegister); returns True if it is
    pinned afterwards.  The registration is undone when ``array`` is garbage collected."""
    if not isinstance(array, np.ndarray) or not array.flags.c_contiguous or array.nbytes == 0:
        return False
    address = array.ctypes.data
    if is_pinned(array):
        return True
    try:
        hip.check(hip.lib().bbt_host_register(address, array.nbytes))
    except hip.HipError:
        return False
    _registered[address] = array.nbytes

    def release(address=address):
        _registered.pop(address, None)
        try:
            hip.lib().bbt_host_unregister(address)
        except Exception:
            pass
    try:
        weakref.finalize(array if array.base is None else array.base, release)
    except TypeError:
        pass
    return True


def is_pinned(array):
    """Does ``array`` (C-contiguous) lie in memory this module page-locked?"""
    if not isinstance(array, np.ndarray) or not array.flags.c_contiguous:
        return False
    address, n = array.ctypes.data, array.nbytes
    for ptr, size in list(_registered.items()):
        if ptr <= address and address + n <= ptr + size:
            return True
    return _pool.covers(address, n)


# --------------------------------------------------------------------------- streams and events
class Stream:
    def __init__(self):
        self._h = C.c_void_p()
        hip.check(hip.lib().bbt_stream_create(C.byref(self._h)))

    @property
    def handle(self):
        return self._h.value

    def synchronize(self):
        hip.check(hip.lib().bbt_stream_sync(self._h))

    def wait(self, event):
        hip.check(hip.lib().bbt_stream_wait_event(self._h, event._h))

    def __del__(self):
        try:
            if self._h:
                hip.lib().bbt_stream_destroy(self._h)
                self._h = None
        except Exception:
            pass


class StreamEvent:
    """An event recorded on an explicit stream (``None``: the package's current stream).

    By default an ordering event (no timing, no system-scope fence: include/bbt_hip.h,
    `bbt_event_create_ordering`), meant for stream-to-stream waits.  ``host_wait=True`` makes a
    default event, the kind the HOST may wait for (`synchronize`) before it touches memory the
    work in front of the event used.  (The hand-over of kernel results to a download queued on
    another stream stays an ordering event: the library holds gfx950 code only, where the
    agent-scope release at the end of a kernel writes its L2 lines back before the copy engine
    reads them; the host waits for that download with `Stream.synchronize`, which fences.)"""

    def __init__(self, host_wait=False):
        self._h = C.c_void_p()
        create = hip.lib().bbt_event_create if host_wait else hip.lib().bbt_event_create_ordering
        hip.check(create(C.byref(self._h)))

    def record(self, stream=None):
        handle = hip.get_stream() if stream is None else stream.handle
        hip.check(hip.lib().bbt_event_record(self._h, handle))
        return self

    def synchronize(self):
        hip.check(hip.lib().bbt_event_sync(self._h))

    def __del__(self):
        try:
            if self._h:
                hip.lib().bbt_event_destroy(self._h)
                self._h = None
        except Exception:
            pass


def current_stream_wait(event):
    hip.check(hip.lib().bbt_stream_wait_event(hip.get_stream(), event._h))


# The copy engine of a HIP stream is chosen at its FIRST copy and kept (measured on MI355X, ROCm
# 7.2, round 4: `H2D on s1; sync; D2H on s2; sync` and both streams share one SDMA engine from then
# on -- 28.6 GB/s each way whatever is done later; first copies issued together: 48.5 GB/s each
# way).  A reader's first run uploads, computes and downloads one after the other, which is the
# bad order.  So the process has ONE upload and ONE download stream, created together and primed
# with a concurrent pair of small copies; every `HostUploader` and every task's download use them
# (one bus, one queue per direction: nothing is lost by sharing).
_copy_pair = None
_copy_pair_lock = threading.Lock()


def copy_streams():
    """(upload stream, download stream) of this process, on different copy engines."""
    global _copy_pair
    if _copy_pair is None:
        with _copy_pair_lock:
            if _copy_pair is None:
                up, down = Stream(), Stream()
                n = 8 << 20
                a, b = pinned_empty((n,), np.uint8), pinned_empty((n,), np.uint8)
                da, db = hip.DeviceArray((n,), np.uint8), hip.DeviceArray((n,), np.uint8)
                lib = hip.lib()
                for _ in range(2):                       # (both queued before either can finish)
                    hip.check(lib.bbt_memcpy_h2d(da.ptr, a.ctypes.data, n, up.handle))
                    hip.check(lib.bbt_memcpy_d2h(b.ctypes.data, db.ptr, n, down.handle))
                up.synchronize()
                down.synchronize()
                _copy_pair = (up, down)
    return _copy_pair


# --------------------------------------------------------------------------- uploads
class _Upload:
    __slots__ = ('start', 'count', 'dev', 'event', 'staging')


class HostUploader:
    """Uploads of sample ranges of one host stream, one range ahead.

    ``fetch(start, count)`` returns the samples as a `hip.DeviceArray`, ordered
    before later work on the package's current stream; ``prefetch(start,
    count)`` announces the range of the fetch after the next one, which is then
    read and uploaded in the background while the caller computes.  The host
    stream is only ever touched by one thread at a time: ``fetch`` waits for a
    running load before anything else.  Every range is loaded once.
    """

    def __init__(self, ih):
        self._ih = weakref.ref(ih)
        self._row = tuple(ih.shape[1:])
        self._dtype = np.dtype(ih.dtype)
        self._stream = copy_streams()[0]
        self._worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix='bbt-upload')
        self._device = hip.get_device() if hasattr(hip, 'get_device') else 0
        self._pending = None            # (start, count, future): the one load in flight
        self._announced = None          # (start, count) to load once the next fetch is served
        self.loads = 0                  # `_load` calls so far (tests count them)
        self._staging = [None, None]    # page-locked staging arrays (streams that are not pinned themselves)
        self._staging_event = [None, None]
        self._turn = 0

    # -- the one place the host stream is read
    def _load(self, start, count, dev, after):
        ih = self._ih()
        up = _Upload()
        up.start, up.count, up.staging = start, count, None
        hip.set_device(self._device)
        self.loads += 1
        shape = (count,) + self._row
        view = None
        if hasattr(ih, 'host_view'):
            view = ih.host_view(start, count)            # zero copy, if the stream's memory is pinned
            if view is not None and not is_pinned(view):
                view = None
        if view is None:
            slot = self._turn
            self._turn ^= 1
            if self._staging_event[slot] is not None:
                self._staging_event[slot].synchronize()   # its previous upload has left the buffer
            buf = self._staging[slot]
            n = int(np.prod(shape, dtype=np.int64))
            if buf is None or buf.size < n:
                buf = self._staging[slot] = pinned_empty((n,), self._dtype)
            view = buf[:n].reshape(shape)
            ih.seek(start)
            got = ih.read(count, out=view)
            if got is not view:
                view[...] = got
            up.staging = slot
        up.dev = dev
        self._stream.wait(after)                          # the block's previous users (pool reuse)
        hip.check(hip.lib().bbt_memcpy_h2d(dev.ptr, view.ctypes.data, view.nbytes, self._stream.handle))
        # (the host waits for this one before it refills the staging buffer: a default event)
        up.event = StreamEvent(host_wait=up.staging is not None).record(self._stream)
        if up.staging is not None:
            self._staging_event[up.staging] = up.event
        return up

    def _submit(self, start, count):
        # Device blocks come from the pool, whose reuse is ordered by the current stream (the pool
        # stream): a block freed a moment ago may still have readers queued there.  So the block
        # is TAKEN HERE, on the calling thread, and the event the upload stream waits for is
        # recorded after that: every reader of the block's previous life was queued before its
        # free, hence before this event.  (Taking it later, in the worker, would let a block
        # through that the main thread freed after the event -- the input of the run whose
        # kernels it has just queued.)
        dev = hip.DeviceArray((int(count),) + self._row, self._dtype)
        after = StreamEvent().record()
        return self._worker.submit(self._load, int(start), int(count), dev, after)

    def prefetch(self, start, count):
        """Announce the range the fetch AFTER the next one will ask for: its load is started as
        soon as the next `fetch` has its own samples -- before the caller queues that run's
        kernels, so the upload waits for the kernels of the run before only and overlaps this
        run's.  (Started right away it would be in the way: the host stream is read by one
        thread at a time, and the next fetch wants its own range first.)"""
        ih = self._ih()
        if ih is None or count <= 0 or start < 0 or start + count > ih.shape[0]:
            return
        self._announced = (int(start), int(count))

    def fetch(self, start, count):
        key = (int(start), int(count))
        up = None
        skip = 0
        if self._pending is not None:
            pending, self._pending = self._pending, None
            got = pending[2].result()                     # (also: the host stream is free again)
            if pending[0] <= key[0] and key[0] + key[1] <= pending[0] + pending[1]:
                # the range that was announced, or part of it: a task that keeps the frame its last
                # run ended in (DeviceTaskMixin._ensure_frames) asks for one frame less than its
                # reader foresaw
                up = got
                skip = key[0] - pending[0]
            else:
                # a read-ahead nobody came for: its block goes back to the pool, whose reuse is
                # ordered by the current stream -- which must therefore come after the upload
                current_stream_wait(got.event)
                del got
        if up is None:
            up = self._submit(*key).result()
        announced, self._announced = self._announced, None
        if announced is not None and announced != key:
            self._pending = announced + (self._submit(*announced),)
        current_stream_wait(up.event)
        return up.dev if (skip == 0 and up.dev.shape[0] == key[1]) else up.dev[skip:skip + key[1]]

    def close(self):
        pending, self._pending = self._pending, None
        if pending is not None:
            try:
                current_stream_wait(pending[2].result().event)     # (see fetch: the block is freed next)
            except Exception:
                pass
        self._announced = None
        self._worker.shutdown(wait=True)


_uploaders = weakref.WeakKeyDictionary()


def uploader_for(ih):
    """The `HostUploader` of host stream ``ih`` (created on first use), or None when the
    pipeline is off or the stream cannot be weakly referenced."""
    if not ENABLED:
        return None
    try:
        up = _uploaders.get(ih)
        if up is None:
            up = _uploaders[ih] = HostUploader(ih)
        return up
    except TypeError:
        return None
