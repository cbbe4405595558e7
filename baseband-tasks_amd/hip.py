"""ctypes binding of libbbt_hip.so (C ABI: include/bbt_hip.h) and thin device
helpers.  There is deliberately NO fallback: if the library is missing or a
call fails, an exception is raised.
"""
import ctypes as C
import importlib.util
import os
import sys
import threading
import weakref

import numpy as np

__all__ = ['HipError', 'HipLibraryMissing', 'lib', 'available', 'DeviceArray',
           'OsmPlan', 'ChanPlan', 'PfbPlan', 'set_stream', 'get_stream',
           'synchronize', 'Event', 'device_count', 'set_device']

_HERE = os.path.dirname(os.path.abspath(__file__))
# BBT_HIP_LIB points at another build of the same library (the sanitizer build
# of tools/build_sanitize.sh); never at a different implementation.
LIB_PATH = os.environ.get('BBT_HIP_LIB') or os.path.join(_HERE, 'lib', 'libbbt_hip.so')


class HipError(RuntimeError):
    """A call into libbbt_hip.so failed."""


class HipLibraryMissing(ImportError):
    """libbbt_hip.so has not been built (run ``python __graft_entry__.py``)."""


_vp, _i64, _i32, _int, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_size_t
_pvp = C.POINTER(C.c_void_p)
_pi64, _pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)

# name -> argtypes ; every entry point declared in include/bbt_hip.h
SIGNATURES = {
    'bbt_version': [],
    'bbt_rtc_info': [C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double)],
    'bbt_tune_scratch': [C.c_int, C.POINTER(C.c_int64)],
    'bbt_device_count': [C.POINTER(_int)],
    'bbt_set_device': [_int],
    'bbt_get_device': [C.POINTER(_int)],
    'bbt_device_name': [C.c_char_p, _int],
    'bbt_malloc': [_pvp, _sz],
    'bbt_free': [_vp],
    'bbt_pool_set_stream': [_vp],
    'bbt_pool_trim': [],
    'bbt_pool_info': [C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    'bbt_host_alloc': [_pvp, _sz],
    'bbt_host_free': [_vp],
    'bbt_memset': [_vp, _int, _sz, _vp],
    'bbt_memcpy_h2d': [_vp, _vp, _sz, _vp],
    'bbt_memcpy_d2h': [_vp, _vp, _sz, _vp],
    'bbt_memcpy_d2d': [_vp, _vp, _sz, _vp],
    'bbt_memcpy2d': [_vp, _sz, _vp, _sz, _sz, _sz, _int, _vp],
    'bbt_pad_streams': [_vp, _vp, _i64, _int, _int, _int, _vp],
    'bbt_detect_power_axis': [_vp, _vp, _i64, _i64, _int, _int, _int, _vp],
    'bbt_stream_create': [_pvp],
    'bbt_stream_destroy': [_vp],
    'bbt_stream_sync': [_vp],
    'bbt_device_sync': [],
    'bbt_event_create': [_pvp],
    'bbt_event_create_ordering': [_pvp],
    'bbt_event_destroy': [_vp],
    'bbt_event_record': [_vp, _vp],
    'bbt_event_sync': [_vp],
    'bbt_event_query': [_vp, C.POINTER(C.c_int)],
    'bbt_stream_wait_event': [_vp, _vp],
    'bbt_host_register': [_vp, _sz],
    'bbt_host_unregister': [_vp],
    'bbt_event_elapsed_ms': [_vp, _vp, C.POINTER(C.c_float)],
    'bbt_osm_plan_create': [_pvp, _i64, _int, _int, _vp, _int, _pi32],
    'bbt_osm_plan_destroy': [_vp],
    'bbt_osm_plan_info': [_vp, _pi64, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)],
    'bbt_osm_plan_fusable': [_vp, _int],
    'bbt_osm_plan_defer': [_vp, _vp],
    'bbt_osm_execute': [_vp, _vp, _vp, _i64, _pi64, _pi64, _pi32, _pi32, _vp],
    'bbt_osm_execute_flat': [_vp, _vp, _vp, _i64, _pi64, _pi64, _pi32, _i32, _pi32, _vp],
    'bbt_osm_plan_set_layout': [_vp, _i64, _i64],
    'bbt_osm_execute_channelized': [_vp, _vp, _vp, _i64, _pi64, _pi64, _pi32, _pi32, _int, _i64,
                                    _i64, _vp],
    'bbt_osm_execute_channelized_detect': [_vp, _vp, _vp, _i64, _pi64, _pi64, _pi32, _pi32, _int,
                                           _i64, _i64, _int, _int, _int, _vp],
    'bbt_osm_detect_bins_max': [_vp, _int, _int],
    'bbt_osm_execute_regular': [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    'bbt_osm_timing_enable': [_vp, _int],
    'bbt_osm_timing_read': [_vp, C.POINTER(C.c_double), _pi64],
    'bbt_osm_timing_read_passes': [_vp, C.POINTER(C.c_double), _pi64, _pi64],
    'bbt_chan_plan_create': [_pvp, _int, _int, _int],
    'bbt_chan_plan_destroy': [_vp],
    'bbt_chan_execute': [_vp, _vp, _vp, _i64, _vp],
    'bbt_pfb_plan_create': [_pvp, _int, _int, _int, C.POINTER(C.c_float)],
    'bbt_pfb_plan_destroy': [_vp],
    'bbt_pfb_execute': [_vp, _vp, _vp, _i64, _vp],
    'bbt_detect_integrate': [_vp, _vp, _i64, _i64, _i64, _int, _int, _vp],
    'bbt_shift_plan_create': [_pvp, _int, _int, _pi32],
    'bbt_shift_plan_destroy': [_vp],
    'bbt_shift_execute': [_vp, _vp, _vp, _i64, _vp],
    'bbt_fir_plan_create': [_pvp, _int, _int, _vp],
    'bbt_fir_plan_destroy': [_vp],
    'bbt_fir_execute': [_vp, _vp, _vp, _i64, _vp],
    'bbt_real_op': [_vp, _vp, _int, _i64, _int, _int, _vp],
    'bbt_chirp': [_vp, _i64, _int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                  C.c_double, C.c_double, C.c_double, _vp],
    'bbt_scale_streams': [_vp, _vp, _i64, _int, _vp, _vp],
    'bbt_unpack': [_vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _vp],
    'bbt_unpack_masked': [_vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp],
    'bbt_comm_unique_id': [_vp, _sz],
    'bbt_comm_init': [_pvp, _int, _int, _vp, _sz],
    'bbt_comm_destroy': [_vp],
    'bbt_bcast_chirp': [_vp, _vp, _i64, _int, _vp],
    'bbt_gather_output': [_vp, _vp, _vp, _i64, _vp],
}

#: oldest libbbt_hip.so whose entry points and argument meanings this binding assumes
MIN_LIB_VERSION = 152

_lib = None
_lock = threading.Lock()


_torch_lib_dir = None      # set when PyTorch's bundled ROCm runtime was preloaded


def _preload_torch_runtime(name='libamdhip64.so'):
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64 / libhsa-runtime64 / librccl; libbbt_hip.so is linked against
    the system's.  If this library is loaded first it pulls in the system
    runtime, and a later ``import torch`` brings a second one that finds no GPU
    ("No HIP GPUs are available", measured on the MI355X box).  So when torch is
    installed but not imported yet, its copy is loaded first, globally: this
    library and torch then share it whatever the import order (the same sonames;
    it is what `bench.py`, which imports torch first, runs on anyway).
    BBT_HIP_RUNTIME=system skips this."""
    global _torch_lib_dir
    if os.environ.get('BBT_HIP_RUNTIME', '') == 'system':
        return
    if _torch_lib_dir is None:
        if 'torch' in sys.modules and name == 'libamdhip64.so':
            return                       # already in the process
        try:
            spec = importlib.util.find_spec('torch')
        except (ImportError, ValueError):
            spec = None
        where = list(getattr(spec, 'submodule_search_locations', None) or [])
        if not where or not os.path.exists(os.path.join(where[0], 'lib', 'libamdhip64.so')):
            return
        _torch_lib_dir = os.path.join(where[0], 'lib')
    path = os.path.join(_torch_lib_dir, name)
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


# ROCm maps the HIP streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a
# process of this package can have more streams than that (a plan's lanes and tail, the stream of
# every one-kernel plan of a chain, upload, download, the caller's); streams that share a queue
# take turns.  Rounds 4-5 set the variable to 16 here when it was unset, for the host path (an
# upload and a download that share a queue: 2.38 instead of 2.57 Gsamples/s host to host).  Measured
# at the end of round 5 (profiles/r05_hw_queues.txt, alternating runs in one box): with 8 or 16
# queues reads of device-resident chains are SLOWER -- `Dedisperse` on 160 blocks 38 against 47 G,
# `Power(Channelize(Dedisperse))` 35 against 44 G, `Resample` 133 against 154 G -- and the headline
# (768 blocks per call) does not care (51.3-51.9 G either way).  The resident path is the product,
# so the package leaves ROCm's default alone (the host path of the final round-5 run read 2.53
# Gsamples/s with 4 queues: profiles/r05_bench_final.json).


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _preload_torch_runtime()
                _preload_torch_runtime('libhiprtc.so')     # (csrc/rtc.hpp opens it by soname)
                if not os.path.exists(LIB_PATH):
                    raise HipLibraryMissing(
                        f"{LIB_PATH} not found: build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
                handle = C.CDLL(LIB_PATH)
                handle.bbt_last_error.restype = C.c_char_p
                handle.bbt_last_error.argtypes = []
                # (the version first: an older library lacks entry points the table below names)
                if handle.bbt_version() < MIN_LIB_VERSION:
                    raise HipLibraryMissing(
                        f"{LIB_PATH} is version {handle.bbt_version()}, this package needs "
                        f">= {MIN_LIB_VERSION}: rebuild it (python -c 'import __graft_entry__ as g; g.build()').")
                for name, argtypes in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.argtypes = argtypes
                    fn.restype = _int
                _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise HipError(lib().bbt_last_error().decode(errors='replace'))


def device_count():
    n = _int(0)
    check(lib().bbt_device_count(C.byref(n)))
    return n.value


def available():
    """True if the library loads and sees at least one GPU."""
    try:
        return device_count() > 0
    except (HipError, HipLibraryMissing, OSError):
        return False


def set_device(index):
    check(lib().bbt_set_device(int(index)))


def get_device():
    index = _int(0)
    check(lib().bbt_get_device(C.byref(index)))
    return index.value


def rtc_info():
    """Run-time specialisation of the kernels for lengths that are not powers of two
    (include/bbt_hip.h: bbt_rtc_info): dict(mode, modules, seconds)."""
    mode, modules, seconds = _int(0), _i64(0), C.c_double(0)
    check(lib().bbt_rtc_info(C.byref(mode), C.byref(modules), C.byref(seconds)))
    return dict(mode=('off', 'on', 'required')[mode.value], modules=modules.value, seconds=seconds.value)


def tuning_scratch(release=False):
    """Bytes of scratch device memory the library keeps on the current device for plans that time
    their candidate kernels when they are made (include/bbt_hip.h: bbt_tune_scratch);
    ``release=True`` frees them."""
    held = _i64(0)
    check(lib().bbt_tune_scratch(int(bool(release)), C.byref(held)))
    return held.value


def device_name():
    buf = C.create_string_buffer(256)
    check(lib().bbt_device_name(buf, 256))
    return buf.value.decode()


# The stream all work of this package is queued on (a hipStream_t as int;
# None / 0 = the default stream).  bench.py points it at torch's stream.
_stream = None
_NO_STREAM = object()


def set_stream(stream):
    """Queue all further work of this package on ``stream`` (a hipStream_t as
    int; None / 0 = the default stream).  Also becomes the stream the device
    memory pool orders the reuse of freed blocks by (include/bbt_hip.h)."""
    global _stream
    _stream = int(stream) if stream else None
    check(lib().bbt_pool_set_stream(_stream))


def get_stream():
    return _stream


def synchronize():
    check(lib().bbt_stream_sync(_stream))


class Event:
    def __init__(self):
        self._h = C.c_void_p()
        check(lib().bbt_event_create(C.byref(self._h)))

    def record(self):
        check(lib().bbt_event_record(self._h, _stream))
        return self

    def synchronize(self):
        check(lib().bbt_event_sync(self._h))

    def elapsed_ms(self, later):
        ms = C.c_float()
        check(lib().bbt_event_elapsed_ms(self._h, later._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self._h:
                lib().bbt_event_destroy(self._h)
        except Exception:
            pass


def pool_trim():
    """Return every idle block of the device memory pool to the driver."""
    check(lib().bbt_pool_trim())


def pool_info():
    """(idle bytes kept for reuse, bytes in use) of the device memory pool."""
    cached, live = C.c_int64(), C.c_int64()
    check(lib().bbt_pool_info(C.byref(cached), C.byref(live)))
    return cached.value, live.value


#: Plan calls whose output this package owns are issued with a deferred join
#: (include/bbt_hip.h: bbt_osm_plan_defer): the stream is not ordered after the plan's lanes, so
#: consecutive calls -- a reader taking run after run of frames -- flow into each other without
#: a drain at the call boundary; the completion event travels with the output's allocation and
#: is waited for by whatever touches that memory next (`DeviceArray.ptr`).  ``BBT_DEFER=0``:
#: every call joins its stream (round 3's behaviour).
DEFER_JOIN = os.environ.get('BBT_DEFER', '1') != '0'


class _EventPool:
    """Completion events of deferred plan calls: ordering events (no timing, no system-scope fence
    -- streams wait for them, the host never does).  An event goes back to the pool as soon as the
    wait for it has been queued: a stream's wait refers to the record that preceded it."""

    def __init__(self):
        self._idle = []
        self._lock = threading.Lock()

    def take(self):
        with self._lock:
            if self._idle:
                return self._idle.pop()
        h = C.c_void_p()
        check(lib().bbt_event_create_ordering(C.byref(h)))
        return h

    def give(self, h):
        with self._lock:
            self._idle.append(h)


_events = _EventPool()


class _Done:
    """The completion event of one deferred plan call, shared by the allocations the call reads
    and writes; back to the pool once each of them has queued its wait."""
    __slots__ = ('event', 'refs')

    def __init__(self, event, refs):
        self.event = event
        self.refs = refs

    def release(self):
        self.refs -= 1
        if self.refs == 0:
            _events.give(self.event)


class _Pending:
    """What a deferred plan call still owes an allocation -- it is writing it, or reading it --:
    the event that marks the call's end and the objects (the input, the plan) that must outlive
    the call."""
    __slots__ = ('done', 'keep')

    def __init__(self, done, keep):
        self.done = done
        self.keep = keep


def _prune(owed):
    """Let go of the leading entries of a list of deferred READERS whose calls have finished
    (`bbt_event_query`: never blocks): each keeps its plan -- and through the plan its streams and
    work buffers -- alive, and a task that was dropped long ago would otherwise be destroyed (a
    device-wide wait: hipFree) in the middle of whatever read happens to push its entry out, 64
    calls later.  Found with tools/chain_host_probe.py: `Dedisperse(Resample(x))` after other chains
    on the same tensor, two calls of every read blocked 2.5-3 ms in `bbt_osm_plan_destroy` (18.6
    instead of 29 G).  Readers only: a finished reader leaves nothing to make visible.  Returns
    what is left (a tuple for a tuple, the list itself, shortened, for a list)."""
    n, done = 0, C.c_int(0)
    while n < len(owed):
        if lib().bbt_event_query(owed[n].done.event, C.byref(done)) != 0 or not done.value:
            break
        n += 1
    if not n:
        return owed
    gone = owed[:n]
    if isinstance(owed, list):
        del owed[:n]
    else:
        owed = owed[n:]
    for p in gone:
        p.done.release()
    return owed


class _Allocation:
    """Owns one block of the device memory pool, and the calls it is still owed: deferred plan
    calls that are WRITING somewhere in it (`writes`) or READING it (`reads`).  A later reader
    must come after the writes, a later writer -- or the return of the block to the pool -- after
    both; calls that only read may share it, and so may calls that write regions the caller knows
    to be disjoint (`DeviceArray.fresh`: the runs of one big read)."""
    writes = ()
    reads = ()

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().bbt_malloc(C.byref(self.ptr), max(self.nbytes, 1)))

    @property
    def pending(self):
        return bool(self.writes or self.reads)

    _MAX_OWED = 64

    def owe(self, pending, write):
        """Note a deferred call that writes (reads) this block.  The lists stay short: beyond
        _MAX_OWED entries the oldest is waited for on the current stream (it finished long ago:
        the wait costs nothing) and let go."""
        owed = (self.writes if write else _prune(self.reads)) + (pending,)
        if len(owed) > self._MAX_OWED:
            old, owed = owed[0], owed[1:]
            check(lib().bbt_stream_wait_event(_stream, old.done.event))
            old.done.release()
        if write:
            self.writes = owed
        else:
            self.reads = owed

    def settle(self, stream=_NO_STREAM, reads=True):
        """Order ``stream`` (default: the package's current stream) after the deferred calls that
        still write this block and -- unless ``reads`` is False: the caller only wants to read --
        after those that still read it."""
        owed = self.writes + (self.reads if reads else ())
        if not owed:
            return
        self.writes = ()
        if reads:
            self.reads = ()
        target = _stream if stream is _NO_STREAM else stream
        for p in owed:
            check(lib().bbt_stream_wait_event(target, p.done.event))
            p.done.release()

    def __del__(self):
        try:
            if self.ptr:
                # (reuse of a freed block is ordered by the pool stream: it must come after the
                # deferred writer too; the writer's inputs are released after this block)
                self.settle()
                lib().bbt_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


# Deferred plan calls that still READ memory this package does not own (a torch tensor handed in
# as a stream's samples): id(owner) -> (weak reference to the owner, [_Pending, ...]).  The owner
# has no `_Allocation` to carry the events, so whoever is about to OVERWRITE such memory in place
# asks `wait_for_readers` first (include/bbt_hip.h: a deferred call's input stays untouched until
# its completion event).
_foreign_reads = {}
_FOREIGN_MAX = 64


def _owe_foreign_read(owner, pending):
    key = id(owner)
    entry = _foreign_reads.get(key)
    if entry is None or entry[0]() is not owner:
        try:
            ref = weakref.ref(owner, lambda _r, k=key: _drop_foreign(k))
        except TypeError:
            return False
        entry = _foreign_reads[key] = (ref, [])
    _prune(entry[1])
    entry[1].append(pending)
    if len(entry[1]) > _FOREIGN_MAX:
        old = entry[1].pop(0)
        check(lib().bbt_stream_wait_event(_stream, old.done.event))
        old.done.release()
    return True


def _drop_foreign(key):
    entry = _foreign_reads.pop(key, None)
    if entry is not None:
        for p in entry[1]:
            try:
                p.done.release()
            except Exception:
                pass


def wait_for_readers(array, stream=_NO_STREAM):
    """Order ``stream`` (default: the package's current stream) after every deferred plan call
    that still reads ``array`` -- a `DeviceArray`, or the foreign object (a torch tensor) that was
    handed to `DeviceStream` / `as_device_array`.  Call it before refilling such an input in
    place on that stream; memory of this package needs no call (`DeviceArray.ptr` waits)."""
    owner = array.owner if isinstance(array, DeviceArray) else array
    if owner.__class__ is _Allocation:
        owner.settle(stream)
        return
    entry = _foreign_reads.get(id(owner))
    if entry is None or entry[0]() is not owner:
        return
    target = _stream if stream is _NO_STREAM else stream
    owed, entry[1][:] = list(entry[1]), []
    for p in owed:
        check(lib().bbt_stream_wait_event(target, p.done.event))
        p.done.release()


class DeviceArray:
    """C-contiguous array in HBM: (ptr, shape, dtype); leading-axis slices are
    zero-copy views.  ``owner`` keeps the allocation (or a foreign object such
    as a torch tensor) alive."""

    def __init__(self, shape, dtype=np.complex64, ptr=None, owner=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        if ptr is None:
            owner = _Allocation(self.nbytes)
            ptr = owner.ptr.value
        self._ptr = int(ptr) if ptr else 0
        # (a view of a view is kept alive by -- and shares the pending state of -- the root)
        self.owner = owner.owner if isinstance(owner, DeviceArray) else owner

    @property
    def ptr(self):
        """Device address.  Asking for it is the announcement of a use: if a deferred plan call
        still owes this memory (`_Allocation.pending`), the current stream is first ordered after
        that call -- every binding, copy and interop path goes through here."""
        o = self.owner
        if o.__class__ is _Allocation and (o.writes or o.reads):
            o.settle()
        return self._ptr

    def ptr_to_read(self):
        """The address for a use that only READS the array: ordered after the deferred calls that
        still write its allocation, not after those that merely read it too."""
        o = self.owner
        if o.__class__ is _Allocation and o.writes:
            o.settle(reads=False)
        return self._ptr

    #: Set by a caller that hands this view to a plan call as output and KNOWS that no call still
    #: owed to the allocation touches the view's region (the consecutive runs of one big read fill
    #: disjoint slices of a fresh array): the call is then not ordered after those calls.
    fresh = False

    @property
    def pending(self):
        """Is a deferred plan call still writing or reading this array's allocation (nobody has
        waited for it yet)?"""
        o = self.owner
        return o.__class__ is _Allocation and bool(o.writes or o.reads)

    @property
    def size(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def row_bytes(self):
        n = self.dtype.itemsize
        for d in self.shape[1:]:
            n *= d
        return n

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        if not isinstance(item, slice):
            raise TypeError("DeviceArray supports only leading-axis slices")
        start, stop, step = item.indices(self.shape[0])
        if step != 1:
            raise ValueError("DeviceArray slices must be contiguous")
        stop = max(stop, start)
        view = DeviceArray((stop - start,) + self.shape[1:], self.dtype,
                           self._ptr + start * self.row_bytes, self.owner)
        if self.fresh:
            view.fresh = True              # (part of a region nobody is owed is such a region)
        return view

    def reshape(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            known = 1
            for d in shape:
                if d != -1:
                    known *= d
            shape[shape.index(-1)] = self.size // known if known else 0
        out = DeviceArray(shape, self.dtype, self._ptr, self.owner)
        if out.size != self.size:
            raise ValueError(f"cannot reshape {self.shape} into {tuple(shape)}")
        if self.fresh:
            out.fresh = True
        return out

    @classmethod
    def from_host(cls, array, dtype=None):
        array = np.ascontiguousarray(array, dtype=dtype)
        out = cls(array.shape, array.dtype)
        out.copy_from_host(array)
        return out

    def copy_from_host(self, array):
        array = np.ascontiguousarray(array, dtype=self.dtype)
        if array.nbytes != self.nbytes:
            raise ValueError("size mismatch in host to device copy")
        if array.nbytes:
            check(lib().bbt_memcpy_h2d(self.ptr, array.ctypes.data, array.nbytes, _stream))
            # pageable host memory: make sure the source may be released
            check(lib().bbt_stream_sync(_stream))
        return self

    def copy_from_device(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError("size mismatch in device to device copy")
        if self.nbytes:
            # (the source is only read: after the calls that still write it, beside those that read it too)
            check(lib().bbt_memcpy_d2d(self.ptr, other.ptr_to_read(), self.nbytes, _stream))
        return self

    def to_host(self, out=None):
        if out is None:
            out = np.empty(self.shape, self.dtype)
        elif (not isinstance(out, np.ndarray) or out.dtype != self.dtype
              or not out.flags.c_contiguous or out.shape != self.shape):
            tmp = self.to_host()
            out[...] = tmp
            return out
        if self.nbytes:
            check(lib().bbt_memcpy_d2h(out.ctypes.data, self.ptr_to_read(), self.nbytes, _stream))
            check(lib().bbt_stream_sync(_stream))
        return out

    def fill_bytes(self, value):
        if self.nbytes:
            check(lib().bbt_memset(self.ptr, int(value), self.nbytes, _stream))
        return self

    @property
    def __cuda_array_interface__(self):
        return dict(shape=self.shape, typestr=self.dtype.str, data=(self.ptr, False),
                    version=3, strides=None)

    def __repr__(self):
        return f"<DeviceArray shape={self.shape} dtype={self.dtype} ptr=0x{self._ptr:x}>"


def as_device_array(obj):
    """DeviceArray view of a DeviceArray or a torch tensor on the GPU."""
    if isinstance(obj, DeviceArray):
        return obj
    if hasattr(obj, 'data_ptr') and hasattr(obj, 'is_contiguous'):   # torch
        if not obj.is_contiguous():
            raise ValueError("tensor must be contiguous")
        import torch
        dt = {torch.complex64: np.complex64, torch.float32: np.float32,
              torch.complex128: np.complex128, torch.float64: np.float64}[obj.dtype]
        return DeviceArray(tuple(obj.shape), dt, obj.data_ptr(), obj)
    raise TypeError(f"cannot interpret {type(obj)} as a device array")


def copy_2d(dst, dst_pitch, src, src_pitch, src_offset, width, rows):
    """Device-to-device copy of ``rows`` runs of ``width`` bytes: run r goes
    from src + src_offset + r * src_pitch to dst + r * dst_pitch."""
    if rows and width:
        check(lib().bbt_memcpy2d(dst.ptr, int(dst_pitch), src.ptr_to_read() + int(src_offset), int(src_pitch),
                                 int(width), int(rows), 2, _stream))
    return dst


def pad_streams_to_even(dev, n_stream):
    """(n, S) complex64 with odd S -> (n, S+1) with a zero stream appended."""
    n = dev.size // n_stream
    out = DeviceArray((n, n_stream + 1), dev.dtype)
    check(lib().bbt_pad_streams(dev.ptr_to_read(), out.ptr, n, n_stream, n_stream + 1, dev.dtype.itemsize, _stream))
    return out


def strip_stream_pad(dev_padded, n_rows, n_stream, out):
    """inverse of pad_streams_to_even for rows of (S+1) -> S elements."""
    isz = dev_padded.dtype.itemsize
    if n_rows:
        check(lib().bbt_memcpy2d(out.ptr, n_stream * isz, dev_padded.ptr_to_read(), (n_stream + 1) * isz,
                                 n_stream * isz, n_rows, 2, _stream))
    return out


def detect_integrate(in_dev, out_dev, n_out, step, n_elem, mode, average=True):
    """Square (mode 0) / Power (1) / plain sum (2) over ``step`` samples."""
    check(lib().bbt_detect_integrate(in_dev.ptr_to_read(), out_dev.ptr, int(n_out), int(step), int(n_elem),
                                     int(mode), int(bool(average)), _stream))


def detect_power_axis(in_dev, out_dev, n_out, step, outer, inner, average=True):
    """Power (mode 1 of `detect_integrate`) for samples (outer, 2, inner)."""
    check(lib().bbt_detect_power_axis(in_dev.ptr_to_read(), out_dev.ptr, int(n_out), int(step), int(outer), int(inner),
                                      int(bool(average)), _stream))


def real_to_complex(x):
    """float32 DeviceArray -> complex64 with zero imaginary part (same shape)."""
    out = DeviceArray(x.shape, np.complex64)
    check(lib().bbt_real_op(x.ptr_to_read(), out.ptr, 0, out.size, 0, 0, _stream))
    return out


def real_part(z, out=None):
    """complex64 DeviceArray -> its real part (float32)."""
    if out is None:
        out = DeviceArray(z.shape, np.float32)
    check(lib().bbt_real_op(z.ptr_to_read(), out.ptr, 1, out.size, 0, 0, _stream))
    return out


def half_to_full_spectrum(z, n_chan, n_stream):
    """(n_spec, n_chan/2+1, n_stream) complex64 -> Hermitian (n_spec, n_chan, n_stream)."""
    n_spec = z.size // ((n_chan // 2 + 1) * n_stream)
    out = DeviceArray((n_spec, n_chan, n_stream), np.complex64)
    check(lib().bbt_real_op(z.ptr_to_read(), out.ptr, 2, out.size, int(n_chan), int(n_stream), _stream))
    return out


def keep_half_spectrum(z, n_chan, n_stream, out):
    """(n_spec, n_chan, n_stream) -> first n_chan/2+1 channels of every spectrum."""
    n_spec = z.size // (n_chan * n_stream)
    half = n_chan // 2 + 1
    if n_spec:
        check(lib().bbt_memcpy2d(out.ptr, half * n_stream * 8, z.ptr_to_read(), n_chan * n_stream * 8,
                                 half * n_stream * 8, n_spec, 2, _stream))
    return out


def split_real_pair_spectra(z, n_chan, n_stream, out, padded=False):
    """Spectra ``(n_spec, n_chan, n_stream/2)`` of complex streams z = a + i b
    -> half spectra ``(n_spec, n_chan/2+1, n_stream)`` of the real streams.
    ``padded``: z carries one more, unused, complex stream."""
    check(lib().bbt_real_op(z.ptr_to_read(), out.ptr, 6 if padded else 4, out.size, int(n_chan), int(n_stream),
                            _stream))
    return out


def merge_real_pair_spectra(z_half, n_chan, n_stream, out):
    """Half spectra ``(n_spec, n_chan/2+1, n_stream)`` of real streams ->
    ``(n_spec, n_chan, n_stream/2)`` spectra of the complex streams a + i b."""
    check(lib().bbt_real_op(z_half.ptr_to_read(), out.ptr, 5, out.size, int(n_chan), int(n_stream), _stream))
    return out


def square_real(x, out):
    check(lib().bbt_real_op(x.ptr_to_read(), out.ptr, 3, out.size, 0, 0, _stream))
    return out


def scale_streams(x, out, n_samples, n_elem, factor_dev):
    """out[i, e] = x[i, e] * factor[e] (complex64)."""
    check(lib().bbt_scale_streams(x.ptr_to_read(), out.ptr, int(n_samples), int(n_elem), factor_dev.ptr, _stream))


class _Plan:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if getattr(self, '_h', None):
            getattr(lib(), self._destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chirp(n, frequency_hz, sideband, reference_hz, rate_hz, d_dm, offset_s=0.):
    """(n_col, n) complex64 dispersion chirps made on the GPU in float64 (bbt_chirp): one
    column per entry of the equal-length 1-d arrays ``frequency_hz, sideband, reference_hz``."""
    f = np.ascontiguousarray(frequency_hz, dtype=np.float64).ravel()
    sb = np.ascontiguousarray(sideband, dtype=np.float64).ravel()
    fr = np.ascontiguousarray(reference_hz, dtype=np.float64).ravel()
    assert f.shape == sb.shape == fr.shape
    out = DeviceArray((f.shape[0], int(n)), np.complex64)
    pd = C.POINTER(C.c_double)
    check(lib().bbt_chirp(out.ptr, int(n), f.shape[0], f.ctypes.data_as(pd), sb.ctypes.data_as(pd),
                          fr.ctypes.data_as(pd), float(rate_hz), float(d_dm), float(offset_s), _stream))
    return out


class OsmPlan(_Plan):
    """ifft(fft(block) * response)[valid] for overlap-save blocks."""
    _destroy = 'bbt_osm_plan_destroy'

    def __init__(self, n_fft, n_stream, response, response_index=None):
        super().__init__()
        self.n_fft, self.n_stream = int(n_fft), int(n_stream)
        if isinstance(response, DeviceArray):
            resp_ptr, on_dev, n_resp = response.ptr, 1, response.shape[0]
            assert response.dtype == np.complex64 and response.shape[1] == n_fft
        else:
            response = np.ascontiguousarray(response, dtype=np.complex64)
            assert response.ndim == 2 and response.shape[1] == n_fft
            resp_ptr, on_dev, n_resp = response.ctypes.data, 0, response.shape[0]
        idx = None
        if response_index is not None:
            idx_arr = np.ascontiguousarray(response_index, dtype=np.int32)
            assert idx_arr.shape == (n_stream,)
            idx = idx_arr.ctypes.data_as(_pi32)
        check(lib().bbt_osm_plan_create(C.byref(self._h), self.n_fft, self.n_stream, n_resp,
                                        resp_ptr, on_dev, idx))

    def info(self):
        ws, chunk, n1, n2 = _i64(), _int(), _int(), _int()
        check(lib().bbt_osm_plan_info(self._h, C.byref(ws), C.byref(chunk), C.byref(n1),
                                      C.byref(n2)))
        return dict(workspace_bytes=ws.value, chunk_blocks=chunk.value, n1=n1.value, n2=n2.value)

    def set_layout(self, in_plane=0, out_plane=0):
        """Pair-planar hand-over between two plans (see bbt_osm_plan_set_layout); 0, 0: interleaved."""
        check(lib().bbt_osm_plan_set_layout(self._h, int(in_plane), int(out_plane)))

    def fusable(self, n_chan):
        """Can `execute_channelized` take Channelize(n_chan) into the row pass?"""
        return bool(lib().bbt_osm_plan_fusable(self._h, int(n_chan)))

    def _call(self, fn, in_dev, out_dev, *args):
        """One execute entry point of the C ABI on (in_dev, out_dev).  When this package owns the
        output's allocation the call is issued with a deferred join (plans with lanes leave the
        lanes running; one-kernel plans run on a stream of their own beside the caller's)
        (`DEFER_JOIN`): its completion event is left with the output's allocation."""
        owner = out_dev.owner
        defer = DEFER_JOIN and owner.__class__ is _Allocation
        # the input is read: after the calls that still write it; the output is written: after
        # every call still owed to its allocation -- unless the caller vouches for the region
        src = in_dev.ptr_to_read()
        dst = out_dev._ptr if (defer and out_dev.fresh) else out_dev.ptr
        if not defer:
            check(fn(self._h, src, dst, *args, _stream))
            return
        ev = _events.take()
        check(lib().bbt_osm_plan_defer(self._h, ev))
        try:
            check(fn(self._h, src, dst, *args, _stream))
        except Exception:
            lib().bbt_osm_plan_defer(self._h, None)         # (if the call never took it)
            check(lib().bbt_event_record(ev, _stream))
            _events.give(ev)
            raise
        # The call's end is owed to the output (it is being written) and to the input (it is being
        # read: whoever overwrites it next -- the upstream task filling its cache again -- or frees
        # it must come after the lanes).  The input of a foreign owner (a torch tensor) is the
        # caller's to keep untouched, as include/bbt_hip.h says.
        src_owner = in_dev.owner
        shared = src_owner.__class__ is _Allocation and src_owner is not owner
        foreign = src_owner is not None and src_owner.__class__ is not _Allocation
        done = _Done(ev, 2 if (shared or foreign) else 1)
        owner.owe(_Pending(done, (src_owner, self)), write=True)
        if shared:
            src_owner.owe(_Pending(done, (self,)), write=False)
        elif foreign and not _owe_foreign_read(src_owner, _Pending(done, (self,))):
            done.release()              # (an owner that cannot be weakly referenced: not tracked)
        out_dev.fresh = False           # (from now on the region IS owed a call)

    @staticmethod
    def _descriptors(in_off, out_off, valid_start, valid_count):
        in_off = np.ascontiguousarray(in_off, dtype=np.int64)
        out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        valid_start = np.ascontiguousarray(valid_start, dtype=np.int32)
        valid_count = np.ascontiguousarray(valid_count, dtype=np.int32)
        n = in_off.shape[0]
        assert out_off.shape == valid_start.shape == valid_count.shape == (n,)
        return (n, in_off.ctypes.data_as(_pi64), out_off.ctypes.data_as(_pi64),
                valid_start.ctypes.data_as(_pi32), valid_count.ctypes.data_as(_pi32)), \
            (in_off, out_off, valid_start, valid_count)

    def execute(self, in_dev, out_dev, in_off, out_off, valid_start, valid_count):
        desc, _alive = self._descriptors(in_off, out_off, valid_start, valid_count)
        self._call(lib().bbt_osm_execute, in_dev, out_dev, *desc)

    def execute_flat(self, in_dev, out_dev, in_off, out_elem_off, valid_start, first_elem, valid_elems):
        """`execute` with the kept range in elements of the (row, stream) matrix (plans of one
        kernel: power-of-two n_fft <= 4096): see bbt_osm_execute_flat."""
        (n, io, oo, vs, ve), _alive = self._descriptors(in_off, out_elem_off, valid_start, valid_elems)
        self._call(lib().bbt_osm_execute_flat, in_dev, out_dev, n, io, oo, vs, int(first_elem), ve)

    def execute_channelized(self, in_dev, out_dev, in_off, out_off, valid_start, valid_count,
                            n_chan, first_spectrum, n_spectra):
        """Fused Channelize: spectra [first_spectrum, +n_spectra) of the stream
        the blocks would produce (``out_off`` absolute in that stream)."""
        desc, _alive = self._descriptors(in_off, out_off, valid_start, valid_count)
        self._call(lib().bbt_osm_execute_channelized, in_dev, out_dev, *desc, int(n_chan),
                   int(first_spectrum), int(n_spectra))

    def detect_bins_max(self, n_chan, step):
        """Integration bins one workgroup of the last pass would touch (<= 64
        for the fused detection)."""
        return int(lib().bbt_osm_detect_bins_max(self._h, int(n_chan), int(step)))

    def execute_channelized_detect(self, in_dev, out_dev, in_off, out_off, valid_start, valid_count,
                                   n_chan, first_spectrum, n_bins, step, mode, average=True):
        """Fused Channelize + Square/Power + Integrate(step): float32 bins."""
        desc, _alive = self._descriptors(in_off, out_off, valid_start, valid_count)
        self._call(lib().bbt_osm_execute_channelized_detect, in_dev, out_dev, *desc, int(n_chan),
                   int(first_spectrum), int(n_bins), int(step), int(mode), int(bool(average)))

    def execute_regular(self, in_dev, out_dev, n_blocks, in_off0, out_off0, hop, valid_start):
        self._call(lib().bbt_osm_execute_regular, in_dev, out_dev, int(n_blocks), int(in_off0),
                   int(out_off0), int(hop), int(valid_start))

    def timing_enable(self, enable=True):
        """False/0 off, True/1 time the normal (two-lane) schedule, 2 isolated
        passes on a single lane."""
        check(lib().bbt_osm_timing_enable(self._h, int(enable)))

    def timing_read(self):
        ms = (C.c_double * 3)()
        n = _i64()
        check(lib().bbt_osm_timing_read(self._h, ms, C.byref(n)))
        return list(ms), n.value

    def timing_read_passes(self):
        """Per pass: accumulated ms, timed launches, overlap-save blocks those launches covered."""
        ms = (C.c_double * 3)()
        launches, blocks = (_i64 * 3)(), (_i64 * 3)()
        check(lib().bbt_osm_timing_read_passes(self._h, ms, launches, blocks))
        return list(ms), list(launches), list(blocks)


class ChanPlan(_Plan):
    """FFT over groups of n_chan complete samples."""
    _destroy = 'bbt_chan_plan_destroy'

    def __init__(self, n_chan, n_stream, direction=-1):
        super().__init__()
        self.n_chan, self.n_stream = int(n_chan), int(n_stream)
        check(lib().bbt_chan_plan_create(C.byref(self._h), self.n_chan, self.n_stream,
                                         int(direction)))

    def execute(self, in_dev, out_dev, n_spectra):
        check(lib().bbt_chan_execute(self._h, in_dev.ptr_to_read(), out_dev.ptr, int(n_spectra), _stream))


class PfbPlan(_Plan):
    """n_tap FIR across blocks followed by the channelizer FFT."""
    _destroy = 'bbt_pfb_plan_destroy'

    def __init__(self, taps, n_stream):
        super().__init__()
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.n_tap, self.n_chan = taps.shape
        self.n_stream = int(n_stream)
        check(lib().bbt_pfb_plan_create(C.byref(self._h), self.n_tap, self.n_chan, self.n_stream,
                                        taps.ctypes.data_as(C.POINTER(C.c_float))))

    def execute(self, in_dev, out_dev, n_spectra):
        check(lib().bbt_pfb_execute(self._h, in_dev.ptr_to_read(), out_dev.ptr, int(n_spectra), _stream))


class FirPlan(_Plan):
    """Time-domain convolution with a short response ``(n_tap, n_stream)``:
    out[i] = sum_k response[k] * in[i + n_tap - 1 - k] (what Convolve keeps)."""
    _destroy = 'bbt_fir_plan_destroy'

    def __init__(self, response):
        super().__init__()
        response = np.ascontiguousarray(response, dtype=np.complex64)
        assert response.ndim == 2
        self.n_tap, self.n_stream = response.shape
        check(lib().bbt_fir_plan_create(C.byref(self._h), self.n_tap, self.n_stream,
                                        response.ctypes.data_as(C.c_void_p)))

    def execute(self, in_dev, out_dev, n_out):
        check(lib().bbt_fir_execute(self._h, in_dev.ptr_to_read(), out_dev.ptr, int(n_out), _stream))


class ShiftPlan(_Plan):
    """out[i, e] = in[i + offsets[e], e] (per-element integer sample shifts)."""
    _destroy = 'bbt_shift_plan_destroy'

    def __init__(self, offsets, elem_bytes):
        super().__init__()
        offsets = np.ascontiguousarray(offsets, dtype=np.int32).ravel()
        check(lib().bbt_shift_plan_create(C.byref(self._h), offsets.shape[0], int(elem_bytes),
                                          offsets.ctypes.data_as(_pi32)))

    def execute(self, in_dev, out_dev, n_out):
        check(lib().bbt_shift_execute(self._h, in_dev.ptr_to_read(), out_dev.ptr, int(n_out), _stream))


COMM_ID_BYTES = 128


def comm_unique_id():
    """The id rank 0 creates and hands to the other ranks (bytes)."""
    lib()
    if _torch_lib_dir is not None:
        _preload_torch_runtime('librccl.so')         # the RCCL that goes with the preloaded runtime
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().bbt_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


class Comm(_Plan):
    """RCCL communicator behind the C ABI (one process per GPU; call after
    `set_device`).  ``uid``: the bytes of `comm_unique_id()` made on rank 0."""
    _destroy = 'bbt_comm_destroy'

    def __init__(self, n_ranks, rank, uid):
        super().__init__()
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        lib()
        if _torch_lib_dir is not None:
            _preload_torch_runtime('librccl.so')
        uid = bytes(uid)
        check(lib().bbt_comm_init(C.byref(self._h), self.n_ranks, self.rank, uid, len(uid)))

    def bcast_chirp(self, resp_dev, root=0):
        """In-place broadcast of a complex64 DeviceArray from ``root``."""
        assert resp_dev.dtype == np.complex64
        check(lib().bbt_bcast_chirp(self._h, resp_dev.ptr, resp_dev.size, int(root), _stream))
        return resp_dev

    def gather_output(self, local, out=None):
        """All-gather equally sized DeviceArrays along axis 0, in rank order."""
        if out is None:
            out = DeviceArray((self.n_ranks * local.shape[0],) + local.shape[1:], local.dtype)
        check(lib().bbt_gather_output(self._h, local.ptr, out.ptr, local.nbytes, _stream))
        return out
