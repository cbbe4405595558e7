"""Where the HOST is while a chain of tasks is read (dev tool): every plan call of one
`read_device` of Dedisperse(Resample(x)) with default arguments, with the host's clock before and
after it -- a call that takes long on the host is one that blocked.
    python tools/chain_host_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip
dev = torch.device('cuda', 0)
hip.set_stream(torch.cuda.current_stream().cuda_stream)
n = 2**27
x = torch.view_as_complex(torch.randn((n, 2, 2), device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1,
                     polarization=np.array(['X', 'Y']))
mk = lambda: bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1,
                             polarization=np.array(['X', 'Y']))


def read_all(task, reps=3):
    for _ in range(reps):
        u = task
        while u is not None and hasattr(u, 'invalidate_cache'):
            u.invalidate_cache(); u = getattr(u, 'ih', None)
        task.seek(0)
        y = task.read_device(task.shape[0]); _ = y.ptr
        torch.cuda.synchronize()


if '--after-others' in sys.argv:          # the chains tools/default_device_reads.py reads before this one
    for make in (lambda: bt.Channelize(mk(), 1024), lambda: bt.Channelize(bt.Dedisperse(mk(), 100.), 1024),
                 lambda: bt.Power(bt.Channelize(bt.Dedisperse(mk(), 100.), 1024)),
                 lambda: bt.PolyphaseFilterBank(mk(), bt.sinc_hamming(12, 1024)), lambda: bt.Resample(mk(), 0.25),
                 lambda: bt.Dedisperse(mk(), 100.)):
        read_all(make())
    print('pool after the other chains:', hip.pool_info())
t = bt.Dedisperse(bt.Resample(ds, 0.25), 100.)
log = []
slow = []
alloc_init = hip._Allocation.__init__


def timed_alloc(self, nbytes):
    a = time.perf_counter()
    alloc_init(self, nbytes)
    b = time.perf_counter()
    if b - a > 100e-6:
        slow.append((nbytes, a, b))


hip._Allocation.__init__ = timed_alloc
slow_calls = []


class TimedLib:
    """The library with every entry point timed on the host: calls over 200 us are noted."""
    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        f = getattr(self._real, name)

        def call(*args):
            a = time.perf_counter()
            r = f(*args)
            b = time.perf_counter()
            if b - a > 200e-6:
                slow_calls.append((name, a, b))
            return r
        return call


hip._lib = TimedLib(hip.lib())
orig = hip.OsmPlan._call


def logged(self, fn, in_dev, out_dev, *args):
    a = time.perf_counter()
    r = orig(self, fn, in_dev, out_dev, *args)
    log.append((self.n_fft, a, time.perf_counter()))
    return r


hip.OsmPlan._call = logged


def step():
    u = t
    while u is not None and hasattr(u, 'invalidate_cache'):
        u.invalidate_cache(); u = getattr(u, 'ih', None)
    t.seek(0)
    return t.read_device(t.shape[0])


for rep in range(8):
    del log[:]
    del slow[:]
    del slow_calls[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = step()
    t1 = time.perf_counter()
    _ = y.ptr
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'read {rep}: enqueue {1e6 * (t1 - t0):.0f} us, whole {1e6 * (t2 - t0):.0f} us; allocations over 100 us: '
          + ', '.join(f'{n >> 20} MiB at {1e6 * (a - t0):.0f} us for {1e6 * (b - a):.0f} us' for n, a, b in slow))
print(f'enqueue {1e6 * (t1 - t0):.0f} us, whole read {1e6 * (t2 - t0):.0f} us = {t.shape[0] / (t2 - t0) / 1e9:.2f} G')
for name, a, b in slow_calls:
    print(f'  slow library call {name}: at {1e6 * (a - t0):8.0f} us for {1e6 * (b - a):7.0f} us')
for n_fft, a, b in log:
    print(f'  plan {n_fft:8d}: entered at {1e6 * (a - t0):8.0f} us, took {1e6 * (b - a):7.0f} us')
