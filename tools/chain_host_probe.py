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
t = bt.Dedisperse(bt.Resample(ds, 0.25), 100.)
log = []
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


for rep in range(4):
    del log[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = step()
    t1 = time.perf_counter()
    _ = y.ptr
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'enqueue {1e6 * (t1 - t0):.0f} us, whole read {1e6 * (t2 - t0):.0f} us = {t.shape[0] / (t2 - t0) / 1e9:.2f} G')
for n_fft, a, b in log:
    print(f'  plan {n_fft:8d}: entered at {1e6 * (a - t0):8.0f} us, took {1e6 * (b - a):7.0f} us')
