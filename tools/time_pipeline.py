"""Where does the detection pipeline spend its time? (dev tool)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip

dev = torch.device('cuda', 0)
hip.set_stream(torch.cuda.current_stream().cuda_stream)
nblk = 96
x = torch.view_as_complex(torch.randn((nblk * 2**20, 2, 2), device=dev))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1,
                     polarization=['X', 'Y'])
dd = bt.Dedisperse(ds, 100.)
ch = bt.Channelize(dd, 1024, 512)
pw = bt.Integrate(bt.Power(ch), 64)
for t in (dd, ch, pw):
    t.max_frames_per_call = 10**6


def timed(label, fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:50s} {dt * 1e3:8.3f} ms")
    return dt


def inval():
    for t in (dd, ch, pw):
        t.invalidate_cache()


def run_ch(n):
    inval()
    ch.seek(0)
    return ch.read_device(n)


def run_pw():
    inval()
    pw.seek(0)
    return pw.read_device(pw.shape[0])


print('ch.shape', ch.shape, 'pw.shape', pw.shape)
timed('channelize all', lambda: run_ch(ch.shape[0]))
timed('channelize pw.shape[0]*64 spectra', lambda: run_ch(pw.shape[0] * 64))
timed('power+integrate pipeline', run_pw)

# same through a SetAttribute wrapper (as tools/bench_configs.py does)
ds2 = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=900e6, sideband=1)
dsf = bt.SetAttribute(ds2, frequency=1000e6, polarization=['X', 'Y'])
dd = bt.Dedisperse(dsf, 100.)
ch = bt.Channelize(dd, 1024, 512)
pw = bt.Integrate(bt.Power(ch), 64)
for t in (dd, ch, pw):
    t.max_frames_per_call = 10**6
timed('SetAttribute: channelize all', lambda: run_ch(ch.shape[0]))
timed('SetAttribute: power+integrate pipeline', run_pw)
