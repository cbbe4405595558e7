"""Throughput of the generic (non power-of-two) overlap-save path next to the
power-of-two one (dev tool): Dedisperse at 800 / 1000 / 1400 MHz, default block
(the reference's rule) and `power_of_two=True`.

    python tools/bench_generic.py [ded:<MHz> ...] [pow2:<MHz> ...] [chan:<n> ...]      (no arguments: all rows;
    with arguments only those, `ded:` without the power-of-two column: for profiling one kernel)"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baseband_tasks_amd as bt
from baseband_tasks_amd.fourier import HipFFTMaker

dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
x = torch.view_as_complex(torch.randn((96 * 2**20, 2, 2), generator=g, device=dev, dtype=torch.float32))


def rate(task, reps=5):
    task.max_frames_per_call = 10**6
    n = task.shape[0]

    def step():
        t = task
        while t is not None and hasattr(t, 'invalidate_cache'):     # (an unfused chain: every task computes)
            t.invalidate_cache()
            t = getattr(t, 'ih', None)
        task.seek(0)
        return task.read_device(n)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    gc.enable()
    return n * reps / (time.perf_counter() - t0) / 1e6


only = sys.argv[1:]
centres = [float(a[4:]) * 1e6 for a in only if a.startswith('ded:')] if only else (800e6, 1000e6, 1400e6, 600e6)
lengths = [int(a[5:]) for a in only if a.startswith('chan:')] if only else (1000, 1536, 3000, 6561, 8192, 16384)
for n_fft in [int(a[4:]) for a in only if a.startswith('blk:')]:               # short power-of-two blocks
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1400e6, sideband=1)
    probe = bt.Dedisperse(ds, 1.)
    pad = probe._pad_start + probe._pad_end
    dd = bt.Dedisperse(ds, 1., samples_per_frame=n_fft - pad)
    info = dd._get_plan().info()
    print(f"block {dd._ih_samples_per_frame:6d} = {info['n1']} x {info['n2']} (padding {pad}): {rate(dd):9.1f} Msamples/s", flush=True)
    ch = bt.Channelize(dd, 256, 512)
    print(f"{'':10s}... -> Channelize(256) (fused: {ch._fusable_input() is not None}): {rate(ch) * 256:9.1f} Msamples/s", flush=True)
    del dd, probe, ch
for fc in [float(a[5:]) * 1e6 for a in only if a.startswith('pow2:')]:          # power-of-two blocks only
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=fc, sideband=1)
    with bt.fft_maker.set(HipFFTMaker(power_of_two=True)):
        d2 = bt.Dedisperse(ds, 100.)
    info = d2._get_plan().info()
    print(f"{fc / 1e6:6.0f} MHz power-of-two block {d2._ih_samples_per_frame:8d} = {info['n1']} x {info['n2']}: "
          f"{rate(d2):9.1f} Msamples/s", flush=True)
    ch = bt.Channelize(d2, 1024, 512)
    print(f"{'':10s}... -> Channelize(1024) (fused: {ch._fusable_input() is not None}): {rate(ch) * 1024:9.1f} Msamples/s", flush=True)
    del d2, ch
for fc in centres:
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=fc, sideband=1)
    dd = bt.Dedisperse(ds, 100.)
    info = dd._get_plan().info()
    line = f"{fc / 1e6:6.0f} MHz default block {dd._ih_samples_per_frame:8d} = {info['n1']} x {info['n2']}: {rate(dd):9.1f} Msamples/s"
    d2 = None
    if not only:
        with bt.fft_maker.set(HipFFTMaker(power_of_two=True)):
            d2 = bt.Dedisperse(ds, 100.)
        line += f"   | power of two {d2._ih_samples_per_frame:8d}: {rate(d2):9.1f} Msamples/s"
    print(line, flush=True)
    del dd, d2
for n in lengths:
    # 16 Mi samples a call as in rounds 3 and 4 (0.1 ms of kernel: a few thousand transforms, two or
    # three rounds of workgroups, and the call's host time count), and 96 Mi (what the kernel sustains)
    line = f"Channelize({n}):"
    for count in (16, 96):
        ds = bt.DeviceStream(x[:count * 2**20], '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
        ch = bt.Channelize(ds, n, 64)
        line += f" {rate(ch) * n:9.1f} Msamples/s ({count} Mi samples a call)"
        del ch, ds
    print(line, flush=True)
