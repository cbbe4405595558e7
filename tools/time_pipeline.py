"""Per-pass times of the metric pipeline with and without fused detection (dev tool)."""
import sys
import time

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


def run(last, n):
    for t in (dd, ch, pw):
        t.invalidate_cache()
    last.seek(0)
    return last.read_device(n)


plan = dd._get_plan()
for label, last, n in (('spectra stored', ch, ch.shape[0]), ('detected + integrated', pw, pw.shape[0])):
    for mode in (1, 2):
        for _ in range(3):
            run(last, n)
        torch.cuda.synchronize()
        plan.timing_enable(mode)
        t0 = time.perf_counter()
        reps = 8
        for _ in range(reps):
            run(last, n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ms, launches = plan.timing_read()
        plan.timing_enable(0)
        nb = 120 * reps
        print(f"{label:24s} mode {mode}: wall {dt * 1e3:6.3f} ms  passes per block (us): "
              f"A {ms[0] / nb * 1e3:5.2f}  B {ms[1] / nb * 1e3:5.2f}  C {ms[2] / nb * 1e3:5.2f}")
