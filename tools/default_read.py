"""Rate of one big read_device() with the task's DEFAULT run size (max_frames_per_call = 32) against
the bench's setting (one plan call for everything): what a user who tunes nothing gets (dev tool)."""
import sys
import time

import torch

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip

dev = torch.device('cuda', 0)
hip.set_stream(torch.cuda.current_stream().cuda_stream)
blocks, spf = 768, 836100
x = torch.view_as_complex(torch.randn(((blocks - 1) * spf + 2**20, 2, 2), device=dev))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
for label, tune in (('defaults', False), ('one call', True)):
    dd = bt.Dedisperse(ds, 100.)
    ch = bt.Channelize(dd, 1024, 512)
    n_spec = (dd.shape[0] // 1024 // 512) * 512
    if tune:
        dd.max_frames_per_call = blocks
        ch.max_frames_per_call = n_spec // 512 + 1

    def step():
        dd.invalidate_cache()
        ch.invalidate_cache()
        ch.seek(0)
        return ch.read_device(n_spec)
    for _ in range(3):
        z = step()          # (keeps the result, as the timed loop does: both result blocks exist afterwards)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        z = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f'{label:10s} {n_spec * 1024 / dt / 1e9:6.2f} Gsamples/s  ({dt * 1e3:.2f} ms per read of {n_spec} spectra)')
    del dd, ch, z
