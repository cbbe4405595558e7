"""Dedisperse with the reference's default block length (1 666 980 = 1260 x 1323 at 800 MHz): the
generic kernels alone, for profiling (dev tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baseband_tasks_amd as bt

dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
x = torch.view_as_complex(torch.randn((96 * 2**20, 2, 2), generator=g, device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=float(sys.argv[1]) if len(sys.argv) > 1 else 800e6,
                     sideband=1)
dd = bt.Dedisperse(ds, 100.)
dd.max_frames_per_call = 10**6
print(dd._ih_samples_per_frame, dd._get_plan().info())
n = dd.shape[0]
for _ in range(2):
    dd.invalidate_cache(); dd.seek(0); dd.read_device(n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dd.invalidate_cache(); dd.seek(0); dd.read_device(n)
torch.cuda.synchronize()
print(n * 5 / (time.perf_counter() - t0) / 1e6, 'Msamples/s')
