"""Rate of Channelize(Dedisperse(x), 1024) against the frames one call computes (dev tool, GPU box):
the default of `DeviceTaskMixin.max_frames_per_call` is a memory bound, this is what it costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
n = 2**28
x = torch.view_as_complex(torch.randn((n, 2, 2), device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
for per in (None, 20, 40, 80, 160, 320, 10**6):
    ch = bt.Channelize(bt.Dedisperse(ds, 100.), 1024)
    if per is not None:
        ch.max_frames_per_call = per
    def step():
        u = ch
        while u is not None and hasattr(u, 'invalidate_cache'):
            u.invalidate_cache(); u = getattr(u, 'ih', None)
        ch.seek(0)
        return ch.read_device(ch.shape[0])
    for _ in range(2): step()
    torch.cuda.synchronize()
    times = []
    for _ in range(4):
        t0 = time.perf_counter(); y = step(); _ = y.ptr; torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print(f"max_frames_per_call {str(per):>8s} (effective {ch.max_frames_per_call}): {ch.shape[0] * 1024 / sorted(times)[1] / 1e9:6.2f} G samples/s", flush=True)
    del ch, y
