"""Kernel times of default-argument Dedisperse(Resample(x)) (dev tool; run under rocprofv3 --kernel-trace)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
n = 2**27
x = torch.view_as_complex(torch.randn((n, 2, 2), device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1,
                     polarization=np.array(['X', 'Y']))
rs = bt.Resample(ds, 0.25)
t = bt.Dedisperse(rs, 100.)
print('resample frame', rs.samples_per_frame, rs._ih_samples_per_frame, 'dedisperse block', t._ih_samples_per_frame, 'spf', t.samples_per_frame,
      'plan', t._get_plan().info(), 'max frames', t.max_frames_per_call, rs.max_frames_per_call)
def step():
    u = t
    while u is not None and hasattr(u, 'invalidate_cache'):
        u.invalidate_cache(); u = getattr(u, 'ih', None)
    t.seek(0)
    return t.read_device(t.shape[0])
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    y = step()
_ = y.ptr
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f'{t.shape[0] / dt / 1e9:.2f} G samples/s, {dt * 1e3:.2f} ms per pass')
