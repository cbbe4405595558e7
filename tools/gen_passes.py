"""Per-pass kernel times of a generic-length (default-argument) Dedisperse plan, in the two-lane
schedule and isolated (dev tool):  python tools/gen_passes.py [centre MHz ...]"""
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
for fc in [float(a) * 1e6 for a in sys.argv[1:]] or [800e6, 600e6]:
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=fc, sideband=1)
    dd = bt.Dedisperse(ds, 100.)
    dd.max_frames_per_call = 10**6
    plan = dd._get_plan()
    info = plan.info()
    n = dd.shape[0]

    def step():
        dd.invalidate_cache()
        dd.seek(0)
        return dd.read_device(n)
    for mode in (1, 2):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        plan.timing_enable(mode)
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, launches, blocks = plan.timing_read_passes()
        plan.timing_enable(0)
        per = [1e3 * m / max(b, 1) for m, b in zip(ms, blocks)]
        print(f"{fc / 1e6:5.0f} MHz  {dd._ih_samples_per_frame} = {info['n1']} x {info['n2']}  chunk {info['chunk_blocks']}  "
              f"{'lanes   ' if mode == 1 else 'isolated'}: first {per[0]:6.2f}  row {per[1]:6.2f}  last {per[2]:6.2f} us per block   "
              f"sum {sum(per):6.2f}   wall {1e6 * dt / 4 / (n / dd.samples_per_frame):6.2f} us per block  "
              f"{4 * n / dt / 1e6:8.1f} Msamples/s", flush=True)
print(bt.hip.rtc_info())
