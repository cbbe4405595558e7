"""How far ahead of the GPU is the host?  (dev tool, GPU box)

For the headline pipeline (768 blocks per call): the host time of one `read_device` call issued on
an idle GPU (pure enqueue cost: Python + C loop + launches), the GPU time of the call, the host's
return times of a burst of calls issued back to back (does the runtime let the host run ahead?),
and a cProfile of the host side of one call.

    python tools/host_enqueue.py [--blocks 768] [--calls 6]
"""
import argparse
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument('--blocks', type=int, default=768)
ap.add_argument('--calls', type=int, default=6)
ap.add_argument('--own-stream', action='store_true', help='a stream of its own instead of the null stream')
args = ap.parse_args()
dev = torch.device('cuda', 0)
if args.own_stream:
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
hip.set_stream(torch.cuda.current_stream().cuda_stream)
spf = 836100
x = torch.view_as_complex(torch.randn(((args.blocks - 1) * spf + 2**20, 2, 2), device=dev))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
dd = bt.Dedisperse(ds, 100.)
ch = bt.Channelize(dd, 1024, 512)
dd.max_frames_per_call = args.blocks
n_spec = (dd.shape[0] // 1024 // 512) * 512
ch.max_frames_per_call = n_spec // 512 + 1


def step():
    dd.invalidate_cache()
    ch.invalidate_cache()
    ch.seek(0)
    return ch.read_device(n_spec)


for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'idle GPU: host enqueue {1e3 * (t1 - t0):7.3f} ms, call complete after {1e3 * (t2 - t0):7.3f} ms')
t0 = time.perf_counter()
marks = []
for _ in range(args.calls):
    step()
    marks.append(time.perf_counter() - t0)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print('burst: host returned at (ms)', [round(1e3 * m, 2) for m in marks], 'all complete', round(1e3 * total, 2))
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    print(f'12 calls: {1e3 * (time.perf_counter() - t0) / 12:7.3f} ms per call')
prof = cProfile.Profile()
prof.enable()
step()
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats('cumulative').print_stats(18)
