"""Dedisperse (2^20-sample blocks, DM 100 at 1000 MHz) on S complex streams resident in HBM: G complete
samples/s and the fraction of 8 TB/s on the algorithmic bytes (dev tool for the many-stream tile choices,
the pair-grouped column tiles).  python tools/bench_streams.py [S ...]"""
import gc
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baseband_tasks_amd as bt

dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
for s in [int(a) for a in sys.argv[1:]] or [16]:
    blocks = max(4, 384 // s)
    spf = 2**20 - 212476
    n_in = (blocks - 1) * spf + 2**20
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.view_as_complex(torch.randn((n_in, s, 2), generator=g, device=dev, dtype=torch.float32))
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    dd = bt.Dedisperse(ds, 100., samples_per_frame=spf)
    dd.max_frames_per_call = blocks

    def step():
        dd.invalidate_cache()
        dd.seek(0)
        return dd.read_device(dd.shape[0])
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n = dd.shape[0]
    alg = (8 * 2**20 / spf + 8) * s
    print(json.dumps(dict(streams=s, blocks=blocks, gsamples_per_s=round(n / dt / 1e9, 3),
                          frac=round(n / dt * alg / 8e12, 4))), flush=True)
    dd.close()
    del x, ds, dd
    gc.collect()
    bt.hip.pool_trim() if hasattr(bt.hip, 'pool_trim') else None
