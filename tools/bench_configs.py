"""Throughput of the non-headline BASELINE configs on one GPU (dev tool):
config 1 (Channelize 1024), config 3 (PFB 12x1024), config 5 (8-stream
Resample -> Dedisperse), plus Dedisperse alone.  HBM-resident input."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)


def stream(n, S, seed=1):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.view_as_complex(torch.randn((n, S, 2), generator=g, device=dev, dtype=torch.float32))
    return bt.DeviceStream(x if S > 1 else x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20,
                           frequency=1000e6, sideband=1)


def timeit(tasks, last, count, reps=10):
    def step():
        for t in tasks:
            t.invalidate_cache()
        last.seek(0)
        return last.read_device(count)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    nblk = 32
    n = nblk * 2**20
    ds = stream(n, 2)
    # config 1
    ch = bt.Channelize(ds, 1024, 512)
    ch.max_frames_per_call = 10**6
    dt = timeit([ch], ch, ch.shape[0])
    print(f"config1 Channelize(1024): {ch.shape[0] * 1024 / dt / 1e6:9.1f} Msamples/s  ({32 * ch.shape[0] * 1024 / dt / 1e9:.0f} GB/s algorithmic)")
    # config 3
    pfb = bt.PolyphaseFilterBank(ds, bt.sinc_hamming(12, 1024))
    pfb.max_frames_per_call = 10**6
    dt = timeit([pfb], pfb, pfb.shape[0])
    print(f"config3 PFB(12x1024):     {pfb.shape[0] * 1024 / dt / 1e6:9.1f} Msamples/s  ({32 * pfb.shape[0] * 1024 / dt / 1e9:.0f} GB/s algorithmic)")
    # dedisperse alone
    dd = bt.Dedisperse(ds, 100.)
    dd.max_frames_per_call = 10**6
    dt = timeit([dd], dd, dd.shape[0])
    print(f"config2 Dedisperse:       {dd.shape[0] / dt / 1e6:9.1f} Msamples/s")
    del ch, pfb, dd, ds
    # config 5: 8 streams
    nblk = 12
    ds8 = stream(nblk * 2**20, 8)
    rs = bt.Resample(ds8, 0.25, pad=64, samples_per_frame=2**20 - 128)
    rs.seek(0)
    dd = bt.Dedisperse(rs, 100., samples_per_frame=2**20 - 212476)
    rs.max_frames_per_call = dd.max_frames_per_call = 10**6
    dt = timeit([rs, dd], dd, dd.shape[0], reps=5)
    print(f"config5 Resample->Dedisperse, 8 streams: {dd.shape[0] / dt / 1e6:9.1f} Msamples/s "
          f"(complete 8-stream samples; x4 = {4 * dd.shape[0] / dt / 1e6:.0f} 2-pol-equivalent)")
    dd8 = bt.Dedisperse(ds8, 100.)
    dd8.max_frames_per_call = 10**6
    dt = timeit([dd8], dd8, dd8.shape[0], reps=5)
    print(f"        Dedisperse alone, 8 streams:     {dd8.shape[0] / dt / 1e6:9.1f} Msamples/s (x4 = {4 * dd8.shape[0] / dt / 1e6:.0f})")


if __name__ == '__main__':
    main()
