"""One BASELINE config per invocation, HBM-resident, one JSON line (dev tool; the
unit rocprofv3 is wrapped around for the per-config profiles in profiles/).

    python tools/bench_one.py config1|config2|config3|config5 [--reps N] [--blocks B]

config1  Channelize(1024), 2 pol            algorithmic 32 B per complete sample
config2  Dedisperse DM 100, 2^20 blocks     36.07 B
config3  PolyphaseFilterBank 12 x 1024      32 B
config5  Resample(0.25) -> Dedisperse, 8 streams, 2^20 blocks   144.3 B
(config 4 and the headline are `bench.py --workload config4` / `bench.py`.)
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baseband_tasks_amd as bt

ALG = dict(config1=32.0, config2=36.07, config3=32.0, config5=144.3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', choices=sorted(ALG))
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--blocks', type=int, default=None)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
    streams = 8 if args.config == 'config5' else 2
    nblk = args.blocks or (24 if streams == 8 else 384)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.view_as_complex(torch.randn((nblk * 2**20, streams, 2), generator=g, device=dev,
                                          dtype=torch.float32))
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6,
                         sideband=1)
    if args.config == 'config1':
        last = bt.Channelize(ds, 1024, 512)
        tasks, count, unit = [last], last.shape[0], 1024
    elif args.config == 'config2':
        last = bt.Dedisperse(ds, 100.)
        tasks, count, unit = [last], last.shape[0], 1
    elif args.config == 'config3':
        last = bt.PolyphaseFilterBank(ds, bt.sinc_hamming(12, 1024))
        tasks, count, unit = [last], last.shape[0], 1024
    else:
        rs = bt.Resample(ds, 0.25, pad=64, samples_per_frame=2**20 - 128)
        rs.seek(0)
        last = bt.Dedisperse(rs, 100., samples_per_frame=2**20 - 212476)
        tasks, count, unit = [rs, last], last.shape[0], 1
    for t in tasks:
        t.max_frames_per_call = 10**6

    def step():
        for t in tasks:
            t.invalidate_cache()
        last.seek(0)
        return last.read_device(count)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    rate = count * unit / dt
    print(json.dumps(dict(config=args.config, msamples_per_s=round(rate / 1e6, 1), ms_per_step=round(dt * 1e3, 4),
                          complete_samples_per_step=count * unit, streams=streams, blocks=nblk,
                          alg_bytes_per_sample=ALG[args.config],
                          alg_gbps=round(rate * ALG[args.config] / 1e9, 1),
                          roofline_frac=round(rate * ALG[args.config] / 8e12, 4))))


if __name__ == '__main__':
    main()
