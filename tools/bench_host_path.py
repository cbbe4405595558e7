"""Host-to-host rate of the .read() path (NumPy in, NumPy out, PCIe both ways), dev tool.

    python tools/bench_host_path.py [--blocks B] [--run R]

The metric pipeline (and Dedisperse alone) on a host-resident stream: `HostStream` over
page-locked memory -> device tasks -> ``read()`` into a page-locked result, upload /
transforms / download of consecutive runs of R blocks overlapping (host_pipeline.py); then
the same with BBT-style synchronous copies (``out=`` an ordinary array) for comparison.
"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import host_pipeline as hp
from baseband_tasks_amd import units as u


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=192)
    ap.add_argument('--run', type=int, default=16)
    args = ap.parse_args()
    n = (args.blocks - 1) * 836100 + 2**20
    rng = np.random.default_rng(1)
    base = rng.standard_normal((2**20, 4), dtype=np.float32).view(np.complex64)
    x = hp.pinned_empty((n, 2), np.complex64)
    for s in range(0, n, 2**20):
        x[s:s + 2**20] = base[:min(2**20, n - s)]
    nh = bt.HostStream(x, '2020-01-01T00:00:00', 16 * u.MHz, samples_per_frame=2**20,
                       frequency=1000 * u.MHz, sideband=1)
    rows = []
    for name, make in (('Dedisperse', lambda: bt.Dedisperse(nh, 100.)),
                       ('Dedisperse->Channelize(1024)',
                        lambda: bt.Channelize(bt.Dedisperse(nh, 100.), 1024, 512))):
        for mode in ('pipelined', 'synchronous'):
            t = make()
            t.max_frames_per_call = args.run if name == 'Dedisperse' else args.run * 836100 // (512 * 1024) + 1
            if name != 'Dedisperse':
                t.ih.max_frames_per_call = args.run + 2
            t.read(t.samples_per_frame)          # plan + warm-up
            best = None
            for _ in range(3):
                t.invalidate_cache()
                t.seek(0)
                out = None if mode == 'pipelined' else np.empty(t.shape, t.dtype)
                t0 = time.perf_counter()
                out = t.read(out=out) if out is not None else t.read()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            ns = out.shape[0] * (1024 if out.ndim == 3 else 1)
            rows.append(dict(task=name, mode=mode, msamples_per_s=round(ns / best / 1e6, 1),
                             h2d_gbps=round(x.nbytes / best / 1e9, 2), d2h_gbps=round(out.nbytes / best / 1e9, 2),
                             seconds=round(best, 4), blocks=args.blocks, blocks_per_run=args.run))
            print(json.dumps(rows[-1]), flush=True)
            del out
            t.close()


if __name__ == '__main__':
    main()
