"""CPU baseline of every BASELINE config: the oracle
(numpy restatement of the reference path, complex64 FFTs) on one core and on
every core this job may use, P independent processes over disjoint blocks.
Informational table for profiles/; bench.py itself reports pipeline (c)."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 1 << 20


def work(job):
    name, seed, reps = job
    from oracle import bbt_oracle as orc
    rng = np.random.default_rng(seed)
    streams = 8 if name == 'config5' else 2
    x = rng.standard_normal((N, 2 * streams), dtype=np.float32).view(np.complex64)
    if name == 'config1':
        f = lambda: orc.channelize(x, 1024, fft64=False).shape[0] * 1024
    elif name in ('config2', 'metric'):
        g = orc.disperse_geometry(16e6, 1000., 1, -100.)
        spf = N - g['pad_start'] - g['pad_end']
        h = orc.chirp(N, 16e6, 1000., 1, -100., g['reference_frequency'])

        def f():
            y = orc.disperse_block(x, h, g['pad_start'], spf, fft64=False)
            if name == 'metric':
                k = y.shape[0] // 1024 * 1024
                return orc.channelize(y[:k], 1024, fft64=False).shape[0] * 1024
            return y.shape[0]
    elif name == 'config3':
        resp = orc.sinc_hamming(12, 1024)
        f = lambda: orc.polyphase_filter_bank(x, resp, N, fft64=False)[0].shape[0] * 1024
    else:
        g = orc.disperse_geometry(16e6, 1000., 1, -100.)
        spf = N - g['pad_start'] - g['pad_end']
        h = orc.chirp(N, 16e6, 1000., 1, -100., g['reference_frequency'])

        def f():
            r, _ = orc.resample(x, 0.25, pad=64, samples_per_frame=N - 128, ih_samples_per_frame=N, fft64=False)
            xb = np.concatenate([r, r[:128]])
            return orc.disperse_block(xb, h, g['pad_start'], spf, fft64=False).shape[0]
    f()
    t0 = time.perf_counter()
    n = 0
    for _ in range(reps):
        n += f()
    return n, time.perf_counter() - t0


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    procs, total, quota = bench.usable_cores()
    model = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    out = dict(cpu=model, cores_used=procs, os_cpu_count=total, cgroup_quota=quota, numpy=np.__version__, configs={})
    ctx = mp.get_context('spawn')
    with ctx.Pool(procs) as pool:
        for name in ('config1', 'config2', 'metric', 'config3', 'config5'):
            n1, t1 = work((name, 1, 2))
            reps = max(2, int(4.0 / (t1 / 2 * 2)))
            rates = []
            for rep in range(3):
                res = pool.map(work, [(name, 10 * rep + i, reps) for i in range(procs)], chunksize=1)
                rates.append(sum(r[0] for r in res) / max(r[1] for r in res) / 1e6)
            out['configs'][name] = dict(one_core_msamples_per_s=round(n1 / t1 / 1e6, 2),
                                        all_cores_msamples_per_s=round(sorted(rates)[1], 1), blocks_per_process=reps)
            print(name, out['configs'][name], flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
