"""Host-to-host rate of the .read() path (numpy in, numpy out, PCIe both ways)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

nblk = 24
n = (nblk - 1) * 836100 + 2**20
rng = np.random.default_rng(1)
base = rng.standard_normal((2**20, 4), dtype=np.float32).view(np.complex64)
x = np.concatenate([base] * (n // 2**20 + 1))[:n]


def src(fh):
    return x[fh.tell():fh.tell() + fh.samples_per_frame]


nh = bt.StreamGenerator(src, x.shape, '2020-01-01T00:00:00', 16 * u.MHz, samples_per_frame=2**20,
                        frequency=1000 * u.MHz, sideband=1)
for name, make in (('Dedisperse', lambda: bt.Dedisperse(nh, 100.)),
                   ('Dedisperse->Channelize(1024)', lambda: bt.Channelize(bt.Dedisperse(nh, 100.), 1024, 512))):
    t = make()
    t.read(t.samples_per_frame)          # plan + warm-up
    t.seek(0)
    t0 = time.perf_counter()
    out = t.read()
    dt = time.perf_counter() - t0
    ns = out.shape[0] * (1024 if out.ndim == 3 else 1)
    print(f"{name}: {ns / dt / 1e6:.0f} Msamples/s host-to-host ({ns * 16 / dt / 1e9:.1f} GB/s out, {dt:.3f} s)")
