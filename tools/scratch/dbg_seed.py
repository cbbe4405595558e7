import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u
from oracle import bbt_oracle as orc
T0 = '2020-01-01T00:00:00'
seed = int(sys.argv[1])
rng = np.random.default_rng(77 + seed)
for case in range(16):
    n_fft = int(2 ** rng.integers(13, 18))
    n_chan = int(rng.choice([16, 32, 64, 128, 256, 512, 1024, 2048, 4096]))
    fs = 2e6
    dm = float(rng.uniform(2., 8.))
    g = orc.disperse_geometry(fs, 400., 1, -dm)
    pad = g['pad_start'] + g['pad_end']
    if pad >= n_fft // 2 or n_chan > n_fft - pad:
        continue
    spf = n_fft - pad
    n_in = int(spf * rng.integers(2, 5) + pad + rng.integers(0, spf))
    sample_shape = (2,) if case % 3 else (2, 2)
    x = (rng.standard_normal((n_in,) + sample_shape) + 1j * rng.standard_normal((n_in,) + sample_shape)).astype(np.complex64)
    ds = bt.DeviceStream(x, T0, fs, frequency=400 * u.MHz, sideband=1)
    dd = bt.Dedisperse(ds, dm, samples_per_frame=spf)
    ch = bt.Channelize(dd, n_chan, samples_per_frame=int(min(rng.integers(1, 40), dd.shape[0] // n_chan)))
    y, _ = orc.dedisperse(x, fs, 400., 1, dm, samples_per_frame=spf, ih_samples_per_frame=min(n_in, 4096))
    z = ch.read()
    want = orc.channelize(y[:z.shape[0] * n_chan], n_chan)
    err = np.abs(z - want).reshape(z.shape[0], -1).max(axis=1) / np.sqrt(np.mean(np.abs(want) ** 2))
    bad = np.nonzero(err > 1e-4)[0]
    print(f'case {case} n_fft {n_fft} n_chan {n_chan} spf {spf} pad {g["pad_start"]}+{g["pad_end"]} frames {ch.samples_per_frame} nspec {z.shape[0]} bad {bad.tolist()}')
    for b in bad[:4]:
        e = np.abs(z[b] - want[b]).reshape(n_chan, -1).max(axis=1)
        nz = np.nonzero(e > 1e-3 * np.abs(want[b]).max())[0]
        print('   spectrum', b, 'sample range', b * n_chan, (b + 1) * n_chan, 'block seam at', [k * spf for k in range(1, 6)],
              'bad channels', len(nz), nz[:8].tolist(), nz[-4:].tolist())
