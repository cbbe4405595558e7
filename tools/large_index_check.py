"""Index arithmetic beyond 32 bits: a 2-pol stream of 2^30 + 2^21 samples (2.15e9 complex elements,
17 GB) through Channelize, Dedisperse, the fused pair and Power+Integrate; the last outputs (highest
indices) and the first are compared with numpy / the oracle on the matching input slices.  Run by
tests/test_gpu_parity.py::test_streams_with_more_than_2_31_elements in its own process."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip, units as u
from oracle import bbt_oracle as orc
from conftest import rel_l2, max_over_rms

T0 = '2020-01-01T00:00:00'


def assert_parity(got, want, what):
    e2, em = rel_l2(got, want), max_over_rms(got, want)
    assert got.shape == want.shape and e2 <= 1e-6 and em <= 1e-5, (what, got.shape, want.shape, e2, em)


def main():
    n = 2**30 + 2**21
    g = torch.Generator(device='cuda')
    g.manual_seed(7)
    x = torch.empty((n, 2, 2), device='cuda', dtype=torch.float32)
    step = 2**26
    for a in range(0, n, step):                                   # (generated in pieces: bounded temporaries)
        x[a:a + step].normal_(generator=g)
    xc = torch.view_as_complex(x)
    hip.set_stream(torch.cuda.current_stream().cuda_stream)
    ds = bt.DeviceStream(xc, T0, 16 * u.MHz, samples_per_frame=2**20, frequency=1000 * u.MHz, sideband=1,
                         polarization=['X', 'Y'])

    def host(a, b):
        return xc[a:b].cpu().numpy()

    # Channelize: first and last spectra
    ch = bt.Channelize(ds, 1024, 2**10)
    ch.max_frames_per_call = 10**6
    z = ch.read_device(ch.shape[0])
    n_spec = ch.shape[0]
    for s0 in (0, n_spec - 3):
        got = z[s0:s0 + 3].to_host()
        want = np.fft.fft(host(s0 * 1024, (s0 + 3) * 1024).astype(np.complex128).reshape(3, 1024, 2), axis=1)
        assert_parity(got, want.astype(np.complex64), f'channelize at spectrum {s0}')
    del z
    ch.close()
    # Dedisperse (2^20 blocks): the last whole block against the oracle on its own input
    dd = bt.Dedisperse(ds, 100., samples_per_frame=836100)
    assert dd._ih_samples_per_frame == 2**20
    dd.max_frames_per_call = 10**6
    y = dd.read_device(dd.shape[0])
    last = dd.shape[0] // 836100 - 1
    a = last * 836100
    want, _ = orc.dedisperse(host(a, a + 2**20), 16e6, 1000., 1, 100., samples_per_frame=836100,
                             ih_samples_per_frame=2**20)
    assert_parity(y[a:a + 836100].to_host(), want[:836100], 'dedisperse, last whole block')
    del y
    dd.invalidate_cache()
    # fused pair + detection: the last integration bin
    it = bt.Integrate(bt.Power(bt.Channelize(dd, 1024, 64)), 64, samples_per_frame=8)
    it.max_frames_per_call = 10**6
    p = it.read_device(it.shape[0])
    b = it.shape[0] - 1
    while True:                    # the last bin whose two blocks lie wholly inside the stream
        first_sample = b * 64 * 1024
        lo = (first_sample // 836100) * 836100
        if lo + 2 * 836100 + 212476 <= n:
            break
        b -= 1
    seg, _ = orc.dedisperse(host(lo, lo + 2 * 836100 + 212476), 16e6, 1000., 1, 100., samples_per_frame=836100,
                            ih_samples_per_frame=2**20)
    spectra = orc.channelize(seg[first_sample - lo:first_sample - lo + 64 * 1024], 1024)
    got, want = p[b:b + 1].to_host()[0], orc.integrate(orc.power(spectra), 64)[0]
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), 'fused detection, last bin'
    print('large index check ok:', n, 'samples,', n_spec, 'spectra, bin', b, 'of', it.shape[0])


if __name__ == '__main__':
    main()
