"""Soak (dev tool, GPU box): six pipelines -- fused and plain, one-kernel, two-level, generic and
2^20-sample blocks -- read in random pieces with random call sizes (`max_frames_per_call`), through
`read` and `read_device`, every piece bit-identical to the same stream read whole: deferred calls,
alternating caches, regular runs and descriptor chunks in every mix.     python tools/soak_reads.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import baseband_tasks_amd as bt
rng = np.random.default_rng(5)
fs, dm = 1e6, 5.
cases = []
def make_case(n_fft, n_chan):
    x = (rng.standard_normal((40 * n_fft, 2)) + 1j * rng.standard_normal((40 * n_fft, 2))).astype(np.complex64)
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', fs, frequency=300e6, sideband=1)
    p = bt.Dedisperse(ds, dm)
    pad = p._pad_start + p._pad_end
    if pad >= n_fft // 2:
        return None
    def mk():
        dd = bt.Dedisperse(ds, dm, samples_per_frame=n_fft - pad)
        return bt.Channelize(dd, n_chan, 3) if n_chan else dd
    return n_fft, n_chan, mk, mk().read()
for n_fft, n_chan in ((2**15, 256), (2**13, 0), (2**17, 64), (6174, 0), (2**20, 1024), (2**14, 0)):
    c = make_case(n_fft, n_chan)
    if c: cases.append(c)
t0 = time.time()
calls = 0
for it in range(40):
    for n_fft, n_chan, mk, want in cases:
        t = mk()
        t.max_frames_per_call = int(rng.integers(1, 200))
        if n_chan:
            t.ih.max_frames_per_call = int(rng.integers(1, 40))
        pos = 0
        while pos < want.shape[0]:
            n = int(min(want.shape[0] - pos, rng.integers(1, max(2, want.shape[0] // 3))))
            got = t.read_device(n).to_host() if rng.integers(2) else t.read(n)
            assert np.array_equal(got, want[pos:pos + n]), (it, n_fft, n_chan, pos, n)
            pos += n
            calls += 1
print('soak ok:', calls, 'reads over', len(cases), 'pipelines in', round(time.time() - t0, 1), 's')
