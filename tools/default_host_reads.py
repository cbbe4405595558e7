"""Default-argument reads from host memory (dev tool): NumPy in -> task(s) -> NumPy out through
`read()`, for the frame sizes the defaults give -- the cases that showed the run-size and
read-ahead cliffs of round 4.     python tools/default_host_reads.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import baseband_tasks_amd as bt
from baseband_tasks_amd import host_pipeline as hp
n = 2**26
x = hp.pinned_empty((n, 2), np.complex64)
x[:] = (np.random.default_rng(1).standard_normal((2**20, 4), dtype=np.float32).view(np.complex64))[np.arange(n) % 2**20]
def timeit(make, label, unit_per_sample=1):
    t = make()
    for _ in range(2):
        t.seek(0); y = t.read()
    times = []
    for _ in range(9):
        t0 = time.perf_counter()
        t.seek(0); y = t.read()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]                 # median of 9 (a read takes 25 ms: single ones scatter)
    print(f"{label:60s} {t.shape[0] * unit_per_sample / dt / 1e9:7.2f} Gsamples/s  ({y.nbytes / dt / 1e9:5.1f} GB/s down; "
          f"fastest {t.shape[0] * unit_per_sample / min(times) / 1e9:5.2f}, slowest {t.shape[0] * unit_per_sample / max(times) / 1e9:5.2f})", flush=True)
hs = lambda spf: bt.HostStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=spf, frequency=1400e6, sideband=1)
timeit(lambda: bt.Channelize(hs(2**20), 1024), 'Channelize(1024), default frames of one spectrum', 1024)
timeit(lambda: bt.Channelize(hs(20000), 1000), 'Channelize(1000) on 20000-sample input frames', 1000)
timeit(lambda: bt.PolyphaseFilterBank(hs(2**20), bt.sinc_hamming(12, 1024)), 'PolyphaseFilterBank 12 x 1024, defaults', 1024)
timeit(lambda: bt.Dedisperse(hs(20000), 10.), 'Dedisperse DM 10, input frames of 20000 (default block)')
timeit(lambda: bt.Dedisperse(hs(2**20), 100.), 'Dedisperse DM 100 at 1400 MHz, 2^20 frames')
timeit(lambda: bt.Channelize(bt.Dedisperse(hs(20000), 10.), 1024, 64), 'Dedisperse(20000-frames) -> Channelize(1024, 64)', 1024)
timeit(lambda: bt.Resample(hs(2**20), 0.25), 'Resample, defaults')
