"""Default-argument reads of device-resident chains (dev tool): `read_device()` of the whole
stream with nothing tuned -- frames of one spectrum, the defaults' block lengths, chains of
several tasks.     python tools/default_device_reads.py [label substring ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
n = 2**27
x = torch.view_as_complex(torch.randn((n, 2, 2), device=dev, dtype=torch.float32))
ds = lambda fc=1000e6, spf=2**20: bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=spf, frequency=fc, sideband=1,
                                                  polarization=np.array(['X', 'Y']))


ONLY = sys.argv[1:]        # (labels to run, as substrings; none: all)


def timeit(make, label, per_sample=1):
    if ONLY and not any(w in label for w in ONLY):
        return
    t = make()
    def step():
        u = t
        while u is not None and hasattr(u, 'invalidate_cache'):
            u.invalidate_cache(); u = getattr(u, 'ih', None)
        t.seek(0)
        return t.read_device(t.shape[0])
    for _ in range(2):
        y = step()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        t0 = time.perf_counter(); y = step(); _ = y.ptr; torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    dt = sorted(times)[2]
    print(f"{label:64s} {t.shape[0] * per_sample / dt / 1e9:8.2f} G input samples/s", flush=True)


timeit(lambda: bt.Channelize(ds(), 1024), 'Channelize(1024)', 1024)
timeit(lambda: bt.Channelize(bt.Dedisperse(ds(), 100.), 1024), 'Channelize(Dedisperse(DM 100), 1024)', 1024)
timeit(lambda: bt.Power(bt.Channelize(bt.Dedisperse(ds(), 100.), 1024)), 'Power(Channelize(Dedisperse))', 1024)
timeit(lambda: bt.Integrate(bt.Power(bt.Channelize(bt.Dedisperse(ds(), 100.), 1024)), 16), 'Integrate(Power(Channelize(Dedisperse)), 16)', 1024 * 16)
timeit(lambda: bt.PolyphaseFilterBank(ds(), bt.sinc_hamming(12, 1024)), 'PolyphaseFilterBank 12 x 1024', 1024)
timeit(lambda: bt.Resample(ds(), 0.25), 'Resample(0.25)')
timeit(lambda: bt.Dedisperse(ds(), 100.), 'Dedisperse(DM 100)')
timeit(lambda: bt.Dedisperse(bt.Resample(ds(), 0.25), 100.), 'Dedisperse(Resample(0.25), DM 100)')
timeit(lambda: bt.Square(bt.Channelize(ds(), 64)), 'Square(Channelize(64))', 64)
timeit(lambda: bt.Dedisperse(ds(1400e6, 20000), 10.), 'Dedisperse(DM 10) on 20000-sample input frames')
timeit(lambda: bt.Channelize(bt.Dedisperse(ds(1400e6, 20000), 10.), 1000), 'Channelize(Dedisperse(DM 10, 20000-sample frames), 1000)', 1000)
