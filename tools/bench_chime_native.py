"""The CHIME-native form of config 4 ('worth one run'): 1024 sub-bands of 390.625 kHz x 2 pol
(2048 streams), DM 557 with per-sub-band reference frequencies -- the reference's defaults then give
2^16-sample blocks (padding 5499 + 5507) -- followed by Channelize(4) (dev tool, GPU box).
    python tools/bench_chime_native.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
nsub = 1024
fs = 400e6 / nsub                                   # 390.625 kHz
freq = (400e6 + fs * (np.arange(nsub) + 0.5)).reshape(nsub, 1)
n = 2**19
x = torch.view_as_complex(torch.randn((n, nsub, 2, 2), device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', fs, samples_per_frame=2**16, frequency=freq, sideband=1,
                     polarization=np.array(['X', 'Y']))
dd = bt.Dedisperse(ds, 557., reference_frequency=freq)
ch = bt.Channelize(dd, 4)
info = dd._get_plan().info()
print('block', dd._ih_samples_per_frame, 'pad', dd._pad_start, dd._pad_end, 'spf', dd.samples_per_frame, 'plan', info, 'fused', ch._fusable_input() is not None)
for t, name, per in ((dd, 'Dedisperse', 1), (ch, 'Dedisperse -> Channelize(4)', 4)):
    def step():
        u = t
        while u is not None and hasattr(u, 'invalidate_cache'):
            u.invalidate_cache(); u = getattr(u, 'ih', None)
        t.seek(0); return t.read_device(t.shape[0])
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): y = step()
    _ = y.ptr; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    rate = t.shape[0] * per / dt
    eta = dd.samples_per_frame / dd._ih_samples_per_frame
    alg = 2048 * 8 * (1 / eta + 1)
    print(f'{name}: {rate / 1e6:8.2f} M complete samples/s (2048 streams) = {rate * 2048 / 1e9:6.1f} G stream-samples/s; algorithmic {rate * alg / 1e12:5.2f} TB/s = {rate * alg / 8e12:5.3f} of 8 TB/s', flush=True)
