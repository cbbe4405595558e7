"""Dedisperse with the reference's default (non power-of-two) block on several streams, next to the
power-of-two block (dev tool, GPU box).    python tools/bench_generic_streams.py [streams ...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baseband_tasks_amd as bt
from baseband_tasks_amd.fourier import HipFFTMaker
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
for S in [int(a) for a in sys.argv[1:]] or (2, 4, 16, 128):
    n = min(2**29 // S, 2**27)
    x = torch.view_as_complex(torch.randn((n, S, 2), device=dev, dtype=torch.float32))
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=800e6, sideband=1)
    line = f"S={S:4d}"
    for name, maker in (('default', None), ('power of two', HipFFTMaker(power_of_two=True))):
        if maker is None:
            dd = bt.Dedisperse(ds, 100.)
        else:
            with bt.fft_maker.set(maker):
                dd = bt.Dedisperse(ds, 100.)
        dd.max_frames_per_call = 10**6
        info = dd._get_plan().info()
        def step():
            dd.invalidate_cache(); dd.seek(0); return dd.read_device(dd.shape[0])
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): y = step()
        _ = y.ptr; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        line += f"   {name} block {dd._ih_samples_per_frame} = {info['n1']} x {info['n2']}: {dd.shape[0] * S / dt / 1e9:6.1f} G stream-samples/s"
        del dd, y
    print(line, flush=True)
    del ds, x
    gc.collect()
