"""Channelize with 256 / 1024 / 4096 channels on 4 ... 2048 streams (dev tool, GPU box): the case where
one stream pair per workgroup reads 16 bytes of every complete sample (BBT_ROWS_NO_PP=1: that route).
    python tools/bench_many_streams.py          (CHAN_N=1000,3000 CHAN_S=16,128: other channel and stream counts)"""
import sys, time, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
gc.disable()
for S in ([int(a) for a in os.environ.get('CHAN_S', '4,8,16,128,2048').split(',')] if os.environ.get('CHAN', '1') != '0' else ()):
    n = (2**28) // S
    x = torch.view_as_complex(torch.randn((n, S, 2), device=dev, dtype=torch.float32))
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 1e6, samples_per_frame=2**16, frequency=300e6, sideband=1)
    for nc in [int(a) for a in os.environ.get('CHAN_N', '256,1024,2048,4096').split(',')]:
        t = bt.Channelize(ds, nc)
        def step():
            t.invalidate_cache(); t.seek(0); return t.read_device(t.shape[0])
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): y = step()
        _ = y.ptr; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        print(f"S={S:5d} Channelize({nc:4d}): {t.shape[0] * nc * S / dt / 1e9:7.1f} G stream-samples/s", flush=True)
    del ds, x

# PolyphaseFilterBank on the same stream counts (PFB=0 skips; PFB_S=16,128: only those stream counts; BBT_PFB_TWO_PASS=0 / 1 forces the route)
if os.environ.get('PFB', '1') != '0':
    for S in [int(a) for a in os.environ.get('PFB_S', '2,4,8,16,128,2048').split(',')]:
        n = (2**28) // S
        x = torch.view_as_complex(torch.randn((n, S, 2), device=dev, dtype=torch.float32))
        ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 1e6, samples_per_frame=2**16 if S > 2 else 2**20, frequency=300e6, sideband=1)
        for ntap, nc in ((4, 1024), (12, 1024)):
            t = bt.PolyphaseFilterBank(ds, bt.sinc_hamming(ntap, nc))
            def step():
                u = t
                while u is not None and hasattr(u, 'invalidate_cache'):
                    u.invalidate_cache(); u = getattr(u, 'ih', None)
                t.seek(0); return t.read_device(t.shape[0])
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): y = step()
            _ = y.ptr; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            print(f"S={S:5d} PolyphaseFilterBank {ntap:2d} x {nc}: {t.shape[0] * nc * S / dt / 1e9:7.1f} G stream-samples/s", flush=True)
        del ds, x, t
