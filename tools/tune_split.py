"""Throughput of default-argument Dedisperse for every split N = N1 x N2 of its (generic) block
length (dev tool: data for the split rule of bbt_osm_plan_create).

    python tools/tune_split.py <centre MHz> [reps]          (BBT_GEN_N1 is set per candidate)"""
import os
import subprocess
import sys

if len(sys.argv) > 2 and sys.argv[1] == 'one':
    import time
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import baseband_tasks_amd as bt
    fc = float(sys.argv[2]) * 1e6
    dev = torch.device('cuda', 0)
    bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.view_as_complex(torch.randn((64 * 2**20, 2, 2), generator=g, device=dev, dtype=torch.float32))
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=fc, sideband=1)
    spf = os.environ.get('TUNE_SPF')
    dd = bt.Dedisperse(ds, float(os.environ.get('TUNE_DM', '100')), samples_per_frame=int(spf) if spf else None)
    dd.max_frames_per_call = 10**6
    info = dd._get_plan().info()
    n = dd.shape[0]

    def step():
        dd.invalidate_cache()
        dd.seek(0)
        return dd.read_device(n)
    for _ in range(2):
        step()
    rates = []
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        rates.append(3 * n / (time.perf_counter() - t0) / 1e9)
    print(f"{dd._ih_samples_per_frame} = {info['n1']:5d} x {info['n2']:5d}   " + ' '.join(f'{r:6.2f}' for r in rates)
          + f"   median {sorted(rates)[len(rates) // 2]:6.2f} G", flush=True)
    sys.exit(0)

fc = sys.argv[1]
reps = sys.argv[2] if len(sys.argv) > 2 else '3'
# the block length: ask the host layer (no GPU needed for the geometry, but simpler through one run)
out = subprocess.run([sys.executable, __file__, 'one', fc, '1'], capture_output=True, text=True,
                     env=dict(os.environ, BBT_RTC='require'))
print('default:', out.stdout.strip() or out.stderr[-500:], flush=True)
n = int(out.stdout.split()[0])
cands = [d for d in range(int(os.environ.get('TUNE_MIN', 96)), 2049) if n % d == 0 and n // d <= 8192]
for d in cands:
    r = subprocess.run([sys.executable, __file__, 'one', fc, reps], capture_output=True, text=True,
                       env=dict(os.environ, BBT_RTC='require', BBT_GEN_N1=str(d)))
    print(r.stdout.strip() or ('N1 %d failed: ' % d + r.stderr.strip()[-300:]), flush=True)
