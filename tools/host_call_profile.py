"""Where the host time of one `read_device` call goes (dev tool, GPU box): cProfile over many calls of
Channelize(n) on 16 Mi samples, and the wall time per call against the kernel's.
    python tools/host_call_profile.py [n_chan]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baseband_tasks_amd as bt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
x = torch.view_as_complex(torch.randn((16 * 2**20, 2, 2), device=dev, dtype=torch.float32))
ds = bt.DeviceStream(x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
ch = bt.Channelize(ds, n, 64)
ch.max_frames_per_call = 10**6
count = ch.shape[0]
def step():
    ch.invalidate_cache()
    ch.seek(0)
    return ch.read_device(count)
for _ in range(5):
    step()
torch.cuda.synchronize()
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"Channelize({n}) on 16 Mi samples: {t_all / reps * 1e6:.1f} us per call in all, {t_issue / reps * 1e6:.1f} us to issue "
      f"({count * n * reps / t_all / 1e9:.1f} G samples/s)")
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
