"""Time the detection + integration kernel alone (dev tool)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from baseband_tasks_amd import hip

dev = torch.device('cuda', 0)
hip.set_stream(torch.cuda.current_stream().cuda_stream)
n_spec, n_chan = 78336, 1024
x = torch.view_as_complex(torch.randn((n_spec, n_chan, 2, 2), device=dev))
xa = hip.DeviceArray(tuple(x.shape), np.complex64, ptr=x.data_ptr(), owner=x)
for step in (16, 64, 512):
    n_out = n_spec // step
    out = hip.DeviceArray((n_out, n_chan, 4), np.float32)
    for mode, name, n_elem in ((1, 'Power', n_chan), (0, 'Square', 2 * n_chan)):
        for _ in range(3):
            hip.detect_integrate(xa, out, n_out, step, n_elem, mode, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            hip.detect_integrate(xa, out, n_out, step, n_elem, mode, True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:7s} step {step:4d}: {dt * 1e3:7.3f} ms  {n_spec * n_chan * 16 / dt / 1e12:5.2f} TB/s read")
