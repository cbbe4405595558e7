"""Create filter-bank plans over and over (dev tool): every plan on 8 streams and more times its
routes (bbt_hip.hip pfb_pick); BBT_PFB_TRACE=1 prints each step."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip
lib = hip.lib()
torch.zeros(1, device='cuda')
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for n_tap, n_chan, S in ((12, 256, 128), (4, 1024, 16), (16, 512, 128), (12, 256, 16)):
        taps = np.ascontiguousarray(bt.sinc_hamming(n_tap, n_chan), dtype=np.float32)
        plan = ctypes.c_void_p()
        t0 = time.time()
        # (the memo of chosen routes is per (device, channels, taps, streams): vary the stream count to time again)
        S_it = S + 8 * (it % 16)
        rc = lib.bbt_pfb_plan_create(ctypes.byref(plan), n_tap, n_chan, S_it, taps.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        assert rc == 0
        lib.bbt_pfb_plan_destroy(plan)
        print(f"iteration {it}: {n_tap} x {n_chan} on {S_it} streams: plan in {time.time() - t0:.3f} s", flush=True)
