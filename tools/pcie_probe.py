"""Raw host <-> device copy rates through the C ABI (dev tool): page-locked and pageable memory,
each direction alone and both at once on two streams."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import hip
from baseband_tasks_amd import host_pipeline as hp

L = hip.lib()
n = 1 << 30
dev_a, dev_b = hip.DeviceArray((n,), np.uint8), hip.DeviceArray((n,), np.uint8)
pin_a, pin_b = hp.pinned_empty((n,), np.uint8), hp.pinned_empty((n,), np.uint8)
pin_a[:] = 1
pin_b[:] = 2
page = np.ones(n, np.uint8)
reg = np.ones(n, np.uint8)
t0 = time.perf_counter()
ok = hp.pin_array(reg)
print(f'hipHostRegister of 1 GiB: {time.perf_counter() - t0:.3f} s, pinned={ok}')
s1, s2 = hp.Stream(), hp.Stream()


def rate(label, fn, reps=5, nbytes=n):
    fn()
    s1.synchronize(); s2.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    s1.synchronize(); s2.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f'{label:44s} {nbytes / dt / 1e9:7.2f} GB/s')


# Both directions at once FIRST, on the fresh streams: the copy engine of a stream is chosen at its
# first copy and kept; if the first copies of s1 and s2 do not overlap (as in the lines below, run
# in this order in rounds 3 and 4: 28.6 GB/s each way) the two streams share one engine for good.
rate('H2D + D2H pinned, first use of both streams together (each way)',
     lambda: (hip.check(L.bbt_memcpy_h2d(dev_a.ptr, pin_a.ctypes.data, n, s1.handle)),
              hip.check(L.bbt_memcpy_d2h(pin_b.ctypes.data, dev_b.ptr, n, s2.handle))))
s3, s4 = hp.Stream(), hp.Stream()
hip.check(L.bbt_memcpy_h2d(dev_a.ptr, pin_a.ctypes.data, n, s3.handle)); s3.synchronize()
hip.check(L.bbt_memcpy_d2h(pin_b.ctypes.data, dev_b.ptr, n, s4.handle)); s4.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    hip.check(L.bbt_memcpy_h2d(dev_a.ptr, pin_a.ctypes.data, n, s3.handle))
    hip.check(L.bbt_memcpy_d2h(pin_b.ctypes.data, dev_b.ptr, n, s4.handle))
s3.synchronize(); s4.synchronize()
print(f"{'H2D + D2H pinned, streams first used one after the other (each way)':44s} {5 * n / (time.perf_counter() - t0) / 1e9:7.2f} GB/s")
rate('H2D pinned (hipHostMalloc)', lambda: hip.check(L.bbt_memcpy_h2d(dev_a.ptr, pin_a.ctypes.data, n, s1.handle)))
rate('D2H pinned', lambda: hip.check(L.bbt_memcpy_d2h(pin_b.ctypes.data, dev_b.ptr, n, s2.handle)))
rate('H2D + D2H pinned, two streams (each way)', lambda: (hip.check(L.bbt_memcpy_h2d(dev_a.ptr, pin_a.ctypes.data, n, s1.handle)),
                                                          hip.check(L.bbt_memcpy_d2h(pin_b.ctypes.data, dev_b.ptr, n, s2.handle))))
rate('H2D registered (hipHostRegister)', lambda: hip.check(L.bbt_memcpy_h2d(dev_a.ptr, reg.ctypes.data, n, s1.handle)))
rate('H2D pageable', lambda: hip.check(L.bbt_memcpy_h2d(dev_a.ptr, page.ctypes.data, n, s1.handle)), reps=2)
rate('D2H pageable', lambda: hip.check(L.bbt_memcpy_d2h(page.ctypes.data, dev_b.ptr, n, s2.handle)), reps=2)
for mib in (64, 256, 512):
    mm = mib << 20
    rate(f'H2D + D2H pinned, two streams, {mib} MiB copies (each way)',
         lambda: (hip.check(L.bbt_memcpy_h2d(dev_a.ptr, pin_a.ctypes.data, mm, s1.handle)),
                  hip.check(L.bbt_memcpy_d2h(pin_b.ctypes.data, dev_b.ptr, mm, s2.handle))), nbytes=mm)
# in 64 MiB pieces (the pipeline's granularity is a run of blocks)
m = 64 << 20
rate('H2D pinned, 16 x 64 MiB', lambda: [hip.check(L.bbt_memcpy_h2d(dev_a.ptr + i * m, pin_a.ctypes.data + i * m, m, s1.handle)) for i in range(16)])
t0 = time.perf_counter(); page2 = page.copy(); print(f'numpy copy of 1 GiB: {n / (time.perf_counter() - t0) / 1e9:.2f} GB/s')
t0 = time.perf_counter(); pin_a[:] = page; print(f'numpy copy into pinned memory: {n / (time.perf_counter() - t0) / 1e9:.2f} GB/s')
