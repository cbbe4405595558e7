#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03l
mkdir -p $OUT
timeout -k 10 300 python3 - > $OUT/profile.txt 2>&1 <<'PY'
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
import baseband_tasks_amd as bt
from baseband_tasks_amd import host_pipeline as hp, hip
from baseband_tasks_amd import units as u
blocks = 96
n = (blocks - 1) * 836100 + 2**20
x = hp.pinned_empty((n, 2), np.complex64)
x[:] = 1
nh = bt.HostStream(x, '2020-01-01T00:00:00', 16 * u.MHz, samples_per_frame=2**20, frequency=1000 * u.MHz, sideband=1)
t = bt.Dedisperse(nh, 100.)
t.max_frames_per_call = 16
t.read(t.samples_per_frame)
for rep in range(2):
    t.invalidate_cache(); t.seek(0)
    t0 = time.perf_counter(); out = t.read(); print('read', time.perf_counter() - t0); del out
t.invalidate_cache(); t.seek(0)
pr = cProfile.Profile(); pr.enable(); out = t.read(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
# the device-resident equivalent for scale
ds = bt.DeviceStream(hip.DeviceArray.from_host(x), '2020-01-01T00:00:00', 16 * u.MHz, samples_per_frame=2**20, frequency=1000 * u.MHz, sideband=1)
td = bt.Dedisperse(ds, 100.); td.max_frames_per_call = 16
td.read_device(td.samples_per_frame)
for rep in range(2):
    td.invalidate_cache(); td.seek(0)
    t0 = time.perf_counter(); z = td.read_device(td.shape[0]); hip.synchronize(); print('read_device', time.perf_counter() - t0)
PY
cat $OUT/profile.txt | head -80
