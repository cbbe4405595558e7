import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

nh = bt.NoiseGenerator((8 * 2**20, 2), '2020-01-01T00:00:00', 16 * u.MHz, 2**20, seed=1,
                       frequency=1000 * u.MHz, sideband=1, polarization=['X', 'Y'])
ds = bt.DeviceStream(nh, nh.start_time, nh.sample_rate)
ch = bt.Channelize(bt.Dedisperse(ds, 100.), 1024, 512)
spectra = ch.read(1000)
print(spectra.shape, spectra.dtype)
on_device = ch.read_device(1000)
print(type(on_device).__name__, on_device.shape)
waterfall = bt.Integrate(bt.Power(ch), 16).read()
print(waterfall.shape, waterfall.dtype)
w = bt.hdf5.open('/tmp/readme_test.hdf5', 'w', template=ch)
ch.seek(0)
ch.read(512, out=w)
w.close()
r = bt.hdf5.open('/tmp/readme_test.hdf5', 'r')
print(r.shape, r.sample_rate, r.start_time)
assert np.array_equal(r.read(512), spectra[:512])
print('ok')
