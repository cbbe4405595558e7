"""Write tests/golden/reference_style.h5 the way the reference's HDF5 writer does -- the same two
calls, with the REAL h5py and astropy of the build container:

    fh = h5py.File(name, 'w')                                  (io/hdf5/base.py:217-218; default libver)
    fh.create_dataset('header', data=yaml.dump(dict(header)))  (io/hdf5/header.py:76-81)
    fh.create_dataset('payload', shape=(samples_per_frame,) + sample_shape, dtype=dtype)
                                                               (io/hdf5/payload.py:72-79), then filled

so that `baseband_tasks_amd.hdf5.open(name, 'r')` can be checked against a file of the layout the
reference itself produces (superblock 0, symbol-table root group, version-1 object headers, the
header as a variable-length string in the global heap, a contiguous payload).  The header values
are those `HDF5Header.fromvalues(stream)` collects (header.py:84-140): sample_shape,
samples_per_frame (the whole stream), sample_rate, time, dtype, frequency, sideband, polarization.

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_hdf5_fixture.py

Also writes reference_style.json: what the real h5py + astropy read back from it.
"""
import hashlib
import json
import os

import numpy as np

for _name, _fn in (('asscalar', lambda a: np.asarray(a).item()), ('alen', lambda a: len(np.asarray(a)))):
    if not hasattr(np, _name):
        setattr(np, _name, _fn)

import h5py                                  # noqa: E402
from astropy import units as u               # noqa: E402
from astropy.io.misc import yaml             # noqa: E402
from astropy.time import Time                # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2024)
n = 300
data = (rng.standard_normal((n, 2, 2)) * 10).astype(np.float32).view(np.complex64)[..., 0]      # (n, 2)
header = dict(sample_shape=(2,), samples_per_frame=n, sample_rate=16. * u.MHz,
              time=Time('2020-01-01T00:00:00', precision=9), dtype='c8',
              frequency=np.array([1000., 1016.]) * u.MHz, sideband=np.array([1, -1], dtype='i1'),
              polarization=np.array(['X', 'Y']))
name = os.path.join(here, 'reference_style.h5')
with h5py.File(name, 'w') as fh:
    fh.create_dataset('header', data=yaml.dump(header))
    words = fh.create_dataset('payload', shape=(n, 2), dtype='c8')
    words[:] = data
with h5py.File(name, 'r') as fh:
    items = yaml.load(fh['header'][()])
    payload = fh['payload'][()]
    text = fh['header'][()]
json.dump(dict(keys=sorted(items), header_text=text.decode() if isinstance(text, bytes) else text,
               payload_sha256=hashlib.sha256(np.ascontiguousarray(payload).tobytes()).hexdigest(),
               payload_shape=list(payload.shape), sample_rate_hz=float(items['sample_rate'].to_value(u.Hz)),
               time_isot=items['time'].isot, jd1=float(items['time'].jd1), jd2=float(items['time'].jd2),
               frequency_hz=items['frequency'].to_value(u.Hz).tolist(),
               sideband=items['sideband'].astype(int).tolist(),
               polarization=[str(p) for p in items['polarization']]),
          open(os.path.join(here, 'reference_style.json'), 'w'), indent=1)
print(os.path.getsize(name), 'bytes')
