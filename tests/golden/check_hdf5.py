"""Read an HDF5 file written by baseband_tasks_amd.hdf5 with the REAL h5py and astropy YAML
loader -- the two calls the reference's reader makes (io/hdf5/header.py:62-69:
``yaml.load(fh['header'][()])``; io/hdf5/payload.py:121-141: the 'payload' dataset) -- and
print what they see as JSON.  Run by tests/test_hdf5.py in the build container's second
interpreter (the GPU box and the main interpreter have neither package):

    /opt/conda/bin/python3.9 -W ignore tests/golden/check_hdf5.py <file.h5>
"""
import hashlib
import json
import sys

import numpy as np

for _name, _fn in (('asscalar', lambda a: np.asarray(a).item()), ('alen', lambda a: len(np.asarray(a)))):
    if not hasattr(np, _name):
        setattr(np, _name, _fn)

import h5py                                  # noqa: E402
from astropy import units as u               # noqa: E402
from astropy.io.misc import yaml             # noqa: E402

with h5py.File(sys.argv[1], 'r') as fh:
    items = yaml.load(fh['header'][()])
    payload = fh['payload'][()]
out = dict(keys=sorted(items),
           dtype=str(items['dtype']), sample_shape=list(items['sample_shape']),
           samples_per_frame=int(items['samples_per_frame']),
           sample_rate_hz=float(items['sample_rate'].to_value(u.Hz)),
           time_isot=items['time'].isot, time_scale=items['time'].scale,
           payload_dtype=str(payload.dtype), payload_shape=list(payload.shape),
           payload_sha256=hashlib.sha256(np.ascontiguousarray(payload).tobytes()).hexdigest())
if 'frequency' in items:
    out['frequency_hz'] = np.atleast_1d(items['frequency'].to_value(u.Hz)).tolist()
if 'sideband' in items:
    out['sideband'] = np.atleast_1d(items['sideband']).astype(int).tolist()
if 'polarization' in items:
    out['polarization'] = [str(p) for p in np.atleast_1d(items['polarization'])]
print(json.dumps(out))
