"""how far is the ONE-transform-pair form of
Resample -> Dedisperse, H = FT(windowed sinc) . chirp on blocks padded 128 + 212476, from the
reference's two-stage result (sampling.py:211-220 through convolution.py:116-120, then
dispersion.py:135-139)?  CPU, float64 transforms on both sides, so what is measured is the
algorithm, not rounding.          python tools/k7_fused_error.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bbt_oracle as orc

FS, FC, DM, N = 16e6, 1000., 100., 2**20
rng = np.random.default_rng(5)
n_in = 4 * N
x = (rng.standard_normal((n_in, 1)) + 1j * rng.standard_normal((n_in, 1))).astype(np.complex128)

# the reference's two stages (config 5 geometry: SURVEY 8d)
rs, ri = orc.resample(x, 0.25, pad=64, samples_per_frame=N - 128, ih_samples_per_frame=N)
dd, di = orc.dedisperse(rs, FS, FC, 1, DM, samples_per_frame=N - 212476, ih_samples_per_frame=N)
shift = ri['pad_start'] + di['pad_start']                 # output sample j <-> input sample j + shift (+ the quarter sample)
print('two stages: resample pads', ri['pad_start'], ri['pad_end'], ' dedisperse pads', di['pad_start'], di['pad_end'],
      ' output', dd.shape, ' input sample of output 0:', shift)

# one transform pair per 2^20-sample block of the INPUT
d_time = ri['d_time']
resp = orc.windowed_sinc(64, np.array([0. - d_time]))[:, 0]        # 129 taps
conv_offset = 64 - int(round(0. - d_time))
g = orc.disperse_geometry(FS, FC, 1, -DM)
chirp = orc.chirp(N, FS, FC, 1, -DM, g['reference_frequency'], g['sample_offset'], sample_ndim=0, dtype=np.complex128)
long_resp = np.zeros(N, np.complex128)
long_resp[:resp.shape[0]] = resp
H = np.fft.fft(long_resp) * chirp
# (a convolution keeps result[n_response - 1:], convolution.py:116-120: the resampled sample i is the
# transform's output i + 128, whatever the response's offset; the dedisperser then keeps from its pad_start on)
pad_start = (resp.shape[0] - 1) + g['pad_start']
pad_end = g['pad_end']
valid = N - pad_start - pad_end
print('one pair : pads', pad_start, pad_end, ' valid', valid, 'of', N)
out = np.zeros(dd.shape[0], np.complex128)
filled = np.zeros(dd.shape[0], bool)
m = 0
while m * valid + N <= n_in:
    blk = x[m * valid:m * valid + N, 0]
    y = np.fft.ifft(np.fft.fft(blk) * H)[pad_start:pad_start + valid]
    j0 = m * valid                                          # index in the two-stage output
    lo, hi = max(j0, 0), min(j0 + valid, dd.shape[0])
    out[lo:hi] = y[lo - j0:hi - j0]
    filled[lo:hi] = True
    m += 1
ref = dd[:, 0][filled]
got = out[filled]
err = got - ref
rms = np.sqrt(np.mean(np.abs(ref)**2))
print(f'compared {filled.sum()} samples of {m} blocks: rel-L2 {np.linalg.norm(err) / np.linalg.norm(ref):.3e}   '
      f'max|delta| / rms {np.abs(err).max() / rms:.3e}   (tolerance of the path: 1e-6 / 1e-5)')
# where the difference sits: by position inside the one-pair blocks
pos = np.nonzero(filled)[0] % valid
for a, b in ((0, 1000), (1000, valid // 2), (valid // 2, valid - 1000), (valid - 1000, valid)):
    sel = (pos >= a) & (pos < b)
    print(f'  block positions [{a}, {b}): rms of the difference / rms {np.sqrt(np.mean(np.abs(err[sel])**2)) / rms:.3e}')
