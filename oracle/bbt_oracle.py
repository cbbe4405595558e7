"""CPU oracle for the dedispersion -> channelizer path.   TEST INFRASTRUCTURE.

A plain-numpy, function-style restatement of the algorithm the reference
(mhvk/baseband-tasks @ 2025-03-21, mounted at /root/reference when this was
written) uses on the hot path.  It is NOT part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / reported baseline.  The
shipped package (``baseband_tasks_amd``) never imports this module and has no
CPU fallback.

Pinning: every function below is checked in ``tests/test_oracle_golden.py``
against fixtures in ``tests/golden/*.npz`` that were produced by running the
real reference (``tests/golden/make_golden.py``, numpy 1.26.4 + astropy
4.3.1) in the build container, and against the known answers the reference's
own tests hold (tests/test_dm.py, tests/test_pfb.py GUPPI coefficients,
tests/test_base.py next_fast_len table).

The FFT itself is numpy's pocketfft (``numpy.fft``), exactly as in the
reference (baseband_tasks/fourier/numpy.py:33-49 calls ``np.fft.fft`` /
``np.fft.ifft`` and casts to the frequency dtype).  With ``fft64=True`` the
data are promoted to complex128 first, which reproduces what numpy < 2 does
internally (the environment the golden fixtures were generated in); with
``fft64=False`` numpy >= 2 transforms complex64 natively (this is what is
timed as the CPU baseline).

Units: frequencies in MHz, sample rates in Hz unless a name says otherwise,
times in seconds, DM in pc / cm^3.  All citations are relative to
/root/reference/baseband_tasks/.
"""
import numpy as np

# dm.py:37 -- "Constant hardcoded to match assumption made by tempo":
#   dispersion_delay_constant = u.s / 2.41e-4 * u.MHz**2 * u.cm**3 / u.pc
DISPERSION_DELAY_CONSTANT = 1. / 2.41e-4   # s MHz^2 cm^3 / pc


# --------------------------------------------------------------------------
# generators.py:154-190  Noise.__call__ / NoiseGenerator
def noise_frame(seed, frame_offset, samples_per_frame, sample_shape,
                dtype=np.complex64):
    """One frame of NoiseGenerator output.

    generators.py:178-190: the Philox counter word 1 is set to the sample
    offset of the frame start, then ``normal(size=shape)`` is drawn in
    float64 with the last axis doubled for complex data, viewed as
    complex128 and cast to the stream dtype.
    """
    dtype = np.dtype(dtype)
    bg = np.random.Philox(seed)
    state = bg.state
    state['state']['counter'][1] = frame_offset
    bg.state = state
    rng = np.random.Generator(bg)
    shape = (samples_per_frame,) + tuple(sample_shape)
    if dtype.kind == 'c':
        shape = shape[:-1] + (shape[-1] * 2,)
    numbers = rng.normal(size=shape)
    if dtype.kind == 'c':
        numbers = numbers.view(np.complex128)
    return numbers.astype(dtype, copy=False)


def noise_stream(seed, start, count, samples_per_frame, sample_shape,
                 dtype=np.complex64):
    """Samples [start, start+count) of a NoiseGenerator stream (base.py:389-438
    pull loop over generators.py:87-90 frames)."""
    out = np.empty((count,) + tuple(sample_shape), dtype)
    pos = start
    done = 0
    while done < count:
        frame_index, off = divmod(pos, samples_per_frame)
        frame = noise_frame(seed, frame_index * samples_per_frame,
                            samples_per_frame, sample_shape, dtype)
        n = min(count - done, samples_per_frame - off)
        out[done:done + n] = frame[off:off + n]
        done += n
        pos += n
    return out


# --------------------------------------------------------------------------
# dm.py:42-105
def time_delay(dm, freq_mhz, ref_freq_mhz=None):
    """dm.py:74-76:  d * (1/f^2 - 1/f_ref^2)  [s]."""
    d = DISPERSION_DELAY_CONSTANT * dm
    ref_inv2 = 0. if ref_freq_mhz is None else 1. / ref_freq_mhz ** 2
    return d * (1. / np.asanyarray(freq_mhz) ** 2 - ref_inv2)


def phase_delay(dm, freq_mhz, ref_freq_mhz=None):
    """dm.py:103-105:  d * f * (1/f_ref - 1/f)^2  [cycles].

    d is in s MHz^2, so d * f[MHz] * (1/MHz)^2 = s * MHz = 1e6 cycles.
    """
    d = DISPERSION_DELAY_CONSTANT * dm
    ref_inv = 0. if ref_freq_mhz is None else 1. / ref_freq_mhz
    f = np.asanyarray(freq_mhz)
    return d * f * (ref_inv - 1. / f) ** 2 * 1e6


# --------------------------------------------------------------------------
# fourier/numpy.py:99-126
def next_fast_len(n):
    """Smallest 2^a 3^b 5^c 7^d >= n (NumpyFFTMaker.next_fast_len)."""
    n = int(n)
    if n <= 7:
        return n
    best = 2 * n
    f2 = 1
    while f2 < best:
        f23 = f2
        while f23 < best:
            f235 = f23
            while f235 < best:
                f2357 = f235
                while f2357 < best:
                    if f2357 >= n:
                        best = f2357
                    f2357 *= 7
                f235 *= 5
            f23 *= 3
        f2 *= 2
    return best


# --------------------------------------------------------------------------
# base.py:743-795  PaddedTaskBase geometry
def padded_geometry(n_in, ih_samples_per_frame, pad_start, pad_end,
                    samples_per_frame=None, fast_len=None):
    """Returns dict(ih_spf, spf, n_out) as PaddedTaskBase.__init__ computes.

    base.py:750-760: ih_spf = spf + pad if spf given else
    max(ih.samples_per_frame, 4 * pad); then next_fast_len; spf = ih_spf - pad.
    base.py:767: n_out = n_in - pad.
    """
    pad = pad_start + pad_end
    if samples_per_frame is None:
        ih_spf = max(ih_samples_per_frame, 4 * pad)
    else:
        ih_spf = samples_per_frame + pad
    if fast_len is not None:
        ih_spf = fast_len(ih_spf)
    return dict(ih_spf=ih_spf, spf=ih_spf - pad, n_out=n_in - pad,
                pad_start=pad_start, pad_end=pad_end)


def padded_blocks(n_in, geo):
    """Block schedule of a padded task reading its whole output in order.

    base.py:775-795: frame m reads input [m*spf, m*spf + ih_spf); when that
    would run past the end it reads [n_in - ih_spf, n_in) instead and skips
    ``frame_offset`` output samples.  Yields (in_start, frame_offset,
    out_start, out_count).
    """
    ih_spf, spf, n_out = geo['ih_spf'], geo['spf'], geo['n_out']
    max_start = n_in - ih_spf
    m = 0
    while m * spf < n_out:
        ih_index = m * spf
        if ih_index > max_start:
            frame_offset = ih_index - max_start
            in_start = max_start
        else:
            frame_offset = 0
            in_start = ih_index
        out_count = min(spf - frame_offset, n_out - m * spf)
        yield in_start, frame_offset, m * spf, out_count
        m += 1


# --------------------------------------------------------------------------
# dispersion.py:48-129
def disperse_geometry(sample_rate_hz, frequency_mhz, sideband, dm,
                      complex_data=True, reference_frequency_mhz=None):
    """pad_start, pad_end, sample_offset, reference frequency
    (Disperse.__init__, dispersion.py:52-93).  ``dm`` is the dispersing DM:
    pass ``-dm`` for Dedisperse (dispersion.py:182-186)."""
    frequency = np.asanyarray(frequency_mhz, dtype=float)
    sideband = np.asanyarray(sideband)
    half_rate = sample_rate_hz / 1e6 / 2.
    if complex_data:
        freq_low = frequency - half_rate
        freq_high = frequency + half_rate
    else:
        freq_low = frequency + np.minimum(sideband, 0.) * half_rate
        freq_high = frequency + np.maximum(sideband, 0.) * half_rate
    if reference_frequency_mhz is None:
        reference_frequency_mhz = (freq_low + freq_high).mean() / 2.
    delay_low = time_delay(dm, freq_low, reference_frequency_mhz)
    delay_high = time_delay(dm, freq_high, reference_frequency_mhz)
    delay_max = max(np.max(delay_low), np.max(delay_high))
    delay_min = min(np.min(delay_low), np.min(delay_high))
    pad_start = int(np.ceil(delay_max * sample_rate_hz))
    pad_end = int(np.ceil(-delay_min * sample_rate_hz))
    if pad_start < 0:
        assert pad_end > 0
        sample_offset = pad_start
        pad_end += pad_start
        pad_start = 0
    elif pad_end < 0:
        sample_offset = -pad_end
        pad_start += pad_end
        pad_end = 0
    else:
        sample_offset = 0
    return dict(pad_start=pad_start, pad_end=pad_end,
                sample_offset=sample_offset,
                reference_frequency=reference_frequency_mhz)


def fft_frequency(n, sample_rate, ndim_after=0, real=False):
    """fourier/base.py:114-157: fftfreq/rfftfreq with trailing unit axes."""
    f = (np.fft.rfftfreq if real else np.fft.fftfreq)(n, d=1. / sample_rate)
    return f.reshape(f.shape + (1,) * ndim_after)


def chirp(n, sample_rate_hz, frequency_mhz, sideband, dm,
          reference_frequency_mhz, sample_offset=0, sample_ndim=1,
          dtype=np.complex64, real=False):
    """Disperse.phase_factor (dispersion.py:115-129), shape (n,)+broadcast.

    frequency = f0 + fftfreq * sideband; phase = phase_delay * sideband
    (+ sample_offset / fs * fftfreq cycles); exp(2 pi i phase) evaluated in
    float64 then cast to the frequency dtype.
    """
    frequency_mhz = np.asanyarray(frequency_mhz, dtype=float)
    sideband = np.asanyarray(sideband)
    ff = fft_frequency(n, sample_rate_hz / 1e6, sample_ndim, real=real)     # MHz
    freq = frequency_mhz + ff * sideband
    ph = phase_delay(dm, freq, reference_frequency_mhz)
    ph = ph * sideband
    if sample_offset != 0:
        ph = ph + (sample_offset / sample_rate_hz) * (ff * 1e6)
    return np.exp(ph * (2. * np.pi) * 1j).astype(dtype, copy=False)


# --------------------------------------------------------------------------
# fourier/numpy.py:33-39
def _fft(a, axis, fft64):
    if fft64:
        return np.fft.fft(a.astype(np.complex128, copy=False),
                          axis=axis).astype(a.dtype, copy=False)
    return np.fft.fft(a, axis=axis).astype(a.dtype, copy=False)


def _ifft(a, axis, fft64):
    if fft64:
        return np.fft.ifft(a.astype(np.complex128, copy=False),
                           axis=axis).astype(a.dtype, copy=False)
    return np.fft.ifft(a, axis=axis).astype(a.dtype, copy=False)


def _rfft(a, axis):
    """fourier/numpy.py:41-43 (real input; float64 inside, cast to complex64)."""
    return np.fft.rfft(a.astype(np.float64), axis=axis).astype(np.complex64)


def _irfft(a, n, axis):
    """fourier/numpy.py:46-49."""
    return np.fft.irfft(a.astype(np.complex128), n=n, axis=axis).astype(np.float32)


def disperse_block(x, phase_factor, pad_start, spf, fft64=True):
    """Disperse.task (dispersion.py:135-139); real streams go through
    rfft / irfft."""
    if x.dtype.kind == 'f':
        ft = _rfft(x, 0)
        ft *= phase_factor
        result = _irfft(ft, x.shape[0], 0)
    else:
        ft = _fft(x, 0, fft64)
        ft *= phase_factor
        result = _ifft(ft, 0, fft64)
    return result[pad_start:pad_start + spf]


def overlap_save(x, geo, block_task):
    """Whole-stream read of a padded task (base.py:389-438 + 775-795)."""
    n_in = x.shape[0]
    out = np.empty((geo['n_out'],) + x.shape[1:], x.dtype)
    for in_start, frame_offset, out_start, out_count in padded_blocks(n_in, geo):
        frame = block_task(x[in_start:in_start + geo['ih_spf']])
        out[out_start:out_start + out_count] = \
            frame[frame_offset:frame_offset + out_count]
    return out


def dedisperse(x, sample_rate_hz, frequency_mhz, sideband, dm,
               reference_frequency_mhz=None, samples_per_frame=None,
               ih_samples_per_frame=None, fast_len=next_fast_len, fft64=True):
    """Dedisperse(ih, dm).read() for an in-memory stream ``x`` of shape
    (n, *sample_shape) (dispersion.py:149-190).  Returns (y, info)."""
    if ih_samples_per_frame is None:
        ih_samples_per_frame = x.shape[0]
    g = disperse_geometry(sample_rate_hz, frequency_mhz, sideband, -dm,
                          complex_data=x.dtype.kind == 'c',
                          reference_frequency_mhz=reference_frequency_mhz)
    geo = padded_geometry(x.shape[0], ih_samples_per_frame, g['pad_start'],
                          g['pad_end'], samples_per_frame, fast_len)
    h = chirp(geo['ih_spf'], sample_rate_hz, frequency_mhz, sideband, -dm,
              g['reference_frequency'], g['sample_offset'],
              sample_ndim=x.ndim - 1, dtype=np.complex64, real=x.dtype.kind == 'f')
    y = overlap_save(x, geo, lambda blk: disperse_block(
        blk, h, geo['pad_start'], geo['spf'], fft64))
    info = dict(g)
    info.update(geo)
    # start_time shift in input samples (base.py:769-770, dispersion.py:96)
    info['start_shift_samples'] = g['pad_start'] + g['sample_offset']
    return y, info


# --------------------------------------------------------------------------
# channelize.py:50-74
def channelize(x, n, fft64=True):
    """Channelize(ih, n).read(): FFT over groups of n samples; a trailing
    partial group is dropped (base.py:684-687)."""
    nspec = x.shape[0] // n
    blocks = x[:nspec * n].reshape((nspec, n) + x.shape[1:])
    if x.dtype.kind == 'f':
        return _rfft(blocks, 1)
    return _fft(blocks, 1, fft64)


def dechannelize(z, fft64=True, n=None):
    """Dechannelize.task (channelize.py:164-165); pass ``n`` for real output."""
    r = _ifft(z, 1, fft64) if n is None else _irfft(z, n, 1)
    return r.reshape((-1,) + z.shape[2:])


def channel_frequency(n, sample_rate_hz, frequency_mhz, sideband, sample_ndim=1, real=False):
    """channelize.py:60-64: frequency + fft.frequency * sideband (MHz)."""
    ff = fft_frequency(n, sample_rate_hz / 1e6, sample_ndim, real=real)
    return np.asanyarray(frequency_mhz, float) + ff * np.asanyarray(sideband)


# --------------------------------------------------------------------------
# pfb.py:14-45, 72-100, 128-154
def sinc_hamming(n_tap, n_sample, sinc_scale=1.):
    """pfb.py:42-45."""
    n = n_tap * n_sample
    x = n_tap * sinc_scale * np.linspace(-0.5, 0.5, n, endpoint=False)
    return (np.sinc(x) * np.hamming(n)).reshape(n_tap, n_sample)


def pfb_geometry(n_in, ih_samples_per_frame, response_shape, samples_per_frame=None):
    """pfb.py:74-89: inner padded task with pad (n_tap-1)*n split evenly, no
    next_fast_len; then Channelize(padded, n, padded.spf // n)."""
    n_tap, n = response_shape
    pad = (n_tap - 1) * n
    assert pad % 2 == 0
    spf = None if samples_per_frame is None else samples_per_frame * n
    geo = padded_geometry(n_in, ih_samples_per_frame, pad // 2, pad // 2, spf, None)
    geo['n_chan'] = n
    geo['n_tap'] = n_tap
    geo['chan_spf'] = geo['spf'] // n
    return geo


def ppf_samples(data, response):
    """PolyphaseFilterBankSamples.ppf (pfb.py:91-100), the definition:
    result[i] = sum_t data_blk[i + t] * response[t]."""
    n_tap, n = response.shape
    blk = data.reshape((-1, n) + data.shape[1:])
    resp = response.reshape(response.shape + (1,) * (data.ndim - 1))
    nout = blk.shape[0] + 1 - n_tap
    result = np.empty((nout,) + blk.shape[1:], data.dtype)
    for i in range(nout):
        result[i] = (blk[i:i + n_tap] * resp).sum(0)
    return result.reshape((-1,) + result.shape[2:])


def ppf_fourier(data, response, fft64=True):
    """PolyphaseFilterBank.ppf (pfb.py:136-154): FFT along the block axis,
    multiply by conj(FFT(zero-padded response)), inverse, drop wrapped rows."""
    n_tap, n = response.shape
    blk = data.reshape((-1, n) + data.shape[1:])
    long_response = np.zeros(blk.shape[:2], data.dtype)
    long_response[:n_tap] = response
    long_response = long_response.reshape(long_response.shape + (1,) * (data.ndim - 1))
    if data.dtype.kind == 'f':
        ft = _rfft(blk, 0)
        ft *= _rfft(long_response, 0).conj()
        result = _irfft(ft, blk.shape[0], 0)
    else:
        ft_resp_conj = _fft(long_response, 0, fft64).conj()
        ft = _fft(blk, 0, fft64)
        ft *= ft_resp_conj
        result = _ifft(ft, 0, fft64)
    result = result[:result.shape[0] + 1 - n_tap]
    return result.reshape((-1,) + result.shape[2:])


def polyphase_filter_bank(x, response, ih_samples_per_frame=None,
                          samples_per_frame=None, fourier=True, fft64=True):
    """PolyphaseFilterBank(ih, response).read() for an in-memory stream.

    Inner padded stream (pfb.py:82-85) followed by Channelize with
    samples_per_frame = padded.spf // n (pfb.py:86-87); the TaskBase shape
    rule (base.py:684-687) drops a trailing partial channelizer frame."""
    if ih_samples_per_frame is None:
        ih_samples_per_frame = x.shape[0]
    geo = pfb_geometry(x.shape[0], ih_samples_per_frame, response.shape, samples_per_frame)
    ppf = (lambda d: ppf_fourier(d, response, fft64)) if fourier else \
        (lambda d: ppf_samples(d, response))
    padded = overlap_save(x, geo, ppf)
    n = geo['n_chan']
    frame = geo['chan_spf'] * n
    n_keep = (padded.shape[0] // frame) * frame
    return channelize(padded[:n_keep], n, fft64), geo


# --------------------------------------------------------------------------
# convolution.py:65-127, sampling.py:146-227, 308-312
def convolve_geometry(n_in, ih_samples_per_frame, n_response, offset=0,
                      samples_per_frame=None, fast_len=next_fast_len):
    """Convolve.__init__ (convolution.py:94-100)."""
    pad = n_response - 1
    return padded_geometry(n_in, ih_samples_per_frame, pad - offset, offset,
                           samples_per_frame, fast_len)


def convolve_block(x, ft_response, pad, fft64=True):
    """Convolve.task (convolution.py:116-120): keeps result[pad_start+pad_end:]."""
    ft = _fft(x, 0, fft64)
    ft *= ft_response
    result = _ifft(ft, 0, fft64)
    return result[pad:]


def ft_response(response, n, dtype, fft64=True):
    """Convolve._ft_response (convolution.py:108-114)."""
    long_response = np.zeros((n,) + response.shape[1:], dtype)
    long_response[:response.shape[0]] = response
    return _fft(long_response, 0, fft64)


def convolve(x, response, offset=0, samples_per_frame=None,
             ih_samples_per_frame=None, fast_len=next_fast_len, fft64=True):
    """Convolve(ih, response, offset=...).read()."""
    if ih_samples_per_frame is None:
        ih_samples_per_frame = x.shape[0]
    if response.ndim == 1 and x.ndim > 1:          # convolution.py:13-20
        response = response.reshape(response.shape[:1] + (1,) * (x.ndim - 1))
    geo = convolve_geometry(x.shape[0], ih_samples_per_frame, response.shape[0],
                            offset, samples_per_frame, fast_len)
    ftr = ft_response(response, geo['ih_spf'], x.dtype, fft64)
    pad = geo['pad_start'] + geo['pad_end']
    y = overlap_save(x, geo, lambda blk: convolve_block(blk, ftr, pad, fft64))
    return y, geo


def windowed_sinc(pad, sample_shift):
    """ShiftAndResample._windowed_sinc (sampling.py:177-193)."""
    sample_shift = np.asanyarray(sample_shift, dtype=float)
    ishift_max = int(round(float(sample_shift.max())))
    ishift_min = int(round(float(sample_shift.min())))
    n_result = 2 * pad + 1 + ishift_max - ishift_min
    result = np.zeros((n_result,) + sample_shift.shape)
    for shift, res in zip(sample_shift.ravel(), result.reshape(n_result, -1).T):
        ishift = int(round(shift.item()))
        x = np.arange(-pad, pad + 1) - (shift - ishift)
        res[ishift - ishift_min:ishift - ishift_max + n_result] = (
            np.sinc(x) * np.cos(np.pi * x / (2 * pad + 2)) ** 2)
    return result


def resample(x, offset, pad=64, samples_per_frame=None,
             ih_samples_per_frame=None, fast_len=next_fast_len, fft64=True):
    """Resample(ih, offset, pad=pad) (sampling.py:308-312 via 146-175):
    shift 0, grid through ``offset`` (float samples from stream start).

    Returns (y, info): info['d_time'] is the start-time change in samples,
    info['pointer'] the sample pointer left by the final seek.
    """
    if ih_samples_per_frame is None:
        ih_samples_per_frame = x.shape[0]
    shift_mean = 0.
    d_time = offset + np.around(shift_mean - offset)
    sample_shift = np.array(0. - d_time, ndmin=x.ndim - 1, dtype=float)
    response = windowed_sinc(pad, sample_shift)
    if samples_per_frame is None:
        samples_per_frame = max(ih_samples_per_frame, pad * 14)
    conv_offset = pad - int(round(float(sample_shift.min())))
    y, geo = convolve(x, response, conv_offset, samples_per_frame,
                      ih_samples_per_frame, fast_len, fft64)
    info = dict(geo)
    info['d_time'] = float(d_time)
    # start_time = ih.start + pad_start/fs + d_time/fs ; seek(ih.start + offset/fs)
    info['start_shift_samples'] = geo['pad_start'] + float(d_time)
    info['pointer'] = int(round(offset - info['start_shift_samples']))
    return y, info


# --------------------------------------------------------------------------
# functions.py:15-56, 59-143 ; integration.py:52-303 (integer step only)
def square(x):
    """Square.task (functions.py:15-16, 38-44): re^2 + im^2 for complex input
    (output real dtype), x^2 for real input."""
    if x.dtype.kind == 'c':
        return np.square(x.real) + np.square(x.imag)
    return np.square(x)


def power(x, axis=-1):
    """Power.task (functions.py:131-143): along the polarization axis (length
    2 -> 4): |X|^2, |Y|^2, Re(X Y*), Im(X Y*)."""
    xin = np.moveaxis(x, axis, 0)
    out = np.empty((4,) + xin.shape[1:], x.real.dtype)
    out[0] = square(xin[0])
    out[1] = square(xin[1])
    c = xin[0] * xin[1].conj()
    out[2] = c.real
    out[3] = c.imag
    return np.moveaxis(out, 0, axis)


def integrate(x, step, start=0, average=True):
    """Integrate(ih, step, start=start).read() for integer ``step``
    (integration.py:116-130, 252-303): bins of ``step`` consecutive samples
    from ``start``; a trailing partial bin is dropped; sums are accumulated in
    the input dtype, divided by the count if ``average``."""
    n_out = int((x.shape[0] - start) / step + 0.5 / step)
    seg = x[start:start + n_out * step].reshape((n_out, step) + x.shape[1:])
    acc = np.zeros((n_out,) + x.shape[1:], x.dtype)
    for k in range(step):            # sequential accumulation, as np.add.reduceat along axis 0
        acc += seg[:, k]
    if average:
        acc /= step
    return acc


# --------------------------------------------------------------------------
# sampling.py:380-425 ; dispersion.py:193-298
def shift_samples(x, shift):
    """ShiftSamples(ih, shift).read(): out[i, ...] = x[i + shift.max() - shift, ...]
    (sampling.py:407-425); returns (y, start shift in samples)."""
    shift = np.round(np.asanyarray(shift, dtype=float)).astype(int)
    full = np.broadcast_to(shift, x.shape[1:])
    n_out = x.shape[0] - int(np.ptp(shift))
    idx = np.ix_(np.arange(n_out), *[np.arange(s) for s in x.shape[1:]])
    return x[(shift.max() - full + idx[0],) + idx[1:]], int(shift.max())


def disperse_samples_shift(sample_rate_hz, frequency_mhz, sideband, dm, complex_data=True,
                           reference_frequency_mhz=None):
    """Integer shifts DisperseSamples applies (dispersion.py:229-246)."""
    frequency = np.asanyarray(frequency_mhz, dtype=float)
    if not complex_data:
        frequency = frequency + np.asanyarray(sideband) * sample_rate_hz / 1e6 / 2.
    if reference_frequency_mhz is None:
        reference_frequency_mhz = frequency.mean()
    delay = time_delay(dm, frequency, reference_frequency_mhz)
    return np.round(delay * sample_rate_hz).astype(int)


# --------------------------------------------------------------------------
# pfb.py:157-269
def inverse_pfb(z, response, sn, pad_start=128, pad_end=128, samples_per_frame=None,
                ih_samples_per_frame=1, fast_len=next_fast_len, fft64=True):
    """InversePolyphaseFilterBank(ih, response, sn, pad_start, pad_end).read()
    for spectra ``z`` of shape (n_spec, n, ...): dechannelize, then per frame
    FFT along the block axis, multiply by the Wiener inverse of the response,
    inverse FFT, flatten, drop the padding.  Returns (y, geometry)."""
    n_tap, n = response.shape
    x = dechannelize(z, fft64)
    pad_minimum = (n_tap - 1) * n
    assert pad_minimum % 2 == 0
    ps = pad_start * n + pad_minimum // 2
    pe = pad_end * n + pad_minimum // 2

    def nfl(m):                                   # pfb.py:227-232
        m = fast_len(m)
        res = m % n
        return m - res + n if res else m
    geo = padded_geometry(x.shape[0], ih_samples_per_frame * n, ps, pe, samples_per_frame, nfl)
    nblk = geo['ih_spf'] // n
    long_response = np.zeros((nblk, n), x.dtype)
    long_response[:n_tap] = response
    long_response = long_response.reshape(long_response.shape + (1,) * (x.ndim - 1))
    ft_response = _fft(long_response, 0, fft64).conj()
    inv_sn2 = 1. / (sn * sn)
    inverse = (ft_response.conj() / (ft_response.real ** 2 + ft_response.imag ** 2 + inv_sn2)
               * (1 + inv_sn2))

    def task(data):
        blk = data.reshape((nblk, n) + data.shape[1:])
        ft = _fft(blk, 0, fft64)
        ft *= inverse
        result = _ifft(ft, 0, fft64)
        result = result.reshape((-1,) + result.shape[2:])
        return result[ps:result.shape[0] - pe]
    geo['inverse_response'] = inverse
    return overlap_save(x, geo, task), geo


# --------------------------------------------------------------------------
# sampling.py:315-377
def time_delay_stream(x, delay_s, lo_hz, sideband):
    """TimeDelay.task: x * exp(-2 pi i delay lo sideband) (complex64 factor);
    the stream's start time moves by ``delay_s``."""
    factor = np.exp(-2j * np.pi * delay_s * lo_hz * np.asanyarray(sideband)).astype(x.dtype)
    return x * factor
