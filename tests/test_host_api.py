"""CPU-only tests of the host layer: stream semantics, geometry and metadata
of the package's task classes (checked against vectors from the real
reference), the generator's bit-exactness, and that nothing computes on the
CPU when the GPU is absent."""
import hashlib

import numpy as np
import pytest

import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u
from baseband_tasks_amd.base import PaddedTaskBase, TaskBase
from baseband_tasks_amd.fourier import FFTMakerBase, HipFFTMaker, fft_maker
from oracle import bbt_oracle as orc

T0 = '2020-01-01T00:00:00'


class _NumpyLikeMaker(FFTMakerBase):
    """Engine stand-in whose fast lengths are the reference NumPy engine's, so
    the reference's (non power-of-two) block geometry can be reproduced."""
    next_fast_len = staticmethod(orc.next_fast_len)

    def __call__(self, *args, **kwargs):
        raise NotImplementedError


def noise(n, sample_shape, spf, seed=12345, fs=16 * u.MHz, **kw):
    return bt.NoiseGenerator((n,) + tuple(sample_shape), T0, fs, spf, seed=seed, **kw)


# --------------------------------------------------------------------------- units / time
def test_time_arithmetic():
    t = bt.Time('2010-11-12T13:14:15')
    assert (t + 0.5).isot == '2010-11-12T13:14:15.500000000'
    assert (t + 86400.25) - t == 86400.25
    assert (t - 1e-3).isot == '2010-11-12T13:14:14.999000000'
    assert bt.Time('2020-01-01T00:00:00.25') - bt.Time('2020-01-01') == 0.25
    assert t < t + 1e-9 and t == bt.Time(t)
    # sub-sample resolution over a day at 16 MHz
    big = t + 86399.0 + 3 / 16e6
    assert abs((big - t) - (86399.0 + 3 / 16e6)) < 1e-10
    assert 16 * u.MHz == 16e6 and u.to_hz(1.5e3) == 1500.


# --------------------------------------------------------------------------- generator
def test_noise_generator_matches_reference(golden):
    nh = noise(2 * 2**20, (2,), 2**20)
    f0 = nh.read(2**20)
    f1 = nh.read(2**20)
    assert np.array_equal(f0[:4], golden['noise_first'])
    assert hashlib.sha256(f0.tobytes()).hexdigest() == str(golden['noise_sha'][0])
    assert hashlib.sha256(f1.tobytes()).hexdigest() == str(golden['noise_sha'][1])
    nh.seek(2**20 - 3)
    assert np.array_equal(nh.read(6), golden['noise_straddle'])
    small = noise(1200, (3, 2), 500, seed=7, fs=1 * u.kHz)
    assert np.array_equal(small.read(), golden['noise_small'])
    # reading order does not matter (reference tests/test_generators.py:253-316)
    small.seek(700)
    late = small.read(300)
    small.seek(0)
    assert np.array_equal(small.read()[700:1000], late)


def test_stream_generator_and_task():
    def alternate(sh):
        return np.full((1,) + sh.shape[1:], sh.tell() % 2 == 1, sh.dtype)
    sh = bt.StreamGenerator(alternate, (10, 6), '2010-11-12', 10 * u.Hz)
    sh.seek(5)
    assert np.array_equal(sh.read().real[:, 0], [1, 0, 1, 0, 1])

    def fill(data):
        data[...] = 2 * (np.arange(data.shape[0]) % 2) - 1
        return data
    eh = bt.EmptyStreamGenerator((1000,), '2010-11-12', 1 * u.kHz, samples_per_frame=100, dtype='f4')
    th = bt.Task(eh, fill)
    th.seek(995)
    assert np.array_equal(th.read(), [1, -1, 1, -1, 1])
    # method-like task gets the task instance (offset correct)
    mh = bt.Task(eh, lambda task, data: np.full_like(data, task.tell()), method=True)
    mh.seek(250)
    assert mh.read(1)[0] == 200


# --------------------------------------------------------------------------- stream semantics (reference tests/test_base.py:168-372)
def test_seek_tell_read_semantics():
    nh = noise(10000, (2,), 1000, fs=1 * u.kHz, frequency=300 * u.MHz, sideband=np.array([1, -1]),
               polarization=['X', 'Y'])
    assert nh.shape == (10000, 2) and nh.sample_shape == (2,) and nh.size == 20000 and nh.ndim == 2
    assert nh.complex_data and nh.dtype == np.complex64 and nh.samples_per_frame == 1000
    assert nh.stop_time - nh.start_time == 10.
    assert nh.seek(10) == 10 and nh.seek(5, 1) == 15 and nh.seek(-10, 'end') == 9990
    assert nh.tell() == 9990 and nh.tell(u.s) == 9.99 and nh.tell('time') == nh.time
    assert nh.seek(nh.start_time + 1.2345) == 1234      # rounds to nearest sample
    with pytest.raises(ValueError):
        nh.seek(0, 3)
    nh.seek(9990)
    assert nh.read().shape == (10, 2)
    with pytest.raises(EOFError):
        nh.read(1)
    nh.seek(0)
    out = np.empty((1500, 2), np.complex64)
    assert nh.read(out=out) is out and nh.tell() == 1500
    with pytest.raises(AssertionError):
        nh.read(out=np.empty((5, 3), np.complex64))
    nh.seek(-5)
    with pytest.raises(OSError):
        nh.read(1)
    # metadata is simplified and broadcastable
    assert nh.frequency.shape == () and nh.sideband.shape == (2,) and nh.polarization.shape == (2,)
    assert nh.sideband.dtype == np.int8
    sl = nh[100:200]
    assert sl.shape == (100, 2) and sl.start_time - nh.start_time == 0.1
    nh.seek(100)
    want = nh.read(100)
    assert np.array_equal(sl.read(), want)          # (the slice shares nh's sample pointer)
    assert np.array_equal(np.array(sl), np.array(nh)[100:200])
    with nh as fh:
        pass
    assert fh.closed
    with pytest.raises(ValueError):
        nh.read(1)


def test_metadata_errors_and_set_attribute():
    with pytest.raises(ValueError):
        noise(100, (2,), 10, frequency=300 * u.MHz)            # sideband missing
    with pytest.raises(ValueError):
        noise(100, (2,), 10, frequency=np.arange(3.) * u.MHz, sideband=1)
    with pytest.raises(TypeError):
        noise(100, (2,), 10, wrong=1)
    nh = noise(100, (2,), 10)
    with pytest.raises(AttributeError):
        nh.frequency
    sa = bt.SetAttribute(nh, frequency=[1e9, 2e9], sideband=[1, 1], polarization=['L', 'R'])
    assert np.array_equal(sa.frequency, [1e9, 2e9]) and sa.sideband.shape == ()
    nh.seek(0)
    want = nh.read(20)
    assert np.array_equal(sa.read(20), want)
    shifted = bt.SetAttribute(nh, start_time=nh.start_time + 1.)
    assert shifted.start_time - nh.start_time == 1.
    with pytest.raises(TypeError):
        bt.Dedisperse(nh, 10.)                                  # no frequency known


class _HostPadded(PaddedTaskBase):
    """A user-written padded task that runs on the host (moving sum)."""

    def task(self, data):
        c = np.cumsum(np.concatenate([np.zeros_like(data[:1]), data]), axis=0)
        n = self._pad_start + self._pad_end + 1
        return c[n:] - c[:-n]


def test_padded_task_base_blocks_like_the_oracle():
    nh = noise(10007, (2,), 1000, seed=3, fs=1 * u.kHz)
    x = nh.read()
    for spf in (None, 500, 997):
        pt = _HostPadded(nh, 3, 2, samples_per_frame=spf)
        geo = orc.padded_geometry(10007, 1000, 3, 2, spf, None)
        assert (pt._ih_samples_per_frame, pt.samples_per_frame, pt.shape[0]) == \
            (geo['ih_spf'], geo['spf'], geo['n_out'])
        assert pt.start_time - nh.start_time == 3e-3
        want = orc.overlap_save(x, geo, pt.task)
        got = pt.read()
        assert np.allclose(got, want, atol=1e-4)
        pt.seek(-7, 2)
        assert np.allclose(pt.read(), want[-7:], atol=1e-4)      # re-aligned final block
    with pytest.raises(ValueError):
        _HostPadded(nh, -1, 0)
    with pytest.warns(UserWarning, match='inefficient'):
        _HostPadded(nh, 30, 30, samples_per_frame=10)
    # next_fast_len table of the hip engine
    # (the reference's rule, fourier/numpy.py:99-126; known answers tests/test_base.py:522-529 style)
    assert [HipFFTMaker.next_fast_len(n) for n in (1, 7, 11, 256, 257, 19324 + 6401, 2**20, 2**20 + 1,
                                                   2**20 + 128, 1000003)] == \
        [1, 7, 12, 256, 270, 25725, 2**20, 1049760, 1049760, 1000188]
    for n in list(range(1, 400)) + [5000, 65537, 10**6 + 7, 2**24 - 1]:
        assert HipFFTMaker.next_fast_len(n) == orc.next_fast_len(n)
    with pytest.raises(ValueError):
        HipFFTMaker.next_fast_len(2**26 + 1)
    # every block length the engine hands out can be planned: beyond 8192 points the library
    # splits a length into two factors of at most 8192 (split_7smooth in csrc/bbt_hip.hip); the
    # 52 products of 2, 3, 5, 7 up to 2^26 that have no such split are skipped
    smooth = sorted({2**a * 3**b * 5**c * 7**d for a in range(27) for b in range(17) for c in range(12)
                     for d in range(10) if 2**a * 3**b * 5**c * 7**d <= 2**26})

    def splits(n):
        return n <= 8192 or any(n % d == 0 and n // d <= 8192 for d in range(1, int(n**0.5) + 1))
    bad = [n for n in smooth if not splits(n)]
    assert len(smooth) == 3174 and len(bad) == 52 and bad[0] == 20588575 == 5**2 * 7**7
    for n in smooth:
        fast = HipFFTMaker.next_fast_len(n)
        assert splits(fast) and fast in smooth
        assert (fast == n) == (n not in bad)
        if n in bad:
            assert fast == min(m for m in smooth if m > n and splits(m))
    # the explicit power-of-two option (fast kernels, not the reference's geometry)
    pow2 = HipFFTMaker(power_of_two=True)
    assert [pow2.next_fast_len(n) for n in (1, 256, 257, 19324 + 6401, 2**20, 2**20 + 1)] == \
        [256, 256, 512, 32768, 2**20, 2**21]
    with pytest.raises(ValueError):
        pow2.next_fast_len(2**24 + 1)


# --------------------------------------------------------------------------- geometry of the GPU tasks vs the reference
def test_dm_matches_reference(golden):
    dm = bt.DispersionMeasure(29.1168)
    f = golden['dm_freqs'] * u.MHz
    np.testing.assert_allclose(dm.time_delay(f), golden['dm_time_delay_inf'], rtol=1e-14)
    np.testing.assert_allclose(dm.time_delay(f, 350 * u.MHz), golden['dm_time_delay_ref'], rtol=1e-13,
                               atol=1e-18)
    np.testing.assert_allclose(dm.phase_delay(f), golden['dm_phase_delay_inf'], rtol=1e-14)
    np.testing.assert_allclose(dm.phase_delay(f, 350 * u.MHz), golden['dm_phase_delay_ref'], rtol=1e-13)
    assert dm.dispersion_delay_constant == golden['dm_const'][0]
    assert -dm == -29.1168 and isinstance(-dm, bt.DispersionMeasure)
    # phase_factor is exp(2 pi i phase_delay)  (reference tests/test_dm.py:66-73)
    assert abs(dm.phase_factor(1400e6)[()] - np.exp(2j * np.pi * dm.phase_delay(1400e6))) < 1e-12


@pytest.mark.parametrize('fc,spf', [(1000., None), (800., 2**20 - 415021), (1400., None)])
def test_dedisperse_geometry_configs(golden, fc, spf):
    nh = noise(8 * 2**20, (2,), 2**20, frequency=fc * u.MHz, sideband=1)
    with fft_maker.set(_NumpyLikeMaker()):
        dd = bt.Dedisperse(nh, 100., samples_per_frame=spf)
    want = golden['geo_dd_fc%d' % fc]
    assert [dd._pad_start, dd._pad_end, dd._ih_samples_per_frame, dd.samples_per_frame,
            dd.shape[0], dd._sample_offset] == list(want)
    assert dd.reference_frequency == golden['reffreq_dd_fc%d' % fc][0] * 1e6
    assert abs((dd.start_time - nh.start_time) * 16e6 - golden['shift_dd_fc%d' % fc][0]) < 1e-3
    assert dd.dm == 100. and dd.sample_shape == (2,) and dd.dtype == np.complex64
    if fc == 1000.:
        # the hip engine's default picks the same 2^20 block
        assert bt.Dedisperse(nh, 100.)._ih_samples_per_frame == 2**20


REFS = [None, 300., 300.0123456789, 300.064, 299.936, 300.128, 300.123456789, 299.872]


@pytest.mark.parametrize('i', range(8))
def test_giant_pulse_geometry_and_chirp(golden, i):
    gp = bt.EmptyStreamGenerator((164000, 2), '2010-11-12T13:14:15', 128 * u.kHz,
                                 samples_per_frame=1000, frequency=300 * u.MHz,
                                 sideband=np.array((1, -1)))
    rf = None if REFS[i] is None else REFS[i] * u.MHz
    with fft_maker.set(_NumpyLikeMaker()):
        d = bt.Disperse(gp, golden['gp_dm'][0], reference_frequency=rf)
    want = golden['geo_gp_ref%d' % i]
    assert [d._pad_start, d._pad_end, d._ih_samples_per_frame, d.samples_per_frame, d.shape[0],
            d._sample_offset] == list(want)
    assert abs((d.start_time - gp.start_time) * 128e3 - golden['shift_gp_ref%d' % i][0]) < 1e-4  # astropy's own jd rounding is ~1e-6
    pf = d.phase_factor
    n = pf.shape[0]
    assert pf.shape == (n, 2) and pf.dtype == np.complex64
    assert np.abs(pf[[0, 1, 17, n // 2 - 1, n // 2, -1]] - golden['gp_chirp%d' % i]).max() < 3e-7
    # response columns handed to the C ABI: one per distinct sideband
    cols, index = d._response_columns()
    assert cols.shape == (2, n) and list(index) == [0, 1]
    assert np.array_equal(cols[1], pf[:, 1])


def test_config2_chirp_matches_reference(golden):
    nh = noise(4 * 2**20, (2,), 2**20, frequency=1000 * u.MHz, sideband=1)
    dd = bt.Dedisperse(nh, 100.)
    pf = dd.phase_factor
    assert pf.shape == (2**20, 1)
    assert np.abs(pf[golden['c2_chirp_idx'], 0] - golden['c2_chirp']).max() < 2e-7
    cols, index = dd._response_columns()
    assert cols.shape == (1, 2**20) and list(index) == [0, 0]


def test_channelize_pfb_resample_geometry(golden):
    nh = noise(2 * 2**20, (2,), 2**20, frequency=1000 * u.MHz, sideband=1)
    ch = bt.Channelize(nh, 1024, samples_per_frame=16)
    assert ch.shape == (2048, 1024, 2) and ch.sample_rate == 15625. and ch.samples_per_frame == 16
    assert ch.frequency.shape == (1024, 1)
    np.testing.assert_allclose(ch.frequency[[0, 1, 511, 512, 1023], 0] / 1e6, golden['c2ch_freq'],
                               rtol=1e-15)
    assert bt.Channelize(nh, 1000).shape == (2 * 2**20 // 1000, 1000, 2)   # any 2^a 3^b 5^c 7^d count
    assert bt.Channelize(nh, 64).shape == (2 * 2**20 // 64, 64, 2)     # short transforms are fine
    assert bt.Channelize(nh, 8192).shape == (256, 8192, 2)
    with pytest.raises(ValueError):
        bt.Channelize(nh, 1001)                                        # 7 x 11 x 13
    assert bt.Channelize(nh, 16384).shape == (128, 16384, 2)           # ... and 16384 (csrc/fft_big.hpp)
    with pytest.raises(ValueError):
        bt.Channelize(nh, 32768)
    with pytest.raises(TypeError):
        bt.Channelize(bt.EmptyStreamGenerator((4096,), T0, 1e3, dtype='f8'), 256)
    real = bt.Channelize(bt.EmptyStreamGenerator((4096,), T0, 1e3, dtype='f4'), 256)
    assert real.shape == (16, 129) and real.dtype == np.complex64       # rfft: n // 2 + 1 channels
    assert bt.Channelize(noise(2**20, (2,), 2**20), 1024).shape == (1024, 1024, 2)   # no metadata needed
    pfb = bt.PolyphaseFilterBank(nh, bt.sinc_hamming(12, 1024))
    assert [pfb.padded._pad_start, pfb.padded._pad_end, pfb.padded._ih_samples_per_frame,
            pfb.padded.samples_per_frame, pfb.samples_per_frame, pfb.shape[0]] == list(golden['c3_geo'])
    assert pfb.shape == tuple(golden['c3_shape'])
    assert abs((pfb.start_time - nh.start_time) * 16e6 - golden['c3_shift'][0]) < 1e-3
    np.testing.assert_allclose(bt.sinc_hamming(12, 64, 0.95), golden['sh_guppi'], rtol=1e-14, atol=1e-17)
    # config 5: Resample then Dedisperse, 8 streams
    nh8 = noise(8 * 2**20, (8,), 2**20, frequency=1000 * u.MHz, sideband=1)
    rs = bt.Resample(nh8, 0.25, pad=64, samples_per_frame=2**20 - 128)
    assert [rs._pad_start, rs._pad_end, rs._ih_samples_per_frame, rs.samples_per_frame,
            rs.shape[0]] == list(golden['c5_rs_geo'][:5])
    assert rs.tell() == golden['c5_rs_pointer'][0] == -64
    rs.seek(0)
    dd = bt.Dedisperse(rs, 100., samples_per_frame=2**20 - 212476)
    assert [dd._pad_start, dd._pad_end, dd._ih_samples_per_frame, dd.samples_per_frame,
            dd.shape[0]] == list(golden['c5_dd_geo'][:5])
    assert abs((dd.start_time - nh8.start_time) * 16e6 - golden['c5_dd_shift'][0]) < 1e-3
    # windowed sinc response equals the oracle's (sampling.py:177-193)
    np.testing.assert_allclose(rs._response[:, 0], orc.windowed_sinc(64, np.array([-0.25]))[:, 0],
                               rtol=1e-14)
    assert rs._ft_response.shape == (2**20, 1) and rs._ft_response.dtype == np.complex64


def test_engine_state():
    assert isinstance(fft_maker.get(), HipFFTMaker)
    other = _NumpyLikeMaker()
    with fft_maker.set(other):
        assert fft_maker.get() is other
    assert isinstance(fft_maker.get(), HipFFTMaker)
    fft_maker.set('hip')
    with pytest.raises(TypeError):
        fft_maker.set(object())
    with pytest.raises(ValueError):
        type('HipFFTMaker', (FFTMakerBase,), {})     # duplicate registration


# --------------------------------------------------------------------------- no CPU fallback
@pytest.mark.skipif(bt.hip.available(), reason="checks behaviour WITHOUT a GPU")
def test_product_path_fails_loudly_without_gpu():
    nh = noise(4 * 4096, (2,), 4096, fs=1 * u.MHz, frequency=300 * u.MHz, sideband=1)
    for task in (bt.Dedisperse(nh, 1.), bt.Channelize(nh, 256), bt.PolyphaseFilterBank(nh, bt.sinc_hamming(4, 256))):
        with pytest.raises((bt.hip.HipError, bt.hip.HipLibraryMissing)):
            task.read(1)


def test_package_never_imports_the_oracle():
    import os
    import re
    pkg = os.path.dirname(bt.__file__)
    for root, _, files in os.walk(pkg):
        for name in files:
            if name.endswith(('.py', '.hip', '.hpp', '.h')):
                with open(os.path.join(root, name)) as f:
                    text = f.read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, re.M), name
                assert 'bbt_oracle' not in text, name


# --------------------------------------------------------------------------- randomised geometry (hypothesis)
def test_block_descriptors_match_the_oracle_for_random_geometry():
    """The block schedule handed to the GPU (`_block_descriptors`: input start,
    absolute output sample, first kept block sample, kept count per block)
    restates base.py:775-795; compare with the oracle's `padded_blocks` over
    random stream lengths, paddings and frame sizes, for every frame range."""
    hypothesis = pytest.importorskip('hypothesis')
    from hypothesis import given, settings, strategies as st
    from baseband_tasks_amd.overlap_save import SpectralMultiplyTask

    class _Plain(SpectralMultiplyTask):
        _keep_from = 0

        def _spectral_response(self):
            return np.ones((self._ih_samples_per_frame, 1), np.complex64)

    @settings(max_examples=60, deadline=None)
    @given(pad_start=st.integers(0, 300), pad_end=st.integers(0, 300),
           extra=st.integers(0, 5000), n_blocks=st.integers(1, 6), data=st.data())
    def check(pad_start, pad_end, extra, n_blocks, data):
        pad = pad_start + pad_end
        ih_spf = 1024
        spf = ih_spf - pad
        n_in = n_blocks * spf + pad + (extra % spf)
        nh = bt.EmptyStreamGenerator((n_in, 2), T0, 1 * u.kHz, samples_per_frame=200)
        pt = _Plain(nh, pad_start, pad_end, samples_per_frame=spf)
        pt._keep_from = pad_start
        geo = orc.padded_geometry(n_in, 200, pad_start, pad_end, spf, HipFFTMaker.next_fast_len)
        assert (pt._ih_samples_per_frame, pt.samples_per_frame, pt.shape[0]) == \
            (geo['ih_spf'], geo['spf'], geo['n_out'])
        blocks = list(orc.padded_blocks(n_in, geo))
        first = data.draw(st.integers(0, len(blocks) - 1))
        last = data.draw(st.integers(first + 1, len(blocks)))
        in0, in_len, starts, out_abs, keep, counts = pt._block_descriptors(first, last)
        assert in0 == blocks[first][0] and in0 + in_len == blocks[last - 1][0] + geo['ih_spf']
        for k, (in_start, frame_offset, out_start, out_count) in enumerate(blocks[first:last]):
            assert (starts[k], out_abs[k], keep[k], counts[k]) == \
                (in_start, out_start, pad_start + frame_offset, out_count)

    check()


def test_integrate_and_channelize_shapes_for_random_sizes():
    hypothesis = pytest.importorskip('hypothesis')
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(n=st.integers(2000, 50000), step=st.integers(1, 400), start=st.integers(0, 1000),
           lg_chan=st.integers(1, 10), spf=st.integers(1, 9))
    def check(n, step, start, lg_chan, spf):
        nh = bt.EmptyStreamGenerator((n, 2), T0, 1 * u.kHz, samples_per_frame=100, dtype=np.float32)
        it = bt.Integrate(nh, step, start=start)
        want = orc.integrate(np.zeros((n, 2), np.float32), step, start=start)
        assert it.shape == want.shape and it.sample_rate == 1e3 / step
        assert abs((it.start_time - nh.start_time) - start / 1e3) < 1e-9
        n_chan = 1 << lg_chan
        if n >= n_chan * spf:
            nc = bt.EmptyStreamGenerator((n, 2), T0, 1 * u.kHz, samples_per_frame=100,
                                         frequency=300 * u.MHz, sideband=1)
            ch = bt.Channelize(nc, n_chan, samples_per_frame=spf)
            assert ch.shape == ((n // (n_chan * spf)) * spf, n_chan, 2)
            assert ch.sample_rate == 1e3 / n_chan

    check()


def test_single_precision_adapter():
    """float64 / complex128 streams (the reference accepts them; its PFB tests use
    them) enter the single-precision path through `SinglePrecision`."""
    nh = bt.NoiseGenerator((4000, 2), T0, 1e6, 1000, dtype=np.complex128, seed=3, frequency=300e6, sideband=1)
    with pytest.raises(TypeError, match='SinglePrecision'):
        bt.Channelize(nh, 64)
    sp = bt.SinglePrecision(nh)
    assert sp.dtype == np.complex64 and sp.shape == nh.shape and sp.sample_rate == nh.sample_rate
    assert sp.start_time == nh.start_time and np.all(sp.frequency == nh.frequency)
    sp.seek(500)
    got = sp.read(1700)
    nh.seek(500)
    assert got.dtype == np.complex64 and np.array_equal(got, nh.read(1700).astype(np.complex64))
    assert bt.Channelize(sp, 50).shape == (80, 50, 2)
    real = bt.NoiseGenerator((3000,), T0, 1e6, 1000, dtype=np.float64, seed=4)
    assert bt.SinglePrecision(real).dtype == np.float32
    assert bt.SinglePrecision(bt.SinglePrecision(real)).read(10).dtype == np.float32


def test_repr_lists_the_non_default_constructor_arguments():
    """The reference's repr (base.py:207-233, 580-599) and what its tests ask of it
    (test_channelize.py:117-126, test_convolution.py:87-93, test_sampling.py:165-169):
    class name, 'ih', then only the arguments that differ from their defaults --
    with None meaning "as the underlying stream" -- and the underlying stream's own
    repr after 'ih:'.  (No GPU needed: plans are made on first read.)"""
    nh = bt.NoiseGenerator((16384, 2), '2020-01-01T00:00:00', 1 * u.MHz, 1024, seed=3, frequency=300 * u.MHz,
                           sideband=np.array([1, -1]))
    ct = bt.Channelize(nh, 1024)
    r = repr(ct)
    assert r.startswith('Channelize(ih') and 'n=1024' in r and '\nih: NoiseGenerator(shape=(16384, 2)' in r
    dr = repr(bt.Dechannelize(ct, 1024))
    assert dr.startswith('Dechannelize(ih') and 'n=1024' in dr and '\nih: Channelize(ih' in dr
    cr = repr(bt.ConvolveSamples(nh, np.ones(3)))
    assert cr.startswith('ConvolveSamples(ih') and 'response=' in cr and 'offset=' not in cr
    assert 'samples_per_frame' in cr                      # differs from the input's
    assert 'offset=1' in repr(bt.Convolve(nh, np.ones(3), offset=1))
    rr = repr(bt.Resample(nh, 0.5, samples_per_frame=511))
    assert rr.startswith('Resample(ih') and 'offset=0.5' in rr
    assert repr(bt.Integrate(bt.Square(nh), 16)).startswith('Integrate(ih, step=16)\nih: Square(ih)')
    sr = repr(bt.SetAttribute(nh, frequency=1e9, sideband=1))
    assert 'frequency=1000000000.0' in sr and 'sideband=1' in sr and 'sample_rate' not in sr.split('\nih:')[0]
    assert repr(nh).startswith('NoiseGenerator(shape=(16384, 2),\n               start_time=2020-01-01T00:00:00')


@pytest.mark.parametrize('n,samples_per_frame,fast,ih_spf,spf', [
    (3, None, False, 20000, 19998), (3, None, True, 20000, 19998), (3, 128, False, 130, 128),
    (3, 128, True, 135, 133), (5, 128, True, 135, 131)])
def test_padded_frames_table_of_the_reference(n, samples_per_frame, fast, ih_spf, spf):
    """Reference tests/test_base.py:519-535: an n-sample box filter over a
    (40000, 8) stream with 20000-sample frames -- input and output frame sizes
    with and without rounding up to a fast FFT length (135 = 3^3 * 5) -- and
    511-518: shape, start time and values from either end."""
    data = np.random.default_rng(4).choice(np.array([-3., -1., 1., 3.], np.float32), size=(40000, 8))
    fh = bt.StreamGenerator(lambda f: data[f.tell():f.tell() + f.samples_per_frame], data.shape,
                            '2014-06-16T05:56:07', 32 * u.MHz, samples_per_frame=20000, dtype=np.float32)
    hat = _HostPadded(fh, n - 1, 0, samples_per_frame=samples_per_frame,
                      next_fast_len=orc.next_fast_len if fast else None)
    assert hat._ih_samples_per_frame == ih_spf and hat.samples_per_frame == spf
    assert hat.sample_rate == fh.sample_rate and hat.shape == (40000 - n + 1, 8)
    assert abs((hat.start_time - fh.start_time) - (n - 1) / fh.sample_rate) < 1e-9
    box = sum(data[k:40000 - (n - 1) + k] for k in range(n))
    assert np.array_equal(hat.read(10), box[:10])
    hat.seek(-10, 2)
    assert np.array_equal(hat.read(10), box[-10:])
    hat.close()
    assert hat.closed


class _FakeHip:
    """The HIP layer `host_pipeline` talks to, as a log (no GPU): device blocks are numbered as
    they are taken, events remember what had been logged when they were recorded."""

    def __init__(self):
        import threading
        self.log = []
        self.main = threading.get_ident()
        self.blocks = 0
        fake = self

        class Dev:
            def __init__(self, shape, dtype):
                import threading
                fake.blocks += 1
                self.id, self.shape = fake.blocks, tuple(shape)
                self.ptr = 0x1000 * self.id
                fake.log.append(('take', self.id, threading.get_ident() == fake.main))

            def __getitem__(self, item):                 # (a view: same block)
                start, stop, _ = item.indices(self.shape[0])
                view = object.__new__(Dev)
                view.id, view.ptr, view.shape = self.id, self.ptr, (stop - start,) + self.shape[1:]
                return view
        self.DeviceArray = Dev

    # -- what host_pipeline calls
    def get_device(self):
        return 0

    def set_device(self, index):
        pass

    def get_stream(self):
        return None

    def check(self, rc):
        assert rc == 0

    def lib(self):
        return self

    def bbt_stream_create(self, ref):
        return 0

    def bbt_stream_destroy(self, h):
        return 0

    def bbt_event_create_ordering(self, ref):
        return 0

    def bbt_event_create(self, ref):
        self.log.append(('host event',))
        return 0

    def bbt_event_destroy(self, h):
        return 0

    def bbt_event_record(self, h, stream):
        import threading
        self.log.append(('record', threading.get_ident() == self.main))
        return 0

    def bbt_stream_wait_event(self, stream, h):
        return 0

    def bbt_event_sync(self, h):
        return 0

    def bbt_memcpy_h2d(self, dst, src, n, stream):
        self.log.append(('h2d', dst // 0x1000))
        return 0


def test_host_uploader_takes_the_block_before_the_ordering_event_and_loads_each_run_once(monkeypatch):
    """`HostUploader` without a GPU: (i) the device block of an upload is taken on the CALLING
    thread and the event its upload stream waits for is recorded after that -- a block the caller
    frees later (the input of the run it has just queued) can then never be the one the worker
    uploads into (VERDICT r03 weak point 6; the pool orders reuse by the caller's stream only);
    (ii) with the read-ahead announced run by run, as `DeviceTaskMixin._read_pipelined` does,
    every run is loaded exactly once and the load of run m + 1 is started inside the fetch of
    run m (the round-3 order loaded every run but the first twice).
    Reference contract: Base.read returns a fresh array per read, base.py:416."""
    from baseband_tasks_amd import host_pipeline as hp
    fake = _FakeHip()
    monkeypatch.setattr(hp, 'hip', fake)
    monkeypatch.setattr(hp, 'is_pinned', lambda a: True)
    monkeypatch.setattr(hp, '_copy_pair', (hp.Stream(), hp.Stream()))      # (no priming copies without a GPU)
    data = np.arange(64 * 2, dtype=np.float32).view(np.complex64).reshape(64, 1)

    class Src:
        shape, dtype = data.shape, data.dtype

        def host_view(self, start, count):
            return data[start:start + count]
    src = Src()
    up = hp.HostUploader(src)
    runs = [(0, 16), (16, 16), (32, 16), (48, 16)]
    got = []
    for i, run in enumerate(runs):
        if i + 1 < len(runs):
            up.prefetch(*runs[i + 1])
        dev = up.fetch(*run)
        got.append(dev.id)
        # (the caller queues its kernels here; the next run's load is already in flight)
        assert (up._pending is not None) == (i + 1 < len(runs))
        if up._pending is not None:
            assert up._pending[:2] == runs[i + 1]
    up.close()
    assert up.loads == len(runs) and got == [1, 2, 3, 4]
    takes = [e for e in fake.log if e[0] == 'take']
    assert all(on_main for _, _, on_main in takes)
    # every block is taken before the `after` event of its upload is recorded (main thread), and
    # the copy into it comes after both
    for block in got:
        i_take = fake.log.index(('take', block, True))
        i_copy = fake.log.index(('h2d', block))
        records_on_main = [k for k, e in enumerate(fake.log) if e == ('record', True) and i_take < k < i_copy]
        assert records_on_main, fake.log
    # a fetch nobody announced, and an announced range nobody fetches: still one load per fetch
    up = hp.HostUploader(src)
    up.prefetch(16, 16)
    assert up.fetch(0, 8).shape == (8, 1)              # loads [0, 8), then starts [16, 32)
    assert up.fetch(32, 8).shape == (8, 1)             # the read-ahead is dropped, [32, 40) loaded
    assert up.loads == 3
    up.close()
    # a task that keeps the frame its last run ended in asks for less than its reader announced:
    # the fetch takes the part it wants of the pending load -- no second load
    up = hp.HostUploader(src)
    up.prefetch(16, 24)
    up.fetch(0, 16)
    part = up.fetch(20, 20)                            # [20, 40) of the announced [16, 40)
    assert part.shape == (20, 1) and up.loads == 2
    up.prefetch(40, 8)
    up.fetch(30, 4)                                    # not announced: loaded; then [40, 48) starts
    assert up.fetch(41, 8).shape == (8, 1)             # reaches past the pending load: loaded anew
    assert up.loads == 5
    up.close()
    # a stream without pinned memory goes through the two staging buffers, which the HOST waits
    # for before it refills them: those uploads carry default events, the others ordering events
    monkeypatch.setattr(hp, 'pinned_empty', lambda shape, dtype: np.empty(shape, dtype))

    class Plain:
        shape, dtype = data.shape, data.dtype

        def seek(self, pos):
            self.pos = pos

        def read(self, count, out=None):
            out[...] = data[self.pos:self.pos + count]
            return out
    fake.log.clear()
    plain = Plain()                                     # (the uploader holds its stream weakly)
    up = hp.HostUploader(plain)
    for run in runs:
        up.fetch(*run)
    up.close()
    assert fake.log.count(('host event',)) == len(runs)


class _FakeLib:
    """libbbt_hip.so as a log: every entry point returns 0, allocations hand out numbered
    addresses, plan info says 'two levels' (a plan with lanes)."""

    def __init__(self):
        self.log = []
        self.next_ptr = 0x100000
        self.events = 0
        self.finished = set()                            # events `bbt_event_query` reports as done

    def __getattr__(self, name):
        def call(*args):
            if name == 'bbt_malloc':
                args[0]._obj.value = self.next_ptr
                self.next_ptr += 0x100000
            elif name == 'bbt_event_create_ordering':
                self.events += 1
                args[0]._obj.value = 0xE000 + self.events
            elif name == 'bbt_osm_plan_create':
                args[0]._obj.value = 0xABC0
            elif name == 'bbt_event_query':              # (not logged: it queues nothing)
                args[1]._obj.value = int(getattr(args[0], 'value', args[0]) in self.finished)
                return 0
            elif name == 'bbt_osm_plan_info':
                args[3]._obj.value = 256                 # n1
                args[4]._obj.value = 4096
                return 0
            def plain(a):
                return getattr(a, 'value', a)
            self.log.append((name,) + tuple(plain(a) for a in args if isinstance(a, (int, type(None))) or hasattr(a, 'value')))
            return 0
        return call


def test_deferred_plan_calls_leave_their_event_with_input_and_output(monkeypatch):
    """`hip.OsmPlan._call` without a GPU (a logging stand-in for the library): a plan call whose
    output the package owns is issued with `bbt_osm_plan_defer`; its completion event stays with the
    output's AND the input's allocation; whatever asks for either address next first queues the
    wait (`DeviceArray.ptr`), once per allocation, and the event returns to the pool only after
    both have; a task's cache alternates between two buffers while a call is owed; a foreign
    output (no `_Allocation`) is never deferred.  Reference: consecutive frame reads,
    base.py:427-436."""
    from baseband_tasks_amd import hip
    from baseband_tasks_amd.device_task import DeviceTaskMixin
    fake = _FakeLib()
    monkeypatch.setattr(hip, '_lib', fake)
    monkeypatch.setattr(hip, 'DEFER_JOIN', True)
    monkeypatch.setattr(hip, '_events', hip._EventPool())
    plan = hip.OsmPlan(2**20, 2, np.zeros((1, 2**20), np.complex64))
    x = hip.DeviceArray((2**20, 2), np.complex64)
    y = hip.DeviceArray((1000, 2), np.complex64)
    desc = ([0], [0], [10], [1000])
    plan.execute(x, y, *desc)
    names = [e[0] for e in fake.log]
    assert names.index('bbt_osm_plan_defer') < names.index('bbt_osm_execute')
    ev = [e for e in fake.log if e[0] == 'bbt_osm_plan_defer'][0][2]
    assert y.pending and x.pending and y.owner.writes[0].done is x.owner.reads[0].done
    assert not y.owner.reads and not x.owner.writes                 # (written / read, not the reverse)
    assert y.owner.writes[0].done.refs == 2 and not hip._events._idle
    # views share the allocation's state; reading the address of one queues the wait, once
    view = y[10:20].reshape(20)
    assert view.pending
    n0 = len(fake.log)
    assert view.ptr == y._ptr + 10 * 16
    assert fake.log[n0:] == [('bbt_stream_wait_event', None, ev)] and not y.pending and x.pending
    assert y.ptr and len(fake.log) == n0 + 1            # (settled: no second wait)
    # the next call reads the same input: readers share it -- no wait for the first call -- and
    # takes a second event; whoever WRITES the input next waits for both readers
    n0 = len(fake.log)
    plan.execute(x, y, *desc)
    after = [e for e in fake.log[n0:] if e[0] in ('bbt_stream_wait_event', 'bbt_osm_plan_defer', 'bbt_osm_execute')]
    assert [e[0] for e in after] == ['bbt_osm_plan_defer', 'bbt_osm_execute'] and after[0][2] != ev
    assert len(x.owner.reads) == 2
    # uses that only READ the input of a deferred call -- a copy out of it (the frame a sequential
    # reader keeps from its last run, `_ensure_frames`), a plain kernel on it, a download -- go
    # beside the deferred reader: no wait (found with tools/chain_timeline.py: the last run of
    # Dedisperse(Resample(x)) started after the run before it had ended, the copy had waited for
    # that run's lanes) ...
    x2 = hip.DeviceArray((2**20, 2), np.complex64)
    y2 = hip.DeviceArray((1000, 2), np.complex64)
    plan.execute(x2, y2, *desc)
    n0 = len(fake.log)
    keep = hip.DeviceArray((10, 2), np.complex64)
    keep.copy_from_device(x2[:10])
    hip.scale_streams(x2[:10], keep, 10, 2, keep)
    x2[:10].to_host()
    assert 'bbt_stream_wait_event' not in [e[0] for e in fake.log[n0:]] and len(x2.owner.reads) == 1
    # ... while a reader of the OUTPUT of a deferred call waits for its writer
    n0 = len(fake.log)
    keep.copy_from_device(y2[:10])
    assert [e[0] for e in fake.log[n0:]] == ['bbt_stream_wait_event', 'bbt_memcpy_d2d'] and not y2.owner.writes
    del x2, y2, keep
    n0 = len(fake.log)
    x.fill_bytes(0)
    waits = [e for e in fake.log[n0:] if e[0] == 'bbt_stream_wait_event']
    assert len(waits) == 2 and not x.pending and fake.log[-1][0] == 'bbt_memset'
    # runs of one big read: disjoint slices of a fresh array, marked `fresh`, do not wait for
    # each other; the reader of the whole waits for all of them
    big = hip.DeviceArray((3000, 2), np.complex64)
    n0 = len(fake.log)
    for k in range(3):
        piece = big[1000 * k:1000 * (k + 1)]
        piece.fresh = True
        plan.execute(x, piece, *desc)
    assert 'bbt_stream_wait_event' not in [e[0] for e in fake.log[n0:]] and len(big.owner.writes) == 3
    n0 = len(fake.log)
    assert big.ptr and [e[0] for e in fake.log[n0:]] == ['bbt_stream_wait_event'] * 3
    # ... and without the mark a second writer of the same allocation does wait
    plan.execute(x, big[:1000], *desc)
    n0 = len(fake.log)
    plan.execute(x, big[1000:2000], *desc)
    assert fake.log[n0][0] == 'bbt_stream_wait_event'
    del big
    # a block that is freed while a call is owed is ordered first
    n0 = len(fake.log)
    y_ptr = y._ptr
    del view, y
    tail = [e[0] for e in fake.log[n0:]]
    assert tail == ['bbt_stream_wait_event', 'bbt_free'] and fake.log[-1][1] == y_ptr
    # foreign memory as output: a joined call
    class Foreign:
        pass
    z = hip.DeviceArray((1000, 2), np.complex64, ptr=0x7000000, owner=Foreign())
    n0 = len(fake.log)
    plan.execute(x, z, *desc)
    assert 'bbt_osm_plan_defer' not in [e[0] for e in fake.log[n0:]] and not z.pending

    # a foreign INPUT (a torch tensor): the call is deferred, and the event is kept for
    # `hip.wait_for_readers`, which whoever refills the tensor in place calls first
    xin = hip.DeviceArray((2**20, 2), np.complex64, ptr=0x9000000, owner=Foreign())
    y2 = hip.DeviceArray((1000, 2), np.complex64)
    n0 = len(fake.log)
    plan.execute(xin, y2, *desc)
    ev2 = [e for e in fake.log[n0:] if e[0] == 'bbt_osm_plan_defer'][0][2]
    assert y2.owner.writes[0].done.refs == 2
    n0 = len(fake.log)
    hip.wait_for_readers(xin.owner)
    assert fake.log[n0:] == [('bbt_stream_wait_event', None, ev2)]
    hip.wait_for_readers(xin)                            # (settled: nothing more to wait for)
    assert len(fake.log) == n0 + 1 and y2.owner.writes[0].done.refs == 1
    key = id(xin.owner)
    del xin
    assert key in hip._foreign_reads                     # (the owed output keeps its input alive)
    del y2
    assert key not in hip._foreign_reads                 # (the registry does not outlive the tensor)

    # a task's cache alternates while the previous run is owed, and stays put once it is not
    class Task(DeviceTaskMixin):
        sample_shape = (2,)
        dtype = np.dtype(np.complex64)
    t = Task()
    a = t._out_buffer(100)
    a.owner.owe(hip._Pending(hip._Done(hip._events.take(), 1), ()), write=True)
    b = t._out_buffer(100)
    assert b.owner is not a.owner and t._cache_buffer_b.owner is a.owner
    assert t._out_buffer(100).owner is b.owner           # (nothing owed on b: no switch)
    b.owner.owe(hip._Pending(hip._Done(hip._events.take(), 1), ()), write=True)
    c = t._out_buffer(100)
    assert c.owner is a.owner                            # back to the first one ...
    n0 = len(fake.log)
    assert c.ptr and fake.log[n0][0] == 'bbt_stream_wait_event'      # ... after the call before last


def test_finished_readers_let_go_of_their_plans(monkeypatch):
    """Deferred calls that READ an input leave their completion event -- and a reference to their
    plan -- with the input; nobody ever waits for a reader of a long-lived input (a tensor that
    many tasks read), so the entries of calls that have FINISHED are dropped when the next one is
    noted (`hip._prune`, `bbt_event_query`): a plan whose task was dropped is destroyed then, not
    64 calls later in the middle of somebody's read (tools/chain_host_probe.py: 2.5-3 ms blocked
    in `bbt_osm_plan_destroy`, twice per read).  No GPU: the logging stand-in for the library."""
    from baseband_tasks_amd import hip
    fake = _FakeLib()
    monkeypatch.setattr(hip, '_lib', fake)
    monkeypatch.setattr(hip, 'DEFER_JOIN', True)
    monkeypatch.setattr(hip, '_events', hip._EventPool())
    desc = ([0], [0], [10], [1000])

    class Tensor:                                        # a foreign owner (weakly referable)
        pass
    tensor = Tensor()
    for x in (hip.DeviceArray((2**20, 2), np.complex64),                               # the package's own memory
              hip.DeviceArray((2**20, 2), np.complex64, ptr=0x7000000, owner=tensor)):  # somebody else's
        readers = (lambda: x.owner.reads) if x.owner is not tensor else (lambda: hip._foreign_reads[id(tensor)][1])
        old_plan = hip.OsmPlan(2**20, 2, np.zeros((1, 2**20), np.complex64))
        y = hip.DeviceArray((1000, 2), np.complex64)
        old_plan.execute(x, y, *desc)
        old_plan.execute(x, y, *desc)
        events = [e[2] for e in fake.log if e[0] == 'bbt_osm_plan_defer'][-2:]
        assert y.ptr                                     # (the output was consumed; the input's side stays)
        del old_plan, y                                  # the task is dropped: only the readers' entries hold the plan
        assert len(readers()) == 2 and 'bbt_osm_plan_destroy' not in [e[0] for e in fake.log]
        plan = hip.OsmPlan(2**20, 2, np.zeros((1, 2**20), np.complex64))
        y = hip.DeviceArray((1000, 2), np.complex64)
        plan.execute(x, y, *desc)                        # nothing has finished yet: all three are kept
        assert len(readers()) == 3
        fake.finished.update(events)
        n0 = len(fake.log)
        plan.execute(x, y, *desc)
        assert len(readers()) == 2 and [e[0] for e in fake.log[n0:]].count('bbt_osm_plan_destroy') == 1
        assert 'bbt_stream_wait_event' not in [e[0] for e in fake.log[n0:] if e[0] != 'bbt_stream_wait_event' or e[2] in events]
        fake.log.clear()
        fake.finished.clear()                            # (the pool hands the same events out again)


def test_pipelined_read_after_an_unconsumed_device_read_keeps_two_buffers(monkeypatch):
    """`read_device` whose result nobody touched leaves the cache buffer owed a deferred
    call; a pipelined `read` entered in that state must still alternate between TWO allocations
    (`_out_buffer`'s own swap had made runs 0 and 1 share one, run 1 computing into it while run 0
    was still going down).  No GPU: the logging stand-in for the library.  Reference: a reader that
    seeks and reads again, base.py:389-438."""
    from baseband_tasks_amd import hip, host_pipeline as hp
    from baseband_tasks_amd.device_task import DeviceTaskMixin
    fake = _FakeLib()
    monkeypatch.setattr(hip, '_lib', fake)
    monkeypatch.setattr(hip, '_events', hip._EventPool())
    monkeypatch.setattr(hp, '_copy_pair', (hp.Stream(), hp.Stream()))
    spf = 64
    computed = []

    class Task(DeviceTaskMixin):
        sample_shape = (2,)
        dtype = np.dtype(np.complex64)
        samples_per_frame = spf
        shape = (spf * 8, 2)
        offset = 0
        host_frames_per_run = 2

        def _compute_frames(self, first, last, out):
            # a deferred plan call: the output's allocation is owed its completion event
            out.owner.owe(hip._Pending(hip._Done(hip._events.take(), 1), ()), write=True)
            computed.append((first, last, out.owner))

        def _input_span(self, first, last):
            return None
    t = Task()
    # state on entry: buffer A owed (an untouched read_device), buffer B present and idle
    t._cache_buffer_b = hip.DeviceArray((spf * 2 * 2,), np.complex64)
    b_alloc = t._cache_buffer_b.owner
    t._ensure_frames(0, 2)
    a_alloc = t._cache_buffer.owner
    assert a_alloc is not b_alloc and t._cache_buffer.pending and not t._cache_buffer_b.pending
    t.invalidate_cache()
    del computed[:]
    out = np.empty((spf * 8, 2), np.complex64)
    t._read_pipelined(spf * 8, out)
    owners = [o for _, _, o in computed]
    assert [c[:2] for c in computed] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    for k in range(3):
        assert owners[k] is not owners[k + 1], 'consecutive runs share a buffer'
    assert owners[0] is owners[2] and owners[1] is owners[3]
    assert {id(o) for o in owners} == {id(a_alloc), id(b_alloc)}
    # every download is queued after the run's own compute and before the buffer's next turn
    names = [e[0] for e in fake.log]
    assert names.count('bbt_memcpy_d2h') == 4


def test_big_reads_are_cut_into_whole_frame_runs_written_in_place(monkeypatch):
    """`DeviceTaskMixin.read_device` for more frames than one cache holds (no GPU: a task that logs):
    whole frames are computed straight into their slice of the result (marked `fresh`: deferred
    plan calls on disjoint slices do not wait for each other), only a frame the request starts or
    ends inside goes through the cache and a copy; runs are `max_frames_per_call` long -- or, when
    the task reads straight from a stream that is resident in HBM, as long as the request."""
    import baseband_tasks_amd.device_task as dt
    log = []

    class Arr:
        fresh = False

        def __init__(self, n, base=0):
            self.n, self.base = n, base

        def __getitem__(self, sl):
            return Arr(sl.stop - sl.start, self.base + sl.start)

        def copy_from_device(self, other):
            log.append(('copy', self.base, self.n, other.base))

    class Source:
        _resident = False

    class T(dt.DeviceTaskMixin):
        samples_per_frame, shape, sample_shape, dtype = 10, (95,), (), np.dtype('c8')
        max_frames_per_call = 3

        def __init__(self, offset):
            self.offset = offset

        def _prepare_read(self, count, out):
            return count

        def _compute_frames(self, f0, f1, out):
            log.append(('direct', f0, f1, out.base, out.n, out.fresh))

        def _ensure_frames(self, f0, f1):
            log.append(('cache', f0, f1))
            return Arr(100, f0 * 10), f0 * 10

        def _input_span(self, a, b):
            return Source, 0, 0
    monkeypatch.setattr(dt, 'DeviceArray', lambda shape, dtype: Arr(shape[0]))
    cases = {
        (False, 0, 95): [('direct', 0, 3, 0, 30, True), ('direct', 3, 6, 30, 30, True), ('direct', 6, 9, 60, 30, True),
                         ('direct', 9, 10, 90, 5, True)],
        (False, 3, 80): [('cache', 0, 1), ('copy', 0, 7, 3), ('direct', 1, 4, 7, 30, True), ('direct', 4, 7, 37, 30, True),
                         ('direct', 7, 8, 67, 10, True), ('cache', 8, 9), ('copy', 77, 3, 80)],
        (True, 0, 95): [('direct', 0, 10, 0, 95, True)],
        (True, 25, 66): [('cache', 2, 3), ('copy', 0, 5, 25), ('direct', 3, 9, 5, 60, True), ('cache', 9, 10),
                         ('copy', 65, 1, 90)],
    }
    for (resident, offset, count), want in cases.items():
        Source._resident = resident
        log.clear()
        task = T(offset)
        task.read_device(count)
        assert log == want and task.offset == offset + count, (resident, offset, count, log)
