"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (the reference does not travel):

    cd /root/repo/tests/golden && \
    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        /opt/conda/bin/python3.9 -W ignore make_golden.py

Environment used for the committed fixtures: CPython 3.9.7, numpy 1.26.4,
astropy 4.3.1, reference mhvk/baseband-tasks @ 2025-03-21, NumPy FFT engine
(``fft_maker.set('numpy')``).  astropy 4.3.1 touches two numpy attributes
removed in numpy >= 1.23; they are defined below before importing astropy.

Only inputs/outputs (data) are written; no reference source is copied.
"""
import hashlib
import os

import numpy as np

for _name, _fn in (('asscalar', lambda a: np.asarray(a).item()),
                   ('alen', lambda a: len(np.asarray(a)))):
    if not hasattr(np, _name):
        setattr(np, _name, _fn)

from astropy import units as u            # noqa: E402
from astropy.time import Time             # noqa: E402

from baseband_tasks.fourier import fft_maker                      # noqa: E402
from baseband_tasks.fourier.numpy import NumpyFFTMaker            # noqa: E402
from baseband_tasks.generators import NoiseGenerator, StreamGenerator  # noqa: E402
from baseband_tasks.dispersion import Disperse, Dedisperse        # noqa: E402
from baseband_tasks.dm import DispersionMeasure                   # noqa: E402
from baseband_tasks.channelize import Channelize, Dechannelize    # noqa: E402
from baseband_tasks.pfb import (sinc_hamming, PolyphaseFilterBank,  # noqa: E402
                                PolyphaseFilterBankSamples, InversePolyphaseFilterBank)
from baseband_tasks.convolution import Convolve                   # noqa: E402
from baseband_tasks.sampling import Resample, ShiftSamples, TimeDelay  # noqa: E402
from baseband_tasks.dispersion import DisperseSamples, DedisperseSamples  # noqa: E402
from baseband_tasks.functions import Square, Power               # noqa: E402
from baseband_tasks.integration import Integrate                 # noqa: E402

fft_maker.set('numpy')
T0 = Time('2020-01-01T00:00:00', precision=9)
SEED = 12345


def noise(shape, fs, spf, frequency=None, sideband=None, seed=SEED):
    kw = {}
    if frequency is not None:
        kw = dict(frequency=frequency, sideband=sideband)
    return NoiseGenerator(shape, T0, fs, spf, dtype=np.complex64, seed=seed, **kw)


def SetPol(nh):
    from baseband_tasks.base import SetAttribute
    return SetAttribute(nh, polarization=np.array(['X', 'Y']))


def stats(a):
    a = np.asarray(a)
    return np.array([a.sum(dtype=np.complex128).real, a.sum(dtype=np.complex128).imag,
                     (np.abs(a.astype(np.complex128)) ** 2).sum()])


def geometry(task, ih):
    return np.array([task._pad_start, task._pad_end, task._ih_samples_per_frame,
                     task.samples_per_frame, task.shape[0],
                     getattr(task, '_sample_offset', 0)], dtype=np.int64), \
        np.array([((task.start_time - ih.start_time) * ih.sample_rate).to_value(u.one)])


def main():
    out = {}

    # ---- generator known answers (generators.py:178-190)
    nh = noise((2 * 2**20, 2), 16 * u.MHz, 2**20)
    f0 = nh.read(2**20)
    f1 = nh.read(2**20)
    out['noise_first'] = f0[:4].copy()
    out['noise_f1_first'] = f1[:4].copy()
    out['noise_sha'] = np.array([hashlib.sha256(f0.tobytes()).hexdigest(),
                                 hashlib.sha256(f1.tobytes()).hexdigest()])
    nh.seek(2**20 - 3)
    out['noise_straddle'] = nh.read(6)
    small = noise((1200, 3, 2), 1 * u.kHz, 500, seed=7)
    out['noise_small'] = small.read()

    # ---- DM math (dm.py)
    dm = DispersionMeasure(29.1168)
    freqs = np.array([300., 327.5, 1400., 1000.1234]) * u.MHz
    out['dm_freqs'] = freqs.value
    out['dm_time_delay_inf'] = dm.time_delay(freqs).to_value(u.s)
    out['dm_time_delay_ref'] = dm.time_delay(freqs, 350. * u.MHz).to_value(u.s)
    out['dm_phase_delay_inf'] = dm.phase_delay(freqs).to_value(u.cycle)
    out['dm_phase_delay_ref'] = dm.phase_delay(freqs, 350. * u.MHz).to_value(u.cycle)
    out['dm_const'] = np.array([DispersionMeasure.dispersion_delay_constant.to_value(
        u.s * u.MHz**2 * u.cm**3 / u.pc)])

    # ---- next_fast_len table (fourier/numpy.py:99-126)
    ns = np.concatenate([np.arange(1, 300), np.array([19324, 19200, 1000003, 2**20 + 128,
                                                      11059200, 1048577, 999999, 3981828])])
    out['nfl_n'] = ns
    out['nfl_out'] = np.array([NumpyFFTMaker.next_fast_len(int(n)) for n in ns])

    # ---- Dedisperse geometry for the BASELINE configs
    geo_cases = {}
    for fc in (1000., 800., 1400.):
        nh = noise((8 * 2**20, 2), 16 * u.MHz, 2**20, fc * u.MHz, 1)
        kw = {}
        if fc == 800.:
            kw = dict(samples_per_frame=2**20 - 415021)
        dd = Dedisperse(nh, 100., **kw)
        g, shift = geometry(dd, nh)
        geo_cases['dd_fc%d' % fc] = (g, shift, dd.reference_frequency.to_value(u.MHz))
    # impulse-test geometry, 8 reference frequencies (tests/test_dispersion.py:14-46)
    gp = StreamGenerator(lambda sh: np.zeros((sh.samples_per_frame,) + sh.shape[1:], sh.dtype),
                         shape=(164000, 2), start_time=Time('2010-11-12T13:14:15'),
                         sample_rate=128. * u.kHz, samples_per_frame=1000, dtype=np.complex64,
                         frequency=300 * u.MHz, sideband=np.array((1, -1)))
    gp_dm = 1000. * 0.05 / 0.039342251
    out['gp_dm'] = np.array([gp_dm])
    refs = [None, 300., 300.0123456789, 300.064, 299.936, 300.128, 300.123456789, 299.872]
    for i, rf in enumerate(refs):
        d = Disperse(gp, gp_dm, reference_frequency=None if rf is None else rf * u.MHz)
        g, shift = geometry(d, gp)
        geo_cases['gp_ref%d' % i] = (g, shift, np.atleast_1d(d.reference_frequency.to_value(u.MHz)))
        pf = d.phase_factor
        out['gp_chirp%d' % i] = pf[[0, 1, 17, pf.shape[0] // 2 - 1, pf.shape[0] // 2, -1]]
        out['gp_chirp_sum%d' % i] = stats(pf)
    for k, (g, shift, rf) in geo_cases.items():
        out['geo_' + k] = g
        out['shift_' + k] = shift
        out['reffreq_' + k] = np.atleast_1d(rf)

    # ---- config 2: Dedisperse(DM=100), 16 MHz at 1000 MHz, N = 2^20
    nh = noise((4 * 2**20, 2), 16 * u.MHz, 2**20, 1000. * u.MHz, 1)
    dd = Dedisperse(nh, 100.)
    pf = dd.phase_factor
    n = pf.shape[0]
    idx = np.array([0, 1, 2, 12345, n // 2 - 1, n // 2, n // 2 + 1, n - 2, n - 1])
    out['c2_chirp_idx'] = idx
    out['c2_chirp'] = pf[idx, 0]
    out['c2_chirp_stats'] = stats(pf)
    spf = dd.samples_per_frame
    y = dd.read()
    out['c2_shape'] = np.array(y.shape)
    out['c2_head'] = y[:2048]
    out['c2_seam1'] = y[spf - 1024:spf + 1024]
    out['c2_seam_last'] = y[3 * spf - 1024:3 * spf + 1024]
    out['c2_tail'] = y[-2048:]
    out['c2_stats_blocks'] = np.stack([stats(y[i * spf:(i + 1) * spf]) for i in range(4)])
    # metric pipeline: Channelize(Dedisperse, 1024, spf=512)
    dd.seek(0)
    ch = Channelize(dd, 1024, samples_per_frame=512)
    z = ch.read()
    out['c2ch_shape'] = np.array(z.shape)
    out['c2ch_head'] = z[:2]
    out['c2ch_seam'] = z[spf // 1024 - 1:spf // 1024 + 2]
    out['c2ch_tail'] = z[-2:]
    out['c2ch_stats'] = stats(z)
    out['c2ch_freq'] = ch.frequency.to_value(u.MHz).reshape(-1)[[0, 1, 511, 512, 1023]]

    # ---- config 1: Channelize(1024) on noise
    nh = noise((2**20, 2), 16 * u.MHz, 2**20, 1000. * u.MHz, 1)
    ch = Channelize(nh, 1024, samples_per_frame=16)
    z = ch.read()
    out['c1_shape'] = np.array(z.shape)
    out['c1_head'] = z[:4]
    out['c1_tail'] = z[-4:]
    out['c1_stats'] = stats(z)

    # ---- config 3: PFB 12 x 1024 sinc-hamming
    resp = sinc_hamming(12, 1024)
    out['sh_12_1024_stats'] = np.array([resp.sum(), resp.max(), resp[0, 0], resp[5, 17], resp[11, 1023]])
    out['sh_guppi'] = sinc_hamming(12, 64, sinc_scale=0.95)
    # the known-answer table the reference's own test holds (tests/test_pfb.py:26-35): GUPPI's
    # 12-tap, 64-channel coefficients, arranged as that test arranges them
    import baseband_tasks.tests as _ref_tests
    a = np.loadtxt(os.path.join(os.path.dirname(_ref_tests.__file__), 'data',
                                'bGDSP_U1_0032_T12_W095_get_pfb_coeffs.txt'))
    out['sh_guppi_table'] = a.reshape(8, -1).T.reshape(12, 64)
    out['sh_chime_stats'] = np.array([sinc_hamming(4, 2048).sum(), sinc_hamming(4, 2048)[1, 5]])
    nh = noise((2 * 2**20, 2), 16 * u.MHz, 2**20, 1000. * u.MHz, 1)
    pfb = PolyphaseFilterBank(nh, resp)
    out['c3_geo'] = np.array([pfb.padded._pad_start, pfb.padded._pad_end,
                              pfb.padded._ih_samples_per_frame, pfb.padded.samples_per_frame,
                              pfb.samples_per_frame, pfb.shape[0]])
    out['c3_shift'] = np.array([((pfb.start_time - nh.start_time) * nh.sample_rate).to_value(u.one)])
    z = pfb.read()
    out['c3_shape'] = np.array(z.shape)
    out['c3_head'] = z[:3]
    out['c3_seam'] = z[pfb.samples_per_frame - 1:pfb.samples_per_frame + 2]
    out['c3_tail'] = z[-3:]
    out['c3_stats'] = stats(z)

    # ---- small complete-output cases (entire arrays)
    # (a) Dedisperse, two sidebands, default (non power-of-two) geometry
    nh = noise((10000, 2), 1. * u.MHz, 4000, 300. * u.MHz, np.array([1, -1]), seed=11)
    for tag, kw in (('sa', {}), ('sb', dict(samples_per_frame=4096 - 767 - 771)),
                    ('sc', dict(reference_frequency=300.4 * u.MHz)),
                    ('sd', dict(reference_frequency=300.7 * u.MHz))):
        dd = Dedisperse(nh, 5., **kw)
        g, shift = geometry(dd, nh)
        out['%s_geo' % tag] = g
        out['%s_shift' % tag] = shift
        out['%s_out' % tag] = dd.read()
    # (b) Disperse with per-stream frequencies (4 streams), explicit power-of-two block
    freq = np.array([[300.], [301.]]) * u.MHz
    nh = noise((12000, 2, 2), 1. * u.MHz, 4000, freq, np.array([[1], [-1]]), seed=12)
    d = Disperse(nh, 3., samples_per_frame=8192 - 923 - 913)
    g, shift = geometry(d, nh)
    out['se_geo'] = g
    out['se_shift'] = shift
    out['se_out'] = d.read()
    # (c) Channelize / Dechannelize small
    nh = noise((20 * 256, 2), 1. * u.MHz, 1000, 300. * u.MHz, 1, seed=13)
    ch = Channelize(nh, 256, samples_per_frame=3)
    z = ch.read()
    out['sf_chan'] = z
    out['sf_freq'] = ch.frequency.to_value(u.MHz)
    nh.seek(0)
    out['sf_dechan'] = Dechannelize(ch).read()
    # (d) PFB small, both forms
    resp = sinc_hamming(4, 256)
    nh = noise((40 * 256, 2), 1. * u.MHz, 2560, 300. * u.MHz, 1, seed=14)
    p = PolyphaseFilterBank(nh, resp, samples_per_frame=8)
    out['sg_geo'] = np.array([p.padded._pad_start, p.padded._pad_end,
                              p.padded._ih_samples_per_frame, p.padded.samples_per_frame,
                              p.samples_per_frame, p.shape[0]])
    out['sg_pfb'] = p.read()
    nh1 = noise((40 * 256,), 1. * u.MHz, 2560, seed=15)
    out['sg_pfb_samples_1d'] = PolyphaseFilterBankSamples(nh1, resp, samples_per_frame=8).read()
    out['sg_pfb_fourier_1d'] = PolyphaseFilterBank(nh1, resp, samples_per_frame=8).read()
    # (e) Convolve and Resample small
    nh = noise((9000, 2), 1. * u.MHz, 3000, 300. * u.MHz, 1, seed=16)
    rng = np.random.RandomState(3)
    response = rng.normal(size=(33,))
    out['sh_response'] = response
    cv = Convolve(nh, response, offset=5)
    g, shift = geometry(cv, nh)
    out['sh_geo'] = g
    out['sh_shift'] = shift
    out['sh_out'] = cv.read()
    rs = Resample(nh, 0.25, pad=32, samples_per_frame=2048 - 64)
    g, shift = geometry(rs, nh)
    out['si_geo'] = g
    out['si_shift'] = shift
    out['si_pointer'] = np.array([rs.tell()])
    rs.seek(0)
    out['si_out'] = rs.read()
    # (f) fused chain small: Resample -> Dedisperse
    rs.seek(0)
    dd = Dedisperse(rs, 5., samples_per_frame=4096 - 767 - 771)
    g, shift = geometry(dd, rs)
    out['sj_geo'] = g
    out['sj_out'] = dd.read()

    # ---- detection and integration (functions.py, integration.py), small complete outputs
    nh = noise((40 * 256, 2), 1. * u.MHz, 2560, 300. * u.MHz, 1, seed=17)
    nh = SetPol(nh)
    ch = Channelize(nh, 256, samples_per_frame=4)
    sq = Square(ch)
    out['sk_square'] = sq.read()
    out['sk_square_pol'] = np.array([str(p) for p in sq.polarization.ravel()])
    ch.seek(0)
    pw = Power(ch)
    out['sk_power'] = pw.read()
    out['sk_power_pol'] = np.array([str(p) for p in pw.polarization.ravel()])
    for tag, kw in (('a', dict(step=8)), ('b', dict(step=5, start=3)), ('c', dict())):
        pw.seek(0)
        it = Integrate(pw, **kw)
        out['sk_int_%s' % tag] = it.read()
        out['sk_int_%s_meta' % tag] = np.array([it.shape[0], it.sample_rate.to_value(u.Hz),
                                               ((it.start_time - nh.start_time)
                                                * nh.sample_rate).to_value(u.one)])
    sq.seek(0)
    out['sk_int_sq'] = Integrate(sq, 4, samples_per_frame=3).read()

    # ---- integer shifts (sampling.py:380-425, dispersion.py:193-298)
    nh = noise((3000, 3, 2), 1. * u.kHz, 1000, np.array([[300.], [300.4], [301.]]) * u.MHz,
               np.array([[1], [1], [-1]]), seed=18)
    sh = ShiftSamples(nh, np.array([[-2], [0], [3]]), samples_per_frame=700)
    out['sl_shift'] = sh.read()
    out['sl_shift_meta'] = np.array([sh.shape[0], sh._pad_end,
                                     ((sh.start_time - nh.start_time) * nh.sample_rate).to_value(u.one)])
    nh.seek(0)
    ds_ = DisperseSamples(nh, 50., samples_per_frame=500)
    out['sl_disp_shift'] = ds_._shift
    out['sl_disp'] = ds_.read()
    out['sl_disp_meta'] = np.array([ds_.shape[0], ds_._pad_end,
                                    ((ds_.start_time - nh.start_time) * nh.sample_rate).to_value(u.one),
                                    ds_.reference_frequency.to_value(u.MHz)])

    # ---- real-valued streams (rfft paths: fourier/numpy.py:41-49, dispersion.py:58-61, channelize.py)
    nr = NoiseGenerator((12000, 2), T0, 1. * u.MHz, 4000, dtype=np.float32, seed=19,
                        frequency=300. * u.MHz, sideband=np.array([1, -1]))
    chr_ = Channelize(nr, 256, samples_per_frame=3)
    out['sr_chan'] = chr_.read()
    out['sr_chan_freq'] = chr_.frequency.to_value(u.MHz)
    chr_.seek(0)
    out['sr_dechan'] = Dechannelize(chr_, n=256, dtype=np.dtype('f4')).read()
    nr.seek(0)
    ddr = Dedisperse(nr, 5., samples_per_frame=4096 - 767 - 771)
    g, shift = geometry(ddr, nr)
    out['sr_dd_geo'] = g
    out['sr_dd_shift'] = shift
    out['sr_dd'] = ddr.read()
    out['sr_dd_chirp'] = ddr.phase_factor[[0, 1, 1000, 2047, 2048]]
    nr.seek(0)
    ddr2 = Disperse(nr, 5., reference_frequency=300.2 * u.MHz, samples_per_frame=4096 - 767 - 771)
    g, shift = geometry(ddr2, nr)
    out['sr_dd2_geo'] = g
    out['sr_dd2'] = ddr2.read()
    nr1 = NoiseGenerator((40 * 256,), T0, 1. * u.MHz, 2560, dtype=np.float32, seed=20)
    out['sr_pfb'] = PolyphaseFilterBank(nr1, sinc_hamming(4, 256), samples_per_frame=8).read()
    nr.seek(0)
    out['sr_square'] = Square(nr).read(1000)

    # ---- inverse polyphase filter bank (pfb.py:157-269), Wiener deconvolution along blocks
    resp = sinc_hamming(4, 32)
    nh = noise((25000, 2), 1. * u.MHz, 5000, 300. * u.MHz, 1, seed=22)
    pfb = PolyphaseFilterBank(nh, resp, samples_per_frame=100)
    out['sm_pfb_shape'] = np.array(pfb.shape)
    ipfb = InversePolyphaseFilterBank(pfb, resp, sn=10., pad_start=16, pad_end=16,
                                      samples_per_frame=8192 - 32 * 32 - 96)
    g, shift = geometry(ipfb, pfb)
    out['sm_ipfb_geo'] = g
    out['sm_ipfb_shift'] = np.array([((ipfb.start_time - nh.start_time) * nh.sample_rate).to_value(u.one)])
    out['sm_ipfb_rate'] = np.array([ipfb.sample_rate.to_value(u.Hz)])
    out['sm_ipfb'] = ipfb.read()
    out['sm_ipfb_resp'] = ipfb._ft_inverse_response[[0, 1, 100, 255], :, 0][:, [0, 5, 31]]

    # ---- TimeDelay (sampling.py:315-377)
    nh = noise((3000, 2), 1. * u.MHz, 1000, 300. * u.MHz, np.array([1, -1]), seed=23)
    td = TimeDelay(nh, 1.234 * u.us, lo=300. * u.MHz)
    out['st_delay'] = td.read()
    out['st_delay_shift'] = np.array([((td.start_time - nh.start_time) * nh.sample_rate).to_value(u.one)])

    # ---- config 5 geometry: Resample + Dedisperse, 8 streams
    nh = noise((8 * 2**20, 8), 16 * u.MHz, 2**20, 1000. * u.MHz, 1)
    rs = Resample(nh, 0.25, pad=64, samples_per_frame=2**20 - 128)
    g, shift = geometry(rs, nh)
    out['c5_rs_geo'] = g
    out['c5_rs_shift'] = shift
    out['c5_rs_pointer'] = np.array([rs.tell()])
    rs.seek(0)
    dd = Dedisperse(rs, 100., samples_per_frame=2**20 - 212476)
    g, shift = geometry(dd, rs)
    out['c5_dd_geo'] = g
    out['c5_dd_shift'] = np.array([((dd.start_time - nh.start_time) * nh.sample_rate).to_value(u.one)])

    # ---- default arguments at full scale: the reference's own (non power-of-two) blocks
    # (a) Dedisperse(DM=100) of a 16 MHz band at 800 MHz: block next_fast_len(4 * pad) = 1 666 980
    nh = noise((4 * 2**20, 2), 16 * u.MHz, 2**20, 800. * u.MHz, 1)
    dd = Dedisperse(nh, 100.)
    g, shift = geometry(dd, nh)
    out['d8_geo'] = g
    out['d8_shift'] = shift
    spf = dd.samples_per_frame
    y = dd.read()
    out['d8_head'] = y[:1024]
    out['d8_seam1'] = y[spf - 512:spf + 512]
    out['d8_tail'] = y[-1024:]
    out['d8_stats_blocks'] = np.stack([stats(y[i * spf:(i + 1) * spf]) for i in range(-(-y.shape[0] // spf))])
    # (b) config 5's first stage with default arguments: Resample picks blocks of 1 049 760 samples
    nh = noise((3 * 2**20, 2), 16 * u.MHz, 2**20, 1000. * u.MHz, 1)
    rs = Resample(nh, 0.25, pad=64)
    g, shift = geometry(rs, nh)
    out['r5_geo'] = g
    out['r5_shift'] = shift
    rs.seek(0)
    r = rs.read()
    spf = rs.samples_per_frame
    out['r5_head'] = r[:1024]
    out['r5_seam1'] = r[spf - 512:spf + 512]
    out['r5_tail'] = r[-1024:]
    out['r5_stats'] = stats(r)

    out.update(config4())
    np.savez_compressed('reference_vectors.npz', **out)
    total = sum(v.nbytes for v in out.values())
    print('wrote reference_vectors.npz with %d arrays, %.2f MB raw' % (len(out), total / 1e6))


def config4():
    """Config 4: ONE sub-band (k = 0, 403.125 MHz, the worst case) of 6.25 MHz, 2 pol,
    DM 557 with the sub-band centre as reference frequency, blocks of 2^24 samples; one full block
    and a re-aligned last one; then Channelize(64).  Added in round 3 (`make_golden.py --only c4`
    merges these arrays into the existing file)."""
    out = {}
    n_fft, pad = 2**24, 2756522
    spf = n_fft - pad
    nh = noise((n_fft + 2**20, 2), 6.25 * u.MHz, 2**20, 403.125 * u.MHz, 1)
    dd = Dedisperse(nh, 557., reference_frequency=403.125 * u.MHz, samples_per_frame=spf)
    g, shift = geometry(dd, nh)
    out['c4_geo'] = g
    out['c4_shift'] = shift
    pf = dd.phase_factor
    idx = np.array([0, 1, 4095, 4096, n_fft // 2 - 1, n_fft // 2, n_fft - 4096, n_fft - 1])
    out['c4_chirp_idx'] = idx
    out['c4_chirp'] = pf[idx, 0]
    out['c4_chirp_stats'] = stats(pf)
    y = dd.read()
    out['c4_shape'] = np.array(y.shape)
    out['c4_head'] = y[:1024]
    out['c4_mid'] = y[spf // 2:spf // 2 + 1024]
    out['c4_seam'] = y[spf - 512:spf + 512]
    out['c4_tail'] = y[-1024:]
    out['c4_stats_blocks'] = np.stack([stats(y[:spf]), stats(y[spf:])])
    dd.seek(0)
    ch = Channelize(dd, 64, samples_per_frame=4096)
    z = ch.read()
    out['c4ch_shape'] = np.array(z.shape)
    out['c4ch_head'] = z[:8]
    out['c4ch_seam'] = z[spf // 64 - 4:spf // 64 + 4]
    out['c4ch_tail'] = z[-8:]
    out['c4ch_stats'] = stats(z)
    return out


if __name__ == '__main__':
    import sys
    if sys.argv[1:] == ['--only', 'c4']:
        old = dict(np.load('reference_vectors.npz'))
        old.update(config4())
        np.savez_compressed('reference_vectors.npz', **old)
        print('merged config 4 arrays; %d arrays in all' % len(old))
    else:
        main()
