"""Known answers of the reference's channelizer tests on the HIP path (reference
baseband_tasks/tests/test_channelize.py:14-193; same channel count, framings,
frequency rules, round trips and error cases).

The reference reads `baseband`'s sample VDIF file (real, 8 threads, 40000
samples at 32 MHz) and sample DADA file (complex, 2 polarisations, 16000
samples at 16 MHz) there; neither the package nor the files are in this image,
so streams of the same shapes, rates and value levels are generated.  Where
the reference compares with its own FFT bit for bit, this compares with
numpy's float64 FFT to float32 rounding (the tolerance of test_gpu_parity.py).
"""
import numpy as np
import pytest

import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

pytestmark = pytest.mark.gpu

N = 1024
LEVELS_2BIT = np.array([-3.3359, -1., 1., 3.3359], np.float32)


def close_to(got, want):
    want = np.asarray(want)
    err = np.abs(got - want)
    rms = np.sqrt(np.mean(np.abs(want) ** 2))
    return np.sqrt(np.mean(err ** 2)) <= 1e-6 * rms and err.max() <= 1e-5 * rms


@pytest.fixture(scope='module')
def vdif_like():
    """Real 2-bit levels, (40000, 8) at 32 MHz, as the sample VDIF file decodes."""
    rng = np.random.default_rng(1)
    data = rng.choice(LEVELS_2BIT, size=(40000, 8))
    fh = bt.StreamGenerator(lambda f: data[f.tell():f.tell() + f.samples_per_frame], data.shape,
                            bt.Time('2014-06-16T05:56:07'), 32 * u.MHz, samples_per_frame=20000, dtype=np.float32)
    with_freq = bt.SetAttribute(fh, frequency=(311.25 + 16. * (np.arange(8) // 2)) * u.MHz,
                                sideband=np.tile([-1, 1], 4))
    whole = data[:N * (data.shape[0] // N)].reshape(-1, N, 8)
    return fh, with_freq, data, np.fft.rfft(whole.astype(np.float64), axis=1)


@pytest.fixture(scope='module')
def dada_like():
    """Complex 8-bit integers, (16000, 2) at 16 MHz, 320 MHz, as the sample DADA file decodes."""
    rng = np.random.default_rng(2)
    data = (rng.integers(-60, 60, size=(16000, 2)) + 1j * rng.integers(-60, 60, size=(16000, 2))).astype(np.complex64)
    fh = bt.StreamGenerator(lambda f: data[f.tell():f.tell() + f.samples_per_frame], data.shape,
                            bt.Time('2013-07-02T01:39:20'), 16 * u.MHz, samples_per_frame=16000,
                            dtype=np.complex64)
    return fh, bt.SetAttribute(fh, frequency=320. * u.MHz, sideband=np.array([1, -1])), data


def test_everything_and_the_tail(vdif_like):
    """test_channelize.py:41-69."""
    fh, _, _, ref = vdif_like
    ct = bt.Channelize(fh, N)
    everything = ct.read()
    assert ct.tell() == ct.shape[0]
    assert abs((ct.time - ct.start_time) - ct.shape[0] / ct.sample_rate) < 1e-9
    assert everything.dtype == np.complex64 and everything.shape == ref.shape
    assert close_to(everything, ref)
    ct.seek(-3, 2)
    assert ct.tell() == ct.shape[0] - 3
    tail = ct.read()
    assert tail.shape[0] == 3 and np.array_equal(tail, everything[-3:])
    ct.seek(-2, 2)
    with pytest.raises(EOFError):
        ct.read(10)
    ct.close()
    assert ct.closed
    with pytest.raises(ValueError):
        ct.read(1)
    with pytest.raises(AttributeError):
        ct.ih


@pytest.mark.parametrize('samples_per_frame', [1, 16, 33])
def test_framing_only_trims_the_end(vdif_like, samples_per_frame):
    """test_channelize.py:71-88: whole frames of spectra only; values unchanged."""
    fh, _, _, ref = vdif_like
    ct = bt.Channelize(fh, N, samples_per_frame=samples_per_frame)
    got = ct.read()
    assert len(got) % samples_per_frame == 0 and len(got) // samples_per_frame == len(ref) // samples_per_frame
    assert close_to(got, ref[:len(got)])
    ct.seek(-3, 2)
    assert ct.tell() == ct.shape[0] - 3
    tail = ct.read()
    assert tail.shape[0] == 3 and np.array_equal(tail, got[-3:])


def test_channel_frequencies_of_real_streams(vdif_like):
    """test_channelize.py:90-94: band edge + sideband * rfftfreq."""
    _, with_freq, _, _ = vdif_like
    ct = bt.Channelize(with_freq, N)
    sideband = np.tile([-1, 1], 4)
    want = ((311.25 + 16 * (np.arange(8) // 2)) * u.MHz
            + sideband * np.fft.rfftfreq(N, 1. / (32 * u.MHz))[:, np.newaxis])
    assert np.array_equal(ct.sideband, sideband) and np.array_equal(ct.frequency, want)


def test_real_round_trip(vdif_like):
    """test_channelize.py:96-110: to 1e-5 for samples of order 1."""
    _, with_freq, data, _ = vdif_like
    ct = bt.Channelize(with_freq, N)
    dt = bt.Dechannelize(ct, N, dtype=np.float32)
    n_rec = (data.shape[0] // N) * N
    assert dt.shape == (n_rec, 8)
    back = dt.read()
    assert back.dtype == np.float32 and np.allclose(back, data[:n_rec], atol=1e-5)
    assert np.array_equal(dt.frequency, with_freq.frequency) and np.array_equal(dt.sideband, with_freq.sideband)
    assert np.array_equal(ct.inverse(ct).read(), back)


def test_refusals(vdif_like):
    """test_channelize.py:112-115 (a channel count the engine cannot transform; the reference
    asserts on one that does not divide the stream), 128-133, 135-139."""
    fh, with_freq, _, _ = vdif_like
    with pytest.raises((AssertionError, ValueError)):
        bt.Channelize(fh, 400001)
    with bt.Channelize(fh, N) as ct:
        with pytest.raises(AttributeError):
            ct.frequency
        with pytest.raises(AttributeError):
            ct.sideband
    with pytest.raises(ValueError):
        bt.Dechannelize(bt.Channelize(with_freq, N), dtype=np.float32)      # real output needs n


def test_repr_names_the_task_and_n(vdif_like):
    """test_channelize.py:117-126."""
    ct = bt.Channelize(vdif_like[0], N)
    assert repr(ct).startswith('Channelize(ih') and f'n={N}' in repr(ct)
    dt = bt.Dechannelize(ct, N)
    assert repr(dt).startswith('Dechannelize(ih') and f'n={N}' in repr(dt)


def test_channel_frequencies_of_complex_streams(dada_like):
    """test_channelize.py:153-167: centre + sideband * fftfreq, for either sideband."""
    fh, with_freq, _ = dada_like
    grid = np.fft.fftfreq(N, 1. / (16 * u.MHz))[:, np.newaxis]
    ct = bt.Channelize(with_freq, N)
    assert np.array_equal(ct.sideband, with_freq.sideband)
    assert np.array_equal(ct.frequency, 320. * u.MHz + grid * with_freq.sideband)
    flipped = bt.SetAttribute(fh, frequency=320. * u.MHz, sideband=-np.asarray(with_freq.sideband))
    ct = bt.Channelize(flipped, N)
    assert np.array_equal(ct.frequency, 320. * u.MHz - grid * with_freq.sideband)


def test_complex_round_trip_and_inverse_of_inverse(dada_like):
    """test_channelize.py:169-193."""
    _, with_freq, data = dada_like
    ct = bt.Channelize(with_freq, N)
    dt = bt.Dechannelize(ct)
    n_rec = (data.shape[0] // N) * N
    assert dt.shape == (n_rec, 2)
    back = dt.read()
    assert np.allclose(back, data[:n_rec], atol=1e-5 * 60)               # the reference's 1e-5 is for samples of order 1
    assert np.array_equal(dt.frequency, with_freq.frequency) and np.array_equal(dt.sideband, with_freq.sideband)
    dt2 = ct.inverse(ct)
    assert np.array_equal(dt2.read(), back)
    ct2 = dt2.inverse(with_freq)
    ct.seek(0)
    assert np.array_equal(ct.read(), ct2.read())
    dt2.close()
