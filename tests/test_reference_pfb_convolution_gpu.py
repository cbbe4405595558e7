"""Known answers of the reference's filter-bank and convolution tests on the HIP
path (reference baseband_tasks/tests/test_pfb.py:38-102 and
test_convolution.py:42-98; same streams, seeds, responses and expectations).

The reference runs these on float64 / complex128 noise; the kernels compute in
single precision, so the double-precision streams pass through
`SinglePrecision` and the comparisons use float32 tolerances (stated at each
assertion).  The inversion tests (test_pfb.py:104-236, CHIME's 4 x 2048 and
GUPPI's 12 x 64 filter banks on real noise, with and without digitization) are
at the end.
"""
import numpy as np
import pytest

import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ test_pfb.py:38-102
N_CHAN, N_TAP = 2048, 4


def noise_stream(dtype):
    nh = bt.NoiseGenerator(shape=(2500 * N_CHAN,), start_time=bt.Time('2010-01-01T00:00:00'),
                           sample_rate=1. * u.kHz, seed=12345, samples_per_frame=128, dtype=dtype)
    return nh, bt.SinglePrecision(nh)


@pytest.mark.parametrize('offset', [0, 1000])
@pytest.mark.parametrize('dtype', ['f8', 'c16'])
def test_filter_bank_is_sum_of_weighted_blocks_then_fft(offset, dtype):
    """test_pfb.py:54-102: multiplying 4 blocks by the 4 x 2048 response, summing
    them and transforming equals transforming the long weighted array and
    keeping every 4th frequency -- and both filter-bank classes return that, for
    real and complex noise, at offsets 0 and 1000 spectra."""
    chime = bt.sinc_hamming(N_TAP, N_CHAN)
    nh, single = noise_stream(dtype)
    nh.seek(offset * N_CHAN)
    blocks = nh.read(5 * N_CHAN).reshape(-1, N_CHAN)
    fft = np.fft.rfft if dtype == 'f8' else np.fft.fft
    weighted = chime * blocks[:4]
    summed = fft(weighted.sum(0))
    assert np.allclose(fft(weighted.ravel())[::4], summed)               # the identity itself (float64)
    second = fft((chime * blocks[1:]).sum(0))
    scale = np.sqrt(np.mean(np.abs(summed) ** 2))
    for cls in (bt.PolyphaseFilterBankSamples, bt.PolyphaseFilterBank):
        pfb = cls(single, chime)
        pfb.seek(offset)
        got = pfb.read(2)
        assert got.shape == (2, summed.shape[0]) and got.dtype == np.complex64
        # float32 arithmetic on 4 x 2048-sample sums: 1e-5 of the spectrum's rms (reference: rtol 1e-7 in float64)
        assert np.abs(got[0] - summed).max() < 1e-5 * scale
        assert np.abs(got[1] - second).max() < 1e-5 * scale


# ------------------------------------------------------------------ test_convolution.py:42-98
class TestSmoothingFilter:
    """Three-tap box filter over 16000 x 2 samples of real noise."""

    @classmethod
    def setup_class(cls):
        cls.start = bt.Time('2010-11-12T13:14:15')
        cls.rate = 10. * u.kHz
        cls.nh = bt.SinglePrecision(bt.NoiseGenerator(shape=(16000, 2), start_time=cls.start, sample_rate=cls.rate,
                                                      samples_per_frame=200, dtype=float, seed=12345))
        cls.data = cls.nh.read().astype(np.float64)
        cls.box = cls.data[:-2] + cls.data[1:-1] + cls.data[2:]

    @pytest.mark.parametrize('cls_name', ['ConvolveSamples', 'Convolve'])
    @pytest.mark.parametrize('offset', [1, 2])
    def test_offset_moves_only_the_start_time(self, cls_name, offset):
        """test_convolution.py:57-75."""
        task = getattr(bt, cls_name)(self.nh, np.ones(3), offset=offset, samples_per_frame=1024)
        assert abs(task.start_time - self.start - (2 - offset) / self.rate) < 1e-9
        head = task.read(10)
        assert np.allclose(head, self.box[:10], atol=2e-6)              # float32 sums of three values of order 1
        task.seek(-10, 2)
        assert np.allclose(task.read(10), self.box[-10:], atol=2e-6)
        per_stream = getattr(bt, cls_name)(self.nh, np.ones((3, 2)), offset=offset, samples_per_frame=1024)
        per_stream.seek(5)
        assert np.allclose(per_stream.read(5), head[5:], atol=2e-6)

    @pytest.mark.parametrize('cls_name', ['ConvolveSamples', 'Convolve'])
    def test_a_response_per_stream(self, cls_name):
        """test_convolution.py:77-85: the second stream's filter lacks its last tap."""
        response = np.array([[1., 1., 1.], [1., 1., 0.]]).T
        task = getattr(bt, cls_name)(self.nh, response, samples_per_frame=512)
        assert abs(task.start_time - self.start - 2 / self.rate) < 1e-9
        want = self.data[:-2] * np.array([1, 0]) + self.data[1:-1] + self.data[2:]
        assert np.allclose(task.read(), want, atol=2e-6)

    @pytest.mark.parametrize('cls_name', ['ConvolveSamples', 'Convolve'])
    def test_a_response_of_the_wrong_shape_is_refused(self, cls_name):
        """test_convolution.py:95-98."""
        with pytest.raises(ValueError):
            getattr(bt, cls_name)(self.nh, np.ones((3, 3)))

    @pytest.mark.parametrize('cls_name', ['ConvolveSamples', 'Convolve'])
    def test_whole_stream_and_a_read_from_the_end(self, cls_name):
        """test_convolution.py:16-39 on generated data (the reference reads a DADA sample
        file there): shape, start and stop times, everything and the last three samples."""
        task = getattr(bt, cls_name)(self.nh, np.ones(3), samples_per_frame=1024)
        everything = task.read()
        assert task.tell() == task.shape[0] == self.nh.shape[0] - 2
        assert abs(task.start_time - self.nh.start_time - 2 / self.rate) < 1e-9
        assert abs(task.stop_time - self.nh.stop_time) < 1e-9
        assert np.allclose(everything, self.box, atol=1e-4)
        task.seek(-3, 2)
        assert task.tell() == task.shape[0] - 3
        tail = task.read()
        assert tail.shape[0] == 3 and np.allclose(tail, self.box[-3:], atol=1e-4)


# ------------------------------------------------------------------ test_pfb.py:104-236
def digitize(spectra, level):
    """Round real and imaginary parts to multiples of ``level`` (test_pfb.py:22-23)."""
    flat = np.ascontiguousarray(spectra)
    return (np.round(flat.view(flat.real.dtype) / level) * level).view(flat.dtype)


class TestInversion:
    """Undoing CHIME's 4 x 2048 and GUPPI's 12 x 64 filter banks on real noise
    (test_pfb.py:104-236).  The reference runs these in float64; here the noise
    is the same (same generator and seed) but cast to float32 and every
    transform is single precision, which the tolerances -- set by the
    deconvolution, not by rounding -- absorb."""

    @classmethod
    def setup_class(cls):
        nh = bt.NoiseGenerator(shape=(2500 * N_CHAN,), start_time=bt.Time('2010-01-01T00:00:00'),
                               sample_rate=1. * u.kHz, seed=12345, samples_per_frame=128, dtype='f8')
        cls.nh = bt.SinglePrecision(nh)
        cls.chime = bt.sinc_hamming(4, 2048)
        cls.guppi = bt.sinc_hamming(12, 64, sinc_scale=0.95)

    def input_blocks(self, first_sample, n_blocks, n):
        self.nh.seek(first_sample)
        return self.nh.read(n_blocks * n).reshape(-1, n).astype(np.float64)

    def test_by_hand_with_and_without_wiener_filter(self):
        """test_pfb.py:104-133: dechannelize the filter bank output, divide by the
        response's transform along the block axis; the band edges (the middle
        of each 2048-sample block) cannot be recovered, a Wiener filter with
        threshold 0.05 tames them."""
        n_sample = 128
        d_in = self.input_blocks(3 * 2048, n_sample, 2048)
        pfb = bt.PolyphaseFilterBank(self.nh, self.chime, samples_per_frame=n_sample)
        ft_pfb = pfb.read(n_sample + 3).astype(np.complex128)
        d_pfb = np.fft.irfft(ft_pfb, axis=1)
        ft_fine = np.fft.rfft(d_pfb, axis=0)
        long_response = np.zeros((n_sample + 3, 2048))
        long_response[:4] = self.chime
        ft_resp = np.fft.rfft(long_response, axis=0).conj()
        d_out = np.fft.irfft(ft_fine / ft_resp, axis=0, n=d_pfb.shape[0])[3:]
        assert np.allclose(d_in[32:-32, :900], d_out[32:-32, :900], atol=0.001)
        assert np.allclose(d_in[32:-32, 1150:], d_out[32:-32, 1150:], atol=0.001)
        threshold = 0.05
        inverse = ft_resp.conj() / (threshold ** 2 + np.abs(ft_resp) ** 2) * (1 + threshold ** 2)
        d_out2 = np.fft.irfft(ft_fine * inverse, axis=0, n=d_pfb.shape[0])[3:]
        assert np.allclose(d_in[32:-32], d_out2[32:-32], atol=0.3)

    def test_chime_filter_bank_inverted(self):
        """test_pfb.py:170-183: sn = 100, 48 blocks of padding; all but the 50
        samples at either edge of each block to 0.01."""
        n_sample, pad = 128, 48
        d_in = self.input_blocks(pad * 2048 + 3 * 2048 // 2, n_sample, 2048)
        pfb = bt.PolyphaseFilterBank(self.nh, self.chime)
        ipfb = bt.InversePolyphaseFilterBank(pfb, self.chime, sn=100, pad_start=pad, pad_end=pad,
                                             samples_per_frame=n_sample * 2048, dtype=self.nh.dtype)
        d_out = ipfb.read(n_sample * 2048).reshape(-1, 2048)
        assert d_out.dtype == np.float32
        assert np.allclose(d_in[:, 50:-50], d_out[:, 50:-50], atol=0.01)

    def test_chime_filter_bank_inverted_after_digitization(self):
        """test_pfb.py:185-202: spectra rounded at a third of their rms, sn = 10:
        residual rms 0.125 +- 0.01, everything within 1.1."""
        n_sample, pad = 128, 32
        d_in = self.input_blocks(pad * 2048 + 3 * 2048 // 2, n_sample, 2048)
        pfb = bt.PolyphaseFilterBank(self.nh, self.chime)
        level = pfb.read(n_sample).real.std() / 3.
        rounded = bt.Task(pfb, lambda ft: digitize(ft, level), samples_per_frame=n_sample)
        ipfb = bt.InversePolyphaseFilterBank(rounded, self.chime, sn=10, pad_start=pad, pad_end=pad,
                                             samples_per_frame=n_sample * 2048, dtype=self.nh.dtype)
        d_out = ipfb.read(n_sample * 2048).reshape(-1, 2048)
        assert np.isclose((d_out - d_in).std(), 0.125, atol=0.01)
        assert np.allclose(d_in, d_out, atol=1.1)

    def test_guppi_filter_bank_inverted(self):
        """test_pfb.py:204-222: 12 x 64 with sinc scale 0.95 cuts the channel edges so
        hard that sn = 30 only gives 0.15; without regularisation all but the
        two samples at either edge of each block come back to 0.005."""
        n_sample, pad = 512, 128
        d_in = self.input_blocks(pad * 64 + 11 * 64 // 2, n_sample, 64)
        pfb = bt.PolyphaseFilterBank(self.nh, self.guppi)
        ipfb = bt.InversePolyphaseFilterBank(pfb, self.guppi, sn=30, pad_start=pad, pad_end=pad,
                                             samples_per_frame=n_sample * 64, dtype=self.nh.dtype)
        d_out = ipfb.read(n_sample * 64).reshape(-1, 64)
        assert np.allclose(d_in, d_out, atol=0.15)
        ipfb2 = bt.InversePolyphaseFilterBank(pfb, self.guppi, sn=1e9, pad_start=pad, pad_end=pad,
                                              samples_per_frame=n_sample * 64, dtype=self.nh.dtype)
        d_out2 = ipfb2.read(n_sample * 64).reshape(-1, 64)
        assert np.allclose(d_in[:, 2:-2], d_out2[:, 2:-2], atol=0.005)

    def test_guppi_filter_bank_inverted_after_digitization(self):
        """test_pfb.py:224-236: rounding at a thirtieth of the rms hardly matters."""
        n_sample, pad = 512, 128
        d_in = self.input_blocks(pad * 64 + 11 * 64 // 2, n_sample, 64)
        pfb = bt.PolyphaseFilterBank(self.nh, self.guppi)
        level = pfb.read(n_sample).real.std() / 30.
        rounded = bt.Task(pfb, lambda ft: digitize(ft, level), samples_per_frame=n_sample)
        ipfb = bt.InversePolyphaseFilterBank(rounded, self.guppi, sn=30, pad_start=pad, pad_end=pad,
                                             samples_per_frame=n_sample * 64, dtype=self.nh.dtype)
        d_out = ipfb.read(n_sample * 64).reshape(-1, 64)
        assert np.allclose(d_in, d_out, atol=0.15)
