"""The reference's delay-and-resample tests on the HIP path (reference
baseband_tasks/tests/test_sampling.py:264-618; same receiver simulation,
signals, delays, channel counts and tolerances).

A real voltage stream (a tone or band-limited noise) is "observed" by two
telescopes: mixed with a local oscillator (real mixing, or complex quadrature
mixing), low-pass filtered and downsampled on the host -- or, CHIME-like, just
channelized.  One telescope sees the signal delayed.  `ShiftAndResample`,
`TimeDelay` and `Resample` (on the GPU) must then turn the delayed telescope's
stream into the other one's.  Delays are in samples of the raw stream; in this
package a shift is given in samples of the stream it is applied to (there are
no unit quantities without astropy), i.e. ``delay / downsample`` (``/ n`` after
channelization).  The reference compares the two telescopes through `Stack`;
here both are read over their common time range.
"""
import numpy as np
import pytest

import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

pytestmark = pytest.mark.gpu

FULL_RATE = 204.8 * u.kHz                 # of the raw, real-valued signal (test_sampling.py:277)
FULL_FRAME = 1024
N_FRAMES = 32
START = bt.Time('2010-11-12T13:14:15')
PHI0_MIXER = np.deg2rad(-12.3456789)
PHI0_SIGNAL = np.deg2rad(98.7654321)


def tone(frequency, phi0):
    """Callable ``f(stream, dtype)``: cos (real) or exp(i .) (complex) of the phase at the
    times of the stream's next frame (test_sampling.py:20-39)."""
    def samples(fh, dtype=None):
        dtype = np.dtype(dtype or fh.dtype)
        t = (fh.time - START) + np.arange(fh.samples_per_frame) / fh.sample_rate
        phi = phi0 + 2. * np.pi * np.asarray(frequency) * t.reshape((-1,) + (1,) * len(fh.sample_shape))
        return (np.cos(phi) if dtype.kind == 'f' else np.exp(1j * phi)).astype(dtype)
    return samples


class Receiver:
    """One of the reference's set-ups: how the raw signal is made and observed."""

    def __init__(self, signal, mixing, chime=False):
        self.dtype = np.dtype(np.complex64 if mixing == 'complex' else np.float32)
        self.chime = chime
        if chime:                                   # test_sampling.py:557-577: no mixing at all
            self.sideband, self.lo, self.phi0_mixer = np.array(-1), 0., 0.
            self.f_signal = 7 / 8 * FULL_RATE
        else:
            self.sideband = np.array([-1, 1])
            self.lo = FULL_RATE * (7 / 16 - self.sideband / 128)     # on the right side of the signal
            self.phi0_mixer = PHI0_MIXER
            self.f_signal = 7 / 16 * FULL_RATE
        self.downsample = 16 if self.dtype.kind == 'c' else 8
        self.rate = FULL_RATE / self.downsample
        self.mixer = tone(self.lo, self.phi0_mixer)
        self.kind = signal
        if signal == 'tone':
            make = tone(self.f_signal, PHI0_SIGNAL)
        else:
            noise = bt.Noise(seed=12345)
            cut = {'real': 'upper', 'complex': None}[mixing] if not chime else 'lower'

            def make(fh, noise=noise, cut=cut):
                data = noise(fh)
                if cut is None:
                    return data
                ft = np.fft.rfft(data, axis=0)             # keep the signal inside the observed band
                if cut == 'upper':
                    ft[ft.shape[0] // 2:] = 0
                else:
                    ft[:ft.shape[0] // 2] = 0
                return np.fft.irfft(ft, axis=0).astype(data.dtype)
        self.raw = bt.StreamGenerator(make, shape=(FULL_FRAME * N_FRAMES, 2), start_time=START,
                                      sample_rate=FULL_RATE, dtype=np.float32, samples_per_frame=FULL_FRAME)

    # -- the simulated receiver (test_sampling.py:300-342) -----------------------------------
    def mix_downsample(self, task, data):
        raw = task.ih
        if task.complex_data:
            # quadrature mix: data * cos, data * -/+ sin for lower / upper sideband
            mixed = data * self.mixer(raw, dtype=task.dtype)
            mixed = np.where(np.asarray(task.sideband) > 0, mixed.conj(), mixed)
            mixed = np.stack([mixed.real, mixed.imag], axis=-1).astype(data.dtype)
        else:
            # real mixing: first remove the wrong sideband
            ft = np.fft.rfft(data, axis=0)
            f = np.fft.rfftfreq(data.shape[0], 1. / raw.sample_rate).reshape((-1,) + (1,) * (ft.ndim - 1))
            ft[(f < self.lo) ^ (np.asarray(task.sideband) < 0)] = 0
            mixed = np.fft.irfft(ft, axis=0) * self.mixer(raw)
        # keep f - f_mix only (low pass), restore the half of the signal that went with f + f_mix
        ft = np.fft.rfft(mixed, axis=0)
        ft[ft.shape[0] // self.downsample:] = 0
        filtered = np.fft.irfft(2. * ft, axis=0).astype(data.dtype)[::self.downsample]
        return filtered[..., 0] + 1j * filtered[..., 1] if task.complex_data else filtered

    def telescope(self, delay=None, n=None):
        """The stream one telescope records: the signal arrives ``delay`` raw samples late."""
        fh = self.raw if delay is None else bt.SetAttribute(self.raw, start_time=START - delay / FULL_RATE)
        if self.chime:
            return bt.Channelize(fh, 32, frequency=FULL_RATE, sideband=self.sideband)
        obs = bt.Task(fh, self.mix_downsample, method=True, dtype=self.dtype, sample_rate=self.rate,
                      frequency=self.lo, sideband=self.sideband)
        return obs if n is None else bt.Channelize(obs, n)

    def samples_per(self, n):
        """Raw samples per sample of a telescope stream."""
        if self.chime:
            return 32
        return self.downsample * (1 if n is None else n)


def common(a, b):
    """Both streams over the time range they share."""
    start = max(a.start_time, b.start_time, key=lambda t: t - START)
    stop = min(a.stop_time, b.stop_time, key=lambda t: t - START)
    count = int(round((stop - start) * a.sample_rate))
    assert count * int(np.prod(a.sample_shape)) > 500          # (we do compare something)
    out = []
    for fh in (a, b):
        fh.seek(start)
        assert abs(fh.time - start) < 1e-9
        out.append(fh.read(count))
    return out


SETUPS = {(s, m): None for s in ('tone', 'noise') for m in ('real', 'complex')}


@pytest.fixture(scope='module', params=[('tone', 'real'), ('tone', 'complex'), ('noise', 'real'),
                                        ('noise', 'complex'), ('tone', 'chime'), ('noise', 'chime')],
                ids=lambda p: '-'.join(p))
def rx(request):
    signal, mixing = request.param
    return Receiver(signal, 'complex' if mixing == 'chime' else mixing, chime=mixing == 'chime')


def tolerance(rx, n):
    """test_sampling.py:276, 436, 525, 559: 1e-2 per sample; channelized 4e-4 (tone), 1e-4 (noise, CHIME)."""
    if n is None and not rx.chime:
        return 1e-2
    return 4e-4 if (rx.kind == 'tone' and not rx.chime) else 1e-4


def channel_counts(rx):
    return [32] if rx.chime else [None, 32]


@pytest.mark.parametrize('delay', [-18.25, -np.pi, -8, 0.1, 65.4321])
def test_shift_and_resample_undoes_the_delay(rx, delay):
    """test_sampling.py:365-387 (and 589-593 CHIME-like)."""
    for n in channel_counts(rx):
        tel1, tel2 = rx.telescope(None, n), rx.telescope(delay, n)
        shift = delay / rx.samples_per(n)
        if n is None:
            assert np.all(np.asarray(tel1.frequency) == rx.lo)
            undone = bt.ShiftAndResample(tel2, shift, tel1.start_time, lo=rx.lo)
        else:
            undone = bt.ShiftAndResample(tel2, shift, tel1.start_time, lo=rx.lo, samples_per_frame=32, pad=6)
        a, b = common(tel1, undone)
        assert np.abs(a - b).max() < tolerance(rx, n), (n, delay)


@pytest.mark.parametrize('delay', [-8, 16])
def test_time_delay_by_whole_samples(rx, delay):
    """test_sampling.py:398-409 (579-583: CHIME-like delays are multiples of its 32-sample spectra)."""
    if rx.dtype.kind != 'c':
        pytest.skip('TimeDelay needs complex samples')
    if rx.chime:
        delay *= 4                                            # -32, 64
    tel1, tel2 = rx.telescope(None), rx.telescope(delay)
    delayed = bt.TimeDelay(tel2, delay / rx.samples_per(None), lo=rx.lo)
    a, b = common(tel1, delayed)
    assert np.abs(a - b).max() < tolerance(rx, None if not rx.chime else 32)


@pytest.mark.parametrize('delay', [-1, 15.4321])
def test_time_delay_then_resample_onto_the_grid(rx, delay):
    """test_sampling.py:411-426 (585-587)."""
    if rx.dtype.kind != 'c':
        pytest.skip('TimeDelay needs complex samples')
    for n in channel_counts(rx):
        tel1, tel2 = rx.telescope(None, n), rx.telescope(delay, n)
        delayed = bt.TimeDelay(tel2, delay / rx.samples_per(n), lo=rx.lo)
        if n is None:
            aligned = bt.Resample(delayed, tel1.start_time)
        else:
            aligned = bt.Resample(delayed, tel1.start_time, samples_per_frame=32, pad=6)
        a, b = common(tel1, aligned)
        assert np.abs(a - b).max() < tolerance(rx, n), (n, delay)


@pytest.mark.parametrize('delay', [-8., 12.3456789])
def test_time_delay_is_shift_and_resample_without_a_filter(rx, delay):
    """test_sampling.py:512-521: pad = 0 and no grid to hit."""
    if rx.dtype.kind != 'c' or rx.kind != 'tone':
        pytest.skip('complex tone set-ups only')
    tel = rx.telescope(delay)
    shift = delay / rx.samples_per(None)
    one = bt.TimeDelay(tel, shift, lo=rx.lo)
    two = bt.ShiftAndResample(tel, shift, offset=None, lo=rx.lo, pad=0)
    assert one.shape == two.shape and abs(one.start_time - two.start_time) < 1e-12
    a, b = common(one, two)
    assert np.abs(a - b).max() < tolerance(rx, None if not rx.chime else 32)


@pytest.mark.parametrize('delay', [None, -13, -2, 1, 111])
def test_the_simulated_tone_is_what_it_should_be(rx, delay):
    """test_sampling.py:446-506 (600-607): the recorded tone has the signal's phase --
    taken `delay` later -- minus the mixer's, times the sideband."""
    if rx.kind != 'tone':
        pytest.skip('tone set-ups only')
    for n in channel_counts(rx):
        tel = rx.telescope(delay, n)
        late = 0. if delay is None else delay / FULL_RATE
        assert abs((tel.start_time - START) + late) < 1e-9
        data = tel.read()
        i = np.arange(data.shape[0]).reshape((-1,) + (1,) * 1)
        dt = (tel.start_time - START) + i / tel.sample_rate
        phi = PHI0_SIGNAL + 2. * np.pi * (dt + late) * rx.f_signal
        phi = (phi - (rx.phi0_mixer + 2. * np.pi * dt * rx.lo)) * rx.sideband
        expected = np.cos(phi) if (n is None and rx.dtype.kind == 'f') else np.exp(1j * phi)
        if n is None and not rx.chime:
            assert np.abs(data - expected.astype(data.dtype)).max() < 1e-2
        else:
            where = np.isclose(np.asarray(tel.frequency), abs(rx.f_signal))
            picked = data[:, np.broadcast_to(where, data.shape[1:])].reshape(data.shape[0], -1)
            complex_input = tel.ih.complex_data
            factor = (n or 32) if complex_input else (n or 32) // 2
            if not complex_input and not rx.chime:
                expected = np.exp(1j * phi)                  # a real tone's positive-frequency half
            assert np.abs(picked - expected * factor).max() < tolerance(rx, n or 32) * factor
