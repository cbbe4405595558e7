"""Known answers of the reference's integration tests on the HIP path, for
what the accelerated `Integrate` covers: whole-stream, from-a-start and
integer-step integration of `Square` output, averaged or as sums with counts
(reference baseband_tasks/tests/test_integration.py:17-168 and 256-263).

Integration over time intervals, pulse phase (`Fold`, `Stack`) and the
non-integer ratio case (171-255, 265-520) need the reference's host-side
`phase` machinery and are outside SURVEY 8(f); they raise
NotImplementedError here.  The reference's streams are float64; the kernels
sum in float32, so comparisons use rtol 1e-5.
"""
import numpy as np
import pytest

import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

pytestmark = pytest.mark.gpu

RATE = 10. * u.kHz
PERIOD = 125                                  # samples between the fake pulses (test_integration.py:31)


@pytest.fixture(scope='module')
def pulsar():
    """16000 x 2 samples: 10 every 125th sample, 0.125 otherwise (test_integration.py:17-44)."""
    def frame(fh):
        here = fh.tell() + np.arange(fh.samples_per_frame)
        return np.repeat(np.where(here % PERIOD == 0, 10., 0.125)[:, None], 2, axis=1).astype(np.float32)
    sh = bt.StreamGenerator(frame, (16000, 2), bt.Time('2010-11-12T13:14:15'), RATE, samples_per_frame=200,
                            dtype=np.float32)
    power = sh.read().astype(np.float64) ** 2
    return sh, power


def test_everything_at_once(pulsar):
    """test_integration.py:54-70."""
    sh, power = pulsar
    ip = bt.Integrate(bt.Square(sh))
    assert ip.start_time == sh.start_time and abs(ip.stop_time - sh.stop_time) < 1e-9
    assert abs((ip.stop_time - sh.start_time) - 1. / ip.sample_rate) < 1e-9
    data = ip.read()
    assert ip.tell() == ip.shape[0] and data.shape == (1, 2) and data.dtype == np.float32
    assert np.allclose(data, power.mean(0), rtol=1e-5)


def test_from_a_start_sample(pulsar):
    """test_integration.py:72-89."""
    sh, power = pulsar
    ip = bt.Integrate(bt.Square(sh), start=151)
    assert abs((ip.start_time - sh.start_time) - 151 / RATE) < 1e-9 and abs(ip.stop_time - sh.stop_time) < 1e-9
    assert np.allclose(ip.read(), power[151:].mean(0), rtol=1e-5)
    r = repr(ip)
    assert r.startswith('Integrate(ih') and 'start=151' in r and 'step' not in r


def test_sums_and_counts_instead_of_averages(pulsar):
    """test_integration.py:91-110."""
    sh, power = pulsar
    st = bt.Square(sh)
    ip = bt.Integrate(st, average=False)
    assert ip.start_time == sh.start_time and abs(ip.stop_time - sh.stop_time) < 1e-9
    integrated = ip.read()
    assert ip.tell() == ip.shape[0]
    assert integrated['data'].dtype == st.dtype and integrated['data'].shape == (1, 2)
    assert np.allclose(integrated['data'], power.sum(0), rtol=1e-5)
    assert np.all(integrated['count'] == sh.shape[0])


@pytest.mark.parametrize('seek', [121, -10])
@pytest.mark.parametrize('samples_per_frame', [1, 4, 10])
@pytest.mark.parametrize('n', [1, 3])
def test_integer_steps(pulsar, n, samples_per_frame, seek):
    """test_integration.py:112-144."""
    sh, power = pulsar
    n_sample = power.shape[0] // n
    seek = seek if seek > 0 else n_sample + seek
    want = power[seek * n:(seek + 10) * n].reshape(-1, n, 2).sum(1)
    st = bt.Square(sh)
    ip = bt.Integrate(st, n, average=False, samples_per_frame=samples_per_frame)
    assert ip.shape[0] == n_sample and ip.start_time == sh.start_time and ip.sample_rate == RATE / n
    ip.seek(seek)
    assert abs((ip.time - sh.start_time) - seek / ip.sample_rate) < 1e-9
    integrated = ip.read(10)
    assert ip.tell() == seek + 10
    assert integrated['data'].dtype == st.dtype and integrated['data'].shape == want.shape
    assert np.allclose(integrated['data'], want, rtol=1e-5) and np.all(integrated['count'] == n)
    r = repr(ip)
    assert f'step={n}' in r and 'average=False' in r


@pytest.mark.parametrize('samples_per_frame', [1, 4, 10])
@pytest.mark.parametrize('n', [1, 3])
def test_integer_steps_from_a_start_sample(pulsar, n, samples_per_frame):
    """test_integration.py:146-168 (the reference passes the step as the duration n / rate there)."""
    sh, power = pulsar
    want = power[121 * n:131 * n].reshape(-1, n, 2).sum(1)
    ip = bt.Integrate(bt.Square(sh), n, start=121 * n, average=False, samples_per_frame=samples_per_frame)
    assert abs((ip.start_time - sh.start_time) - 121 * n / RATE) < 1e-9 and ip.sample_rate == RATE / n
    integrated = ip.read(10)
    assert ip.tell() == 10 and abs((ip.time - sh.start_time) - 131 * n / RATE) < 1e-9
    assert np.allclose(integrated['data'], want, rtol=1e-5) and np.all(integrated['count'] == n)


def test_a_start_outside_the_stream_is_refused(pulsar):
    """test_integration.py:256-263."""
    sh, _ = pulsar
    with pytest.raises(ValueError):
        bt.Integrate(sh, start=sh.start_time - 1.)
    with pytest.raises(ValueError):
        bt.Integrate(sh, start=sh.start_time + 3.)
    with pytest.raises(AssertionError):
        bt.Integrate(sh, step=36_000_000)                    # an hour of samples
