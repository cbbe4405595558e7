"""Known answers of the reference's detection tests on the HIP path (reference
baseband_tasks/tests/test_functions.py:16-195: `Square` and `Power` values,
pointer behaviour, and how frequency / sideband / polarization labels carry
over -- same labels, shapes and refusals).  The reference reads `baseband`'s
sample files; streams of the same shapes are generated here.
"""
import numpy as np
import pytest

import baseband_tasks_amd as bt
from baseband_tasks_amd import units as u

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dada_like():
    """complex64 (16000, 2) at 16 MHz, like the sample DADA file."""
    rng = np.random.default_rng(5)
    data = (rng.integers(-60, 60, size=(16000, 2)) + 1j * rng.integers(-60, 60, size=(16000, 2))).astype(np.complex64)
    fh = bt.StreamGenerator(lambda f: data[f.tell():f.tell() + f.samples_per_frame], data.shape,
                            bt.Time('2013-07-02T01:39:20'), 16 * u.MHz, samples_per_frame=16000, dtype=np.complex64)
    return fh, data


@pytest.fixture(scope='module')
def vdif_like():
    """float32 (40000, 8) at 32 MHz, like the sample VDIF file."""
    rng = np.random.default_rng(6)
    data = rng.choice(np.array([-3.3359, -1., 1., 3.3359], np.float32), size=(40000, 8))
    return bt.StreamGenerator(lambda f: data[f.tell():f.tell() + f.samples_per_frame], data.shape,
                              bt.Time('2014-06-16T05:56:07'), 32 * u.MHz, samples_per_frame=20000, dtype=np.float32)


def empty(shape, dtype=np.complex64, **kwargs):
    return bt.EmptyStreamGenerator(shape, sample_rate=1. * u.Hz, start_time=bt.Time('2018-01-01T00:00:00'),
                                   dtype=dtype, **kwargs)


def test_square_values_and_pointer(dada_like):
    """test_functions.py:19-42."""
    fh, data = dada_like
    want = (data.real.astype(np.float64) ** 2 + data.imag.astype(np.float64) ** 2)
    st = bt.Square(fh)
    everything = st.read()
    assert st.tell() == st.shape[0]
    assert abs((st.time - st.start_time) - st.shape[0] / st.sample_rate) < 1e-9
    assert st.dtype == np.float32 and everything.dtype == np.float32
    assert np.allclose(everything, want, rtol=1e-6)
    st.seek(-3, 2)
    assert st.tell() == st.shape[0] - 3
    tail = st.read()
    assert tail.shape[0] == 3 and np.allclose(tail, want[-3:], rtol=1e-6)
    st.close()


def test_square_carries_labels_and_doubles_polarization(vdif_like):
    """test_functions.py:45-67."""
    labelled = bt.SetAttribute(vdif_like, frequency=311.25 * u.MHz + (np.arange(8.) // 2) * 16. * u.MHz,
                               sideband=np.tile([-1, +1], 4), polarization=np.tile(['L', 'R'], 4))
    st = bt.Square(labelled)
    assert np.array_equal(st.frequency, labelled.frequency) and np.array_equal(st.sideband, labelled.sideband)
    assert np.array_equal(st.polarization, np.tile(['LL', 'RR'], 4))
    st.close()
    bare = bt.Square(vdif_like)
    for name in ('frequency', 'sideband', 'polarization'):
        with pytest.raises(AttributeError):
            getattr(bare, name)


def test_power_values_and_labels(dada_like):
    """test_functions.py:73-95: |X|^2, |Y|^2, Re X conj(Y), Im X conj(Y)."""
    fh, data = dada_like
    r0, i0, r1, i1 = data.view('f4').astype(np.float64).T
    want = np.stack([r0 * r0 + i0 * i0, r1 * r1 + i1 * i1, r0 * r1 + i0 * i1, i0 * r1 - r0 * i1], axis=1)
    pt = bt.Power(fh, polarization=['LL', 'RR', 'LR', 'RL'])
    assert np.array_equal(pt.polarization, np.array(['LL', 'RR', 'LR', 'RL']))
    got = pt.read()
    assert abs((pt.time - fh.start_time) - fh.shape[0] / fh.sample_rate) < 1e-9
    assert pt.dtype == np.float32 and got.dtype == np.float32
    assert np.allclose(got, want, rtol=1e-6, atol=1e-3)
    assert repr(pt).startswith('Power(ih, polarization=')
    pt.close()


def test_power_takes_polarization_from_the_stream(dada_like):
    """test_functions.py:97-109."""
    fh, _ = dada_like
    pt = bt.Power(bt.SetAttribute(fh, polarization=np.array(['L', 'R'])))
    assert np.array_equal(pt.polarization, np.array(['LL', 'RR', 'LR', 'RL']))
    assert repr(pt).startswith('Power(ih)\n')
    pt = bt.Power(bt.SetAttribute(fh, polarization=np.array(['R', 'L'])))
    assert np.array_equal(pt.polarization, np.array(['RR', 'LL', 'RL', 'LR']))
    pt.close()


def test_polarization_on_another_axis():
    """test_functions.py:111-125."""
    eh = empty((10000, 2, 4), polarization=[['L'], ['R']])
    expected = np.array([['LL'], ['RR'], ['LR'], ['RL']])
    assert np.array_equal(bt.Power(eh).polarization, expected)
    detailed = np.array([['LL'] * 4, ['RR'] * 4, ['LR'] * 4, ['RL'] * 4])
    assert np.array_equal(bt.Power(eh, polarization=detailed).polarization, expected)


def test_power_values_with_the_polarization_axis_first():
    """Values for the layout of test_functions.py:111-125 (the reference only checks labels there):
    samples (2 pol, 4 bands); also through Integrate."""
    rng = np.random.default_rng(9)
    data = (rng.standard_normal((600, 2, 4)) + 1j * rng.standard_normal((600, 2, 4))).astype(np.complex64)
    fh = bt.StreamGenerator(lambda f: data[f.tell():f.tell() + f.samples_per_frame], data.shape,
                            bt.Time('2018-01-01T00:00:00'), 1. * u.kHz, samples_per_frame=100, dtype=np.complex64,
                            polarization=[['L'], ['R']])
    x, y = data[:, 0].astype(np.complex128), data[:, 1].astype(np.complex128)
    cross = x * y.conj()
    want = np.stack([np.abs(x) ** 2, np.abs(y) ** 2, cross.real, cross.imag], axis=1)
    pt = bt.Power(fh)
    assert pt.shape == (600, 4, 4)
    assert np.allclose(pt.read(), want, rtol=1e-5, atol=1e-6)
    summed = bt.Integrate(bt.Power(fh), 20).read()
    assert np.allclose(summed, want.reshape(30, 20, 4, 4).mean(1), rtol=1e-5, atol=1e-6)


def test_frequency_and_sideband_survive_power():
    """test_functions.py:127-144 (the reference's regression test for its issue 60)."""
    frequency = np.array([[320.25], [320.25], [336.25], [336.25]]) * u.MHz
    sideband = np.array([[-1], [1], [-1], [1]])
    eh = empty((10000, 4, 2), frequency=frequency, sideband=sideband, polarization=['R', 'L'])
    pt = bt.Power(eh)
    for task in (pt, bt.Power(eh, polarization=pt.polarization)):
        assert np.array_equal(task.polarization, np.array(['RR', 'LL', 'RL', 'LR']))
        assert np.array_equal(task.frequency, eh.frequency) and np.array_equal(task.sideband, eh.sideband)


def test_power_refusals(dada_like, vdif_like):
    """test_functions.py:146-195."""
    fh, _ = dada_like
    with pytest.raises(AttributeError):
        bt.Power(fh)                                              # no polarization anywhere
    with pytest.raises(ValueError):
        bt.Power(fh, polarization=['L'])                          # only one
    with pytest.raises(ValueError):
        bt.Power(fh, polarization=['L', 'L', 'R', 'R'])           # duplicates
    with pytest.raises(ValueError):
        bt.Power(fh, polarization=[['LL'], ['RR'], ['LR'], ['RL']])     # wrong axis
    with pytest.raises(ValueError):
        bt.Power(empty((10000, 2, 4), dtype=np.float32, polarization=[['L'], ['R']]))     # real samples
    sideband = np.array([[-1], [1], [-1], [1]])
    labels = ['RR', 'LL', 'RL', 'LR']
    bad_freq = np.array([[320, 320], [320, 320], [336, 336], [336, 337]]) * u.MHz
    with pytest.raises(ValueError):
        bt.Power(empty((10000, 4, 2), frequency=bad_freq, sideband=sideband), polarization=labels)
    frequency = np.array([[320.25], [320.25], [336.25], [336.25]]) * u.MHz
    bad_side = np.array([[-1, -1], [1, -1], [-1, -1], [1, 1]])
    with pytest.raises(ValueError):
        bt.Power(empty((10000, 4, 2), frequency=frequency, sideband=bad_side), polarization=labels)
    with pytest.raises(AttributeError):
        bt.Power(vdif_like)
    with pytest.raises(ValueError):
        bt.Power(bt.SetAttribute(vdif_like, polarization=np.array(['L', 'R'] * 4)))       # too many
