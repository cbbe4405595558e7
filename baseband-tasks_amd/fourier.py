"""FFT-engine selection, mirroring the reference's plugin seam.

The reference lets tasks pick an FFT engine through the `fft_maker` state
(baseband_tasks/fourier/base.py:348-466) and asks the engine for
``next_fast_len`` when sizing overlap-save blocks (base.py:757-758).  This
package has a single engine, the hand-written gfx950 FFT inside
libbbt_hip.so.  Its fast lengths are powers of two, up to 2^24 for
overlap-save blocks.
"""
import operator

import numpy as np

__all__ = ['HipFFTMaker', 'fft_maker', 'FFT_MAKER_CLASSES']

FFT_MAKER_CLASSES = {}

MIN_FFT_LEN = 256
MAX_BLOCK_LEN = 1 << 24
MAX_WG_FFT_LEN = 4096


class FFTMakerBase:
    """Engines register under their lower-cased name minus 'fftmaker'
    (reference fourier/base.py:221-253)."""

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        key = cls.__name__.lower()
        if key.endswith('fftmaker') and len(key) > 8:
            key = key[:-8]
        if key in FFT_MAKER_CLASSES:
            raise ValueError(f"key {key} already registered in FFT_MAKER_CLASSES.")
        FFT_MAKER_CLASSES[key] = cls


class HipFFT:
    """One pre-defined transform (reference fourier/base.py:59-218): FFT along
    ``axis`` of complex64 arrays of ``time_shape``; host in, host out."""

    def __init__(self, time_shape, axis, direction, ortho, sample_rate):
        self.time_shape = self.frequency_shape = tuple(time_shape)
        self.time_dtype = self.frequency_dtype = np.dtype(np.complex64)
        axis = operator.index(axis)
        if not -len(self.time_shape) <= axis < len(self.time_shape):
            raise ValueError(f"axis {axis} is out of bounds for shape {self.time_shape}.")
        axis %= len(self.time_shape)
        self.axis, self.ortho, self.sample_rate = axis, bool(ortho), sample_rate
        self.direction = 'backward' if direction == 'backward' else 'forward'
        n = self.time_shape[axis]
        if n < 2 or n > MAX_WG_FFT_LEN or n & (n - 1):
            raise ValueError("the hip engine transforms power-of-two lengths "
                             f"2..{MAX_WG_FFT_LEN} along an axis (got {n}).")
        self._plan = None

    @property
    def frequency(self):
        rate = 1. if self.sample_rate is None else self.sample_rate
        f = np.fft.fftfreq(self.time_shape[self.axis], d=1. / rate)
        return f.reshape(f.shape + (1,) * (len(self.time_shape) - self.axis - 1))

    def inverse(self):
        return HipFFT(self.time_shape, self.axis,
                      'forward' if self.direction == 'backward' else 'backward',
                      self.ortho, self.sample_rate)

    def __call__(self, a):
        from . import hip
        a = np.ascontiguousarray(a, dtype=np.complex64)
        assert a.shape == self.time_shape
        n = self.time_shape[self.axis]
        outer = int(np.prod(self.time_shape[:self.axis], dtype=np.int64))
        inner = int(np.prod(self.time_shape[self.axis + 1:], dtype=np.int64))
        flat = a.reshape(outer * n, inner)
        odd = inner % 2 == 1
        if odd:
            flat = np.concatenate([flat, np.zeros_like(flat[:, :1])], axis=1)
        streams = flat.shape[1]
        if self._plan is None:
            self._plan = hip.ChanPlan(n, streams, -1 if self.direction == 'forward' else +1)
        din = hip.DeviceArray.from_host(flat)
        dout = hip.DeviceArray(flat.shape, np.complex64)
        self._plan.execute(din, dout, outer)
        res = dout.to_host()
        if odd:
            res = res[:, :inner]
        res = np.ascontiguousarray(res).reshape(self.time_shape)
        if self.ortho:
            res *= np.float32(np.sqrt(n) if self.direction == 'backward' else 1. / np.sqrt(n))
        return res

    def __repr__(self):
        return (f"<HipFFT direction={self.direction},\n    axis={self.axis}, ortho={self.ortho},"
                f" sample_rate={self.sample_rate}\n    Time domain: shape={self.time_shape},"
                f" dtype=complex64\n    Frequency domain: shape={self.frequency_shape},"
                " dtype=complex64>")


class HipFFTMaker(FFTMakerBase):
    """The gfx950 engine."""

    def __call__(self, shape, dtype, direction='forward', axis=0, ortho=False,
                 sample_rate=None):
        if np.dtype(dtype) != np.complex64:
            raise TypeError("the hip engine transforms complex64 data only.")
        return HipFFT(tuple(shape), operator.index(axis), direction, ortho, sample_rate)

    @staticmethod
    def next_fast_len(n):
        """Smallest supported block length >= n (a power of two >= 256)."""
        n = operator.index(n)
        fast = MIN_FFT_LEN
        while fast < n:
            fast *= 2
        if fast > MAX_BLOCK_LEN:
            raise ValueError(f"block length {n} needs a transform of {fast} points; this "
                             f"build supports up to {MAX_BLOCK_LEN} (reduce samples_per_frame "
                             "or the padding).")
        return fast

    def __repr__(self):
        return "HipFFTMaker()"


class _FFTMakerState:
    """`fft_maker.get()` / `fft_maker.set(...)` (reference fourier/base.py:397-466)."""
    system_default = HipFFTMaker()

    def __init__(self):
        self._value = self.system_default

    def get(self):
        return self._value

    def set(self, fft_engine=None, **kwargs):
        if fft_engine is None:
            fft_engine = self.system_default
        elif isinstance(fft_engine, str):
            fft_engine = FFT_MAKER_CLASSES[fft_engine](**kwargs)
        elif not isinstance(fft_engine, FFTMakerBase):
            raise TypeError("Can only set the default to an instance of a FFT maker "
                            "such as HipFFTMaker().")
        elif kwargs:
            raise TypeError("cannot pass keyword arguments except if fft_engine is "
                            "the name of an FFT maker.")
        state, previous = self, self._value
        self._value = fft_engine

        class _Restore:
            def __enter__(self_inner):
                return fft_engine

            def __exit__(self_inner, *exc):
                state._value = previous

            def __repr__(self_inner):
                return f"<ScienceState fft_maker: {fft_engine!r}>"
        return _Restore()

    def __call__(self, shape, dtype, *, direction='forward', axis=0, ortho=False,
                 sample_rate=None):
        return self.get()(shape, dtype, direction=direction, axis=axis, ortho=ortho,
                          sample_rate=sample_rate)


fft_maker = _FFTMakerState()
