"""FFT-engine selection, mirroring the reference's plugin seam.

The reference lets tasks pick an FFT engine through the `fft_maker` state
(baseband_tasks/fourier/base.py:348-466) and asks the engine for
``next_fast_len`` when sizing overlap-save blocks (base.py:757-758).  This
package has a single engine, the hand-written gfx950 FFT inside
libbbt_hip.so.  Like the reference's NumPy engine its fast lengths are the
products of 2, 3, 5 and 7 (fourier/numpy.py:99-126), so default-argument tasks
get the reference's block geometry; powers of two run on the tuned kernels,
other lengths on a generic LDS Stockham transform.
"""
import operator

import numpy as np

__all__ = ['HipFFTMaker', 'fft_maker', 'FFT_MAKER_CLASSES']

FFT_MAKER_CLASSES = {}

MIN_FFT_LEN = 256               # shortest transform the fused (power-of-two) paths take
MAX_BLOCK_LEN = 1 << 26         # overlap-save blocks: two factors of at most MAX_WG_FFT_LEN
MAX_WG_FFT_LEN = 8192           # longest transform done by one workgroup


def is_fast_len(n):
    """True for n = 2^a 3^b 5^c 7^d."""
    n = operator.index(n)
    if n < 1:
        return False
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


def check_transform_length(n, what='transform length'):
    """One-workgroup transforms: n = 2^a 3^b 5^c 7^d up to 8192, and 16384 (csrc/fft_big.hpp)."""
    if n == 2 * MAX_WG_FFT_LEN:
        return
    if n < 2 or n > MAX_WG_FFT_LEN or not is_fast_len(n):
        raise ValueError(f"the hip engine handles {what} n = 2^a 3^b 5^c 7^d with 2 <= n <= "
                         f"{MAX_WG_FFT_LEN}, and {2 * MAX_WG_FFT_LEN}; got {n}.")


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
    ``axis`` of arrays of ``time_shape``; host in, host out.  complex64 time
    data transform to complex64 of the same shape; float32 time data to
    ``n // 2 + 1`` complex64 frequencies (fourier/base.py:313-340), computed as
    the complex transform of the zero-imaginary-part data (forward) or of the
    Hermitian-extended half spectrum (backward)."""

    def __init__(self, time_shape, time_dtype, axis, direction, ortho, sample_rate):
        self.time_shape = tuple(operator.index(d) for d in time_shape)
        self.time_dtype = np.dtype(time_dtype)
        axis = operator.index(axis)
        if not -len(self.time_shape) <= axis < len(self.time_shape):
            raise ValueError(f"axis {axis} is out of bounds for shape {self.time_shape}.")
        axis %= len(self.time_shape)
        self.axis, self.ortho, self.sample_rate = axis, bool(ortho), sample_rate
        self.direction = 'backward' if direction == 'backward' else 'forward'
        n = self.time_shape[axis]
        check_transform_length(n)
        self._real = self.time_dtype.kind == 'f'
        self.frequency_dtype = np.dtype(np.complex64)
        fshape = list(self.time_shape)
        if self._real:
            fshape[axis] = n // 2 + 1
        self.frequency_shape = tuple(fshape)
        self._plan = None

    @property
    def frequency(self):
        rate = 1. if self.sample_rate is None else self.sample_rate
        n = self.time_shape[self.axis]
        f = (np.fft.rfftfreq if self._real else np.fft.fftfreq)(n, d=1. / rate)
        return f.reshape(f.shape + (1,) * (len(self.time_shape) - self.axis - 1))

    def inverse(self):
        return HipFFT(self.time_shape, self.time_dtype, self.axis,
                      'forward' if self.direction == 'backward' else 'backward',
                      self.ortho, self.sample_rate)

    def _transform(self, a, sign):
        """Complex transform of ``a`` (time_shape, complex64) along the axis."""
        from . import hip
        n = self.time_shape[self.axis]
        outer = int(np.prod(self.time_shape[:self.axis], dtype=np.int64))
        inner = int(np.prod(self.time_shape[self.axis + 1:], dtype=np.int64))
        flat = np.ascontiguousarray(a, dtype=np.complex64).reshape(outer * n, inner)
        odd = inner % 2 == 1
        if odd:
            flat = np.concatenate([flat, np.zeros_like(flat[:, :1])], axis=1)
        if self._plan is None:
            self._plan = hip.ChanPlan(n, flat.shape[1], sign)
        din = hip.DeviceArray.from_host(flat)
        dout = hip.DeviceArray(flat.shape, np.complex64)
        self._plan.execute(din, dout, outer)
        res = dout.to_host()
        if odd:
            res = res[:, :inner]
        res = np.ascontiguousarray(res).reshape(self.time_shape)
        if self.ortho:
            res *= np.float32(np.sqrt(n) if sign > 0 else 1. / np.sqrt(n))
        return res

    def __call__(self, a):
        a = np.asanyarray(a)
        n = self.time_shape[self.axis]
        if self.direction == 'forward':
            assert a.shape == self.time_shape
            res = self._transform(a, -1)
            if self._real:
                res = np.ascontiguousarray(np.take(res, range(n // 2 + 1), axis=self.axis))
            return res
        assert a.shape == self.frequency_shape
        if not self._real:
            return self._transform(a, +1)
        # irfft semantics: imaginary parts of the DC (and Nyquist) bins are ignored
        half = np.moveaxis(np.array(a, dtype=np.complex64), self.axis, 0)
        half[0] = half[0].real
        if n % 2 == 0:
            half[-1] = half[-1].real
        full = np.concatenate([half, half[-2 if n % 2 == 0 else -1:0:-1].conj()])
        res = self._transform(np.moveaxis(full, 0, self.axis), +1)
        return np.ascontiguousarray(res.real)

    def __repr__(self):
        return (f"<HipFFT direction={self.direction},\n    axis={self.axis}, ortho={self.ortho},"
                f" sample_rate={self.sample_rate}\n    Time domain: shape={self.time_shape},"
                f" dtype={self.time_dtype}\n    Frequency domain: shape={self.frequency_shape},"
                " dtype=complex64>")


class HipFFTMaker(FFTMakerBase):
    """The gfx950 engine.

    Parameters
    ----------
    power_of_two : bool
        Size overlap-save blocks to the next power of two (>= 256) instead of
        the next product of 2, 3, 5, 7.  Those blocks run on the tuned kernels
        and can host the fused channelizer, but the block geometry -- and with
        it the output at the 1e-3 level, a chirp not being time limited -- then
        differs from the reference's default.  Default False.
    """

    def __init__(self, power_of_two=False):
        self.power_of_two = bool(power_of_two)
        if self.power_of_two:
            self.next_fast_len = self._next_power_of_two

    def __call__(self, shape, dtype, direction='forward', axis=0, ortho=False,
                 sample_rate=None):
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the hip engine transforms complex64 and float32 data only.")
        return HipFFT(tuple(shape), dtype, operator.index(axis), direction, ortho, sample_rate)

    @staticmethod
    def _transformable(n):
        """Can a plan be made for block length ``n``?  Powers of two up to 2^24; other
        products of 2, 3, 5, 7 up to 8192, or beyond that if they split into two factors of
        at most 8192 each (split_7smooth in csrc/bbt_hip.hip).  52 of the 3174 such lengths
        up to 2^26 do not (the smallest: 20 588 575 = 5^2 7^7)."""
        if n & (n - 1) == 0 or n <= 8192:
            return True
        d = int(np.sqrt(n))
        while d * d > n:
            d -= 1
        while d >= 1:
            if n % d == 0:
                return n // d <= 8192
            d -= 1
        return False

    @classmethod
    def next_fast_len(cls, n):
        """Smallest 2^a 3^b 5^c 7^d >= n -- the rule of the reference's NumPy
        engine (fourier/numpy.py:99-126), so that default block lengths agree --
        skipping the few such lengths this library cannot split into two transforms
        (`_transformable`; the reference has no such limit)."""
        fast = cls._next_smooth(n)
        while not cls._transformable(fast):
            fast = cls._next_smooth(fast + 1)
        return fast

    @staticmethod
    def _next_smooth(n):
        n = operator.index(n)
        if n <= 7:
            return n
        best = None
        p7 = 1
        while best is None or p7 < best:
            p57 = p7
            while best is None or p57 < best:
                p357 = p57
                while best is None or p357 < best:
                    # smallest power of two that lifts p357 to >= n
                    need = -(-n // p357)
                    cand = p357 << max(need - 1, 0).bit_length()
                    if best is None or cand < best:
                        best = cand
                    if p357 >= n:
                        break
                    p357 *= 3
                if p57 >= n:
                    break
                p57 *= 5
            if p7 >= n:
                break
            p7 *= 7
        if best > MAX_BLOCK_LEN:
            raise ValueError(f"block length {n} needs a transform of {best} points; this "
                             f"build supports up to {MAX_BLOCK_LEN} (reduce samples_per_frame "
                             "or the padding).")
        return best

    @staticmethod
    def _next_power_of_two(n):
        n = operator.index(n)
        fast = MIN_FFT_LEN
        while fast < n:
            fast *= 2
        if fast > (1 << 24):
            raise ValueError(f"block length {n} needs a transform of {fast} points; the "
                             "power-of-two path supports up to 2^24.")
        return fast

    def __repr__(self):
        return f"HipFFTMaker(power_of_two={self.power_of_two})"


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
