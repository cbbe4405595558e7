"""Polyphase filter bank on the GPU (reference baseband_tasks/pfb.py:14-154)."""
import numpy as np

from . import hip
from .base import PaddedTaskBase, getattr_if_none, _stream_rate
from .channelize import Dechannelize, _RowFFTTask, _check_n, _prod
from .device_task import DeviceTaskMixin, fetch_device
from .fourier import fft_maker

__all__ = ['sinc_hamming', 'PolyphaseFilterBank', 'PolyphaseFilterBankSamples',
           'InversePolyphaseFilterBank']


def sinc_hamming(n_tap, n_sample, sinc_scale=1.):
    """sinc(n_tap * s * (k/N - 1/2)) * hamming(N), N = n_tap * n_sample,
    shaped ``(n_tap, n_sample)`` (reference pfb.py:14-45)."""
    n = n_tap * n_sample
    x = n_tap * sinc_scale * np.linspace(-0.5, 0.5, n, endpoint=False)
    return (np.sinc(x) * np.hamming(n)).reshape(n_tap, n_sample)


class _PaddedSource(PaddedTaskBase):
    """The inner padded stream of a filter bank (reference pfb.py:80-82): its
    ``task`` is the owning filter bank's ``ppf``."""
    _owner = None

    def task(self, data):
        return self._owner.ppf(data)


class _FewChannelPfbPlan:
    """The filter bank for channel counts the fused FIR + FFT kernels do not take -- n < 256
    (GUPPI's 12 x 64, reference tests/test_pfb.py:33-35), any 2^a 3^b 5^c 7^d that is not a power
    of two, 8192 and 16384 -- with the interface of `hip.PfbPlan`: seen as
    ``(blocks, n * S)`` the stream is n * S streams of blocks, the polyphase sum
    is an ``n_tap``-tap filter along the block axis with its own taps per phase
    (`hip.FirPlan`, the direct filter kernel), and the channelizer transform
    follows (`hip.ChanPlan`, the short-transform kernel)."""

    def __init__(self, response, n_stream):
        response = np.asarray(response, dtype=np.float64)
        self.n_tap, self.n_chan = response.shape
        self.n_stream = int(n_stream)
        # out[i] = sum_t x[i + t] h[t]  ==  FirPlan's sum_k r[k] x[i + n_tap - 1 - k] with r = h reversed
        taps = np.repeat(response[::-1], self.n_stream, axis=1)          # (n_tap, n * S), streams innermost
        self._fir = hip.FirPlan(np.ascontiguousarray(taps, dtype=np.complex64))
        self._fft = hip.ChanPlan(self.n_chan, self.n_stream, -1)

    def execute(self, in_dev, out_dev, n_spectra):
        n, s = self.n_chan, self.n_stream
        rows = n_spectra + self.n_tap - 1
        blocks = hip.DeviceArray((rows, n * s), np.complex64, ptr=in_dev.ptr, owner=in_dev)
        summed = hip.DeviceArray((n_spectra, n * s), np.complex64)
        self._fir.execute(blocks, summed, n_spectra)
        self._fft.execute(summed.reshape(n_spectra * n, s), out_dev, n_spectra)

    def close(self):
        self._fir.close()
        self._fft.close()


class PolyphaseFilterBank(_RowFFTTask):
    """Channelize with a polyphase filter: spectrum ``i`` is the FFT over ``c``
    of ``sum_t x[(i + t) n + c] * response[t, c]`` (the definition in
    reference pfb.py:91-100; the reference's Fourier-domain class,
    pfb.py:103-154, computes the same thing).

    Geometry follows the reference: an inner padded stream (``.padded``) with
    ``(n_tap - 1) n / 2`` samples of padding on each side, channelized with
    ``padded.samples_per_frame // n`` spectra per frame; the time stamp of
    spectrum 0 is therefore ``(n_tap - 1) n / 2`` input samples after the
    start of ``ih``.

    Parameters
    ----------
    ih : stream (complex64)
    response : array (n_tap, n)
    samples_per_frame : int, optional
        Spectra per frame.
    frequency, sideband : optional overrides of the stream metadata.
    """


    def _even(self, count):
        """One stream runs unpadded on the sliding-window kernels (n 256..2048 with
        4, 8, 12 or 16 taps); every other odd count is padded to even."""
        if count == 1 and self._n in (256, 512, 1024, 2048) and self._response.shape[0] in (4, 8, 12, 16):
            return 1
        return count + count % 2

    def __init__(self, ih, response, samples_per_frame=None, frequency=None, sideband=None):
        response = np.asanyarray(response)
        n_tap, n = response.shape
        self._fused_kernel = 256 <= n <= 4096 and not n & (n - 1)
        _check_n(n)                                      # (other counts: filter + transform, two kernels)
        if np.dtype(ih.dtype) not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated filter bank handles complex64 and float32 streams; "
                            f"got {ih.dtype} (wrap the stream in SinglePrecision(...)).")
        self._real = np.dtype(ih.dtype).kind == 'f'
        n_out = n // 2 + 1 if self._real else n
        pad = (n_tap - 1) * n
        assert pad % 2 == 0
        if samples_per_frame is not None:
            samples_per_frame = samples_per_frame * n
        self.padded = _PaddedSource(ih, pad_start=pad // 2, pad_end=pad // 2,
                                    samples_per_frame=samples_per_frame)
        self.padded._owner = self
        if self.padded._ih_samples_per_frame % n:
            raise ValueError("the input block of the polyphase filter "
                             f"({self.padded._ih_samples_per_frame} samples) must be a "
                             f"multiple of n={n}; pass samples_per_frame.")
        self._response = response
        self._source = ih
        rate = _stream_rate(ih)
        frequency = getattr_if_none(ih, 'frequency', frequency, required=False)
        sideband = getattr_if_none(ih, 'sideband', sideband, required=False)
        if frequency is not None:
            fft_freq = (np.fft.rfftfreq if self._real else np.fft.fftfreq)(n, d=1. / rate)
            frequency = frequency + fft_freq.reshape((n_out,) + (1,) * (ih.ndim - 1)) * sideband
        self._setup_streams(n, _prod(ih.shape[1:]))
        self._reshape = (self.padded._ih_samples_per_frame // n, n) + tuple(ih.shape[1:])
        super().__init__(self.padded, shape=(-1, n_out) + tuple(ih.shape[1:]),
                         sample_rate=rate / n,
                         samples_per_frame=self.padded.samples_per_frame // n,
                         frequency=frequency, sideband=sideband, dtype=np.complex64)

    def _get_plan(self):
        if self._plan is None:
            self._plan = self._make_plan(self._n_stream_even)
        return self._plan

    def _split_plan_ok(self, p):
        return self._even(1) == 1                     # (taps, channels) of the sliding-window kernels

    def _make_split_plan(self, p):
        return hip.PfbPlan(self._response, -p)

    def _make_plan(self, n_stream_even):
        if not self._fused_kernel:
            return _FewChannelPfbPlan(self._response, n_stream_even)
        return hip.PfbPlan(self._response, n_stream_even)

    def _compute_frames(self, first, last, out):
        start, stop = self._frame_span(first, last)
        n_spectra = stop - start
        n, n_tap = self._n, self._response.shape[0]
        x = fetch_device(self._source, start * n, (n_spectra + n_tap - 1) * n)
        x = x.reshape((n_spectra + n_tap - 1) * n, self._n_stream)
        s, se = self._n_stream, self._n_stream_even
        if self._real and self._pairs():
            # the taps are real, so the filter bank too takes two real streams as one complex one
            p = self._pairs()
            rows = (n_spectra + n_tap - 1) * n
            self._pair_spectra_to_half(hip.DeviceArray((rows, p), np.complex64, ptr=x.ptr, owner=x),
                                       n_spectra, out)
            return
        if self._real:
            x = hip.real_to_complex(x)
            final, out = out, hip.DeviceArray((n_spectra * n, s), np.complex64)
        flat = out.reshape(n_spectra * n, s)
        if se != s:
            x = hip.pad_streams_to_even(x, s)
            tmp = hip.DeviceArray((n_spectra * n, se), np.complex64)
            self._get_plan().execute(x, tmp, n_spectra)
            hip.strip_stream_pad(tmp, n_spectra * n, s, flat)
        else:
            self._get_plan().execute(x, flat, n_spectra)
        if self._real:
            hip.keep_half_spectrum(flat, n, s, final)

    def _spectra_of_frame(self, data):
        """Filter bank spectra (device) of one padded input frame given on the host."""
        n, n_tap = self._n, self._response.shape[0]
        data = np.ascontiguousarray(data, dtype=np.complex64)      # real input: zero imaginary part
        rows = data.shape[0] // n
        n_spectra = rows + 1 - n_tap
        s, se = self._n_stream, self._n_stream_even
        x = hip.DeviceArray.from_host(data.reshape(rows * n, s))
        if se != s:
            x = hip.pad_streams_to_even(x, s)
        z = hip.DeviceArray((n_spectra * n, se), np.complex64)
        self._get_plan().execute(x, z, n_spectra)
        return z, n_spectra

    def ppf(self, data):
        """Apply the polyphase filter to one padded input frame given on the
        host (the reference's hook, pfb.py:91-100 / 145-154): the filtered time
        stream, ``samples_per_frame * n`` samples.  The filter and the FFT are
        one kernel on the GPU, so this is that kernel followed by the inverse
        channelizer transform."""
        n = self._n
        z, n_spectra = self._spectra_of_frame(data)
        se = self._n_stream_even
        y = hip.DeviceArray((n_spectra * n, se), np.complex64)
        if getattr(self, '_ppf_inverse', None) is None:
            self._ppf_inverse = hip.ChanPlan(n, se, +1)
        self._ppf_inverse.execute(z, y, n_spectra)
        res = y.to_host()[:, :self._n_stream].reshape((n_spectra * n,) + tuple(self.sample_shape[1:]))
        return np.ascontiguousarray(res.real if self._real else res)

    def task(self, data):
        """Channelize one filtered frame given on the host (what
        `Channelize.task` does for the reference's class, channelize.py:73-74)."""
        data = np.ascontiguousarray(data, dtype=np.complex64)
        n, s, se = self._n, self._n_stream, self._n_stream_even
        n_spectra = data.shape[0] // n
        x = hip.DeviceArray.from_host(data.reshape(n_spectra * n, s))
        if se != s:
            x = hip.pad_streams_to_even(x, s)
        if getattr(self, '_task_forward', None) is None:
            self._task_forward = hip.ChanPlan(n, se, -1)
        y = hip.DeviceArray((n_spectra * n, se), np.complex64)
        self._task_forward.execute(x, y, n_spectra)
        z = y.to_host()[:, :s].reshape((n_spectra, n) + tuple(self.sample_shape[1:]))
        return np.ascontiguousarray(z[:, :n // 2 + 1] if self._real else z)

    def close(self):
        super().close()
        self.__dict__.pop('_source', None)
        for name in ('_ppf_inverse', '_task_forward'):
            plan = self.__dict__.pop(name, None)
            if plan is not None:
                plan.close()


#: The GPU evaluates the time-domain definition directly, so both reference
#: classes map to the same task.
PolyphaseFilterBankSamples = PolyphaseFilterBank


class InversePolyphaseFilterBank(DeviceTaskMixin, PaddedTaskBase):
    """Undo a polyphase filter bank by Wiener deconvolution (reference
    pfb.py:157-269): dechannelize, then, frame by frame, FFT along the block
    axis, multiply by ``conj(R) / (|R|^2 + 1/sn^2) * (1 + 1/sn^2)`` with ``R``
    the transform of the zero-extended response, inverse FFT.

    On the GPU this is the overlap-save spectral-multiply plan run along the
    block axis with ``n * prod(sample_shape)`` streams (one response column per
    polyphase phase); a real-valued output (``dtype`` float32, for the half
    spectra of a real stream's filter bank) goes through the same plan with zero
    imaginary parts -- the Wiener filter of real taps is Hermitian.

    Parameters
    ----------
    ih : stream of spectra, shape (n_spec, n, ...)
    response : array (n_tap, n)
    sn : float
        Effective signal-to-noise ratio of the Wiener filter.
    pad_start, pad_end : int
        Extra blocks of padding on each side of a frame (default 128).
    samples_per_frame : int, optional
        Output samples per frame (a multiple of n).
    frequency, sideband, dtype : as for `Dechannelize`.
    """
    _plan = None

    def __init__(self, ih, response, sn, pad_start=128, pad_end=128, samples_per_frame=None,
                 frequency=None, sideband=None, dtype=None):
        response = np.asanyarray(response)
        n_tap, n = response.shape
        dtype = np.dtype(np.complex64 if dtype is None else dtype)
        if dtype.kind == 'f' and dtype.itemsize == 8 or dtype.kind == 'c' and dtype.itemsize == 16:
            dtype = np.dtype(np.float32 if dtype.kind == 'f' else np.complex64)     # computed in single precision
        if dtype not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated inverse filter bank produces complex64 or float32.")
        self._real = dtype.kind == 'f'
        self.dechannelized = Dechannelize(ih, n=n, frequency=frequency, sideband=sideband, dtype=dtype)
        self._response = response
        self._n = n
        pad_minimum = (n_tap - 1) * n
        assert pad_minimum % 2 == 0
        self._FFT = fft_maker.get()
        super().__init__(self.dechannelized, pad_start=pad_start * n + pad_minimum // 2,
                         pad_end=pad_end * n + pad_minimum // 2,
                         samples_per_frame=samples_per_frame, next_fast_len=self.next_fast_len)
        if self._ih_samples_per_frame % n or self.samples_per_frame % n:
            raise ValueError(f"frames must hold whole blocks of n={n} samples.")
        self._reshape = (self._ih_samples_per_frame // n, n) + tuple(self.sample_shape)
        self._inv_sn2 = 1. / (sn * sn)
        self._n_stream = _prod(self.sample_shape)
        self._ft_inverse_cache = None

    def next_fast_len(self, m):
        """A fast length that is a whole number of blocks of n samples, the number
        of blocks itself being a length the engine transforms (the transform here
        runs along the block axis).

        DEVIATION from the reference (pfb.py:236-241), which only rounds the fast
        length up to a multiple of n: its number of blocks can be any integer
        (6208 = 97 x 64 samples for m = 6144 + 1, n = 64), here it is the next
        product of 2, 3, 5, 7 (6272 = 98 x 64).  The Wiener deconvolution is
        circular along the block axis, so with default arguments the frame
        geometry -- and the samples near frame edges, below the padding's
        accuracy -- can differ from the reference's; pass ``samples_per_frame``
        such that ``(samples_per_frame + pad) / n`` is 7-smooth (as the reference's
        own tests do) for identical frames."""
        blocks = -(-self._FFT.next_fast_len(m) // self._n)
        return self._FFT.next_fast_len(blocks) * self._n

    @property
    def _ft_inverse_response(self):
        """Wiener deconvolution filter, shape (n_block, n, 1, ...) (pfb.py:234-246);
        float64 arithmetic, cast to complex64."""
        if self._ft_inverse_cache is None:
            long_response = np.zeros(self._reshape[:2], np.complex128)
            long_response[:self._response.shape[0]] = self._response
            ft_response = np.fft.fft(long_response, axis=0).conj()
            inverse = (ft_response.conj() / (ft_response.real ** 2 + ft_response.imag ** 2
                                             + self._inv_sn2)) * (1 + self._inv_sn2)
            self._ft_inverse_cache = inverse.astype(np.complex64).reshape(
                inverse.shape + (1,) * len(self.sample_shape))
        return self._ft_inverse_cache

    def _get_plan(self):
        if self._plan is None:
            n, s = self._n, self._n_stream
            columns = np.ascontiguousarray(self._ft_inverse_response.reshape(-1, n).T)
            self._plan = hip.OsmPlan(self._reshape[0], n * s, columns,
                                     np.repeat(np.arange(n, dtype=np.int32), s))
        return self._plan

    def _compute_frames(self, first, last, out):
        plan = self._get_plan()
        n, s, spf, n_in = self._n, self._n_stream, self.samples_per_frame, self._ih_samples_per_frame
        frames = np.arange(first, last)
        blocks = [self._block_start(m) for m in frames]
        starts = np.array([b[0] for b in blocks], dtype=np.int64)
        skips = np.array([b[1] for b in blocks], dtype=np.int64)
        counts = np.minimum(spf - skips, self.shape[0] - frames * spf)       # samples kept per frame
        keep = self._pad_start + skips                                        # first kept sample
        off = int(keep[0] % n)                         # same for all frames (skips are whole blocks)
        assert np.all(keep % n == off) and np.all(starts % n == 0)
        n_blk = -(-(off + counts) // n)                                       # blocks computed per frame
        tmp_off = np.concatenate([[0], np.cumsum(n_blk)[:-1]])
        in0 = int(starts[0])
        geo = plan.info()
        n_rows = self._reshape[0]
        flat_ok = (not self._real and geo['n1'] == 1 and 256 <= n_rows <= 4096 and not n_rows & (n_rows - 1)
                   and int(counts.max()) * s < 2**31 and (off * s) % 2 == 0
                   and (spf * s) % 2 == 0 and np.all((counts * s) % 2 == 0))
        x = fetch_device(self.dechannelized, in0, int(starts[-1]) + n_in - in0)
        if self._real:
            x = hip.real_to_complex(x)
        x = x.reshape(x.shape[0] // n, n * s)
        if flat_ok:
            # (bbt_osm_execute_flat takes even element offsets and counts: one stream with an
            # odd number of kept samples -- a short last frame -- takes the route below)
            # one kernel per block: it writes the kept samples -- from the middle of a row of the
            # block axis on -- straight to their place (no intermediate rows, no copy)
            flat = out.reshape(out.shape[0], s)
            plan.execute_flat(x, flat, (starts - in0) // n, (frames * spf - first * spf) * s, keep // n,
                              off * s, counts * s)
            return
        tmp = hip.DeviceArray((int(n_blk.sum()), n * s), np.complex64)
        plan.execute(x, tmp, (starts - in0) // n, tmp_off, keep // n, n_blk)
        tmp = tmp.reshape(int(n_blk.sum()) * n, s)
        if self._real:
            tmp = hip.real_part(tmp)
        flat = out.reshape(out.shape[0], s)
        base = first * spf
        for m, t0, cnt in zip(frames, tmp_off, counts):
            o = int(m * spf - base)
            flat[o:o + int(cnt)].copy_from_device(tmp[int(t0) * n + off:int(t0) * n + off + int(cnt)])

    def task(self, data):
        """Deconvolve one input frame given on the host (the reference's hook,
        pfb.py:255-269): ``_ih_samples_per_frame`` dechannelized samples in,
        ``samples_per_frame`` out, padding removed."""
        plan = self._get_plan()
        n, s, spf, n_in = self._n, self._n_stream, self.samples_per_frame, self._ih_samples_per_frame
        data = np.ascontiguousarray(data, dtype=np.complex64)
        assert data.shape[0] == n_in
        keep = self._pad_start
        off = keep % n
        n_blk = -(-(off + spf) // n)
        x = hip.DeviceArray.from_host(data.reshape(n_in // n, n * s))
        tmp = hip.DeviceArray((n_blk, n * s), np.complex64)
        plan.execute(x, tmp, [0], [0], [keep // n], [n_blk])
        res = tmp.to_host().reshape((n_blk * n,) + tuple(self.sample_shape))
        res = res[off:off + spf]
        return np.ascontiguousarray(res.real if self._real else res)

    def close(self):
        super().close()
        self._drop_cache()
        self._ft_inverse_cache = None
        if self._plan is not None:
            self._plan.close()
            self._plan = None
