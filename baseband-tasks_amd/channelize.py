"""Block-FFT channelizer and its inverse on the GPU (reference
baseband_tasks/channelize.py:12-178)."""
import operator
import os

import numpy as np

from . import hip
from .base import TaskBase, getattr_if_none, _stream_rate
from .device_task import DeviceTaskMixin, fetch_device
from .fourier import MIN_FFT_LEN, check_transform_length

__all__ = ['Channelize', 'Dechannelize', 'FUSE_WITH_OVERLAP_SAVE', 'FUSE_DETECTION']

#: When the input of a `Channelize` is a GPU overlap-save task (`Dedisperse`,
#: `Disperse`, `Convolve`, `Resample`) the channelizer FFT is folded into that
#: task's row pass (libbbt_hip: bbt_osm_execute_channelized) and the
#: intermediate stream never exists in memory.  Results are the same to
#: rounding; set to False (or BBT_FUSE=0) to run the two tasks separately.
FUSE_WITH_OVERLAP_SAVE = os.environ.get('BBT_FUSE', '1') != '0'
#: Let Integrate(Square|Power(Channelize(overlap-save task))) run inside that task's last pass.
FUSE_DETECTION = os.environ.get('BBT_FUSE_DETECT', '1') != '0'


def _prod(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def _check_n(n):
    """Channel counts the kernels take: any 2^a 3^b 5^c 7^d up to 8192, and 16384."""
    check_transform_length(n, 'channel counts')


class _RowFFTTask(DeviceTaskMixin, TaskBase):
    """Shared frame computation: ``in_per_out`` input samples per output
    sample (n for channelizing, 1/n for the inverse)."""
    _plan = None
    _direction = -1

    #: the plain transform kernels take ONE complex stream as it is (two consecutive
    #: transforms side by side instead of a stream pair) for power-of-two n in
    #: [256, 4096]; other odd stream counts are padded to even
    _SINGLE_STREAM = True
    #: the plain channelizer's kernel can separate the spectra of two real streams itself
    _SPLIT_IN_TRANSFORM = True

    def _even(self, count):
        if count == 1 and self._SINGLE_STREAM and 256 <= self._n <= 4096 and not self._n & (self._n - 1):
            return 1
        return count + count % 2

    def _setup_streams(self, n, n_stream):
        self._n = n
        self._n_stream = n_stream
        self._n_stream_even = self._even(n_stream)

    #: Transform two neighbouring real streams as one complex stream a + i b
    #: (one transform for two: the spectra are separated afterwards).
    PAIR_REAL_STREAMS = True
    _pair_plan = None

    def _get_plan(self):
        if self._plan is None:
            self._plan = hip.ChanPlan(self._n, self._n_stream_even, self._direction)
        return self._plan

    def _pairs(self):
        """Number of complex streams when real streams go in pairs, else 0."""
        return self._n_stream // 2 if (self.PAIR_REAL_STREAMS and self._n_stream % 2 == 0) else 0

    def _run_pairs(self, x, n_spectra, out_flat):
        """As `_run` for the S/2 complex streams that pairs of real streams form."""
        p = self._pairs()
        pe = self._even(p)
        if self._pair_plan is None:
            self._pair_plan = self._make_plan(pe)
        if pe != p:
            x = hip.pad_streams_to_even(x, p)
            tmp = hip.DeviceArray((n_spectra * self._n, pe), np.complex64)
            self._pair_plan.execute(x, tmp, n_spectra)
            hip.strip_stream_pad(tmp, n_spectra * self._n, p, out_flat)
        else:
            self._pair_plan.execute(x, out_flat, n_spectra)

    def _pair_spectra_to_half(self, x, n_spectra, out):
        """Real streams in pairs: transform (n_rows, S/2) complex ``x`` and write
        the half spectra (n_spectra, n/2+1, S) of the S real streams to ``out``.
        An odd number of pairs goes through padded; the split reads the padded
        spectra directly."""
        p = self._pairs()
        pe = self._even(p)
        if self._SPLIT_IN_TRANSFORM and pe == p and self._split_plan_ok(p):
            # pairs of real streams = complex streams: the transform kernel pairs k with n - k
            # itself and writes the half spectra (no separate pass over the spectra)
            if self._pair_plan is None:
                self._pair_plan = self._make_split_plan(p)
            self._pair_plan.execute(x, out, n_spectra)
            return
        if self._pair_plan is None:
            self._pair_plan = self._make_plan(pe)
        if pe != p:
            x = hip.pad_streams_to_even(x, p)
        z = hip.DeviceArray((n_spectra * self._n, pe), np.complex64)
        self._pair_plan.execute(x, z, n_spectra)
        hip.split_real_pair_spectra(z, self._n, self._n_stream, out, padded=pe != p)

    def _make_plan(self, n_stream_even):
        return hip.ChanPlan(self._n, n_stream_even, self._direction)

    def _split_plan_ok(self, p):
        return 256 <= self._n <= 4096 and not self._n & (self._n - 1)

    def _make_split_plan(self, p):
        """Plan that takes streams z = a + i b of two real streams each and writes their half spectra."""
        return hip.ChanPlan(self._n, p, -2)

    def _run(self, x, n_spectra, out_flat):
        """x: (n_spectra * n, S) -> out_flat: (n_spectra * n, S)."""
        s, se = self._n_stream, self._n_stream_even
        if se != s:
            x = hip.pad_streams_to_even(x, s)
            tmp = hip.DeviceArray((n_spectra * self._n, se), np.complex64)
            self._get_plan().execute(x, tmp, n_spectra)
            hip.strip_stream_pad(tmp, n_spectra * self._n, s, out_flat)
        else:
            self._get_plan().execute(x, out_flat, n_spectra)

    def close(self):
        super().close()
        self._drop_cache()
        if self._plan is not None:
            self._plan.close()
            self._plan = None
        if self._pair_plan is not None:
            self._pair_plan.close()
            self._pair_plan = None


class Channelize(_RowFFTTask):
    """Fourier transform blocks of ``n`` samples: output sample shape
    ``(n,) + ih.sample_shape``, sample rate ``ih.sample_rate / n``, channel
    frequencies in `numpy.fft.fftfreq` order (unnormalised forward FFT, no
    shift) -- reference channelize.py:50-74.

    Parameters
    ----------
    ih : stream (complex64)
    n : int
        Channels: any product of 2, 3, 5, 7 up to 8192, and 16384 (powers of two up to
        4096 run on the tuned kernels and, from 256, fuse into an upstream
        overlap-save task).
    samples_per_frame : int
        Spectra per frame (default 1); only affects framing.
    frequency, sideband : optional overrides of the stream metadata.
    """

    def __init__(self, ih, n, samples_per_frame=1, *, frequency=None, sideband=None):
        n = operator.index(n)
        samples_per_frame = operator.index(samples_per_frame)
        if np.dtype(ih.dtype) not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated channelizer handles complex64 and float32 streams; "
                            f"got {ih.dtype} (wrap the stream in SinglePrecision(...)).")
        _check_n(n)
        # real streams: n // 2 + 1 channels (rfft); computed as the complex
        # transform of the zero-extended stream, upper half dropped
        self._real = np.dtype(ih.dtype).kind == 'f'
        n_out = n // 2 + 1 if self._real else n
        rate = _stream_rate(ih)
        frequency = getattr_if_none(ih, 'frequency', frequency, required=False)
        sideband = getattr_if_none(ih, 'sideband', sideband, required=False)
        if frequency is not None:
            fft_freq = (np.fft.rfftfreq if self._real else np.fft.fftfreq)(n, d=1. / rate)
            frequency = frequency + fft_freq.reshape((n_out,) + (1,) * (ih.ndim - 1)) * sideband
        self._setup_streams(n, _prod(ih.shape[1:]))
        super().__init__(ih, shape=(-1, n_out) + tuple(ih.shape[1:]), sample_rate=rate / n,
                         samples_per_frame=samples_per_frame, frequency=frequency,
                         sideband=sideband, dtype=np.complex64)

    @property
    def max_frames_per_call(self):
        """As `DeviceTaskMixin.max_frames_per_call`; a channelizer that is folded into an upstream
        overlap-save task computes whole blocks of that task with every call, so a call covers at
        least two of them (with the default bound of 512 MiB a call of 2048 streams on 2^16-sample
        blocks covered 0.6 of a block, and every block was computed twice)."""
        base = DeviceTaskMixin.max_frames_per_call.fget(self)
        if self._max_frames_per_call is None:
            try:
                dd = self._fusable_input()
            except Exception:
                dd = None
            if dd is not None:
                base = max(base, -(-2 * int(dd.samples_per_frame) // (self._n * int(self.samples_per_frame))))
        return base

    @max_frames_per_call.setter
    def max_frames_per_call(self, value):
        self._max_frames_per_call = None if value is None else int(value)

    def _fusable_input(self):
        """The upstream overlap-save task if its row pass can take over the
        channelizer FFT, else None."""
        from .overlap_save import SpectralMultiplyTask
        dd = self.ih
        if not (FUSE_WITH_OVERLAP_SAVE and isinstance(dd, SpectralMultiplyTask)) or dd.closed:
            return None
        if self._real != dd._real:
            return None
        if dd.samples_per_frame < self._n:
            return None
        plan = dd._get_plan()
        if dd._real:
            # real streams in pairs are complex streams to the plan (a + i b); their spectra are
            # separated afterwards, so the pair route of both tasks must be on
            if not (dd._paired and self._pairs()):
                return None
            n_plan = dd._n_stream // 2
            if n_plan % 2 and not dd._single:
                return None
        elif dd._n_stream != dd._n_stream_even and not dd._single:    # (one stream runs unpadded)
            return None
        if not plan.fusable(self._n):
            return None
        return dd

    def _input_span(self, first, last):
        start, stop = self._frame_span(first, last)
        n = self._n
        dd = self._fusable_input()
        if dd is not None:
            spf = dd.samples_per_frame
            m0, m1 = (start * n) // spf, (stop * n - 1) // spf + 1
            in0, in_len, _, _, _, counts = dd._block_descriptors(m0, m1)
            if np.all(counts >= n):
                return dd.ih, in0, in_len
        return self.ih, start * n, (stop - start) * n

    def _compute_frames(self, first, last, out):
        start, stop = self._frame_span(first, last)
        n_spectra = stop - start
        n = self._n
        flat = None if self._real else out.reshape(n_spectra * n, self._n_stream)
        dd = self._fusable_input()
        if dd is not None:
            spf = dd.samples_per_frame
            m0, m1 = (start * n) // spf, (stop * n - 1) // spf + 1
            in0, in_len, starts, out_abs, keep, counts = dd._block_descriptors(m0, m1)
            if np.all(counts >= n):      # (a short final frame cannot host a whole spectrum)
                x = fetch_device(dd.ih, in0, in_len)
                if self._real:
                    # (n, S) float32 == (n, S/2) complex64 z = a + i b: spectra of z from the fused plan,
                    # then the half spectra of a and b
                    p = self._n_stream // 2
                    x = hip.DeviceArray((in_len, p), np.complex64, ptr=x.ptr, owner=x)
                    z = hip.DeviceArray((n_spectra * n, p), np.complex64)
                    dd._get_plan().execute_channelized(x, z, starts - in0, out_abs, keep, counts, n,
                                                       start, n_spectra)
                    hip.split_real_pair_spectra(z, n, self._n_stream, out)
                    return
                dd._get_plan().execute_channelized(x, flat, starts - in0, out_abs, keep, counts, n,
                                                   start, n_spectra)
                return
        x = fetch_device(self.ih, start * n, n_spectra * n).reshape(n_spectra * n, self._n_stream)
        if self._real and self._pairs():
            p = self._pairs()            # (n, S) float32 == (n, S/2) complex64, byte for byte
            self._pair_spectra_to_half(hip.DeviceArray((n_spectra * n, p), np.complex64, ptr=x.ptr, owner=x),
                                       n_spectra, out)
            return
        if self._real:
            full = hip.DeviceArray((n_spectra * n, self._n_stream), np.complex64)
            self._run(hip.real_to_complex(x), n_spectra, full)
            hip.keep_half_spectrum(full, n, self._n_stream, out)
            return
        self._run(x, n_spectra, flat)

    def _compute_detected(self, first_spectrum, n_bins, step, mode, average, out):
        """Detect (mode 0 Square, 1 Power) and integrate spectra
        [first_spectrum, first_spectrum + n_bins * step) inside the upstream
        overlap-save plan, without storing them.  Returns False when that
        route does not apply (the caller then detects the stored spectra)."""
        dd = self._fusable_input() if FUSE_DETECTION else None
        if dd is None or dd._single or dd._ih_samples_per_frame > (1 << 20) or self._n < MIN_FFT_LEN:
            return False
        plan = dd._get_plan()
        if plan.info()['n1'] != 256 or (step > 1 and plan.detect_bins_max(self._n, step) > 64):
            return False
        n, spf = self._n, dd.samples_per_frame
        start, stop = first_spectrum, first_spectrum + n_bins * step
        m0, m1 = (start * n) // spf, (stop * n - 1) // spf + 1
        in0, in_len, starts, out_abs, keep, counts = dd._block_descriptors(m0, m1)
        if not np.all(counts >= n):
            return False
        x = fetch_device(dd.ih, in0, in_len)
        plan.execute_channelized_detect(x, out, starts - in0, out_abs, keep, counts, n, start,
                                        n_bins, step, mode, average)
        return True

    def task(self, data):
        """Channelize one frame given on the host (reference channelize.py:73-74)."""
        data = np.ascontiguousarray(data, dtype=np.complex64)       # real input: zero imaginary part
        n_spectra = data.shape[0] // self._n
        x = hip.DeviceArray.from_host(data.reshape(n_spectra * self._n, self._n_stream))
        y = hip.DeviceArray(x.shape, np.complex64)
        self._run(x, n_spectra, y)
        y = y.to_host().reshape((n_spectra, self._n) + tuple(self.sample_shape[1:]))
        return np.ascontiguousarray(y[:, :self._n // 2 + 1]) if self._real else y

    def inverse(self, ih):
        """`Dechannelize` that undoes this channelization."""
        return Dechannelize(ih, n=self._n, dtype=np.float32 if self._real else None)


class Dechannelize(_RowFFTTask):
    """Inverse FFT over the channel axis, back to a time stream (complex
    output only; reference channelize.py:90-178)."""
    _direction = +1

    def __init__(self, ih, n=None, samples_per_frame=None, *, dtype=None, frequency=None,
                 sideband=None):
        assert np.dtype(ih.dtype).kind == 'c', "Dechannelization needs complex spectra."
        dtype = np.dtype(np.complex64 if dtype is None else dtype)
        if dtype not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated dechannelizer produces complex64 or float32.")
        self._real = dtype.kind == 'f'
        if n is None:
            if self._real:
                raise ValueError("need explicit 'n' for real transform.")
            n = ih.shape[1]
        n = operator.index(n)
        # (like the reference, a channel count that does not fit n is only an error once data are read)
        self._mismatch = None
        if ih.shape[1] != (n // 2 + 1 if self._real else n):
            self._mismatch = f"{ih.shape[1]} channels do not match n={n} for {dtype} output."
        _check_n(n)
        if samples_per_frame is None:
            ih_spf = ih.samples_per_frame
        else:
            ih_spf = max(int(round(samples_per_frame / n)), 1)
        if frequency is None and getattr(ih, 'frequency', None) is not None:
            frequency = ih.frequency[0] if np.ndim(ih.frequency) >= len(ih.shape) - 1 \
                else ih.frequency
        self._setup_streams(n, _prod(ih.shape[2:]))
        super().__init__(ih, shape=(-1,) + tuple(ih.shape[2:]),
                         sample_rate=_stream_rate(ih) * n, ih_samples_per_frame=ih_spf,
                         frequency=frequency, sideband=sideband, dtype=dtype)

    def _spectra_to_stream(self, x, n_spectra, out):
        n, s = self._n, self._n_stream
        p = self._pairs() if self._real else 0
        if p and self._even(p) == p and 256 <= n <= 4096 and not n & (n - 1):
            # half spectra of pairs of real streams -> streams z = a + i b in one pass (the kernel mirrors k > n/2)
            if self._pair_plan is None:
                self._pair_plan = hip.ChanPlan(n, p, +2)
            self._pair_plan.execute(x, hip.DeviceArray((n_spectra * n, p), np.complex64, ptr=out.ptr, owner=out),
                                    n_spectra)
            return
        if self._real and self._pairs():
            p = self._pairs()
            z = hip.merge_real_pair_spectra(x, n, s, hip.DeviceArray((n_spectra * n, p), np.complex64))
            self._run_pairs(z, n_spectra,
                            hip.DeviceArray((n_spectra * n, p), np.complex64, ptr=out.ptr, owner=out))
            return
        if self._real:
            full = hip.half_to_full_spectrum(x, n, s).reshape(n_spectra * n, s)
            tmp = hip.DeviceArray((n_spectra * n, s), np.complex64)
            self._run(full, n_spectra, tmp)
            hip.real_part(tmp, out)
        else:
            self._run(x.reshape(n_spectra * n, s), n_spectra, out.reshape(n_spectra * n, s))

    def _input_span(self, first, last):
        start, stop = self._frame_span(first, last)
        return self.ih, start // self._n, (stop - start) // self._n

    def _compute_frames(self, first, last, out):
        if self._mismatch:
            raise ValueError(self._mismatch)
        start, stop = self._frame_span(first, last)
        n_spectra = (stop - start) // self._n
        x = fetch_device(self.ih, start // self._n, n_spectra)
        self._spectra_to_stream(x, n_spectra, out)

    def task(self, data):
        data = np.ascontiguousarray(data, dtype=np.complex64)
        n_spectra = data.shape[0]
        out = hip.DeviceArray((n_spectra * self._n,) + tuple(self.sample_shape), self.dtype)
        self._spectra_to_stream(hip.DeviceArray.from_host(data), n_spectra, out)
        return out.to_host()

    def inverse(self, ih):
        return Channelize(ih, n=self._n)
