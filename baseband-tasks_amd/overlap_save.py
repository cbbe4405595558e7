"""GPU overlap-save tasks of the form  ifft(fft(block) * H)[valid].

Base of `Disperse`/`Dedisperse` (reference dispersion.py:135-139) and
`Convolve` (convolution.py:116-120): the geometry is `PaddedTaskBase`'s, the
arithmetic is one `bbt_osm_execute` call for a whole run of frames.
"""
import os

import numpy as np

from . import hip
from .base import PaddedTaskBase
from .device_task import DeviceTaskMixin, fetch_device
from .fourier import fft_maker

__all__ = ['SpectralMultiplyTask']


def _prod(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class SpectralMultiplyTask(DeviceTaskMixin, PaddedTaskBase):
    """Subclasses provide

    ``_spectral_response()`` -> complex64 array ``(N,) + b`` with ``b``
    broadcastable to the sample shape, in FFT-natural order, unscaled; and
    set ``_keep_from`` (block index of the first kept sample of a regular
    frame: ``pad_start`` for dispersion, ``pad_start + pad_end`` for
    convolution).
    """
    _plan = None
    _keep_from = 0

    def __init__(self, ih, pad_start, pad_end, *, samples_per_frame=None, **kwargs):
        if np.dtype(ih.dtype) not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise TypeError("the accelerated path handles complex64 and float32 streams; got "
                            f"{ih.dtype} (wrap the stream in SinglePrecision(...)).")
        # float32 streams run through the same complex kernels with the
        # Hermitian-extended response (what rfft -> multiply -> irfft computes
        # in the reference).  Its impulse response is real, so it acts on real
        # and imaginary parts separately: two neighbouring real streams that
        # share a response column ARE one complex stream (a reinterpretation of
        # the same bytes, no copies, half the transforms).  Streams that cannot
        # be paired get a zero imaginary part in and drop it on the way out.
        self._real = np.dtype(ih.dtype).kind == 'f'
        self._FFT = fft_maker.get()
        super().__init__(ih, pad_start=pad_start, pad_end=pad_end,
                         samples_per_frame=samples_per_frame,
                         next_fast_len=self._FFT.next_fast_len, **kwargs)
        self._n_stream = _prod(self.sample_shape)
        self._n_stream_even = self._n_stream + (self._n_stream % 2)

    # -- plan ------------------------------------------------------------------
    def _response_columns(self):
        """(C, N) response columns and the column index of every stream."""
        resp = np.asarray(self._spectral_response(), dtype=np.complex64)
        n = self._ih_samples_per_frame
        if self._real:
            # keep the non-negative frequencies (all irfft looks at) and extend
            half = resp[:n // 2 + 1].copy()
            half[0] = half[0].real
            if n % 2 == 0:
                half[-1] = half[-1].real
            resp = np.concatenate([half, half[-2 if n % 2 == 0 else -1:0:-1].conj()])
        assert resp.shape[0] == n
        bshape = resp.shape[1:]
        ncol = _prod(bshape)
        index = np.broadcast_to(np.arange(ncol).reshape(bshape), self.sample_shape).ravel()
        columns = np.ascontiguousarray(resp.reshape(n, ncol).T)
        index = index.astype(np.int32)
        if self._n_stream_even != self._n_stream:
            index = np.concatenate([index, index[-1:]])
        return columns, index

    #: Run pairs of real streams as single complex streams when they share a response.
    PAIR_REAL_STREAMS = True
    _paired = None
    #: Do not pad a lone complex stream to a pair (power-of-two blocks only).
    SINGLE_STREAM_UNPADDED = True
    _single = False

    def _plan_layout(self):
        """(response columns, column index per stream the plan sees, number of
        those streams incl. the pad to even).  For float32 streams 2k, 2k+1
        with the same response the plan sees S/2 complex streams."""
        columns, index = self._response_columns()
        s = self._n_stream
        self._paired = bool(self._real and self.PAIR_REAL_STREAMS and s % 2 == 0
                            and np.array_equal(index[0:s:2], index[1:s:2]))
        if self._paired:
            index = index[0:s:2]
            if index.shape[0] % 2:
                index = np.concatenate([index, index[-1:]])
        # ONE complex stream (or two float32 streams) on power-of-two blocks runs as it is:
        # the library pairs consecutive blocks instead of streams (include/bbt_hip.h)
        n = self._ih_samples_per_frame
        self._single = bool(self.SINGLE_STREAM_UNPADDED and (s // 2 if self._paired else s) == 1
                            and 256 <= n <= (1 << 24) and not n & (n - 1))
        if self._single:
            index = index[:1]
        return columns, np.ascontiguousarray(index, dtype=np.int32), index.shape[0]

    def _get_plan(self):
        if self._plan is None:
            columns, index, n_plan = self._plan_layout()
            self._plan = hip.OsmPlan(self._ih_samples_per_frame, n_plan, columns, index)
        return self._plan

    def _run_plan(self, x, out, n_in, n_out, *block_args, executor=None):
        """plan.execute on (n_in, S) input / (n_out, S) output in this task's
        dtype, taking care of real streams and odd stream counts.
        ``executor(plan, x, target)`` replaces the one ``plan.execute`` call."""
        plan = self._get_plan()
        s = self._n_stream
        if self._real and self._paired:
            # (n, S) float32 == (n, S/2) complex64, byte for byte
            s //= 2
            x = hip.DeviceArray((n_in, s), np.complex64, ptr=x.ptr, owner=x)
            final, out = None, hip.DeviceArray((n_out, s), np.complex64, ptr=out.ptr, owner=out)
        elif self._real:
            x = hip.real_to_complex(x.reshape(n_in, s))
            final, out = out, hip.DeviceArray((n_out, s), np.complex64)
        else:
            final = None
        se = s if self._single else s + s % 2
        if se != s:
            x = hip.pad_streams_to_even(x, s)
            target = hip.DeviceArray((n_out, se), np.complex64)
        else:
            target = out
        if executor is None:
            plan.execute(x, target, *block_args)
        else:
            executor(plan, x, target)
        if se != s:
            hip.strip_stream_pad(target, n_out, s, out)
        if final is not None:
            hip.real_part(out, final)

    # -- frames ------------------------------------------------------------------
    def _block_descriptors(self, first, last):
        """Overlap-save blocks for output frames [first, last): input span
        (start, length) and per-block arrays (input start, absolute output
        sample, first kept block sample, kept count)."""
        spf, n = self.samples_per_frame, self._ih_samples_per_frame
        frames = np.arange(first, last, dtype=np.int64)
        # `PaddedTaskBase._block_start` for all frames at once: the last frame is re-aligned to
        # end with the input and skips what the frame before it has produced
        wanted = frames * spf
        starts = np.minimum(wanted, self.ih.shape[0] - n)
        skips = wanted - starts
        counts = np.minimum(spf - skips, self.shape[0] - frames * spf)
        in0 = int(starts[0])
        return in0, int(starts[-1]) + n - in0, starts, frames * spf, self._keep_from + skips, counts

    #: When the input is a `Convolve` (`Resample`) on its short-block route and both plans work on
    #: the same S >= 4 complex streams, the filtered stream -- which only the two plans see -- is
    #: handed over pair-planar (libbbt_hip: bbt_osm_plan_set_layout): S / 2 arrays of two-stream
    #: samples, so that this task's first column pass reads 256-byte runs of one pair, as it does
    #: for two streams, instead of 16 bytes out of every 8 S-byte row.  Same arithmetic, same
    #: result.  ``BBT_PLANAR=0`` switches it off.
    PLANAR_HANDOVER = os.environ.get('BBT_PLANAR', '1') != '0'

    def _planar_input(self):
        """The upstream short-block convolution that can hand its result over pair-planar, else None."""
        from .convolution import Convolve
        up = self.ih
        if not (self.PLANAR_HANDOVER and isinstance(up, Convolve)) or up.closed:
            return None
        s = self._n_stream
        if self._real or up._real or s % 2 or s < 4 or up._n_stream != s:
            return None
        n = self._ih_samples_per_frame
        if n & (n - 1) or not (1 << 17) <= n <= (1 << 20):
            return None
        if up._short_blocks() is None or self._get_plan().info()['n1'] != 256:
            return None
        return up

    def _input_span(self, first, last):
        in0, in_len = self._block_descriptors(first, last)[:2]
        up = self._planar_input()
        if up is not None:
            return (up.ih,) + up._short_blocks()._span_blocks(in0, in_len)[:2]
        return self.ih, in0, in_len

    def _compute_frames(self, first, last, out):
        in0, in_len, starts, out_abs, keep, counts = self._block_descriptors(first, last)
        out_off = out_abs - first * self.samples_per_frame
        up = self._planar_input()
        if up is not None:
            x = up.read_planar(in0, in_len)                   # (S / 2, in_len, 2)
            plan = self._get_plan()
            plan.set_layout(in_plane=in_len)
            try:
                plan.execute(x, out, starts - in0, out_off, keep, counts)
            finally:
                plan.set_layout()
            return
        x = fetch_device(self.ih, in0, in_len)
        self._run_plan(x, out, in_len, out.shape[0], starts - in0, out_off, keep, counts)

    def _span_blocks(self, start, n_out):
        """Blocks of the ABSOLUTE grid that hold output samples [start, start + n_out): block k
        reads input [k hop, k hop + N) and yields outputs [k hop, (k + 1) hop); past the last
        block that fits the input, one block re-aligned to end with the stream yields the rest
        (as `PaddedTaskBase._block_start` does for frames).  The same sample is always computed
        from the same block, whatever the request -- results do not depend on how a caller cuts
        its reads.  Returns (first input sample, input length, first regular block, number of
        regular whole blocks, [(in_off, out_off, valid_start, count), ...] for the clipped or
        re-aligned blocks), offsets relative to the fetched input / to ``start``."""
        n, pad, keep = self._ih_samples_per_frame, self._pad_start + self._pad_end, self._keep_from
        hop, total = n - pad, self.ih.shape[0]
        stop = start + n_out
        k_fit = (total - n) // hop                       # last block of the grid that fits the input
        k0, k1 = start // hop, (stop - 1) // hop
        in0 = min(k0 * hop, total - n)
        in_end = min(k1 * hop + n, total)
        odd = []

        def clipped(block_in, first_out, last_out):      # outputs [first_out, last_out) of a block at block_in
            lo, hi = max(first_out, start), min(last_out, stop)
            if hi > lo:
                odd.append((block_in - in0, lo - start, keep + lo - block_in, hi - lo))
        ka, kb = k0, min(k1, k_fit)                       # regular blocks ka .. kb
        if ka <= kb and (ka * hop < start or (ka + 1) * hop > stop):
            clipped(ka * hop, ka * hop, (ka + 1) * hop)
            ka += 1
        if ka <= kb and (kb + 1) * hop > stop:
            clipped(kb * hop, kb * hop, (kb + 1) * hop)
            kb -= 1
        if k1 > k_fit:                                    # the end of the stream: one re-aligned block
            clipped(total - n, (k_fit + 1) * hop, total - pad)
        return in0, in_end - in0, ka, max(kb - ka + 1, 0), odd

    def _compute_span(self, start, n_out, out, planar=False):
        """Output samples [start, start + n_out) into ``out``, whatever this task's frame
        boundaries, on the absolute block grid of `_span_blocks` (whole blocks in one launch for
        plans of one kernel, bbt_osm_execute_regular).  For tasks whose result does not depend
        on the block geometry (exact linear convolutions).  ``planar``: ``out`` is (S / 2, n_out, 2),
        one array of two-stream samples per stream pair (complex streams, S even)."""
        hop = self._ih_samples_per_frame - self._pad_start - self._pad_end
        in0, in_len, ka, n_regular, odd = self._span_blocks(start, n_out)
        x = fetch_device(self.ih, in0, in_len)

        def run(plan, x, target):
            if planar:
                plan.set_layout(out_plane=n_out)
            try:
                if n_regular:
                    plan.execute_regular(x, target, n_regular, ka * hop - in0, ka * hop - start, hop,
                                         self._keep_from)
                if odd:
                    plan.execute(x, target, *(list(col) for col in zip(*odd)))
            finally:
                if planar:
                    plan.set_layout()

        self._run_plan(x, out, in_len, n_out, executor=run)

    def close(self):
        super().close()
        self._drop_cache()
        if self._plan is not None:
            self._plan.close()
            self._plan = None

    def task(self, data):
        """Process one input block given on the host (the reference's hook,
        base.py:699-706).  Runs the same kernels on an uploaded copy."""
        n, spf = self._ih_samples_per_frame, self.samples_per_frame
        assert data.shape == (n,) + tuple(self.sample_shape)
        data = np.ascontiguousarray(data, dtype=self.dtype)
        x = hip.DeviceArray.from_host(data.reshape(n, self._n_stream))
        y = hip.DeviceArray((spf, self._n_stream), self.dtype)
        self._run_plan(x, y, n, spf, [0], [0], [self._keep_from], [spf])
        return y.to_host().reshape((spf,) + tuple(self.sample_shape))
