"""Device-side ingest: packed sampler frames -> complex64 / float32 streams.

SURVEY 8(f) rank 3.  The reference is fed by `baseband` stream readers
(docs/index.rst:26-31), which decode VDIF / DADA payloads on the host; neither
`baseband` nor any sample file is part of the reference tree or of this image.
These readers therefore follow the published formats -- VDIF 1.1.1 (header
words 0-3, extended data version 1/3 sample rate; little-endian 32-bit payload
words with the first sample in the least significant bits; levels as
`baseband` decodes them: 2-bit -3.3359, -1, +1, +3.3359; 4-bit (v - 8) / 2.95;
8-bit (v - 127.5) / 35.5; 16-bit offset binary) and the PSRDADA ASCII header
with signed 8-bit samples -- and are checked only against this package's own
encoder in the tests.  PARITY UNPINNED: nothing here has been compared with
`baseband` output, and nothing can be in this image: the reference's tests take
their VDIF / DADA sample files from that package (tests/common.py:6-25), which
is absent, so no reference-held vector exists for this row.

The raw bytes are uploaded as they are; unpacking happens in HBM
(libbbt_hip: bbt_unpack), so a chain ``Dedisperse(open_vdif(...))`` moves
packed samples over PCIe (16 x fewer bytes than complex64 for 2-bit data).
"""
import operator
import os

import numpy as np

from . import hip
from . import host_pipeline
from . import units as u
from .base import Base
from .device_task import DeviceTaskMixin
from .units import Time

__all__ = ['RawFrameStream', 'open_vdif', 'open_dada', 'vdif_header', 'encode_vdif_frames']


class _RawFrameSets:
    """The packed bytes as a host stream of frame sets (one row = the ``n_thread`` frames of a
    set, uint8): what `host_pipeline.HostUploader` needs to read ahead -- a worker thread copies
    run m + 1 to page-locked memory (or takes it from there, if ``raw`` was pinned with
    `host_pipeline.pin_array`) and uploads it while run m is unpacked and processed."""

    def __init__(self, raw, set_nbytes, n_sets):
        self._rows = raw[:n_sets * set_nbytes].reshape(n_sets, set_nbytes)
        self.shape = self._rows.shape
        self.dtype = np.dtype(np.uint8)
        self._pos = 0

    def seek(self, offset):
        self._pos = int(offset)
        return self._pos

    def read(self, count, out=None):
        rows = self._rows[self._pos:self._pos + count]
        self._pos += count
        if out is None:
            return np.ascontiguousarray(rows)
        out[...] = rows
        return out

    def host_view(self, start, count):
        return self._rows[start:start + count]


class RawFrameStream(DeviceTaskMixin, Base):
    """A stream whose samples sit packed in equally sized frames of ``raw``
    (any buffer: bytes, ``np.memmap``, uint8 array) and are unpacked on the GPU.

    Frame ``f`` is thread ``f % n_thread`` of frame set ``f // n_thread``; a
    frame holds ``samples_per_frame`` complete samples of ``n_chan`` channels
    (complex: I, Q adjacent).  Stream shape ``(n, n_thread, n_chan)`` with unit
    axes dropped (``squeeze``), dtype complex64 or float32.
    """

    def __init__(self, raw, *, frame_nbytes, header_nbytes, samples_per_frame, bits, n_chan=1,
                 n_thread=1, complex_data=False, code=0, start_time, sample_rate, squeeze=True,
                 sample_shape=None, first_byte=0, frame_map=None, **kwargs):
        raw = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8)
        self._raw = raw[first_byte:]
        self._frame_nbytes = operator.index(frame_nbytes)
        self._header_nbytes = operator.index(header_nbytes)
        self._bits, self._code = operator.index(bits), operator.index(code)
        self._n_chan, self._n_thread = operator.index(n_chan), operator.index(n_thread)
        self._n_elem = self._n_chan * (2 if complex_data else 1)
        # frame_map[set, thread] = index of that frame in ``raw``, -1 where the file has none
        # (or only an invalid one): those samples read as 0.  None: frames in order, all valid.
        self._frame_map = None if frame_map is None else np.ascontiguousarray(frame_map, dtype=np.int64)
        if self._frame_map is not None:
            assert self._frame_map.ndim == 2 and self._frame_map.shape[1] == self._n_thread
            n_sets = self._frame_map.shape[0]
        else:
            n_sets = self._raw.shape[0] // (self._frame_nbytes * self._n_thread)
        if n_sets < 1:
            raise ValueError("the buffer holds less than one complete frame set.")
        if sample_shape is None:
            sample_shape = (self._n_thread, self._n_chan)
        assert int(np.prod(sample_shape)) == self._n_thread * self._n_chan
        if squeeze:
            sample_shape = tuple(d for d in sample_shape if d != 1)
        self._sets = None
        if self._frame_map is None and isinstance(self._raw, np.ndarray) and self._raw.flags.c_contiguous:
            self._sets = _RawFrameSets(self._raw, self._frame_nbytes * self._n_thread, n_sets)
        super().__init__(shape=(n_sets * samples_per_frame,) + sample_shape, start_time=start_time,
                         sample_rate=sample_rate, samples_per_frame=samples_per_frame,
                         dtype=np.complex64 if complex_data else np.float32, **kwargs)

    #: frame sets unpacked by one call (bounds the upload)
    max_frames_per_call = 4096

    def _input_span(self, first, last):
        """Frames in file order: the bytes of frame sets [first, last) (host_pipeline read-ahead)."""
        if self._sets is None:
            return None
        return self._sets, first, last - first

    def _compute_frames(self, first, last, out):
        nb = self._frame_nbytes * self._n_thread
        valid_dev = None
        up = host_pipeline.uploader_for(self._sets) if self._sets is not None else None
        if up is not None:
            raw_dev = up.fetch(first, last - first)          # (may be on its way already)
            hip.check(hip.lib().bbt_unpack_masked(raw_dev.ptr, out.ptr, (last - first) * self._n_thread,
                                                  self._frame_nbytes, self._header_nbytes, self._bits,
                                                  self.samples_per_frame, self._n_thread, self._n_elem,
                                                  self._code, None, hip.get_stream()))
            return
        if self._frame_map is None:
            chunk = np.ascontiguousarray(self._raw[first * nb:last * nb])
        else:
            # gather the frames of these sets in (set, thread) order; absent ones stay zero bytes
            index = self._frame_map[first:last].ravel()
            present = index >= 0
            frames = self._raw[:(self._raw.shape[0] // self._frame_nbytes) * self._frame_nbytes]
            frames = frames.reshape(-1, self._frame_nbytes)
            chunk = np.zeros((index.shape[0], self._frame_nbytes), np.uint8)
            chunk[present] = frames[index[present]]
            if not present.all():
                valid_dev = hip.DeviceArray.from_host(present.astype(np.uint8))
        raw_dev = hip.DeviceArray.from_host(chunk)
        hip.check(hip.lib().bbt_unpack_masked(raw_dev.ptr, out.ptr, (last - first) * self._n_thread,
                                              self._frame_nbytes, self._header_nbytes, self._bits,
                                              self.samples_per_frame, self._n_thread, self._n_elem,
                                              self._code, valid_dev.ptr if valid_dev is not None else None,
                                              hip.get_stream()))

    def close(self):
        super().close()
        self._drop_cache()
        self._raw = self._sets = None


# --------------------------------------------------------------------------- VDIF
def vdif_header(words):
    """Fields of a VDIF header from its first 32-bit words (VDIF 1.1.1, section 5)."""
    w = [int(x) for x in words[:8]]
    legacy = bool((w[0] >> 30) & 1)
    h = dict(invalid=bool((w[0] >> 31) & 1), legacy=legacy, seconds=w[0] & 0x3fffffff,
             ref_epoch=(w[1] >> 24) & 0x3f, frame_nr=w[1] & 0xffffff,
             vdif_version=(w[2] >> 29) & 0x7, n_chan=1 << ((w[2] >> 24) & 0x1f),
             frame_nbytes=(w[2] & 0xffffff) * 8, complex_data=bool((w[3] >> 31) & 1),
             bits=((w[3] >> 26) & 0x1f) + 1, thread_id=(w[3] >> 16) & 0x3ff, station=w[3] & 0xffff,
             header_nbytes=16 if legacy else 32, edv=None, sample_rate=None)
    if not legacy and len(w) >= 5:
        h['edv'] = (w[4] >> 24) & 0xff
        if h['edv'] in (1, 3):
            rate = w[4] & 0x7fffff
            h['sample_rate'] = rate * (1e6 if (w[4] >> 23) & 1 else 1e3)
    return h


def _epoch_time(ref_epoch, seconds):
    year, half = 2000 + ref_epoch // 2, ref_epoch % 2
    return Time('%04d-%02d-01T00:00:00' % (year, 1 + 6 * half)) + seconds


def open_vdif(raw, sample_rate=None, **kwargs):
    """`RawFrameStream` for a VDIF file / buffer of equally long frames.

    Every header is read: frames are put in time order by (seconds, frame
    number) and in thread order by thread id, whatever their order in the file;
    frames flagged invalid, and (set, thread) slots the file has no frame for,
    read as zeros (the fill value of `baseband`'s VDIF reader); of duplicates the
    first one counts.  A file that is complete and in order is read without the
    indirection.

    ``sample_rate`` (Hz, per channel; complex samples count once) is taken
    from an EDV 1 / 3 header if not given, else from the number of frames per
    second found in the buffer.  Metadata (``frequency``, ``sideband``,
    ``polarization``) may be passed on.
    """
    if isinstance(raw, (str, os.PathLike)):
        raw = np.memmap(raw, dtype=np.uint8, mode='r')
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8)
    h0 = vdif_header(buf[:32].view('<u4'))
    nb = h0['frame_nbytes']
    if nb <= h0['header_nbytes'] or buf.shape[0] < nb:
        raise ValueError("not a VDIF stream: bad frame length in the first header.")
    n_frames = buf.shape[0] // nb
    heads = buf[:n_frames * nb].reshape(n_frames, nb)[:, :16].copy().view('<u4').astype(np.int64)
    invalid = (heads[:, 0] >> 31) & 1
    secs = heads[:, 0] & 0x3fffffff
    frame_nr = heads[:, 1] & 0xffffff
    length = (heads[:, 2] & 0xffffff) * 8
    bits = ((heads[:, 3] >> 26) & 0x1f) + 1
    thread_id = (heads[:, 3] >> 16) & 0x3ff
    usable = (invalid == 0)
    if np.any((length != nb) | (bits != h0['bits'])):
        raise ValueError("frames of different lengths or sample sizes in one VDIF stream are not supported.")
    if not usable.any():
        raise ValueError("no valid frame in the buffer.")
    payload_bits = (nb - h0['header_nbytes']) * 8
    ncomp = 2 if h0['complex_data'] else 1
    spf = payload_bits // (h0['bits'] * h0['n_chan'] * ncomp)
    if sample_rate is None:
        sample_rate = h0['sample_rate']
    if sample_rate is None:
        # frames per second: highest frame number seen in a second that is complete
        complete = secs < secs.max()
        if not complete.any():
            raise ValueError("cannot infer the sample rate from less than a second of frames; "
                             "pass sample_rate.")
        sample_rate = float((frame_nr[complete].max() + 1) * spf)
    rate = u.to_hz(sample_rate)
    fps = int(round(rate / spf))
    # position of every frame: frame set (time) and thread
    threads = np.unique(thread_id[usable])
    n_thread = threads.shape[0]
    when = secs * fps + frame_nr
    first = when[usable].min()
    which_set = when - first
    which_thread = np.searchsorted(threads, thread_id)
    n_sets = int(which_set[usable].max()) + 1
    frame_map = np.full((n_sets, n_thread), -1, np.int64)
    take = usable & (which_set >= 0)
    # later duplicates must not overwrite earlier ones: assign in reverse file order
    order = np.nonzero(take)[0][::-1]
    frame_map[which_set[order], which_thread[order]] = order
    in_order = n_sets * n_thread == n_frames and np.array_equal(frame_map.ravel(), np.arange(n_frames))
    start = _epoch_time(h0['ref_epoch'], int(first // fps)) + int(first % fps) * spf / rate
    return RawFrameStream(buf[:n_frames * nb], frame_nbytes=nb, header_nbytes=h0['header_nbytes'],
                          samples_per_frame=spf, bits=h0['bits'], n_chan=h0['n_chan'],
                          n_thread=n_thread, complex_data=h0['complex_data'], code=0,
                          start_time=start, sample_rate=sample_rate,
                          frame_map=None if in_order else frame_map, **kwargs)


def encode_vdif_frames(data, bits, *, seconds=0, ref_epoch=40, frame_nr0=0, frames_per_second=None,
                       samples_per_frame=None, edv=0, sample_rate=None, station=0):
    """Pack ``data`` (n, n_thread, n_chan) float32 or complex64, already in
    decoder levels, into VDIF frames (bytes) -- the inverse of `open_vdif`'s
    decoding, used by the tests and for writing synthetic files."""
    data = np.asarray(data)
    complex_data = data.dtype.kind == 'c'
    n, n_thread, n_chan = data.shape
    comp = data.astype(np.complex64).view(np.float32).reshape(n, n_thread, n_chan * 2) \
        if complex_data else data.astype(np.float32)
    if bits == 1:
        codes = (comp > 0).astype(np.uint32)
    elif bits == 2:
        codes = np.searchsorted(np.array([-2., 0., 2.]), comp).astype(np.uint32)
    elif bits == 4:
        codes = np.clip(np.rint(comp * 2.95 + 8.), 0, 15).astype(np.uint32)
    elif bits == 8:
        codes = np.clip(np.rint(comp * 35.5 + 127.5), 0, 255).astype(np.uint32)
    else:
        codes = np.clip(np.rint(comp + (1 << (bits - 1))), 0, (1 << bits) - 1).astype(np.uint32)
    spf = n if samples_per_frame is None else samples_per_frame
    assert n % spf == 0 and (spf * comp.shape[2] * bits) % 64 == 0
    per_word = 32 // bits
    out = bytearray()
    for s in range(n // spf):
        fnr = frame_nr0 + s
        sec = seconds + (fnr // frames_per_second if frames_per_second else 0)
        fnr = fnr % frames_per_second if frames_per_second else fnr
        for t in range(n_thread):
            c = codes[s * spf:(s + 1) * spf, t].reshape(-1, per_word)
            words = np.zeros(c.shape[0], np.uint32)
            for k in range(per_word):
                words |= c[:, k] << np.uint32(k * bits)
            nb = 32 + words.nbytes
            w = np.zeros(8, np.uint32)
            w[0] = sec & 0x3fffffff
            w[1] = (ref_epoch << 24) | (fnr & 0xffffff)
            w[2] = (int(np.log2(n_chan)) << 24) | (nb // 8)
            w[3] = (int(complex_data) << 31) | ((bits - 1) << 26) | (t << 16) | station
            if edv in (1, 3):
                w[4] = (edv << 24) | (1 << 23) | int(round(sample_rate / 1e6))
            out += w.astype('<u4').tobytes() + words.astype('<u4').tobytes()
    return bytes(out)


# --------------------------------------------------------------------------- DADA
def open_dada(raw, **kwargs):
    """`RawFrameStream` for a PSRDADA file / buffer: ASCII header of HDR_SIZE
    bytes (keys NBIT, NDIM, NPOL, NCHAN, TSAMP in us, UTC_START, OBS_OFFSET),
    then samples ordered (time, pol, channel[, re / im]) as signed integers."""
    if isinstance(raw, (str, os.PathLike)):
        raw = np.memmap(raw, dtype=np.uint8, mode='r')
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8)
    text = bytes(buf[:4096]).split(b'\0')[0].decode('ascii', errors='replace')
    hdr = {}
    for line in text.splitlines():
        parts = line.split('#')[0].split(None, 1)
        if len(parts) == 2:
            hdr[parts[0]] = parts[1].strip()
    hdr_size = int(hdr.get('HDR_SIZE', 4096))
    bits, ndim = int(hdr['NBIT']), int(hdr.get('NDIM', 1))
    npol, nchan = int(hdr.get('NPOL', 1)), int(hdr.get('NCHAN', 1))
    if bits not in (8, 16) or ndim not in (1, 2):
        raise ValueError(f"DADA files with NBIT={bits}, NDIM={ndim} are not handled here.")
    sample_rate = 1e6 / float(hdr['TSAMP'])
    sample_bytes = npol * nchan * ndim * bits // 8
    offset = int(hdr.get('OBS_OFFSET', 0))
    utc = hdr['UTC_START'].replace('-', 'T', 3).replace('T', '-', 2) if hdr['UTC_START'].count('-') == 3 \
        else hdr['UTC_START']
    start = Time(utc) + (offset / sample_bytes) / sample_rate
    n_samples = (buf.shape[0] - hdr_size) // sample_bytes
    # one "frame" = a run of complete samples (whole 32-bit words), no per-frame header
    spf = 1
    while spf < 65536 and n_samples % (spf * 2) == 0:
        spf *= 2
    while (spf * sample_bytes) % 4:
        spf *= 2
    n_samples -= n_samples % spf
    if n_samples <= 0:
        raise ValueError("the file holds no whole frame of samples.")
    return RawFrameStream(buf[hdr_size:hdr_size + n_samples * sample_bytes], frame_nbytes=spf * sample_bytes,
                          header_nbytes=0, samples_per_frame=spf, bits=bits, n_chan=npol * nchan,
                          n_thread=1, complex_data=ndim == 2, code=1, start_time=start,
                          sample_rate=sample_rate, sample_shape=(npol, nchan), **kwargs)
