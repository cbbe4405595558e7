"""Sink (and matching source) for the reference's intermediate HDF5 format.

SURVEY 8(f) rank 3, the downstream side: `baseband_tasks.io.hdf5` stores a
stream as one HDF5 file with two datasets -- ``header``, a YAML text holding
``sample_shape, samples_per_frame, sample_rate, time, dtype`` and the optional
``frequency, sideband, polarization`` (io/hdf5/header.py:23-130, written with
astropy's YAML dumper), and ``payload``, the samples as a plain array
(io/hdf5/payload.py:121-178; complex numbers as h5py stores them, a compound
of two floats ``r`` and ``i``) -- and copies a stream into it through
``fw.write(data)`` / ``fw[a:b] = data`` (io/hdf5/base.py:102-126, which is
what `integration.Integrate` and friends write through).

Neither h5py nor astropy is available where this package runs, so the file is
written directly: the subset of the HDF5 file format needed for those two
datasets (version-2 superblock, version-2 object headers with compact links,
contiguous layouts, Jenkins lookup3 checksums), the YAML text in the tags
astropy's loader resolves (``!astropy.units.Quantity``, ``!astropy.time.Time``,
``!numpy.ndarray``).  tests/test_hdf5.py reads such files back with the real
h5py + astropy YAML loader (in the build container's second interpreter, the
way the reference's ``HDF5Header.fromfile`` does); the reference's own reader
class cannot be run here (it imports `baseband`).  `open(name, 'r')` reads the
files this module writes (not HDF5 files in general).
"""
import base64
import os
import struct

import numpy as np

from . import units as u
from .base import Base
from .units import Time

__all__ = ['open', 'HDF5StreamWriter', 'HDF5StreamReader', 'header_yaml', 'lookup3']

_UNDEF = 0xFFFFFFFFFFFFFFFF
_SIGNATURE = b'\x89HDF\r\n\x1a\n'
_M = 0xFFFFFFFF


# --------------------------------------------------------------------------- checksums
def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & _M


def lookup3(data, initval=0):
    """Bob Jenkins' lookup3 `hashlittle`, the metadata checksum of the HDF5 format."""
    data = bytes(data)
    length = len(data)
    a = b = c = (0xdeadbeef + length + initval) & _M
    i = 0
    while length > 12:
        a = (a + int.from_bytes(data[i:i + 4], 'little')) & _M
        b = (b + int.from_bytes(data[i + 4:i + 8], 'little')) & _M
        c = (c + int.from_bytes(data[i + 8:i + 12], 'little')) & _M
        a = (a - c) & _M; a ^= _rot(c, 4); c = (c + b) & _M
        b = (b - a) & _M; b ^= _rot(a, 6); a = (a + c) & _M
        c = (c - b) & _M; c ^= _rot(b, 8); b = (b + a) & _M
        a = (a - c) & _M; a ^= _rot(c, 16); c = (c + b) & _M
        b = (b - a) & _M; b ^= _rot(a, 19); a = (a + c) & _M
        c = (c - b) & _M; c ^= _rot(b, 4); b = (b + a) & _M
        i += 12
        length -= 12
    if length == 0:
        return c
    tail = data[i:] + b'\0' * (12 - length)
    a = (a + int.from_bytes(tail[0:4], 'little')) & _M
    b = (b + int.from_bytes(tail[4:8], 'little')) & _M
    c = (c + int.from_bytes(tail[8:12], 'little')) & _M
    c ^= b; c = (c - _rot(b, 14)) & _M
    a ^= c; a = (a - _rot(c, 11)) & _M
    b ^= a; b = (b - _rot(a, 25)) & _M
    c ^= b; c = (c - _rot(b, 16)) & _M
    a ^= c; a = (a - _rot(c, 4)) & _M
    b ^= a; b = (b - _rot(a, 14)) & _M
    c ^= b; c = (c - _rot(b, 24)) & _M
    return c


# --------------------------------------------------------------------------- HDF5 structures
def _message(kind, data, flags=0):
    return struct.pack('<BHB', kind, len(data), flags) + data


def _object_header(messages, room=0):
    body = b''.join(messages)
    if room:
        body += _message(0, b'\0' * room)                   # NIL message: space for later growth
    head = b'OHDR' + struct.pack('<BBI', 2, 0x02, len(body))   # version 2, 4-byte chunk size, no times
    blob = head + body
    return blob + struct.pack('<I', lookup3(blob))


_FLOAT_TYPES = {4: (bytes([0x11, 0x20, 31, 0]), struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)),
                8: (bytes([0x11, 0x20, 63, 0]), struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023))}


def _datatype(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == 'f' and dtype.itemsize in _FLOAT_TYPES:
        head, props = _FLOAT_TYPES[dtype.itemsize]
        return head + struct.pack('<I', dtype.itemsize) + props
    if dtype.kind == 'c' and dtype.itemsize in (8, 16):
        part = _datatype(np.dtype('<f%d' % (dtype.itemsize // 2)))
        members = b'r\0' + bytes([0]) + part + b'i\0' + bytes([dtype.itemsize // 2]) + part
        return bytes([0x36, 2, 0, 0]) + struct.pack('<I', dtype.itemsize) + members
    if dtype.kind == 'i' or dtype.kind == 'u':
        signed = 0x08 if dtype.kind == 'i' else 0
        return (bytes([0x10, signed, 0, 0]) + struct.pack('<I', dtype.itemsize)
                + struct.pack('<HH', 0, 8 * dtype.itemsize))
    if dtype.kind == 'S':
        return bytes([0x13, 0x01, 0, 0]) + struct.pack('<I', dtype.itemsize)       # null-padded ASCII
    raise TypeError(f"no HDF5 datatype for {dtype}")


def _dataset_header(shape, dtype, address, nbytes):
    rank = len(shape)
    space = struct.pack('<BBBB', 2, rank, 0, 1 if rank else 0) + b''.join(struct.pack('<Q', d) for d in shape)
    layout = struct.pack('<BBQQ', 3, 1, address, nbytes)
    return _object_header([_message(0x01, space), _message(0x03, _datatype(dtype), 1),
                           _message(0x05, bytes([3, 0x0a]), 1), _message(0x08, layout)])


def _link(name, address):
    name = name.encode()
    return _message(0x06, struct.pack('<BBB', 1, 0, len(name)) + name + struct.pack('<Q', address))


def _layout(header_text, shape, dtype):
    """The file's metadata for the two datasets: (bytes, payload address, end of file)."""
    text = header_text.encode()
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    pos_root = 48
    root_size = len(_object_header([_message(0x02, b''), _message(0x0A, b''), _link('header', 0),
                                    _link('payload', 0)])) + 18 + 2        # (message bodies added below)
    pos_h = pos_root + root_size
    oh_h = _dataset_header((), np.dtype('S%d' % len(text)), 0, len(text))
    pos_p = pos_h + len(oh_h)
    oh_p = _dataset_header(shape, dtype, 0, nbytes)
    data_h = pos_p + len(oh_p)
    data_p = -(-(data_h + len(text)) // 4096) * 4096                       # payload on a page boundary
    eof = data_p + nbytes
    root = _object_header([_message(0x02, struct.pack('<BBQQ', 0, 0, _UNDEF, _UNDEF)),
                           _message(0x0A, bytes([0, 0]), 1), _link('header', pos_h), _link('payload', pos_p)])
    assert len(root) == root_size
    oh_h = _dataset_header((), np.dtype('S%d' % len(text)), data_h, len(text))
    oh_p = _dataset_header(shape, dtype, data_p, nbytes)
    sb = _SIGNATURE + struct.pack('<BBBBQQQQ', 2, 8, 8, 0, 0, _UNDEF, eof, pos_root)
    sb += struct.pack('<I', lookup3(sb))
    meta = sb + root + oh_h + oh_p + text
    return meta + b'\0' * (data_p - len(meta)), data_p, eof


# --------------------------------------------------------------------------- YAML header
def _yaml_array(a, indent):
    a = np.ascontiguousarray(a)
    pad = ' ' * indent
    inner = base64.b64encode(a.tobytes())
    text = base64.b64encode(inner).decode()
    lines = [text[i:i + 76] for i in range(0, len(text), 76)] or ['']
    out = '!numpy.ndarray\n' + pad + 'buffer: !!binary |\n'
    out += ''.join(pad + '  ' + ln + '\n' for ln in lines)
    out += pad + f'dtype: {a.dtype.name if a.dtype.kind in "fiub" else a.dtype.str}\n'
    out += pad + 'order: C\n'
    out += pad + 'shape: !!python/tuple [' + ', '.join(str(d) for d in a.shape) + ']\n'
    return out


def _yaml_quantity(value, unit, indent):
    pad = ' ' * indent
    out = '!astropy.units.Quantity\n' + pad + 'unit: !astropy.units.Unit {unit: ' + unit + '}\n'
    if np.ndim(value) == 0:
        return out + pad + f'value: {float(value)!r}\n'
    return out + pad + 'value: ' + _yaml_array(np.asarray(value, dtype=np.float64), indent + 2)


def header_yaml(sample_shape, samples_per_frame, sample_rate_hz, start_time, dtype, frequency_hz=None,
                sideband=None, polarization=None):
    """The ``header`` dataset: the reference's header keywords (io/hdf5/header.py:44-45, 215-216)
    as YAML in the tags of astropy's dumper; rates and frequencies in Hz."""
    t = Time(start_time)
    jd1, jd2 = t.jd1_jd2()
    items = {'dtype': np.dtype(dtype).str}
    if frequency_hz is not None:
        items['frequency'] = _yaml_quantity(frequency_hz, 'Hz', 2)
    if polarization is not None:
        pol = np.asarray(polarization)
        items['polarization'] = _yaml_array(pol.astype('<U%d' % max(1, pol.dtype.itemsize // (4 if pol.dtype.kind == 'U' else 1))), 2)
    items['sample_rate'] = _yaml_quantity(sample_rate_hz, 'Hz', 2)
    items['sample_shape'] = '!!python/tuple [' + ', '.join(str(d) for d in sample_shape) + ']\n'
    items['samples_per_frame'] = f'{int(samples_per_frame)}\n'
    if sideband is not None:
        items['sideband'] = _yaml_array(np.asarray(sideband, dtype=np.int8), 2)
    items['time'] = ("!astropy.time.Time {format: isot, in_subfmt: '*', jd1: %r, jd2: %r,\n"
                     "  out_subfmt: '*', precision: 9, scale: utc}\n" % (jd1, jd2))
    out = ''
    for key in sorted(items):
        value = items[key]
        out += f'{key}: {value}' if value.endswith('\n') else f'{key}: {value}\n'
    return out


# --------------------------------------------------------------------------- stream writer / reader
class HDF5StreamWriter:
    """Write a stream of known length into an HDF5 file of the reference's layout.

    ``template`` supplies shape, start time, sample rate, dtype and the optional
    frequency / sideband / polarization (any stream of this package or one
    with the same attributes); keywords override it.  Like the reference's
    writer (io/hdf5/base.py:102-126) it takes samples in order, through
    ``write(data)`` or ``fw[a:b] = data``; `hip.DeviceArray` pieces are copied
    down first.
    """

    def __init__(self, name, template=None, *, shape=None, start_time=None, sample_rate=None, dtype=None,
                 frequency=None, sideband=None, polarization=None):
        get = lambda key, given: given if given is not None else getattr(template, key, None)
        shape = tuple(get('shape', shape))
        self.shape = shape
        self.sample_shape = shape[1:]
        self.dtype = np.dtype(get('dtype', dtype))
        self.sample_rate = u.to_hz(get('sample_rate', sample_rate))
        self.start_time = Time(get('start_time', start_time))
        self.frequency = get('frequency', frequency)
        self.sideband = get('sideband', sideband)
        self.polarization = get('polarization', polarization)
        freq = None if self.frequency is None else np.asarray(u.to_hz(self.frequency), dtype=np.float64)
        text = header_yaml(self.sample_shape, shape[0], self.sample_rate, self.start_time, self.dtype,
                           frequency_hz=freq, sideband=self.sideband, polarization=self.polarization)
        meta, self._data_at, self._eof = _layout(text, shape, self.dtype)
        self._fh = builtins_open(name, 'wb')
        self._fh.write(meta)
        self.offset = 0
        self.closed = False

    def tell(self):
        return self.offset

    def write(self, data):
        if self.closed:
            raise ValueError("I/O operation on closed stream.")
        if hasattr(data, 'to_host'):
            data = data.to_host()
        data = np.ascontiguousarray(data, dtype=self.dtype)
        assert data.shape[1:] == self.sample_shape, f"'data' must have trailing shape {self.sample_shape}"
        if self.offset + data.shape[0] > self.shape[0]:
            raise EOFError("cannot write beyond the length given in the header.")
        self._fh.write(data.tobytes())
        self.offset += data.shape[0]

    def __setitem__(self, item, value):
        start, stop, step = item.indices(self.shape[0])
        assert start == self.offset, 'Can only assign right following pointer.'
        assert step == 1, 'unity step size only is supported'
        assert len(value) == stop - start, 'number of samples should match.'
        self.write(value)

    def close(self):
        if not self.closed:
            self.closed = True
            if self.offset < self.shape[0]:            # (a short file still has the promised size)
                self._fh.truncate(self._eof)
            self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _parse_yaml_scalar(text, key):
    for line in text.splitlines():
        if line.startswith(key + ':'):
            return line[len(key) + 1:].strip()
    return None


def _parse_array(text, key):
    lines = text.splitlines()
    for i, line in enumerate(lines):
        if line.startswith(key + ':'):
            block = []
            for ln in lines[i + 1:]:
                if ln and not ln.startswith(' '):
                    break
                block.append(ln)
            body = '\n'.join(block)
            if 'buffer:' not in body:
                return None
            j = next(k for k, ln in enumerate(block) if 'buffer:' in ln)
            b64 = ''
            for ln in block[j + 1:]:
                s = ln.strip()
                if ':' in s:
                    break
                b64 += s
            dt = next(ln.split(':', 1)[1].strip() for ln in block if ln.strip().startswith('dtype:'))
            shp = next(ln.split('[', 1)[1].split(']')[0] for ln in block if ln.strip().startswith('shape:'))
            shape = tuple(int(s) for s in shp.split(',') if s.strip())
            raw = base64.b64decode(base64.b64decode(b64))
            return np.frombuffer(raw, dtype=np.dtype(dt)).reshape(shape).copy()
    return None


class HDF5StreamReader(Base):
    """Read a file written by `HDF5StreamWriter` (samples through a memory map)."""

    def __init__(self, name):
        raw = np.memmap(name, mode='r')
        head = bytes(raw[:48])
        if head[:8] != _SIGNATURE or head[8] not in (2, 3) or lookup3(head[:44]) != struct.unpack('<I', head[44:48])[0]:
            raise OSError(f"{name}: not an HDF5 file this reader understands (superblock).")
        root = struct.unpack('<Q', head[36:44])[0]
        links = dict(self._links(raw, root))
        if not {'header', 'payload'} <= set(links):
            raise OSError(f"{name}: no 'header' and 'payload' datasets.")
        h_addr, h_size, _, _ = self._dataset(raw, links['header'])
        text = bytes(raw[h_addr:h_addr + h_size]).decode()
        p_addr, p_size, shape, _ = self._dataset(raw, links['payload'])
        dtype = np.dtype(_parse_yaml_scalar(text, 'dtype'))
        self._text = text
        self._data = np.ndarray(shape, dtype, buffer=raw, offset=p_addr)
        line = next(ln for ln in text.replace('\n  ', ' ').splitlines() if ln.startswith('time:'))
        jd1 = float(line.split('jd1:')[1].split(',')[0])
        jd2 = float(line.split('jd2:')[1].split(',')[0])
        rate = float(text.split('sample_rate:')[1].split('value:')[1].split()[0])
        kwargs = {}
        freq = _parse_array(text, 'frequency')
        if freq is None and 'frequency:' in text:
            freq = float(text.split('frequency:')[1].split('value:')[1].split()[0])
        if freq is not None:
            kwargs['frequency'] = freq
        for key in ('sideband', 'polarization'):
            value = _parse_array(text, key)
            if value is not None:
                kwargs[key] = value
        super().__init__(shape=tuple(shape), start_time=Time.from_jd(jd1, jd2), sample_rate=rate,
                         samples_per_frame=min(shape[0], 1 << 20) if shape[0] else 1, dtype=dtype, **kwargs)

    @staticmethod
    def _messages(raw, addr):
        head = bytes(raw[addr:addr + 16])
        if head[:4] != b'OHDR' or head[4] != 2:
            raise OSError("object header version not supported by this reader.")
        flags = head[5]
        pos = addr + 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
        width = 1 << (flags & 3)
        size = int.from_bytes(bytes(raw[pos:pos + width]), 'little')
        pos += width
        end = pos + size
        while pos + 4 <= end:
            kind, n, _ = struct.unpack('<BHB', bytes(raw[pos:pos + 4]))
            pos += 4 + (2 if flags & 0x04 else 0)
            yield kind, bytes(raw[pos:pos + n])
            pos += n

    @classmethod
    def _links(cls, raw, addr):
        for kind, data in cls._messages(raw, addr):
            if kind == 0x06 and data[0] == 1 and data[1] == 0:
                n = data[2]
                yield data[3:3 + n].decode(), struct.unpack('<Q', data[3 + n:11 + n])[0]

    @classmethod
    def _dataset(cls, raw, addr):
        shape, address, size = (), None, None
        for kind, data in cls._messages(raw, addr):
            if kind == 0x01:
                rank = data[1]
                shape = tuple(struct.unpack('<Q', data[4 + 8 * i:12 + 8 * i])[0] for i in range(rank))
            elif kind == 0x08 and data[1] == 1:
                address, size = struct.unpack('<QQ', data[2:18])
        if address is None:
            raise OSError("dataset is not stored contiguously.")
        return address, size, shape, None

    def host_view(self, start, count):
        return None

    def read(self, count=None, out=None):
        count = self._prepare_read(count, out)
        piece = self._data[self.offset:self.offset + count]
        self.offset += count
        if out is None:
            return np.array(piece)
        out[...] = piece
        return out

    def _read_frame(self, frame_index):
        start = frame_index * self.samples_per_frame
        return np.array(self._data[start:min(start + self.samples_per_frame, self.shape[0])])

    def close(self):
        super().close()
        self._data = None


builtins_open = open


def open(name, mode='r', **kwargs):
    """Open an HDF5 file of the reference's intermediate format as a stream
    (reference io/hdf5/base.py:129-222): ``mode='w'`` with ``template=`` (and /
    or the header values as keywords) gives a writer, ``'r'`` a reader for
    files written by this module."""
    if mode == 'w':
        return HDF5StreamWriter(name, **kwargs)
    if mode == 'r':
        if kwargs:
            raise TypeError("no keywords for reading.")
        return HDF5StreamReader(name)
    raise ValueError("mode must be 'r' or 'w'.")
