"""The SURVEY 8(f) "next" rows, HBM-resident, one JSON line each (dev tool).

    python tools/bench_next.py [--reps N] [row ...]

f1_detect   Integrate(Power(stream), 16) on a channelized 2-pol stream  (read 16 B, write 16/16 B per complete sample)
f1_fused    Integrate(Power(Channelize(Dedisperse, 1024)), 64): the metric pipeline with the powers summed in the
            last pass (algorithmic: 16 B in per input sample, output negligible)
f2_shift    ShiftSamples, 64 sub-bands x 2 pol, integer shifts          (read 8 B + write 8 B per element)
f3_vdif     k_unpack on resident frames, 2-bit complex, 8 channels     (read 0.5 B + write 8 B per complex sample)
f3_vdif_read  the same through open_vdif(...).read_device, upload of the raw frames included (PCIe-inclusive)
f3_dada     k_unpack on resident samples, 8-bit complex, 2 pol          (read 2 B + write 8 B per complex sample)
f4_dechan   Dechannelize(1024), 2 pol                                   (read 16 B + write 16 B per complete sample)
f4_ipfb     InversePolyphaseFilterBank 4 x 1024, 2 pol                  (Dechannelize + overlap-save along the block axis)

chan_* / pfb_* / dedisperse_real   variants of the path's own kernels the BASELINE configs do not reach (few and many
            channels, channel counts that are not powers of two, other tap counts, float32 streams)

`frac` prices the algorithmic bytes against 8 TB/s.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baseband_tasks_amd as bt
from baseband_tasks_amd import ingest
from baseband_tasks_amd import units as u

T0 = '2020-01-01T00:00:00'
DEV = torch.device('cuda', 0)


def randn_c64(n, shape, seed=1):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return torch.view_as_complex(torch.randn((n,) + tuple(shape) + (2,), generator=g, device=DEV, dtype=torch.float32))


def timed(step, reps):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def restart(tasks, last, count):
    for t in tasks:
        t.max_frames_per_call = 10**6
        t.invalidate_cache()
    last.seek(0)
    return last.read_device(count)


def f1_detect(reps):
    n_spec, n_chan = 2**18, 1024
    x = randn_c64(n_spec, (n_chan, 2))
    ds = bt.DeviceStream(x, T0, 16e6 / n_chan, samples_per_frame=2**12, polarization=['X', 'Y'])
    it = bt.Integrate(bt.Power(ds), 16, samples_per_frame=2**8)
    dt = timed(lambda: restart([it.ih, it], it, it.shape[0]), reps)
    units = n_spec * n_chan
    return dict(units=units, unit='complete samples', bytes_per_unit=16 + 16 / 16, seconds=dt)


def f1_fused(reps):
    nblk = 384
    x = randn_c64(nblk * 2**20, (2,))
    ds = bt.DeviceStream(x, T0, 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1, polarization=['X', 'Y'])
    dd = bt.Dedisperse(ds, 100.)
    ch = bt.Channelize(dd, 1024, 64)
    it = bt.Integrate(bt.Power(ch), 64, samples_per_frame=8)
    dt = timed(lambda: restart([dd, ch, it.ih, it], it, it.shape[0]), reps)
    units = it.shape[0] * 64 * 1024
    return dict(units=units, unit='complete samples (of the dedispersed stream)',
                bytes_per_unit=16 * 2**20 / 836100, seconds=dt)


def f2_shift(reps):
    n = 2**21
    x = randn_c64(n, (64, 2))
    ds = bt.DeviceStream(x, T0, 6.25e6, samples_per_frame=2**16)
    shift = (np.arange(64).reshape(64, 1) * 37) % 1000
    sh = bt.ShiftSamples(ds, shift, samples_per_frame=2**16)
    dt = timed(lambda: restart([sh], sh, sh.shape[0]), reps)
    return dict(units=sh.shape[0] * 128, unit='elements', bytes_per_unit=16, seconds=dt)


def f2_dedisperse_samples(reps):
    """Incoherent dedispersion of a channelized band: 64 sub-bands of 6.25 MHz at 400-800 MHz, 2 pol,
    DM chosen so that the delays span ~3000 samples (monotone in frequency, as real ones are)."""
    n = 2**21
    x = randn_c64(n, (64, 2))
    freq = (403.125e6 + 6.25e6 * np.arange(64)).reshape(64, 1)
    ds = bt.DeviceStream(x, T0, 6.25e6, samples_per_frame=2**16, frequency=freq, sideband=1)
    sh = bt.DedisperseSamples(ds, 0.025, samples_per_frame=2**16)
    dt = timed(lambda: restart([sh], sh, sh.shape[0]), reps)
    return dict(units=sh.shape[0] * 128, unit='elements', bytes_per_unit=16, seconds=dt,
                note=f'delays up to {int(np.ptp(sh._shift))} samples')


def _unpack_resident(raw, n_frames, frame_nbytes, header_nbytes, bits, spf, n_thread, n_elem, code, reps):
    """bbt_unpack on frames already in HBM (the kernel alone)."""
    hip = bt.hip
    raw_dev = hip.DeviceArray.from_host(np.frombuffer(raw, np.uint8))
    out = hip.DeviceArray((n_frames // n_thread * spf, n_thread, n_elem), np.float32)

    def step():
        hip.check(hip.lib().bbt_unpack(raw_dev.ptr, out.ptr, n_frames, frame_nbytes, header_nbytes, bits, spf,
                                       n_thread, n_elem, code, hip.get_stream()))
    return timed(step, reps)


def _vdif_frames():
    spf, n_chan, n_frames = 4000, 8, 4096
    rng = np.random.default_rng(2)
    levels = np.array([-3.3359, -1., 1., 3.3359], np.float32)
    comp = rng.choice(levels, size=(spf * n_frames, 1, n_chan * 2))
    raw = ingest.encode_vdif_frames(comp.view(np.complex64), 2, seconds=100, ref_epoch=41, frame_nr0=0,
                                    frames_per_second=8000, samples_per_frame=spf, edv=3, sample_rate=32e6)
    return raw, spf, n_chan, n_frames


def f3_vdif(reps):
    raw, spf, n_chan, n_frames = _vdif_frames()
    frame_nbytes = len(raw) // n_frames
    dt = _unpack_resident(raw, n_frames, frame_nbytes, 32, 2, spf, 1, 2 * n_chan, 0, reps)
    return dict(units=n_frames * spf * n_chan, unit='complex samples', bytes_per_unit=0.5 + 8 + 32 / (spf * n_chan),
                seconds=dt, note='k_unpack alone, 2-bit complex x 8 channels, frames resident in HBM')


def f3_vdif_read(reps):
    raw, spf, n_chan, n_frames = _vdif_frames()
    fh = bt.open_vdif(raw, frequency=300 * u.MHz, sideband=1)
    n = fh.shape[0]
    dt = timed(lambda: restart([fh], fh, n), max(reps // 3, 2))
    return dict(units=n * n_chan, unit='complex samples', bytes_per_unit=0.5 + 8 + 32 / (spf * n_chan), seconds=dt,
                note='open_vdif(...).read_device: includes the upload of the raw frames from pageable host memory '
                     'each step (PCIe-inclusive)')


def f3_dada(reps):
    n = 2**26
    rng = np.random.default_rng(8)
    samples = rng.integers(-128, 128, size=(n, 2, 2), dtype=np.int8)
    dt = _unpack_resident(samples.tobytes(), n // 4096, 4096 * 4, 0, 8, 4096, 1, 4, 1, reps)
    return dict(units=n * 2, unit='complex samples', bytes_per_unit=2 + 8, seconds=dt,
                note='k_unpack alone, signed 8-bit complex x 2 pol, resident in HBM')


def f4_dechan(reps):
    n_spec, n_chan = 2**18, 1024
    x = randn_c64(n_spec, (n_chan, 2))
    ds = bt.DeviceStream(x, T0, 16e6 / n_chan, samples_per_frame=2**12, frequency=1000e6 * np.ones((n_chan, 1)),
                         sideband=1)
    dc = bt.Dechannelize(ds, n_chan)
    dt = timed(lambda: restart([dc], dc, dc.shape[0]), reps)
    return dict(units=n_spec * n_chan, unit='complete samples', bytes_per_unit=32, seconds=dt)


def f4_ipfb(reps):
    n_spec, n_chan = 2**17, 1024
    # block length along the block axis: a parameter of the task (samples_per_frame).  Measured on
    # MI355X (round 4, complete samples/s): 512 rows 64.5 G, 1024 67.6 G, 2048 56.4 G, 4096 54.8 G --
    # shorter blocks waste more on the 67 rows of padding but their transforms take 4 or 8 stream
    # pairs per workgroup (64- / 128-byte runs) at three workgroups per CU
    rows = int(os.environ.get('IPFB_ROWS', '1024'))
    x = randn_c64(n_spec, (n_chan, 2))
    ds = bt.DeviceStream(x, T0, 16e6 / n_chan, samples_per_frame=2**12, frequency=1000e6 * np.ones((n_chan, 1)),
                         sideband=1)
    ipfb = bt.InversePolyphaseFilterBank(ds, bt.sinc_hamming(4, n_chan), sn=10., pad_start=32, pad_end=32,
                                         samples_per_frame=(rows - 67) * n_chan)
    dt = timed(lambda: restart([ipfb.dechannelized, ipfb], ipfb, ipfb.shape[0]), reps)
    return dict(units=ipfb.shape[0], unit='complete samples', bytes_per_unit=32 * rows / (rows - 67), seconds=dt,
                note=f'blocks of {rows} spectra, 64 + 3 padding; algorithmic = channelized stream in + samples out')


def _channelize_row(n_chan, reps, real=False, n_samples=2**28):
    n_spec = n_samples // n_chan
    if real:
        g = torch.Generator(device=DEV)
        g.manual_seed(3)
        x = torch.randn((n_spec * n_chan, 2), generator=g, device=DEV, dtype=torch.float32)
    else:
        x = randn_c64(n_spec * n_chan, (2,))
    ds = bt.DeviceStream(x, T0, 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    ch = bt.Channelize(ds, n_chan, 2**20 // n_chan if 2**20 % n_chan == 0 else 1000)
    dt = timed(lambda: restart([ch], ch, ch.shape[0]), reps)
    in_b = 8 if real else 16
    out_b = 16 * (n_chan // 2 + 1) / n_chan if real else 16
    return dict(units=ch.shape[0] * n_chan, unit='complete samples', bytes_per_unit=in_b + out_b, seconds=dt,
                note=f'Channelize({n_chan}) of 2 {"float32" if real else "complex64"} streams')


def chan_64(reps):
    return _channelize_row(64, reps)


def chan_8(reps):
    return _channelize_row(8, reps)


def chan_1000(reps):
    return _channelize_row(1000, reps, n_samples=2**27)


def chan_6000(reps):
    return _channelize_row(6000, reps, n_samples=2**27)


def chan_8192(reps):
    return _channelize_row(8192, reps)


def chan_16(reps):
    return _channelize_row(16, reps)


def chan_32(reps):
    return _channelize_row(32, reps)


def chan_128(reps):
    return _channelize_row(128, reps)


def chan_16384(reps):
    return _channelize_row(16384, reps)


def chan_real_1024(reps):
    return _channelize_row(1024, reps, real=True)


def chan_real4_1024(reps):
    n_chan, n = 1024, 2**27
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    x = torch.randn((n, 2, 2), generator=g, device=DEV, dtype=torch.float32)
    ds = bt.DeviceStream(x, T0, 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    ch = bt.Channelize(ds, n_chan, 1024)
    dt = timed(lambda: restart([ch], ch, ch.shape[0]), reps)
    return dict(units=ch.shape[0] * n_chan, unit='complete samples', bytes_per_unit=16 + 32 * 513 / 1024, seconds=dt,
                note='Channelize(1024) of 4 float32 streams (2 x 2)')


def _pfb_row(n_tap, n_chan, reps, real=False):
    nblk = 192
    if real:
        g = torch.Generator(device=DEV)
        g.manual_seed(3)
        x = torch.randn((nblk * 2**20, 2), generator=g, device=DEV, dtype=torch.float32)
    else:
        x = randn_c64(nblk * 2**20, (2,))
    ds = bt.DeviceStream(x, T0, 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    pfb = bt.PolyphaseFilterBank(ds, bt.sinc_hamming(n_tap, n_chan))
    dt = timed(lambda: restart([pfb], pfb, pfb.shape[0]), reps)
    in_b = 8 if real else 16
    out_b = 16 * (n_chan // 2 + 1) / n_chan if real else 16
    return dict(units=pfb.shape[0] * n_chan, unit='complete samples', bytes_per_unit=in_b + out_b, seconds=dt,
                note=f'PolyphaseFilterBank {n_tap} x {n_chan}, 2 {"float32" if real else "complex64"} streams')


def pfb_4x1024(reps):
    return _pfb_row(4, 1024, reps)


def pfb_8x2048(reps):
    return _pfb_row(8, 2048, reps)


def pfb_16x4096(reps):
    return _pfb_row(16, 4096, reps)


def pfb_12x256(reps):
    return _pfb_row(12, 256, reps)


def pfb_real_12x1024(reps):
    return _pfb_row(12, 1024, reps, real=True)


def _dedisperse_row(reps, kind):
    """2^20-sample blocks as in config 2: 'real' = 2 float32 streams (one complex stream, padded
    to a pair inside), 'single' = 1 complex64 stream (padded), 'default' = 2 complex64 streams with the
    reference's default block length (not a power of two: the generic kernels)."""
    nblk = 192
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    if kind == 'real':
        x = torch.randn((nblk * 2**20, 2), generator=g, device=DEV, dtype=torch.float32)
        fs, per = 32e6, 8
    elif kind == 'single':
        x = randn_c64(nblk * 2**20, ())
        fs, per = 16e6, 8
    else:
        x = randn_c64(nblk * 2**20, (2,))
        fs, per = 16e6, 16
    ds = bt.DeviceStream(x, T0, fs, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    if kind == 'default':
        dd = bt.Dedisperse(ds, 100.)
    else:
        probe = bt.Dedisperse(ds, 100.)
        pad = probe._ih_samples_per_frame - probe.samples_per_frame
        dd = bt.Dedisperse(ds, 100., samples_per_frame=2**20 - pad)
        assert dd._ih_samples_per_frame == 2**20
    dt = timed(lambda: restart([dd], dd, dd.shape[0]), reps)
    return dict(units=dd.shape[0], unit='complete samples',
                bytes_per_unit=per * (1 + dd._ih_samples_per_frame / dd.samples_per_frame), seconds=dt,
                note=f'Dedisperse DM 100, {kind}: blocks of {dd._ih_samples_per_frame}, {dd.samples_per_frame} kept')


def dedisperse_real(reps):
    return _dedisperse_row(reps, 'real')


def dedisperse_single(reps):
    return _dedisperse_row(reps, 'single')


def dedisperse_default(reps):
    return _dedisperse_row(reps, 'default')


def fused_single(reps):
    nblk = 384
    x = randn_c64(nblk * 2**20, ())
    ds = bt.DeviceStream(x, T0, 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    dd = bt.Dedisperse(ds, 100., samples_per_frame=836100)
    ch = bt.Channelize(dd, 1024, 64)
    assert ch._fusable_input() is dd
    dt = timed(lambda: restart([dd, ch], ch, ch.shape[0]), reps)
    return dict(units=ch.shape[0] * 1024, unit='complete samples', bytes_per_unit=8 * (1 + 2**20 / 836100), seconds=dt,
                note='the metric pipeline on ONE complex64 stream: fused, consecutive blocks paired')


def fused_real(reps):
    nblk = 384
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    x = torch.randn((nblk * 2**20, 2), generator=g, device=DEV, dtype=torch.float32)
    ds = bt.DeviceStream(x, T0, 32e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    probe = bt.Dedisperse(ds, 100.)
    pad = probe._ih_samples_per_frame - probe.samples_per_frame
    dd = bt.Dedisperse(ds, 100., samples_per_frame=2**20 - pad)
    ch = bt.Channelize(dd, 1024, 64)
    assert ch._fusable_input() is dd
    dt = timed(lambda: restart([dd, ch], ch, ch.shape[0]), reps)
    return dict(units=ch.shape[0] * 1024, unit='complete samples',
                bytes_per_unit=8 * 2**20 / dd.samples_per_frame + 16 * 513 / 1024, seconds=dt,
                note='the metric pipeline on two float32 streams: fused as one complex stream, then split')


def chan_single_1024(reps):
    n_chan, n_spec = 1024, 2**18
    x = randn_c64(n_spec * n_chan, ())
    ds = bt.DeviceStream(x, T0, 16e6, samples_per_frame=2**20, frequency=1000e6, sideband=1)
    ch = bt.Channelize(ds, n_chan, 1024)
    dt = timed(lambda: restart([ch], ch, ch.shape[0]), reps)
    return dict(units=ch.shape[0] * n_chan, unit='complete samples', bytes_per_unit=16, seconds=dt,
                note='Channelize(1024) of ONE complex64 stream (padded to a pair inside)')


ROWS = dict(chan_64=chan_64, chan_8=chan_8, chan_1000=chan_1000, chan_6000=chan_6000, chan_8192=chan_8192, chan_16384=chan_16384, chan_16=chan_16, chan_32=chan_32, chan_128=chan_128,
            chan_real_1024=chan_real_1024, chan_real4_1024=chan_real4_1024, pfb_4x1024=pfb_4x1024, pfb_8x2048=pfb_8x2048, pfb_16x4096=pfb_16x4096,
            pfb_12x256=pfb_12x256, pfb_real_12x1024=pfb_real_12x1024, dedisperse_real=dedisperse_real,
            dedisperse_single=dedisperse_single, dedisperse_default=dedisperse_default, chan_single_1024=chan_single_1024,
            fused_single=fused_single, fused_real=fused_real,
            f1_detect=f1_detect, f1_fused=f1_fused, f2_shift=f2_shift, f2_dedisperse_samples=f2_dedisperse_samples, f3_vdif=f3_vdif, f3_vdif_read=f3_vdif_read,
            f3_dada=f3_dada,
            f4_dechan=f4_dechan, f4_ipfb=f4_ipfb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('rows', nargs='*', default=sorted(ROWS))
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)
    for name in args.rows:
        try:
            r = ROWS[name](args.reps)
        except Exception as exc:                      # one row failing does not hide the others
            print(json.dumps(dict(row=name, error=repr(exc))), flush=True)
            continue
        gbps = r['units'] * r['bytes_per_unit'] / r['seconds'] / 1e9
        print(json.dumps(dict(row=name, munits_per_s=round(r['units'] / r['seconds'] / 1e6, 1), unit=r['unit'],
                              ms_per_step=round(r['seconds'] * 1e3, 4), alg_bytes_per_unit=round(r['bytes_per_unit'], 3),
                              alg_gbps=round(gbps, 1), frac=round(gbps / 8000., 4), note=r.get('note'))), flush=True)
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
