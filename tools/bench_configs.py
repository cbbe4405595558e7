"""Throughput of the non-headline BASELINE configs on one GPU (dev tool):
config 1 (Channelize 1024), config 3 (PFB 12x1024), config 5 (8-stream
Resample -> Dedisperse), plus Dedisperse alone.  HBM-resident input."""
import gc
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import baseband_tasks_amd as bt

dev = torch.device('cuda', 0)
bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)


def stream(n, S, seed=1):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.view_as_complex(torch.randn((n, S, 2), generator=g, device=dev, dtype=torch.float32))
    return bt.DeviceStream(x if S > 1 else x, '2020-01-01T00:00:00', 16e6, samples_per_frame=2**20,
                           frequency=1000e6, sideband=1)


def timeit(tasks, last, count, reps=10):
    def step():
        for t in tasks:
            t.invalidate_cache()
        last.seek(0)
        return last.read_device(count)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()            # a full collection of torch's object graph takes ~50 ms: keep it out of the loop
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gc.enable()
    return dt


def single_stream_configs():
    nblk = 96
    ds = stream(nblk * 2**20, 2)
    # config 1
    ch = bt.Channelize(ds, 1024, 512)
    ch.max_frames_per_call = 10**6
    dt = timeit([ch], ch, ch.shape[0])
    print(f"config1 Channelize(1024): {ch.shape[0] * 1024 / dt / 1e6:9.1f} Msamples/s  ({32 * ch.shape[0] * 1024 / dt / 1e9:.0f} GB/s algorithmic)")
    del ch
    # config 3
    pfb = bt.PolyphaseFilterBank(ds, bt.sinc_hamming(12, 1024))
    pfb.max_frames_per_call = 10**6
    dt = timeit([pfb], pfb, pfb.shape[0])
    print(f"config3 PFB(12x1024):     {pfb.shape[0] * 1024 / dt / 1e6:9.1f} Msamples/s  ({32 * pfb.shape[0] * 1024 / dt / 1e9:.0f} GB/s algorithmic)")
    del pfb
    # dedisperse alone
    dd = bt.Dedisperse(ds, 100.)
    dd.max_frames_per_call = 10**6
    dt = timeit([dd], dd, dd.shape[0])
    print(f"config2 Dedisperse:       {dd.shape[0] / dt / 1e6:9.1f} Msamples/s")
    del dd
    for label in ('direct', 'Fourier'):
        limit = bt.Convolve.FIR_MAX_TAPS
        if label == 'Fourier':
            bt.Convolve.FIR_MAX_TAPS = 0
        rs = bt.Resample(ds, 0.25, pad=64, samples_per_frame=2**20 - 128)
        rs.seek(0)
        rs.max_frames_per_call = 10**6
        dt = timeit([rs], rs, rs.shape[0] - 64)
        bt.Convolve.FIR_MAX_TAPS = limit
        print(f"Resample(129 taps) alone, {label}: {(rs.shape[0] - 64) / dt / 1e6:9.1f} Msamples/s")
    del rs
    # the metric pipeline at other band centres (SURVEY 8d variants) and with detection
    for fc, spf in ((800e6, 2**20 - 415021), (1400e6, None), (1000e6, None)):
        dsf = bt.SetAttribute(ds, frequency=fc, polarization=['X', 'Y'])
        kw = {} if spf is None else dict(samples_per_frame=spf)
        dd = bt.Dedisperse(dsf, 100., **kw)
        chd = bt.Channelize(dd, 1024, 512)
        dd.max_frames_per_call = chd.max_frames_per_call = 10**6
        dt = timeit([dd, chd], chd, chd.shape[0])
        print(f"metric pipeline at {fc / 1e6:.0f} MHz (valid {dd.samples_per_frame} of {dd._ih_samples_per_frame}): "
              f"{chd.shape[0] * 1024 / dt / 1e6:9.1f} Msamples/s")
    pw = bt.Integrate(bt.Power(chd), 64)
    pw.max_frames_per_call = 10**6
    dt = timeit([dd, chd, pw], pw, pw.shape[0])
    print(f"Dedisperse->Channelize(1024)->Power->Integrate(64): {pw.shape[0] * 64 * 1024 / dt / 1e6:9.1f} Msamples/s "
          f"({dt * 1e3:.3f} ms for {pw.shape[0]} x 64 spectra)")


def main():
    only5 = 'config5' in sys.argv
    if not only5:
        single_stream_configs()
    # config 5: 8 streams
    nblk = 12
    ds8 = stream(nblk * 2**20, 8)
    rs = bt.Resample(ds8, 0.25, pad=64, samples_per_frame=2**20 - 128)
    rs.seek(0)
    dd = bt.Dedisperse(rs, 100., samples_per_frame=2**20 - 212476)
    rs.max_frames_per_call = dd.max_frames_per_call = 10**6
    dt = timeit([rs, dd], dd, dd.shape[0], reps=5)
    print(f"config5 Resample->Dedisperse, 8 streams: {dd.shape[0] / dt / 1e6:9.1f} Msamples/s "
          f"(complete 8-stream samples; x4 = {4 * dd.shape[0] / dt / 1e6:.0f} 2-pol-equivalent)")
    fir_limit = bt.Convolve.FIR_MAX_TAPS
    bt.Convolve.FIR_MAX_TAPS = 0                  # the same through the Fourier-domain plan
    rs = bt.Resample(ds8, 0.25, pad=64, samples_per_frame=2**20 - 128)
    rs.seek(0)
    dd = bt.Dedisperse(rs, 100., samples_per_frame=2**20 - 212476)
    rs.max_frames_per_call = dd.max_frames_per_call = 10**6
    dt = timeit([rs, dd], dd, dd.shape[0], reps=5)
    bt.Convolve.FIR_MAX_TAPS = fir_limit
    print(f"        with the resampler in the Fourier domain: {dd.shape[0] / dt / 1e6:9.1f} Msamples/s")
    for label in ('direct', 'Fourier'):
        if label == 'Fourier':
            bt.Convolve.FIR_MAX_TAPS = 0
        rs = bt.Resample(ds8, 0.25, pad=64, samples_per_frame=2**20 - 128)
        rs.seek(0)
        rs.max_frames_per_call = 10**6
        dt = timeit([rs], rs, rs.shape[0] - 64, reps=5)
        bt.Convolve.FIR_MAX_TAPS = fir_limit
        print(f"        Resample alone ({label}), 8 streams:  {(rs.shape[0] - 64) / dt / 1e6:9.1f} Msamples/s")
    dd8 = bt.Dedisperse(ds8, 100.)
    dd8.max_frames_per_call = 10**6
    dt = timeit([dd8], dd8, dd8.shape[0], reps=5)
    print(f"        Dedisperse alone, 8 streams:     {dd8.shape[0] / dt / 1e6:9.1f} Msamples/s (x4 = {4 * dd8.shape[0] / dt / 1e6:.0f})")
    del ds8, rs, dd, dd8
    if only5:
        return
    # config 4, one GPU's share: 8 sub-bands x 2 pol of 6.25 MHz, DM 557, 2^24 blocks -> Channelize(64)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    spf4 = 2**24 - 2756522
    n4 = 2 * spf4 + 2756522
    x = torch.view_as_complex(torch.randn((n4, 8, 2, 2), generator=g, device=dev, dtype=torch.float32))
    freq = (403.125e6 + 6.25e6 * np.arange(8)).reshape(8, 1)
    ds4 = bt.DeviceStream(x, '2020-01-01T00:00:00', 6.25e6, samples_per_frame=2**24, frequency=freq, sideband=1)
    dd4 = bt.Dedisperse(ds4, 557., reference_frequency=freq, samples_per_frame=spf4)
    ch4 = bt.Channelize(dd4, 64, 4096)
    dd4.max_frames_per_call = ch4.max_frames_per_call = 10**6
    dt = timeit([dd4, ch4], ch4, ch4.shape[0], reps=3)
    print(f"config4 share (8 sub-bands x 2 pol, DM 557, N=2^24 -> Channelize(64)): "
          f"{ch4.shape[0] * 64 / dt / 1e6:9.1f} M complete samples/s (x8 = {8 * ch4.shape[0] * 64 / dt / 1e6:.0f} 2-pol-equivalent)")


if __name__ == '__main__':
    main()
