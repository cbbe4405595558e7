#!/usr/bin/env python
"""Headline benchmark: Msamples/s through Dedisperse(DM=100) + Channelize(1k
channels) on a 2-pol complex64 stream resident in HBM (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: BLOCKS overlap-save
blocks of 2^20 samples per GPU (16 MHz band at 1000 MHz, DM 100: pad
104963 + 107513, 836100 valid samples per block), coherently dedispersed and
channelized to 1024 channels, through the package's task objects
(Channelize(Dedisperse(DeviceStream))).read_device(), i.e. through the C ABI.
Frame caches are invalidated every step so all work is redone.

For N > 1 the driver launches one process per GPU (torch.distributed, backend
nccl = RCCL); time blocks are independent, so each rank owns its own batch
(weak scaling, no data-path collective).  The chirp is computed on rank 0
and broadcast over RCCL at plan time.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT = 1 << 20
N_CHAN = 1024
DM = 100.
FS_HZ = 16e6
FC_HZ = 1000e6
ALG_BYTES_PER_SAMPLE = 36.07        # (8/eta + 8) * 2 pol, eta = 836100 / 2^20
HBM_PEAK_GBPS = 8000.               # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--blocks', type=int, default=768, help='overlap-save blocks per step per GPU')
    ap.add_argument('--cpu-blocks', type=int, default=40, help='blocks per process for the CPU baseline (x1/2)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='keep the HIP-event kernel timing out of the timed region (A/B check)')
    ap.add_argument('--gather', action='store_true', help='also time an all-gather of the outputs')
    return ap.parse_args()


def _cpu_worker(n_blocks):
    """Dedisperse -> Channelize(1024) on n_blocks blocks of 2^20 through the
    oracle, one process; returns (valid samples, seconds)."""
    from oracle import bbt_oracle as orc
    g = orc.disperse_geometry(FS_HZ, FC_HZ / 1e6, 1, -DM)
    spf = N_FFT - g['pad_start'] - g['pad_end']
    h = orc.chirp(N_FFT, FS_HZ, FC_HZ / 1e6, 1, -DM, g['reference_frequency'])
    rng = np.random.default_rng(1)
    x = rng.standard_normal((N_FFT, 4), dtype=np.float32).view(np.complex64)
    carry = np.empty((0, 2), np.complex64)
    orc.disperse_block(x, h, g['pad_start'], spf, fft64=False)      # warm-up
    t0 = time.perf_counter()
    n_out = 0
    for _ in range(n_blocks):
        y = orc.disperse_block(x, h, g['pad_start'], spf, fft64=False)
        y = np.concatenate([carry, y])
        k = (y.shape[0] // N_CHAN) * N_CHAN
        z = orc.channelize(y[:k], N_CHAN, fft64=False)
        carry = y[k:]
        n_out += z.shape[0] * N_CHAN
    return n_out, time.perf_counter() - t0


def cpu_baseline(n_blocks):
    """The oracle (numpy restatement of the reference path) on the host cores:
    P independent processes over disjoint runs of blocks, P = the
    GPU box's CPU share (<= 16).  Also quotes the single-process rate."""
    import multiprocessing as mp
    n1, t1 = _cpu_worker(max(4, n_blocks // 8))
    single = n1 / t1 / 1e6
    procs = max(1, min(os.cpu_count() or 1, 16))
    per = max(2, n_blocks // 2)
    ctx = mp.get_context('spawn')
    rates, busy, wall = [], 0.0, 0.0
    with ctx.Pool(procs) as pool:
        for _ in range(3):                              # three repeats, median
            t0 = time.perf_counter()
            res = pool.map(_cpu_worker, [per] * procs)
            wall = time.perf_counter() - t0
            # rate from the slowest worker's own loop time (excludes interpreter start-up)
            busy = max(r[1] for r in res)
            rates.append(sum(r[0] for r in res) / busy / 1e6)
    model = 'unknown CPU'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=sorted(rates)[1], unit='Msamples/s', cores=procs, kind='port',
                sample=f'{procs} processes x {per} blocks of 2^20 x 2 pol through oracle/bbt_oracle.py '
                       f'(numpy {np.__version__} complex64 FFT) on {os.cpu_count()} x {model}; median of 3 '
                       f'repeats, last {busy:.1f} s busy / {wall:.1f} s wall; '
                       f'one process alone: {single:.1f} Msamples/s')


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # BBT_BENCH_BACKEND=gloo rehearses the multi-rank code path on a box with fewer GPUs than
        # ranks (ranks then share devices; RCCL itself refuses two ranks on one device)
        backend = os.environ.get('BBT_BENCH_BACKEND', 'nccl')
        dev_index = local_rank % max(1, torch.cuda.device_count()) if backend != 'nccl' else local_rank
        torch.cuda.set_device(dev_index)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device('cuda', dev_index)

    import baseband_tasks_amd as bt
    from baseband_tasks_amd import sharding
    bt.hip.set_device(dev.index)
    bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- workload: BLOCKS blocks per rank, pre-staged in HBM (synthetic noise)
    pad = 104963 + 107513
    spf = N_FFT - pad
    n_in = (args.blocks - 1) * spf + N_FFT
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345 + rank)
    x = torch.randn((n_in, 2, 2), generator=gen, device=dev, dtype=torch.float32)
    x = torch.view_as_complex(x)                       # (n_in, 2) complex64, unit variance/component
    ds = bt.DeviceStream(x, '2020-01-01T00:00:00', FS_HZ, samples_per_frame=N_FFT,
                         frequency=FC_HZ, sideband=1, polarization=['X', 'Y'])
    dd = bt.Dedisperse(ds, DM)
    assert (dd._ih_samples_per_frame, dd.samples_per_frame) == (N_FFT, spf)
    ch = bt.Channelize(dd, N_CHAN, samples_per_frame=512)
    dd.max_frames_per_call = args.blocks
    n_spec = (dd.shape[0] // N_CHAN // 512) * 512
    ch.max_frames_per_call = n_spec // 512 + 1
    # chirp: rank 0 computes, everyone receives over RCCL (no-op for one rank)
    sharding.share_response(dd, torch, dist if world > 1 else None, dev)
    samples_per_step = n_spec * N_CHAN                  # valid output complete samples

    def step():
        dd.invalidate_cache()
        ch.invalidate_cache()
        ch.seek(0)
        return ch.read_device(n_spec)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel timing of the overlap-save passes: HIP events on the stream
    # each pass is launched on, recorded inside the timed region itself, in the
    # normal two-lane schedule (what rocprofv3 --kernel-trace sees as well).
    plan = dd._get_plan()
    for _ in range(args.warmup):
        step()
    gc.collect()        # CPython's full collection over torch's object graph is a ~50 ms pause;
    gc.disable()        # the steps allocate no cycles, so none is due inside the timed region
    fence()
    if not args.no_kernel_timing:
        plan.timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * samples_per_step * args.steps / elapsed / 1e6
    n_steps_timed = args.steps
    if args.no_kernel_timing:               # A/B switch: time the kernels in extra steps instead
        plan.timing_enable(1)
        n_steps_timed = max(3, args.steps // 2)
        for _ in range(n_steps_timed):
            step()
        fence()
    ms, launches = plan.timing_read()
    # the same passes isolated (one lane, nothing else on the GPU), for reference
    plan.timing_enable(2)
    n_iso = max(3, args.steps // 4)
    for _ in range(n_iso):
        step()
    fence()
    ms_iso, _ = plan.timing_read()
    plan.timing_enable(0)
    info = plan.info()
    names = ['osm_col_forward', 'osm_rowpass', 'osm_col_inverse']
    k = int(np.argmax(ms))
    blocks_per_launch = min(info['chunk_blocks'], args.blocks)
    # launches may be ragged (last chunk smaller): use total blocks / launches
    blocks_timed = n_steps_timed * args.blocks
    avg_ms = ms[k] / launches
    units_per_launch = blocks_timed / launches * spf
    achieved = units_per_launch * ALG_BYTES_PER_SAMPLE / (avg_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(names[k])
        except Exception:
            traffic = None
    roofline = dict(bound='hbm', kernel=names[k], achieved=round(achieved, 1), peak=HBM_PEAK_GBPS,
                    unit='GB/s', frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic,
                    avg_launch_ms=round(avg_ms, 5), launches=launches,
                    pass_ms_per_block={n: round(m / blocks_timed, 6) for n, m in zip(names, ms)},
                    pass_ms_per_block_isolated={n: round(m / (n_iso * args.blocks), 6)
                                                for n, m in zip(names, ms_iso)},
                    note='launch durations from HIP events inside the timed region, two lanes active '
                         '(a pass shares the GPU with a pass of the other lane); isolated = one lane')
    path_gbps = value * 1e6 / world * ALG_BYTES_PER_SAMPLE / 1e9
    roofline_path = dict(bound='hbm', achieved=round(path_gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                         frac=round(path_gbps / HBM_PEAK_GBPS, 4),
                         input_msamples_per_s=round(value * N_FFT / spf, 1),
                         note='whole Dedisperse->Channelize path per GPU: 36.07 B x valid samples / wall time; '
                              'input_msamples_per_s = value / eta, eta = 836100 / 2^20 (all GPUs)')

    gather = None
    if args.gather and world > 1:
        z = step()
        zt = torch.as_tensor(z, device=dev)
        zt = torch.view_as_real(zt)
        out = torch.empty((world * zt.shape[0],) + tuple(zt.shape[1:]), dtype=zt.dtype, device=dev)
        fence()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(out, zt)
        fence()
        dt = time.perf_counter() - t0
        gather = dict(seconds=dt, GBps_per_rank_in=zt.numel() * 4 * (world - 1) / dt / 1e9)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(args.cpu_blocks)

    if rank == 0:
        line = dict(
            metric='Msamples/s through Dedisperse(DM=100)+Channelize(1k ch), 2-pol c64',
            value=round(value, 1), unit='Msamples/s', n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
            higher_is_better=True, scaling='weak', vs_baseline=None, dtype='c64',
            data='synthetic',
            config=dict(workload='configs[1]+metric pipeline: Dedisperse DM=100, 16 MHz BW at 1000 MHz, '
                                 '2^20-sample overlap-save blocks (836100 valid) -> Channelize(1024), '
                                 '2-pol complex64, HBM-resident input',
                        blocks_per_step_per_gpu=args.blocks, n_fft=N_FFT, n_chan=N_CHAN,
                        valid_samples_per_step_per_gpu=samples_per_step,
                        sharding='independent time blocks per rank, chirp broadcast over RCCL'
                        if world > 1 else 'single GPU'),
            roofline=roofline, roofline_path=roofline_path, cpu_baseline=cpu)
        if gather:
            line['gather'] = gather
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
