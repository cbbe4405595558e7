#!/usr/bin/env python
"""Headline benchmark: Msamples/s through Dedisperse(DM=100) + Channelize(1k
channels) on a 2-pol complex64 stream resident in HBM (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload headline|config4]

One "step" = one pass of the hot path over one batch.

``--workload headline`` (default; BASELINE.json configs[1] + the metric
pipeline): BLOCKS overlap-save blocks of 2^20 samples per GPU (16 MHz band at
1000 MHz, DM 100: pad 104963 + 107513, 836100 valid samples per block),
coherently dedispersed and channelized to 1024 channels through the package's
task objects, Channelize(Dedisperse(DeviceStream)).read_device(), i.e. through
the C ABI.  Time blocks are independent, so each rank owns its own batch (weak
scaling, no data-path collective); the chirp is computed on rank 0 and
broadcast at plan time.

``--workload config4`` (BASELINE.json configs[3]): 400-800 MHz as 64 sub-bands
of 6.25 MHz x 2 pol, DM 557, 2^24-sample blocks, Channelize(64) => 4096
channels; each rank takes 8 sub-bands (`sharding.SubbandShard`; all 64 at
--gpus 8), results are concatenated along the sub-band axis with one
all-gather (`sharding.gather_subbands`).

Both workloads print the rate with outputs left sharded (`value`) and, for
more than one rank, the rate including the gather (`with_gather`).

Launching.  Under torchrun (WORLD_SIZE set) this process is one rank.  Plain
``python bench.py --gpus N`` with N > 1 starts the N ranks itself: the parent
touches neither torch nor the GPU, spawns one child per GPU with
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, relays rank 0's JSON line and exits
non-zero if any rank fails.  BBT_BENCH_BACKEND=gloo rehearses the multi-rank
path on a box with fewer GPUs than ranks.

After timing, outside the timed region, the result of the timed call shape is
compared with the CPU oracle at the first blocks, a mid-batch seam and the last
blocks (`verified`); the run fails if that check fails.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.               # MI355X_MICROARCH.md: 8 TB/s spec
TOL_REL_L2, TOL_MAX = 1e-6, 1e-5    # SURVEY 8(d) parity tolerance (float32 path vs float64-FFT oracle)

# headline workload (SURVEY 8d config 2 + metric pipeline)
N_FFT = 1 << 20
N_CHAN = 1024
DM = 100.
FS_HZ = 16e6
FC_HZ = 1000e6
PAD_START, PAD_END = 104963, 107513
ALG_BYTES_PER_SAMPLE = 36.07        # (8/eta + 8) * 2 pol, eta = 836100 / 2^20

# config 4
C4_NFFT = 1 << 24
C4_FS_HZ = 6.25e6
C4_DM = 557.
C4_PAD = 2756522                    # 1362235 + 1394287
C4_NCHAN = 64
C4_NSUB = 64
C4_ALG_BYTES_PER_ELEM = 8.0 / ((C4_NFFT - C4_PAD) / C4_NFFT) + 8.0     # 17.57 B per c64 element


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['headline', 'config4'], default='headline')
    ap.add_argument('--blocks', type=int, default=None,
                    help='overlap-save blocks per step per GPU (default 768 headline, 8 config4)')
    ap.add_argument('--subbands-per-rank', type=int, default=8, help='config4: sub-bands per rank')
    ap.add_argument('--cpu-seconds', type=float, default=20.,
                    help='target wall time of the all-core CPU baseline')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-verify', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='keep the HIP-event kernel timing out of the timed region (A/B check)')
    ap.add_argument('--no-gather', action='store_true', help='skip the timed all-gather (N > 1)')
    ap.add_argument('--no-host-path', action='store_true',
                    help='skip the host-to-host (.read()) measurement of the metric pipeline')
    ap.add_argument('--no-traffic', action='store_true',
                    help='do not measure roofline.traffic with rocprofv3 child runs (N = 1); replay the '
                         'figure on file under profiles/ instead')
    ap.add_argument('--traffic-child', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------
# roofline.traffic, measured: two short child runs of this workload under
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters only:
# MI355X_MICROARCH.md, HBM section), started BEFORE this process touches the GPU.
def measure_traffic(args):
    """{pass name: HBM-side bytes per launch} from the TCC counters, or (None, reason)."""
    import importlib.util
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None, 'this run is itself under a profiler'
    spec = importlib.util.spec_from_file_location('rocprof_db', os.path.join(ROOT, 'tools', 'rocprof_db.py'))
    rdb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rdb)
    tmp = tempfile.mkdtemp(prefix='bbt_traffic_', dir='/tmp')
    child = [sys.executable, os.path.abspath(__file__), '--traffic-child', '--workload', args.workload,
             '--steps', '2', '--warmup', '1', '--no-cpu', '--no-verify', '--no-host-path', '--no-traffic',
             '--no-kernel-timing']
    child += ['--blocks', str(min(args.blocks or 96, 96))] if args.workload == 'headline' else \
        ['--blocks', str(args.blocks or 8), '--subbands-per-rank', str(args.subbands_per_rank)]
    env = dict(os.environ, TMPDIR='/tmp')
    per = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            # (its own process group, so that a run that overstays can be ended together with its children)
            proc = subprocess.Popen([exe, '--pmc', counter, '-d', out, '-o', 'run', '--'] + child, cwd='/tmp',
                                    env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, f'rocprofv3 --pmc {counter} did not finish within 180 s'
            db = os.path.join(out, 'run_results.db')
            if proc.returncode or not os.path.exists(db):
                return None, f'rocprofv3 --pmc {counter} failed (rc {proc.returncode}): ' + err.decode()[-200:]
            import sqlite3
            acc = {}
            for name, cname, disp, value in sqlite3.connect(db).execute(
                    'select kernel_name, counter_name, dispatch_id, value from counters_collection'):
                k = rdb.short(name)
                if k and cname == counter:
                    a = acc.setdefault(k, [0.0, set()])
                    a[0] += value
                    a[1].add(disp)
            per[counter] = {k: a[0] / len(a[1]) for k, a in acc.items()}
    except Exception as exc:
        return None, f'{type(exc).__name__}: {exc}'[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    passes = {}
    for k in set(per['FETCH_SIZE']) | set(per['WRITE_SIZE']):
        # KiB -> bytes; FETCH_SIZE x 2 on gfx950 (the guide's correction for wide coalesced reads)
        b = 2.0 * per['FETCH_SIZE'].get(k, 0.0) * 1024 + per['WRITE_SIZE'].get(k, 0.0) * 1024
        for prefix, pass_name in rdb.PASS_OF:
            if k.startswith(prefix):
                passes[pass_name] = passes.get(pass_name, 0.0) + b
    if not passes:
        return None, 'no overlap-save kernels in the counter collection'
    return passes, ('measured by this invocation before the timed run: two child runs of the same workload (2 steps of '
                    '96 blocks) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; HBM-side bytes '
                    'per launch = FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KiB -> bytes; Infinity-Cache hits are counted')


# ---------------------------------------------------------------------------
# launcher: python bench.py --gpus N  ->  N ranks (no torch, no GPU call here)
def launch_ranks(args):
    n = args.gpus
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f'bench.py: rank {r} exited with {code}; stopping the others', file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------
# CPU baseline: the oracle (numpy restatement of the reference path) on the host
def _cpu_worker(job):
    """Dedisperse -> Channelize(1024) on n_blocks blocks of 2^20 through the
    oracle, one process; every block is fresh input.  Returns (valid samples,
    seconds in the loop)."""
    seed, n_blocks = job
    from oracle import bbt_oracle as orc
    g = orc.disperse_geometry(FS_HZ, FC_HZ / 1e6, 1, -DM)
    spf = N_FFT - g['pad_start'] - g['pad_end']
    h = orc.chirp(N_FFT, FS_HZ, FC_HZ / 1e6, 1, -DM, g['reference_frequency'])
    rng = np.random.default_rng(seed)
    # a ring of distinct input blocks (so the FFT input is not one cache-resident array)
    ring = [rng.standard_normal((N_FFT, 4), dtype=np.float32).view(np.complex64) for _ in range(3)]
    carry = np.empty((0, 2), np.complex64)
    orc.disperse_block(ring[0], h, g['pad_start'], spf, fft64=False)      # warm-up
    t0 = time.perf_counter()
    n_out = 0
    for b in range(n_blocks):
        y = orc.disperse_block(ring[b % 3], h, g['pad_start'], spf, fft64=False)
        y = np.concatenate([carry, y])
        k = (y.shape[0] // N_CHAN) * N_CHAN
        z = orc.channelize(y[:k], N_CHAN, fft64=False)
        carry = y[k:]
        n_out += z.shape[0] * N_CHAN
    return n_out, time.perf_counter() - t0


def usable_cores():
    """Cores this process may really use: min(os.cpu_count(), affinity mask,
    cgroup quota)."""
    total = os.cpu_count() or 1
    n = total
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, period = f.read().split()
            if q != 'max':
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, total, quota


def host_path(bt, torch, blocks=96, reps=3):
    """The metric pipeline host to host: samples in (page-locked) host memory -> ``read()`` ->
    NumPy array, i.e. the reference's own calling convention (base.py:389-438), PCIe both ways.
    Upload of run m + 1, transforms of run m and download of run m - 1 overlap
    (baseband-tasks_amd/host_pipeline.py).  Never part of `value`."""
    from baseband_tasks_amd import host_pipeline as hp
    spf = N_FFT - PAD_START - PAD_END
    n_in = (blocks - 1) * spf + N_FFT
    x = hp.pinned_empty((n_in, 2), np.complex64)
    rng = np.random.default_rng(3)
    base = rng.standard_normal((N_FFT, 4), dtype=np.float32).view(np.complex64)
    for s in range(0, n_in, N_FFT):
        x[s:s + N_FFT] = base[:min(N_FFT, n_in - s)]
    nh = bt.HostStream(x, '2020-01-01T00:00:00', FS_HZ, samples_per_frame=N_FFT, frequency=FC_HZ,
                       sideband=1, polarization=['X', 'Y'])
    dd = bt.Dedisperse(nh, DM)
    ch = bt.Channelize(dd, N_CHAN, samples_per_frame=512)
    run = 16
    dd.max_frames_per_call = run + 2
    ch.max_frames_per_call = run * spf // (512 * N_CHAN) + 1
    ch.read(ch.samples_per_frame)                    # plan, first touch of the pinned pools
    best, z = None, None
    for _ in range(reps):
        ch.invalidate_cache()
        dd.invalidate_cache()
        ch.seek(0)
        z = None
        t0 = time.perf_counter()
        z = ch.read()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n_valid = z.shape[0] * N_CHAN
    # the bus by itself: one direction alone, and both directions at once (what bounds this path)
    m = 256 << 20
    a, b = hp.pinned_empty((m,), np.uint8), hp.pinned_empty((m,), np.uint8)
    da, db = bt.hip.DeviceArray((m,), np.uint8), bt.hip.DeviceArray((m,), np.uint8)
    s1, s2 = hp.Stream(), hp.Stream()
    lib = bt.hip.lib()

    def up():
        bt.hip.check(lib.bbt_memcpy_h2d(da.ptr, a.ctypes.data, m, s1.handle))

    def down():
        bt.hip.check(lib.bbt_memcpy_d2h(b.ctypes.data, db.ptr, m, s2.handle))

    def rate(*copies, reps=6):
        for c in copies:
            c()
        s1.synchronize(), s2.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for c in copies:
                c()
        s1.synchronize(), s2.synchronize()
        return reps * m / (time.perf_counter() - t0) / 1e9
    # (both directions FIRST: a stream keeps the copy engine of its first copy, and two streams
    # whose first copies do not overlap end up sharing one -- host_pipeline.copy_streams)
    duplex = rate(up, down)
    alone = min(rate(up), rate(down))
    packed = None
    try:
        packed = host_path_packed(bt, hp, blocks, run, reps)
    except Exception as exc:
        packed = dict(error=f'{type(exc).__name__}: {exc}'[:300])
    res = dict(value=round(n_valid / best / 1e6, 1), unit='Msamples/s', blocks=blocks, blocks_per_run=run,
               seconds=round(best, 4), h2d_gbps=round(x.nbytes / best / 1e9, 2),
               d2h_gbps=round(z.nbytes / best / 1e9, 2), pcie_gbps_one_way_alone=round(alone, 1),
               pcie_gbps_each_way_together=round(duplex, 1),
               what='Channelize(Dedisperse(HostStream over page-locked memory)).read() -> NumPy array: '
                    'PCIe both ways, three streams (upload / transforms / download of consecutive runs '
                    'overlap).  pcie_gbps_one_way_alone / _each_way_together: 256 MiB copies on two fresh '
                    'streams, both directions at once first (a stream keeps the copy engine of its first '
                    'copy: two streams whose first copies do not overlap share one engine and take turns, '
                    '28.6 GB/s each way, which is what held this path at 1.7 G before the process-wide primed '
                    'stream pair of host_pipeline.copy_streams)',
               packed_input=packed)
    ch.close()
    dd.close()
    return res


def host_path_packed(bt, hp, blocks, run, reps):
    """The same pipeline fed the way a recorder feeds it: 8-bit complex samples in frames (32-byte
    header + 8192 complete samples x 2 pol x (I, Q) bytes) in page-locked host memory, uploaded
    packed and unpacked in HBM (`ingest.RawFrameStream`, bbt_unpack) -- 4 bytes per complete sample
    go up instead of 16; the spectra still come down as complex64."""
    from baseband_tasks_amd import ingest
    spf = N_FFT - PAD_START - PAD_END
    n_in = (blocks - 1) * spf + N_FFT
    per, header = 8192, 32
    frame = header + per * 4
    n_sets = -(-n_in // per)
    raw = hp.pinned_empty((n_sets * frame,), np.uint8)
    rng = np.random.default_rng(5)
    raw.reshape(n_sets, frame)[:] = rng.integers(0, 256, size=(1, frame), dtype=np.uint8)
    rs = ingest.RawFrameStream(raw, frame_nbytes=frame, header_nbytes=header, samples_per_frame=per, bits=8,
                               n_chan=2, complex_data=True, start_time='2020-01-01T00:00:00',
                               sample_rate=FS_HZ, frequency=FC_HZ, sideband=1, polarization=['X', 'Y'])
    dd = bt.Dedisperse(rs, DM, samples_per_frame=spf)
    ch = bt.Channelize(dd, N_CHAN, samples_per_frame=512)
    dd.max_frames_per_call = run + 2
    ch.max_frames_per_call = run * spf // (512 * N_CHAN) + 1
    ch.read(ch.samples_per_frame)
    best, z = None, None
    for _ in range(reps):
        for t in (ch, dd, rs):
            t.invalidate_cache()
        ch.seek(0)
        z = None
        t0 = time.perf_counter()
        z = ch.read()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n_valid = z.shape[0] * N_CHAN
    res = dict(value=round(n_valid / best / 1e6, 1), unit='Msamples/s', seconds=round(best, 4),
               h2d_gbps=round(raw.nbytes / best / 1e9, 2), d2h_gbps=round(z.nbytes / best / 1e9, 2),
               what='the same pipeline on 8-bit complex samples in 32800-byte frames (page-locked host memory), '
                    'unpacked in HBM: RawFrameStream -> Dedisperse -> Channelize -> read() -> NumPy complex64')
    for t in (ch, dd, rs):
        t.close()
    return res


def cpu_baseline(target_seconds):
    """P independent processes over disjoint runs of blocks, P =
    every core this job may use; also the single-process rate."""
    import multiprocessing as mp
    n1, t1 = _cpu_worker((1, 4))
    single = n1 / t1 / 1e6
    procs, total, quota = usable_cores()
    per_block = t1 / 4
    # three repeats of `per` blocks per process; all processes run at once, so
    # a block takes longer than alone (shared memory bandwidth): assume 2x
    per = max(2, int(target_seconds / 3 / (2 * per_block)))
    ctx = mp.get_context('spawn')
    rates, busy, wall = [], 0.0, 0.0
    with ctx.Pool(procs) as pool:
        pool.map(_cpu_worker, [(100 + i, 1) for i in range(procs)])       # start-up + warm-up
        for rep in range(3):                            # three repeats, median
            t0 = time.perf_counter()
            res = pool.map(_cpu_worker, [(1000 * rep + i, per) for i in range(procs)], chunksize=1)
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
    return dict(value=round(sorted(rates)[1], 2), unit='Msamples/s', cores=procs, kind='port',
                single_core=round(single, 2),
                sample=f'{procs} processes (os.cpu_count()={total}, cgroup quota={quota}) x {per} fresh '
                       f'blocks of 2^20 x 2 pol each through oracle/bbt_oracle.py (numpy {np.__version__} '
                       f'complex64 FFT) on {model}; median of 3 repeats, last {busy:.1f} s busy / '
                       f'{wall:.1f} s wall; one process alone: {single:.1f} Msamples/s')


# ---------------------------------------------------------------------------
def _errors(got, want):
    got = np.asarray(got, np.complex128).ravel()
    want = np.asarray(want, np.complex128).ravel()
    rms = np.sqrt(np.mean(np.abs(want) ** 2))
    return (np.linalg.norm(got - want) / np.linalg.norm(want), np.abs(got - want).max() / rms)


def verify_headline(x, z, n_blocks, n_spec):
    """Spot-check the spectra of the timed call against the oracle: around the
    first seam, a mid-batch seam and the last seam (plus the very first and
    last spectra).  x: torch (n_in, 2) complex64 on the device, z: DeviceArray
    (n_spec, 1024, 2)."""
    from oracle import bbt_oracle as orc
    g = orc.disperse_geometry(FS_HZ, FC_HZ / 1e6, 1, -DM)
    assert (g['pad_start'], g['pad_end']) == (PAD_START, PAD_END)
    spf = N_FFT - PAD_START - PAD_END
    h = orc.chirp(N_FFT, FS_HZ, FC_HZ / 1e6, 1, -DM, g['reference_frequency'])
    worst = [0.0, 0.0]
    checked = 0
    pairs = sorted({0, max(0, n_blocks // 2 - 1), max(0, n_blocks - 2)})
    for m in pairs:
        nb = min(2, n_blocks - m)
        xin = x[m * spf:(m + nb - 1) * spf + N_FFT].cpu().numpy()
        y = np.concatenate([orc.disperse_block(xin[b * spf:b * spf + N_FFT], h, PAD_START, spf)
                            for b in range(nb)])
        base = m * spf                                   # stream sample of y[0]
        s_lo = -(-base // N_CHAN)
        s_hi = min((base + nb * spf) // N_CHAN, n_spec)
        seam = (base + spf) // N_CHAN
        want_idx = sorted(set(range(s_lo, min(s_lo + 6, s_hi)))
                          | set(range(max(s_lo, seam - 3), min(seam + 4, s_hi)))
                          | set(range(max(s_lo, s_hi - 6), s_hi)))
        for s in want_idx:
            want = np.fft.fft(y[s * N_CHAN - base:(s + 1) * N_CHAN - base].astype(np.complex128), axis=0)
            got = z[s:s + 1].to_host()[0]
            e = _errors(got, want)
            worst = [max(worst[0], e[0]), max(worst[1], e[1])]
            checked += 1
    ok = worst[0] <= TOL_REL_L2 and worst[1] <= TOL_MAX
    return dict(ok=bool(ok), rel_l2=float(f'{worst[0]:.3e}'), max_over_rms=float(f'{worst[1]:.3e}'),
                spectra_checked=checked, blocks=[int(m) for m in pairs],
                what=f'{n_blocks}-block timed call vs oracle/bbt_oracle.py (float64 FFT) at the first, '
                     f'a mid-batch and the last block seam; tolerance {TOL_REL_L2:g} / {TOL_MAX:g}')


def verify_config4(x, z, freq_mhz, n_blocks, n_spec, spf):
    """config 4: sub-band 0 of this rank (the worst case at rank 0), the first
    spectra of block 0, the spectra around the first seam, the last spectra."""
    from oracle import bbt_oracle as orc
    f0 = float(freq_mhz[0])
    pad_start = 1362235          # the whole band's padding (its lowest sub-band), all ranks
    h = orc.chirp(C4_NFFT, C4_FS_HZ, f0, 1, -C4_DM, f0)
    worst = [0.0, 0.0]
    checked = 0
    nb = min(2, n_blocks)
    xin = x[:(nb - 1) * spf + C4_NFFT, 0].cpu().numpy()         # (n, 2)
    y = np.concatenate([orc.disperse_block(xin[b * spf:b * spf + C4_NFFT], h.reshape(-1, 1), pad_start, spf)
                        for b in range(nb)])
    s_hi = min(nb * spf // C4_NCHAN, n_spec)
    seam = spf // C4_NCHAN
    idx = sorted(set(range(0, 8)) | set(range(max(0, seam - 4), min(seam + 5, s_hi))) | set(range(s_hi - 8, s_hi)))
    for s in idx:
        want = np.fft.fft(y[s * C4_NCHAN:(s + 1) * C4_NCHAN].astype(np.complex128), axis=0)     # (64, 2)
        got = z[s:s + 1].to_host()[0][:, 0]                      # (64, nsub, 2) -> sub-band 0
        e = _errors(got, want)
        worst = [max(worst[0], e[0]), max(worst[1], e[1])]
        checked += 1
    ok = worst[0] <= TOL_REL_L2 and worst[1] <= TOL_MAX
    return dict(ok=bool(ok), rel_l2=float(f'{worst[0]:.3e}'), max_over_rms=float(f'{worst[1]:.3e}'),
                spectra_checked=checked,
                what=f'sub-band {f0} MHz of the {n_blocks}-block timed call vs the oracle at the start, the '
                     f'first block seam and the end of the first two blocks; tolerance {TOL_REL_L2:g} / {TOL_MAX:g}')


# ---------------------------------------------------------------------------
def _inject(what):
    """BBT_BENCH_INJECT=bcast,gather: make that collective fail on every rank (tests of the
    fall-backs below: the sharded `value` must stand without either)."""
    if what in os.environ.get('BBT_BENCH_INJECT', '').split(','):
        raise RuntimeError(f'injected {what} failure (BBT_BENCH_INJECT)')


def chirp_shared_or_local(share, local, rank):
    """The chirp hand-out with its fall-back: ``share()`` (rank 0 evaluates, one broadcast) or, if
    that raises, ``local()`` (every rank evaluates its own: same values, no collective).  Returns
    True when the broadcast was used."""
    try:
        _inject('bcast')
        share()
        return True
    except Exception as exc:          # the measurement stands without it
        print(f'bench.py: rank {rank}: chirp broadcast failed ({type(exc).__name__}: {exc}); '
              'computing it locally', file=sys.stderr)
        local()
        return False


def gathered_or_error(timed_gather, after_failure=None):
    """The `with_gather` object: the timed steps that end with the all-gather, or -- the sharded
    `value` stands either way -- a record of why they are missing."""
    try:
        _inject('gather')
        return timed_gather()
    except Exception as exc:
        if after_failure is not None:
            after_failure()
        return dict(error=f'{type(exc).__name__}: {exc}'[:300])


def headline_collectives(world, blocks, backend='nccl'):
    """What a headline run of `world` ranks sends, and which bound applies (DESIGN, multi-GPU):
    per rank one chirp broadcast at plan time, nothing on the data path; the optional gather of the
    channelized outputs is timed separately."""
    spf = N_FFT - PAD_START - PAD_END
    n_spec = ((blocks * spf) // N_CHAN // 512) * 512
    out_bytes = n_spec * N_CHAN * 2 * 8                   # complex64 spectra of 2 pol per rank and step
    link = 153e9                                          # one xGMI link, bytes/s
    per_gpu = 50e9                                        # complete samples/s one GPU computes (measured order)
    return dict(
        layout=f'time: every rank takes its own {blocks} consecutive overlap-save blocks (weak scaling)',
        calls=[dict(op='broadcast', what='chirp (C x N complex64) + column index', root=0,
                    bytes=int(N_FFT * 8 + 2 * 4), when='plan creation, once',
                    through='bbt_bcast_chirp' if os.environ.get('BBT_BENCH_COMM') == 'cabi' and backend == 'nccl'
                    else f'torch.distributed.broadcast ({"RCCL" if backend == "nccl" else backend})'),
               dict(op='all_gather', what='channelized spectra', bytes_sent_per_rank_per_step=int(out_bytes),
                    bytes_received_per_rank_per_step=int(out_bytes * (world - 1)),
                    when='with_gather only: after every step, outside `value`',
                    through='bbt_gather_output' if os.environ.get('BBT_BENCH_COMM') == 'cabi' and backend == 'nccl'
                    else 'all_gather_into_tensor')],
        data_path_collectives=0,
        bound=dict(sharded='per-GPU HBM roofline x N (no exchange)',
                   produced_gb_per_s_per_rank=round(per_gpu * 16 / 1e9, 1),     # 2 pol x complex64 per complete sample
                   gathered='all-gather over xGMI: a rank receives the other ranks\' shares over its inbound '
                            'links (153 GB/s each, 7 per GPU); it produces several times what one link carries, '
                            'so gathered outputs are link-bound and `value` leaves them sharded',
                   # whole-job complete samples/s a gather can sustain: world x link / (16 B x (world - 1))
                   gathered_cap_msamples_per_s=dict(
                       ring_one_link=round(link / 16 * world / max(world - 1, 1) / 1e6, 1),
                       direct_all_links=round(min(7, max(world - 1, 1)) * link / 16 * world / max(world - 1, 1) / 1e6, 1))))


def _timed_gather(args, torch, dist, sharding, comm, backend, dev, coll_dev, world, step, fence, z,
                  samples_per_step):
    """Steps that end with the collective (outputs gathered)."""
    zt = torch.view_as_real(torch.as_tensor(z, device=dev))
    n_g = max(2, args.steps // 4)

    def gathered_step():
        zz = step()
        t_ = torch.view_as_real(torch.as_tensor(zz, device=dev))
        if backend != 'nccl':
            t_ = t_.cpu()
        if args.workload == 'config4':            # (n_spec, 64 channels, sub-bands, 2 pol, re/im)
            return sharding.gather_subbands(t_, torch, dist, comm, axis=2)
        return sharding.gather_frames(t_, torch, dist, comm)
    g_out = gathered_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(n_g):
        g_out = gathered_step()
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    return dict(value=round(world * samples_per_step * n_g / dt / 1e6, 1), unit='Msamples/s',
                steps=n_g, ms_per_step=round(dt / n_g * 1e3, 4), gathered_shape=list(g_out.shape),
                bytes_received_per_rank_per_step=int(zt.numel() * 4 * (world - 1)),
                collective=('bbt_gather_output (C ABI, RCCL)' if comm is not None else
                            'all_gather_into_tensor over ' + ('RCCL/xGMI' if backend == 'nccl' else backend)))


def dry_run_rank(args):
    """BBT_BENCH_DRYRUN=1: exercise only the launch / rendezvous / exit-code
    plumbing (gloo on the CPU, no GPU, no kernels); used by the CPU tests.
    BBT_BENCH_DRYRUN_FAIL=<rank> makes that rank fail.  With --workload config4
    every rank also lays out its share of the 64 sub-bands exactly as the real
    run does (same classes, geometry only: no plan is built) and the ranks
    gather a stand-in output along the sub-band axis, so the multi-GPU layout --
    the cover of the band, the common padding, the gathered shape -- is checked
    without GPUs."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if os.environ.get('BBT_BENCH_DRYRUN_FAIL') == str(rank):
        sys.exit(7)
    if world > 1:
        dist.init_process_group('gloo')
    t = torch.tensor([rank + 1], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    line = dict(dry_run=True, n_gpus=world, rank_sum=int(t.item()))
    if args.workload == 'headline' and os.environ.get('BBT_BENCH_DRYRUN') == 'collectives':
        # the headline's exchange steps, walked without a GPU: the chirp hand-out with a stand-in
        # payload through the same fall-back helper the real run uses (BBT_BENCH_INJECT applies),
        # a stand-in gather, and the list of calls with their sizes and bounds
        blocks = args.blocks or 768
        payload = torch.arange(16, dtype=torch.float32) if rank == 0 else torch.zeros(16)
        local = torch.arange(16, dtype=torch.float32)

        def share():
            if world > 1:
                dist.broadcast(payload, 0)
        used = chirp_shared_or_local(share, lambda: payload.copy_(local), rank)
        assert torch.equal(payload, local)                # (either way every rank holds the chirp)

        def gather():
            mine = torch.full((2, 3), float(rank))
            got = [torch.zeros_like(mine) for _ in range(world)]
            if world > 1:
                dist.all_gather(got, mine)
            return dict(gathered_ranks=[int(g[0, 0]) for g in got] if world > 1 else [rank])
        line.update(chirp_broadcast_used=used, with_gather=gathered_or_error(gather),
                    collectives=headline_collectives(world, blocks, os.environ.get('BBT_BENCH_BACKEND', 'nccl')))
    if args.workload == 'config4':
        import baseband_tasks_amd as bt
        from baseband_tasks_amd import sharding
        nsub = args.subbands_per_rank
        spf = C4_NFFT - C4_PAD
        band = (403.125e6 + 6.25e6 * np.arange(C4_NSUB)).reshape(C4_NSUB, 1)
        whole = bt.EmptyStreamGenerator((2 * C4_NFFT, C4_NSUB, 2), '2020-01-01T00:00:00', C4_FS_HZ,
                                        samples_per_frame=C4_NFFT, frequency=band, sideband=1)
        mine = sharding.SubbandShard(whole, rank, world) if world * nsub == C4_NSUB else None
        k = (rank * nsub + np.arange(nsub)) % C4_NSUB            # (what the real run takes)
        if mine is not None:
            assert mine.subbands == (int(k[0]), int(k[-1]) + 1)
        else:
            mine = bt.EmptyStreamGenerator((2 * C4_NFFT, nsub, 2), '2020-01-01T00:00:00', C4_FS_HZ,
                                           samples_per_frame=C4_NFFT, frequency=band[k], sideband=1)
        dd = sharding.SubbandDedisperse(mine, C4_DM, band_frequency=band, band_reference_frequency=band,
                                        reference_frequency=band[k], samples_per_frame=spf)
        ch = bt.Channelize(dd, C4_NCHAN, samples_per_frame=4096)
        geo = torch.tensor([int(k[0]), int(k[-1]) + 1, dd._pad_start, dd._pad_end, dd._ih_samples_per_frame,
                            dd.samples_per_frame, dd.shape[0], ch.shape[0]], dtype=torch.int64)
        geos = [torch.zeros_like(geo) for _ in range(world)]
        if world > 1:
            dist.all_gather(geos, geo)
        else:
            geos = [geo]
        # a stand-in for the rank's output: 3 spectra of (channels, its sub-bands, 2 pol), filled with
        # the sub-band numbers, gathered along the sub-band axis like the real output
        local = torch.zeros((3, C4_NCHAN, nsub, 2, 2), dtype=torch.float32)
        local += torch.as_tensor(k, dtype=torch.float32).reshape(1, 1, nsub, 1, 1)
        full = sharding.gather_subbands(local, torch, dist, None, axis=2) if world > 1 else local
        line.update(subbands=[[int(g[0]), int(g[1])] for g in geos],
                    geometry=[[int(v) for v in g[2:]] for g in geos],
                    gathered_shape=list(full.shape),
                    gathered_subband_axis=[int(v) for v in full[0, 0, :, 0, 0]],
                    chirp_columns_per_rank=nsub,
                    chirp_bytes_per_rank=int(nsub * C4_NFFT * 8))
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_rank(args):
    if os.environ.get('BBT_BENCH_DRYRUN'):
        return dry_run_rank(args)
    # (GPU_MAX_HW_QUEUES is left at ROCm's default, as the package leaves it: rounds 4-5 set 16 here
    # for the host-path leg (+8 %), `value` is the same with 4, 6, 8 or 16, and reads of resident
    # chains shorter than the headline's lose up to 20 % with 8 or 16: profiles/r05_hw_queues.txt)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('BBT_BENCH_BACKEND', 'nccl')
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # BBT_BENCH_BACKEND=gloo rehearses the multi-rank code path on a box with fewer GPUs than
        # ranks (ranks then share devices; RCCL itself refuses two ranks on one device)
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
    coll_dev = dev

    import baseband_tasks_amd as bt
    from baseband_tasks_amd import sharding
    bt.hip.set_device(dev.index)
    bt.hip.set_stream(torch.cuda.current_stream().cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # BBT_BENCH_COMM=cabi: chirp broadcast and output gather through the library's own RCCL entry
    # points (bbt_bcast_chirp / bbt_gather_output) instead of torch.distributed's
    comm = None
    if world > 1 and os.environ.get('BBT_BENCH_COMM') == 'cabi' and backend == 'nccl':
        comm = sharding.make_comm(dist)

    gen = torch.Generator(device=dev)
    gen.manual_seed(12345 + rank)

    if args.workload == 'headline':
        blocks = args.blocks or 768
        spf = N_FFT - PAD_START - PAD_END
        n_in = (blocks - 1) * spf + N_FFT
        x = torch.view_as_complex(torch.randn((n_in, 2, 2), generator=gen, device=dev, dtype=torch.float32))
        ds = bt.DeviceStream(x, '2020-01-01T00:00:00', FS_HZ, samples_per_frame=N_FFT,
                             frequency=FC_HZ, sideband=1, polarization=['X', 'Y'])
        dd = bt.Dedisperse(ds, DM)
        assert (dd._ih_samples_per_frame, dd.samples_per_frame) == (N_FFT, spf)
        n_chan, ch_spf = N_CHAN, 512
        # chirp: rank 0 computes, everyone receives it (RCCL broadcast; no-op for one rank)
        shared = chirp_shared_or_local(
            lambda: sharding.share_response(dd, torch, dist if world > 1 else None, dev, comm=comm),
            dd._get_plan, rank)
        sharding_note = ('independent time blocks per rank, chirp ' +
                         (f'broadcast over {"RCCL" if backend == "nccl" else backend}' if shared
                          else 'evaluated on every rank (the broadcast failed)')) if world > 1 else 'single GPU'
        alg_bytes = ALG_BYTES_PER_SAMPLE
        streams = 2
        workload = ('configs[1]+metric pipeline: Dedisperse DM=100, 16 MHz BW at 1000 MHz, 2^20-sample '
                    'overlap-save blocks (836100 valid) -> Channelize(1024), 2-pol complex64, '
                    'HBM-resident input')
    else:
        blocks = args.blocks or 8          # (a call of 6-12 blocks keeps both lanes busy: DESIGN 4.4)
        nsub = args.subbands_per_rank
        spf = C4_NFFT - C4_PAD
        n_in = (blocks - 1) * spf + C4_NFFT
        # this rank's sub-bands of the 64 (rank r: [r * nsub, (r + 1) * nsub) mod 64)
        k = (rank * nsub + np.arange(nsub)) % C4_NSUB
        freq = (403.125e6 + 6.25e6 * k).reshape(nsub, 1)
        x = torch.view_as_complex(torch.randn((n_in, nsub, 2, 2), generator=gen, device=dev,
                                              dtype=torch.float32))
        ds = bt.DeviceStream(x, '2020-01-01T00:00:00', C4_FS_HZ, samples_per_frame=C4_NFFT,
                             frequency=freq, sideband=1)
        # block geometry (padding) is the whole band's, set by its lowest sub-band, on every rank
        band = (403.125e6 + 6.25e6 * np.arange(C4_NSUB)).reshape(C4_NSUB, 1)
        dd = sharding.SubbandDedisperse(ds, C4_DM, band_frequency=band, band_reference_frequency=band,
                                        reference_frequency=freq, samples_per_frame=spf)
        assert (dd._ih_samples_per_frame, dd.samples_per_frame, dd._pad_start) == (C4_NFFT, spf, 1362235), \
            (dd._ih_samples_per_frame, dd.samples_per_frame, dd._pad_start)
        n_chan, ch_spf = C4_NCHAN, 4096
        dd._get_plan()                       # every rank evaluates its own chirp columns: nothing to share
        sharding_note = (f'{nsub} of 64 sub-bands per rank (sharding.SubbandShard layout), own chirp '
                         'columns per rank, all-gather along the sub-band axis') if world > 1 \
            else f'single GPU: {nsub} of 64 sub-bands'
        streams = 2 * nsub
        alg_bytes = C4_ALG_BYTES_PER_ELEM * streams
        workload = (f'configs[3] share: {nsub} sub-bands x 2 pol of 6.25 MHz (400-800 MHz band), DM=557 with '
                    'per-sub-band reference frequency, 2^24-sample blocks (14020694 valid) -> '
                    'Channelize(64), complex64, HBM-resident input')

    ch = bt.Channelize(dd, n_chan, samples_per_frame=ch_spf)
    dd.max_frames_per_call = blocks
    n_spec = (dd.shape[0] // n_chan // ch_spf) * ch_spf
    ch.max_frames_per_call = n_spec // ch_spf + 1
    samples_per_step = n_spec * n_chan                  # valid output complete samples

    def step():
        dd.invalidate_cache()
        ch.invalidate_cache()
        ch.seek(0)
        return ch.read_device(n_spec)

    # Per-kernel timing of the overlap-save passes: HIP events on the stream
    # each pass is launched on, recorded inside the timed region itself, in the
    # normal multi-lane schedule (what rocprofv3 --kernel-trace sees as well).
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
        z = step()
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * samples_per_step * args.steps / elapsed / 1e6
    n_steps_timed = args.steps
    if args.no_kernel_timing:               # A/B switch: time the kernels in extra steps instead
        plan.timing_enable(1)
        n_steps_timed = max(3, args.steps // 2)
        for _ in range(n_steps_timed):
            z = step()
        fence()
    ms, pass_launches, pass_blocks = plan.timing_read_passes()
    plan.timing_enable(0)

    # ---- the timed call shape, checked against the oracle (outside the timed region)
    verified = None
    if not args.no_verify:
        if args.workload == 'headline':
            verified = verify_headline(x, z, blocks, n_spec)
        else:
            verified = verify_config4(x, z, freq.ravel() / 1e6, blocks, n_spec, spf)
        flag = torch.tensor([0 if verified['ok'] else 1], device=coll_dev, dtype=torch.int32)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        verified['all_ranks_ok'] = bool(int(flag.item()) == 0)

    # ---- the same passes isolated (one lane, nothing else on the GPU), for reference
    plan.timing_enable(2)
    n_iso = max(2, args.steps // 4)
    for _ in range(n_iso):
        step()
    fence()
    ms_iso, _, blocks_iso = plan.timing_read_passes()
    plan.timing_enable(0)
    info = plan.info()
    names = ['osm_col_forward', 'osm_rowpass', 'osm_col_inverse']
    # (the stage schedule puts events on a sample of the launches only: per-pass counts)
    per_block = [m / max(b, 1) for m, b in zip(ms, pass_blocks)]
    k = int(np.argmax(per_block))
    launches = pass_launches[k]
    avg_ms = ms[k] / max(launches, 1)
    units_per_launch = pass_blocks[k] / max(launches, 1) * spf      # launches may be ragged
    achieved = units_per_launch * alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    traffic_note = 'no PMC profile on file for this workload'
    live = getattr(args, 'traffic_live', None)
    if live and live[0] and names[k] in live[0]:
        traffic, traffic_note = live[0][names[k]], live[1]
    elif os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get(args.workload, tj).get(names[k])
            traffic_note = ('replayed from profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE x 2 + '
                            'WRITE_SIZE per launch, collected in separate passes with the same kernels; '
                            'not measured in this run), source ' + str(tj.get('source', 'n/a')))
            if live and live[1]:
                traffic_note += '; live measurement: ' + str(live[1])
        except Exception:
            traffic = None
    roofline = dict(bound='hbm', kernel=names[k], achieved=round(achieved, 1), peak=HBM_PEAK_GBPS,
                    unit='GB/s', frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic,
                    traffic_note=traffic_note,
                    avg_launch_ms=round(avg_ms, 5), launches=launches,
                    blocks_per_launch=round(pass_blocks[k] / max(launches, 1), 3),
                    launches_per_step=-(-blocks // info['chunk_blocks']),
                    alg_bytes_per_unit=round(alg_bytes, 2),
                    pass_ms_per_block={n: round(m, 6) for n, m in zip(names, per_block)},
                    pass_ms_per_block_isolated={n: round(m / max(b, 1), 6)
                                                for n, m, b in zip(names, ms_iso, blocks_iso)},
                    note='launch durations from HIP events inside the timed region on the stream each '
                         'pass runs on, in the normal schedule (a pass shares the GPU with the other '
                         'passes in flight; with one stream per pass the events sit on every '
                         'BBT_OSM_TIMING_STRIDE-th launch, `launches` of them); isolated = one stream, '
                         'nothing else running; achieved = alg_bytes_per_unit x valid samples per launch '
                         '/ avg launch duration of the dominant pass')
    path_gbps = value * 1e6 / world * alg_bytes / 1e9
    roofline_path = dict(bound='hbm', achieved=round(path_gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                         frac=round(path_gbps / HBM_PEAK_GBPS, 4),
                         input_msamples_per_s=round(value * dd._ih_samples_per_frame / spf, 1),
                         note='whole path per GPU: alg_bytes_per_unit x valid samples / wall time; '
                              'input_msamples_per_s = value / eta (all GPUs)')

    # ---- outputs gathered: time steps that end with the collective
    with_gather = None
    if world > 1 and not args.no_gather:
        with_gather = gathered_or_error(
            lambda: _timed_gather(args, torch, dist, sharding, comm, backend, dev, coll_dev, world,
                                  step, fence, z, samples_per_step),
            torch.cuda.synchronize)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.workload == 'headline':
        cpu = cpu_baseline(args.cpu_seconds)
    host = None
    if rank == 0 and world == 1 and not args.no_host_path and args.workload == 'headline':
        try:
            host = host_path(bt, torch)
        except Exception as exc:            # (reported, never fails the resident number)
            host = dict(error=f'{type(exc).__name__}: {exc}'[:300])

    failed = bool(verified and not verified['all_ranks_ok'])
    if rank == 0:
        line = dict(
            metric='Msamples/s through Dedisperse(DM=100)+Channelize(1k ch), 2-pol c64'
            if args.workload == 'headline' else
            'M complete samples/s (sub-bands x 2 pol per sample) through Dedisperse(DM=557)+Channelize(64)',
            value=round(value, 1), unit='Msamples/s', n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
            higher_is_better=True, scaling='weak', vs_baseline=None, dtype='c64',
            data='synthetic',
            config=dict(workload=workload, blocks_per_step_per_gpu=blocks,
                        n_fft=N_FFT if args.workload == 'headline' else C4_NFFT, n_chan=n_chan,
                        streams_per_gpu=streams,
                        valid_samples_per_step_per_gpu=samples_per_step,
                        outputs='left sharded on the ranks (value); with_gather includes the all-gather',
                        sharding=sharding_note),
            roofline=roofline, roofline_path=roofline_path, cpu_baseline=cpu, verified=verified,
            with_gather=with_gather, host_path=host)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if failed:
        print('bench.py: verification against the oracle FAILED', file=sys.stderr)
        sys.exit(3)


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if (int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.no_traffic and not args.traffic_child
            and not os.environ.get('BBT_BENCH_DRYRUN')):
        args.traffic_live = measure_traffic(args)          # (children; this process has not touched the GPU yet)
    run_rank(args)


if __name__ == '__main__':
    main()
