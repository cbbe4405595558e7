"""Multi-rank logic on CPU: frame partitioning and the response broadcast /
output gather over torch.distributed with the gloo backend, world size 2."""
import os
import socket
import sys

import numpy as np
import pytest

from baseband_tasks_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_range_partitions_exactly():
    for n in (0, 1, 7, 64, 65, 1000):
        for world in (1, 2, 3, 8):
            spans = [sharding.frame_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import baseband_tasks_amd as bt
    from baseband_tasks_amd import units as u
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_total = 16 * 4096
        nh = bt.NoiseGenerator((n_total, 2), '2020-01-01T00:00:00', 1 * u.MHz, 4096, seed=5,
                               frequency=300 * u.MHz, sideband=np.array([1, -1]))
        # ranks deliberately disagree about the DM before the broadcast
        dd = bt.Dedisperse(nh, 5. if rank == 0 else 4.9, samples_per_frame=4096 - 1538)
        resp, idx = sharding.share_response(dd, torch, dist, torch.device('cpu'))
        first, last = sharding.frame_range(dd._n_frames(), rank, world)
        # fake "outputs": the frame indices owned, gathered in stream order
        local = torch.arange(first, first + 4, dtype=torch.float32).reshape(4, 1)
        gathered = sharding.gather_frames(local, torch, dist)
        # sub-band sharding: 6 sub-bands x 2 pol, each rank takes 3 with their metadata
        freq = (400. + 6.25 * np.arange(6)).reshape(6, 1) * u.MHz
        sb = bt.NoiseGenerator((4096, 6, 2), '2020-01-01T00:00:00', 1 * u.MHz, 1024, seed=9,
                               frequency=freq, sideband=1, polarization=['X', 'Y'])
        mine = sharding.SubbandShard(sb, rank, world)          # (reading it needs the GPU: -m gpu suite)
        lo, hi = mine.subbands
        part = np.ascontiguousarray(sb.read()[:, lo:hi])
        full = sharding.gather_subbands(torch.view_as_real(torch.from_numpy(part)), torch, dist)
        np.savez(os.path.join(result_dir, f'rank{rank}.npz'), resp=resp.numpy(), idx=idx.numpy(),
                 span=np.array([first, last]), gathered=gathered.numpy(),
                 sub_shape=np.array(mine.shape), sub_freq=np.asarray(mine.frequency, dtype=float),
                 sub_pol=np.asarray(mine.polarization), sub_span=np.array(mine.subbands),
                 sub_full=torch.view_as_complex(full.contiguous()).numpy())
    finally:
        dist.destroy_process_group()


def test_share_response_and_gather_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    # every rank ends up with rank 0's response (DM 5), bit for bit
    assert np.array_equal(r0['resp'], r1['resp']) and np.array_equal(r0['idx'], r1['idx'])
    import baseband_tasks_amd as bt
    from baseband_tasks_amd import units as u
    nh = bt.EmptyStreamGenerator((16 * 4096, 2), '2020-01-01T00:00:00', 1 * u.MHz, samples_per_frame=4096,
                                 frequency=300 * u.MHz, sideband=np.array([1, -1]))
    dd = bt.Dedisperse(nh, 5., samples_per_frame=4096 - 1538)
    dd.DEVICE_CHIRP = False                     # (the host evaluation: what rank 0 broadcasts)
    want, idx = dd._response_columns()
    assert np.array_equal(r1['resp'], want) and list(idx) == [0, 1]
    # spans tile the frames; gather is in rank order
    assert r0['span'][0] == 0 and r0['span'][1] == r1['span'][0]
    assert np.array_equal(r0['gathered'], r1['gathered'])
    assert r0['gathered'][:4, 0].tolist() == [0, 1, 2, 3]
    assert r0['gathered'][4, 0] == r1['span'][0]
    # sub-band shards: 3 sub-bands each, metadata follows, concatenation restores the stream
    sb = bt.NoiseGenerator((4096, 6, 2), '2020-01-01T00:00:00', 1 * u.MHz, 1024, seed=9,
                           frequency=(400. + 6.25 * np.arange(6)).reshape(6, 1) * u.MHz, sideband=1,
                           polarization=['X', 'Y'])
    whole = sb.read()
    for r, res in enumerate((r0, r1)):
        assert res['sub_shape'].tolist() == [4096, 3, 2] and res['sub_span'].tolist() == [3 * r, 3 * r + 3]
        assert np.allclose(res['sub_freq'].ravel(), (400. + 6.25 * np.arange(3 * r, 3 * r + 3)) * 1e6)
        assert res['sub_pol'].tolist() == ['X', 'Y']
        assert np.array_equal(res['sub_full'], whole)


def test_subband_dedisperse_keeps_the_whole_bands_block_geometry():
    """Config 4: every rank's shard pads like the unsharded 64-sub-band task
    (set by the lowest sub-band), whichever sub-bands it holds."""
    import baseband_tasks_amd as bt
    band = (403.125e6 + 6.25e6 * np.arange(64)).reshape(64, 1)
    spf = 2**24 - 2756522
    whole = bt.EmptyStreamGenerator((2**25, 64, 2), '2020-01-01T00:00:00', 6.25e6, samples_per_frame=2**24,
                                    frequency=band, sideband=1)
    ref = bt.Dedisperse(whole, 557., reference_frequency=band, samples_per_frame=spf)
    assert (ref._pad_start, ref._pad_end, ref._ih_samples_per_frame) == (1362235, 1394287, 2**24)
    for rank in (0, 3, 7):
        mine = sharding.SubbandShard(whole, rank, 8)
        lo, hi = mine.subbands
        assert (lo, hi) == (8 * rank, 8 * rank + 8)
        plain = bt.Dedisperse(mine, 557., reference_frequency=band[lo:hi], samples_per_frame=spf)
        dd = sharding.SubbandDedisperse(mine, 557., band_frequency=band, band_reference_frequency=band,
                                        reference_frequency=band[lo:hi], samples_per_frame=spf)
        assert (dd._pad_start, dd._pad_end, dd._ih_samples_per_frame, dd.samples_per_frame) == \
            (ref._pad_start, ref._pad_end, 2**24, spf)
        assert dd.shape == (ref.shape[0], 8, 2) and dd.start_time == ref.start_time
        if rank:
            assert plain._pad_start < ref._pad_start          # its own geometry would differ
        # chirp columns are the shard's own
        assert np.all(np.asarray(dd.frequency) == band[lo:hi])


def _bench(env, *argv, timeout=120):
    import subprocess
    e = dict(os.environ, **env)
    e.pop('WORLD_SIZE', None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus N` (no torchrun): the parent spawns N ranks,
    rank 0 prints the one JSON line, a failing rank fails the run.  Dry-run
    mode: rendezvous and exit codes only (gloo, no GPU)."""
    import json
    r = _bench(dict(BBT_BENCH_DRYRUN='1'), '--gpus', '2')
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0]) == dict(dry_run=True, n_gpus=2, rank_sum=3)
    r = _bench(dict(BBT_BENCH_DRYRUN='1', BBT_BENCH_DRYRUN_FAIL='1'), '--gpus', '2')
    assert r.returncode == 7 and 'rank 1 exited with 7' in r.stderr


def test_bench_config4_layout_on_eight_ranks():
    """`python bench.py --gpus 8 --workload config4` in dry-run mode (gloo, no GPU): the eight ranks
    lay their shares out with the classes the real run uses -- together they cover the 64 sub-bands
    once, in order; every rank pads like the whole band (the lowest sub-band sets it); the gathered
    channelized output has the sub-bands back in band order on their axis."""
    import json
    r = _bench(dict(BBT_BENCH_DRYRUN='1'), '--gpus', '8', '--workload', 'config4', timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['rank_sum'] == 36
    assert d['subbands'] == [[8 * r, 8 * r + 8] for r in range(8)]
    spf = 2**24 - 2756522
    n_out = 2 * 2**24 - 2756522
    for g in d['geometry']:              # pad_start, pad_end, block, kept per block, output samples, spectra
        assert g == [1362235, 1394287, 2**24, spf, n_out, n_out // 64 // 4096 * 4096]
    assert d['gathered_shape'] == [3, 64, 64, 2, 2]
    assert d['gathered_subband_axis'] == list(range(64))
    assert d['chirp_columns_per_rank'] == 8 and d['chirp_bytes_per_rank'] == 8 * 2**24 * 8


def test_bench_headline_collectives_dry_run_and_fallbacks():
    """`BBT_BENCH_DRYRUN=collectives python bench.py --gpus 2` (gloo, no GPU) walks the headline's
    two exchange steps through the helpers the real run uses: the chirp hand-out
    (`chirp_shared_or_local`) and the gather (`gathered_or_error`), and lists what each rank would
    send, how much, and which bound applies.  With `BBT_BENCH_INJECT=bcast,gather` both collectives
    fail on every rank: every rank then evaluates the chirp itself, `with_gather` records the
    error, and the run still ends with rank 0's one line and exit code 0 -- the sharded number
    never depends on a collective."""
    import json
    r = _bench(dict(BBT_BENCH_DRYRUN='collectives'), '--gpus', '2')
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['chirp_broadcast_used'] is True and d['with_gather'] == dict(gathered_ranks=[0, 1])
    c = d['collectives']
    assert c['data_path_collectives'] == 0
    ops = {call['op']: call for call in c['calls']}
    assert ops['broadcast']['bytes'] == 2**20 * 8 + 8 and ops['broadcast']['root'] == 0
    n_spec = (768 * 836100 // 1024 // 512) * 512
    assert ops['all_gather']['bytes_sent_per_rank_per_step'] == n_spec * 1024 * 16
    assert ops['all_gather']['bytes_received_per_rank_per_step'] == n_spec * 1024 * 16
    cap = c['bound']['gathered_cap_msamples_per_s']
    assert cap['ring_one_link'] < c['bound']['produced_gb_per_s_per_rank'] * 1e3 / 16 * 2      # link-bound
    r = _bench(dict(BBT_BENCH_DRYRUN='collectives', BBT_BENCH_INJECT='bcast,gather'), '--gpus', '2')
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['chirp_broadcast_used'] is False and 'injected gather failure' in d['with_gather']['error']
    assert r.stderr.count('chirp broadcast failed') == 2 and d['rank_sum'] == 3
