"""Multi-GPU helpers: one process per GPU, independent overlap-save blocks.

The path shards with no halo exchange, in two ways.  (1) Time:
block m reads input [m * spf, m * spf + N) and nothing else (reference
base.py:783-790), so ranks own disjoint runs of blocks and re-read their own
overlap (`frame_range`, `gather_frames`).  (2) Sub-bands: the streams along
the first sample axis are independent series, so each rank takes a run of
them with its own chirp columns (`SubbandShard`) and the results are
concatenated along that axis (`gather_subbands`, the role
`combining.Concatenate` plays on the host, combining.py:176-211).  The only
collectives are one broadcast of the response (chirp) at plan time and an
optional gather of the outputs.  They go through ``torch.distributed``
(backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests) or, with a
`hip.Comm` (`make_comm`), through the library's own C-ABI entry points
``bbt_bcast_chirp`` / ``bbt_gather_output`` (RCCL, include/bbt_hip.h), which
is what a host without torch would bind.
"""
import numpy as np

__all__ = ['frame_range', 'make_comm', 'share_response', 'gather_frames', 'SubbandShard',
           'SubbandDedisperse', 'gather_subbands']


def frame_range(n_frames, rank, world):
    """Contiguous run [first, last) of frames owned by ``rank``; sizes differ
    by at most one, earlier ranks take the larger share."""
    base, extra = divmod(n_frames, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def make_comm(dist, src=0):
    """A `hip.Comm` (RCCL communicator behind the C ABI) for this process
    group: rank ``src`` creates the id, ``dist`` (any backend; only a small
    object broadcast) carries it to the others.  Call after `hip.set_device`."""
    from . import hip
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [hip.comm_unique_id() if rank == src else None]
    dist.broadcast_object_list(box, src)
    return hip.Comm(world, rank, box[0])


def share_response(task, torch, dist, device, src=0, comm=None):
    """Give every rank's overlap-save task the SAME response: rank ``src``
    evaluates it (float64 on the host), the others receive it with a
    broadcast, and each rank builds its plan from the device copy.

    ``dist`` None (single process) just builds the plan locally.  Works with
    CPU tensors under gloo (then the plan is not built: no GPU).
    """
    from . import hip
    if dist is None:
        task._get_plan()
        return None
    n = task._ih_samples_per_frame
    rank = dist.get_rank()
    if rank == src:
        task.DEVICE_CHIRP = False          # (evaluated once, on the host, and broadcast: also without a GPU)
        columns, index, n_plan = task._plan_layout()
        shape = torch.tensor(list(columns.shape) + [n_plan, int(bool(task._paired))],
                             dtype=torch.int64, device=device)
    else:
        shape = torch.zeros(4, dtype=torch.int64, device=device)
    dist.broadcast(shape, src)
    ncol, n_plan = int(shape[0].item()), int(shape[2].item())
    task._paired = bool(shape[3].item())
    task._single = n_plan == 1               # one stream: run unpadded (overlap_save._plan_layout)
    assert int(shape[1].item()) == n, "ranks disagree about the block length"
    if rank == src:
        resp = torch.view_as_real(torch.from_numpy(columns)).to(device).contiguous()
        idx = torch.from_numpy(index.astype(np.int32)).to(device)
    else:
        resp = torch.empty((ncol, n, 2), dtype=torch.float32, device=device)
        idx = torch.empty(n_plan, dtype=torch.int32, device=device)
    if comm is not None:               # the C ABI's own collective (bbt_bcast_chirp)
        comm.bcast_chirp(hip.DeviceArray((ncol, n), np.complex64, resp.data_ptr(), resp), src)
    else:
        dist.broadcast(resp, src)
    dist.broadcast(idx, src)
    if device.type == 'cuda':
        dev_resp = hip.DeviceArray((ncol, n), np.complex64, resp.data_ptr(), resp)
        if task._plan is not None:
            task._plan.close()
        task._plan = hip.OsmPlan(n, n_plan, dev_resp, idx.cpu().numpy())
    return torch.view_as_complex(resp), idx


def gather_frames(local, torch, dist, comm=None):
    """All-gather equally sized per-rank outputs in rank (= stream) order;
    through ``comm`` (bbt_gather_output) if given, else torch.distributed."""
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    local = local.contiguous()
    if comm is not None:
        from . import hip
        comm.gather_output(hip.as_device_array(local), hip.as_device_array(out))
    else:
        dist.all_gather_into_tensor(out, local)
    return out


def _slice_meta(value, lo, hi, n_axis, ndim_sample):
    """Slice metadata (frequency, sideband, polarization) along the first
    sample axis if it extends along it; broadcast (unit) axes stay."""
    if value is None:
        return None
    arr = np.asanyarray(value)
    if arr.ndim < ndim_sample or arr.shape[arr.ndim - ndim_sample] != n_axis:
        return value
    index = [slice(None)] * arr.ndim
    index[arr.ndim - ndim_sample] = slice(lo, hi)
    return arr[tuple(index)]


def SubbandShard(ih, rank, world):
    """The run of sub-bands (entries of the first sample axis) of stream ``ih``
    that ``rank`` of ``world`` processes, as a stream with the same interface:
    shape ``(n, k) + ih.shape[2:]``, metadata sliced to match.  Data stay where
    they are (host reads slice, device reads copy the strided columns once)."""
    from . import hip
    from .base import TaskBase, META_ATTRIBUTES
    from .device_task import DeviceTaskMixin, fetch_device

    n_sub = ih.shape[1]
    lo, hi = frame_range(n_sub, rank, world)
    ndim_sample = len(ih.shape) - 1
    meta = {key: _slice_meta(getattr(ih, key, None), lo, hi, n_sub, ndim_sample)
            for key in META_ATTRIBUTES}

    class _SubbandShard(DeviceTaskMixin, TaskBase):
        subbands = (lo, hi)

        def task(self, data):
            return np.ascontiguousarray(data[:, lo:hi])

        def _compute_frames(self, first, last, out):
            start, stop = self._frame_span(first, last)
            x = fetch_device(self.ih, start, stop - start)
            inner = out.row_bytes // (hi - lo)             # bytes of one sub-band of a sample
            hip.copy_2d(out, out.row_bytes, x, x.row_bytes, lo * inner, (hi - lo) * inner,
                        stop - start)

    return _SubbandShard(ih, shape=(ih.shape[0], hi - lo) + tuple(ih.shape[2:]),
                         **{k: v for k, v in meta.items() if v is not None})


def SubbandDedisperse(ih, dm, *, band_frequency, band_reference_frequency=None,
                      reference_frequency=None, samples_per_frame=None, **kwargs):
    """`Dedisperse` of a run of sub-bands with the block geometry of the WHOLE
    band: the padding of a multi-sub-band task is set by its extreme sub-band
    (reference dispersion.py:66-74 takes the max over all streams), so a rank
    that holds only some sub-bands must pad as the whole task would for its
    blocks -- and therefore its output -- to be those of the unsharded task.
    ``band_frequency`` / ``band_reference_frequency`` are the centre and
    reference frequencies of every sub-band of the band (Hz or quantities;
    reference defaults to the mean band centre like the reference's).  The
    chirp columns are this shard's own."""
    from . import units as u
    from .dispersion import Dedisperse

    band = u.to_hz(band_frequency)
    band_ref = None if band_reference_frequency is None else u.to_hz(band_reference_frequency)

    class _SubbandDedisperse(Dedisperse):
        def _padding_band(self, f_lo, f_hi, ref, half_rate):
            lo, hi = band - half_rate, band + half_rate
            return lo, hi, (np.mean(lo + hi) / 2. if band_ref is None else band_ref)

    if np.dtype(ih.dtype).kind != 'c':
        raise TypeError("SubbandDedisperse handles complex (baseband) sub-bands")
    return _SubbandDedisperse(ih, dm, reference_frequency=reference_frequency,
                              samples_per_frame=samples_per_frame, **kwargs)


def gather_subbands(local, torch, dist, comm=None, axis=1):
    """Concatenate equally shaped per-rank results along the sub-band axis, in
    rank order: ``(n, k, ...)`` -> ``(n, world * k, ...)`` on every rank
    (``axis`` = 1, a dedispersed stream), or along any later axis -- a
    channelized stream ``(n, n_chan, k, ...)`` has its sub-bands on axis 2."""
    world = dist.get_world_size()
    flat = gather_frames(local, torch, dist, comm)                 # (world * n, ...)
    n = local.shape[0]
    stacked = flat.reshape((world, n) + tuple(local.shape[1:]))     # (world, n, a1, a2, ...)
    order = tuple(range(1, axis + 1)) + (0,) + tuple(range(axis + 1, stacked.dim()))
    shape = tuple(local.shape[:axis]) + (world * local.shape[axis],) + tuple(local.shape[axis + 1:])
    return stacked.permute(*order).reshape(shape)
