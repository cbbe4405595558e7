"""Multi-GPU helpers: one process per GPU, independent overlap-save blocks.

The path shards with no halo exchange: block m reads input
[m * spf, m * spf + N) and nothing else (reference base.py:783-790), so ranks
own disjoint runs of blocks and re-read their own overlap.  The only
collectives are one broadcast of the response (chirp) at plan time and an
optional gather of the outputs; both go through ``torch.distributed``
(backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).
"""
import numpy as np

__all__ = ['frame_range', 'share_response', 'gather_frames']


def frame_range(n_frames, rank, world):
    """Contiguous run [first, last) of frames owned by ``rank``; sizes differ
    by at most one, earlier ranks take the larger share."""
    base, extra = divmod(n_frames, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def share_response(task, torch, dist, device, src=0):
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
        columns, index = task._response_columns()
        shape = torch.tensor(list(columns.shape), dtype=torch.int64, device=device)
    else:
        shape = torch.zeros(2, dtype=torch.int64, device=device)
    dist.broadcast(shape, src)
    ncol = int(shape[0].item())
    assert int(shape[1].item()) == n, "ranks disagree about the block length"
    if rank == src:
        resp = torch.view_as_real(torch.from_numpy(columns)).to(device).contiguous()
        idx = torch.from_numpy(index.astype(np.int32)).to(device)
    else:
        resp = torch.empty((ncol, n, 2), dtype=torch.float32, device=device)
        idx = torch.empty(task._n_stream_even, dtype=torch.int32, device=device)
    dist.broadcast(resp, src)
    dist.broadcast(idx, src)
    if device.type == 'cuda':
        dev_resp = hip.DeviceArray((ncol, n), np.complex64, resp.data_ptr(), resp)
        if task._plan is not None:
            task._plan.close()
        task._plan = hip.OsmPlan(n, task._n_stream_even, dev_resp, idx.cpu().numpy())
    return torch.view_as_complex(resp), idx


def gather_frames(local, torch, dist):
    """All-gather equally sized per-rank outputs in rank (= stream) order."""
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out
