"""Batch sharding of independent faces over ranks + the single gather at the end.

One process per GPU (torchrun / torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for
tests).  Faces are independent (no cross-sample op anywhere in CodeFormer.forward: GroupNorm, AdaIN and attention
are per-sample), so the data path needs NO collective; the only exchange is one `gather` of the restored faces to the
destination rank (3.1 MB per face fp32, each peer over its direct xGMI link to the root).

The reference has no multi-GPU inference; this is the MI355X-side addition named by the north star.
"""
import os

import torch
import torch.distributed as dist


# 'gather' (one collective to the destination rank) unless the backend lacks it, then one all_gather (set once, by all ranks)
_GATHER_IMPL = [os.environ.get('CODEFORMER_GATHER', 'gather')]


def env_rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init_distributed(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (no-op for WORLD_SIZE=1).

    Returns (rank, world_size, device)."""
    rank, world, local = env_rank_world()
    use_cuda = torch.cuda.is_available() if device is None else torch.device(device).type == 'cuda'
    if use_cuda:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    else:
        dev = torch.device('cpu')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # the host driver only supports dmabuf IPC
        if backend is None:
            backend = 'nccl' if use_cuda else 'gloo'
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = dev
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, dev


def shard_bounds(n, world):
    """Contiguous, balanced partition of n faces: rank r owns [b[r], b[r+1]).  Earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def shard(x, rank, world):
    b = shard_bounds(x.shape[0], world)
    return x[b[rank]:b[rank + 1]]


class PendingGather:
    """Handle of a gather that may still be in flight (see gather_faces(async_op=True)); wait() returns what the blocking
    call returns: the (total, ...) tensor on the destination rank, None elsewhere."""

    def __init__(self, work, finish, keep):
        self._work, self._finish, self._keep = work, finish, keep   # keep: buffers the collective reads / writes

    def wait(self):
        if self._work is not None:
            self._work.wait()      # NCCL: the current stream waits for the collective's stream; gloo: blocks the host
            self._work = None
        out = self._finish() if self._finish is not None else None
        self._finish = self._keep = None
        return out


def gather_faces(local, total, dst=0, group=None, async_op=False):
    """Gather per-rank outputs (n_r, ...) to `dst` in rank order with ONE collective.

    Shards may be uneven (shard_bounds); every rank pads to the largest shard so a plain `gather` can be used.
    Returns the (total, ...) tensor on dst, None elsewhere.

    async_op=True returns a PendingGather instead: the collective is enqueued (RCCL runs it on its own stream, ordered
    after the kernels that produced `local`) and the caller goes on launching the next batch; `wait()` later joins it.
    A serving loop keeps one gather in flight, so the xGMI transfer of batch i overlaps the compute of batch i+1.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return PendingGather(None, lambda: local, None) if async_op else local
    rank = dist.get_rank(group)
    bounds = shard_bounds(total, world)
    nmax = max(bounds[r + 1] - bounds[r] for r in range(world))
    tail = local.shape[1:]
    if local.shape[0] != nmax:
        pad = torch.zeros((nmax - local.shape[0],) + tuple(tail), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    else:
        send = local.contiguous()
    slab = work = None
    if _GATHER_IMPL[0] == 'gather':
        try:
            if rank == dst:
                slab = torch.empty((world, nmax) + tuple(tail), dtype=local.dtype, device=local.device)
                work = dist.gather(send, gather_list=list(slab.unbind(0)), dst=dst, group=group, async_op=True)
            else:
                work = dist.gather(send, gather_list=None, dst=dst, group=group, async_op=True)
        except (NotImplementedError, RuntimeError) as e:  # a backend without gather: every rank takes the same branch
            if 'gather' not in str(e).lower() and not isinstance(e, NotImplementedError):
                raise
            _GATHER_IMPL[0] = 'all_gather'
    if _GATHER_IMPL[0] == 'all_gather':
        slab = torch.empty((world, nmax) + tuple(tail), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(slab.view((world * nmax,) + tuple(tail)), send, group=group, async_op=True)

    def finish():
        if rank != dst:
            return None
        parts = [slab[r, :bounds[r + 1] - bounds[r]] for r in range(world)]
        if all(p.shape[0] == nmax for p in parts):
            return slab.view((world * nmax,) + tuple(tail))
        return torch.cat(parts, dim=0)

    pending = PendingGather(work, finish, (send, slab))
    return pending if async_op else pending.wait()


@torch.no_grad()
def restore_sharded(net, x_local, total, w=0.5, adain=True, dst=0):
    """Run the local shard through `net` and gather the restored faces on `dst` (one collective).

    x_local: this rank's (n_r,3,512,512) faces already on the rank's device.  Returns (faces or None, local tuple).
    """
    out = net(x_local, w=w, adain=adain)
    return gather_faces(out[0], total, dst=dst), out


@torch.no_grad()
def restore_sharded_async(net, x_local, total, w=0.5, adain=True, dst=0):
    """As restore_sharded, but the gather is left in flight: returns (PendingGather, local tuple).  Call `.wait()` on the
    handle after launching the next batch (bench.py keeps exactly one gather pending)."""
    out = net(x_local, w=w, adain=adain)
    return gather_faces(out[0], total, dst=dst, async_op=True), out
