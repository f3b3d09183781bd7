"""N>1 path on CPU: world_size-2 gloo processes shard the faces, run, and gather with ONE collective."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    from codeformer_amd.parallel import shard_bounds
    assert shard_bounds(128, 8) == [16 * i for i in range(9)]
    assert shard_bounds(5, 2) == [0, 3, 5]
    assert shard_bounds(3, 4) == [0, 1, 2, 3, 3]
    for n in range(0, 40):
        for w in range(1, 9):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(0 <= b[i + 1] - b[i] <= (n + w - 1) // w for i in range(w))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyNet(torch.nn.Module):
    """Per-face function with the CodeFormer call signature (faces are independent, like the real net)."""

    def forward(self, x, w=0.0, adain=False):
        y = x * (1.0 + w) + x.mean(dim=(1, 2, 3), keepdim=True)
        return y, y.flatten(1)[:, :4], y[:, :, :2, :2]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from codeformer_amd import parallel
    r, ws, dev = parallel.init_distributed(backend='gloo', device='cpu')
    assert (r, ws, str(dev)) == (rank, world, 'cpu')
    x = torch.arange(total * 3 * 4 * 4, dtype=torch.float32).view(total, 3, 4, 4) / 100.0
    local = parallel.shard(x, rank, world)
    faces, out = parallel.restore_sharded(TinyNet(), local, total, w=0.5, adain=True, dst=0)
    # the pipelined form bench.py uses: two batches, one gather in flight while the next batch is computed
    h1, _ = parallel.restore_sharded_async(TinyNet(), local, total, w=0.5, adain=True, dst=0)
    h2, _ = parallel.restore_sharded_async(TinyNet(), local * 2, total, w=0.25, adain=True, dst=0)
    f1, f2 = h1.wait(), h2.wait()
    if rank == 0:
        ref = TinyNet()(x, w=0.5)[0]
        ok = bool(torch.equal(faces, ref)) and faces.shape[0] == total
        ok = ok and bool(torch.equal(f1, ref)) and bool(torch.equal(f2, TinyNet()(x * 2, w=0.25)[0]))
        q.put(ok)
    else:
        assert faces is None and f1 is None and f2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [4, 5])
def test_two_rank_shard_and_gather_equals_single_process(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
