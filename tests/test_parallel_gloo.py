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


# ---- the REAL module through the sharded path (CPU tensors take CodeFormer's host branch: stock torch ops) ---------------------
def _real_worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(4)     # summation order of the CPU convolutions depends on the pool size: fixed for every process
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd import parallel
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    from oracle.synth import seeded_input
    parallel.init_distributed(backend='gloo', device='cpu')
    torch.manual_seed(0)
    net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                          connect_list=['32', '64', '128', '256']).eval()
    x = seeded_input(total)
    b = parallel.shard_bounds(total, world)
    # each rank restores its shard face by face (the reference's own call pattern, and on CPU the only batch-invariant one:
    # oneDNN picks its blocking by batch size); the gather is the single collective of the path
    with torch.no_grad():
        local = [net(x[i:i + 1], w=0.5, adain=True) for i in range(b[rank], b[rank + 1])]
    out_local = torch.cat([o[0] for o in local]) if local else x.new_zeros((0, 3, 512, 512))
    faces = parallel.gather_faces(out_local, total, dst=0)
    idx_local = torch.cat([o[1].argmax(-1) for o in local]) if local else torch.zeros((0, 256), dtype=torch.int64)
    idx = parallel.gather_faces(idx_local, total, dst=0)
    if rank == 0:
        with torch.no_grad():
            ref = [net(x[i:i + 1], w=0.5, adain=True) for i in range(total)]     # what ONE process computes
        ok = bool(torch.equal(faces, torch.cat([r[0] for r in ref]))) and bool(torch.equal(idx, torch.cat([r[1].argmax(-1) for r in ref])))
        # and through the batched helper (one forward per shard): same faces up to CPU batch-blocking noise
        full, _ = parallel.restore_sharded(net, parallel.shard(x, rank, world), total, w=0.5, adain=True, dst=0)
        ok = ok and float((full - faces).abs().max()) < 1e-3
        q.put(ok)
    else:
        parallel.restore_sharded(net, parallel.shard(x, rank, world), total, w=0.5, adain=True, dst=0)
    dist.barrier()
    dist.destroy_process_group()


def test_real_codeformer_two_ranks_equal_one_process_bitwise():
    """The actual CodeFormer module, 3 faces over 2 ranks (uneven shards 2 + 1): the gathered faces and code indices are
    bit-identical to the single-process result."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_entrypoint_under_two_ranks_writes_the_same_pngs(tmp_path):
    """inference_codeformer.py --has_aligned launched as 2 ranks (face list sharded, each rank writes its own results) produces
    byte-identical PNGs to the 1-rank run."""
    import subprocess
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(5)
    src = tmp_path / 'cropped_faces'
    os.makedirs(src)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)).save(src / f'f{i}.png')
    base = [sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '--has_aligned', '-i', str(src), '-w', '0.5', '--device', 'cpu',
            '--random_init_seed', '0']
    env = dict(os.environ, OMP_NUM_THREADS='4', MKL_NUM_THREADS='4')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run(base + ['-o', str(tmp_path / 'one')], capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    port = _free_port()
    procs = [subprocess.Popen(base + ['-o', str(tmp_path / 'two')], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(env, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)))
             for k in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()
    names = sorted(os.listdir(tmp_path / 'one' / 'restored_faces'))
    assert names == sorted(os.listdir(tmp_path / 'two' / 'restored_faces')) == ['f0.png', 'f1.png', 'f2.png']
    for n in names:
        a = np.asarray(Image.open(tmp_path / 'one' / 'restored_faces' / n))
        b = np.asarray(Image.open(tmp_path / 'two' / 'restored_faces' / n))
        assert np.array_equal(a, b), n


# ---- frames as the unit of sharding (whole-image / video path, codeformer_amd.video) ----------------------------------------------
def _frame_worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from codeformer_amd import parallel
    from codeformer_amd.video import frame_shard, gather_frames
    parallel.init_distributed(backend='gloo', device='cpu')
    frames = torch.arange(n_frames * 6 * 8 * 3, dtype=torch.int64).view(n_frames, 6, 8, 3).remainder(251).to(torch.uint8)
    mine = frame_shard(n_frames, rank, world)
    local = [255 - frames[i] for i in mine]                       # "restoration" of this rank's block of frames
    got = gather_frames(local, n_frames, dst=0)
    if rank == 0:
        q.put(bool(torch.equal(got, 255 - frames)) and got.dtype == torch.uint8)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [6, 7])
def test_frames_are_sharded_in_contiguous_blocks_and_gathered_once(n_frames):
    from codeformer_amd.video import frame_shard
    assert [list(frame_shard(7, r, 3)) for r in range(3)] == [[0, 1, 2], [3, 4], [5, 6]]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
