"""Single-rank RCCL smoke (GPU box): process-group init with device_id, all_reduce, barrier, and the two collectives of
codeformer_amd.parallel.gather_faces in the asynchronous form bench.py uses (gather / all_gather_into_tensor, async_op=True,
joined later) -- a 1-GPU box cannot run world_size > 1 over RCCL, so this only proves the API path, not the transfer."""
import os

import torch
import torch.distributed as dist

os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29511')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t = torch.ones(4, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
slab = torch.empty(1, 4, device='cuda')
w1 = dist.gather(t * 3, gather_list=list(slab.unbind(0)), dst=0, async_op=True)
busy = torch.randn(4096, 4096, device='cuda') @ torch.randn(4096, 4096, device='cuda')   # compute enqueued while the gather flies
slab2 = torch.empty(1, 4, device='cuda')
w2 = dist.all_gather_into_tensor(slab2.view(4), t * 5, async_op=True)
w1.wait()
w2.wait()
torch.cuda.synchronize()
assert slab.tolist() == [[3.0] * 4] and slab2.tolist() == [[5.0] * 4] and bool(torch.isfinite(busy).all())
print('nccl single-rank ok (async gather / all_gather_into_tensor joined)', slab.tolist(), slab2.tolist())
dist.destroy_process_group()
