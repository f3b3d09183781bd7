import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29511')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t = torch.ones(4, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
slab = torch.empty(1, 4, device='cuda'); dist.gather(t, gather_list=list(slab.unbind(0)), dst=0)
print('nccl single-rank ok', slab.tolist()); dist.destroy_process_group()
