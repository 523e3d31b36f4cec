"""Sanity: an RCCL all-reduce can be recorded into a hipGraph through torch.distributed (1 rank)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1024, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x * 2
    dist.all_reduce(y)
g.replay(); torch.cuda.synchronize()
print("nccl-in-graph ok", float(y.sum()))
dist.destroy_process_group()
