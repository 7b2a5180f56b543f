"""Can RCCL collectives issued through torch.distributed be captured into a hipGraph and
replayed?  Runs with ONE rank (the c10d capture path is the same as with N ranks)."""
import os, time, sys
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29581")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(256, 4096, device="cuda", dtype=torch.bfloat16)
pair = torch.randn(256, 2, device="cuda")
dist.all_reduce(x)                       # warm up the communicator outside capture
outs = [torch.empty_like(pair)]
dist.all_gather(outs, pair)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x * 2
    dist.all_reduce(y)
    parts = [torch.empty_like(pair)]
    dist.all_gather(parts, pair)
    z = y + parts[0].sum()
for i in range(20):
    g.replay()
torch.cuda.synchronize()
time.sleep(3)                            # give the watchdog thread time to complain
for i in range(20):
    g.replay()
torch.cuda.synchronize()
print("captured+replayed ok", float(z.float().mean()), float((x * 2).float().mean() + pair.sum()))
dist.destroy_process_group()
print("done")
