"""Does gloo all_reduce on CUDA tensors work when two ranks share ONE GPU, for which sizes?"""
import os, sys, time, faulthandler
import torch, torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="env://")
for n in (1 << 10, 1 << 16, 1 << 20, 1 << 21, 1 << 22):
    for dt in (torch.float32, torch.bfloat16):
        x = torch.ones(n, device="cuda", dtype=dt)
        y = torch.mm(torch.randn(512, 512, device="cuda"), torch.randn(512, 512, device="cuda"))
        t0 = time.time()
        dist.all_reduce(x)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"n={n} {dt} ok {x[0].item()} {time.time()-t0:.3f}s", flush=True)
dist.destroy_process_group()
