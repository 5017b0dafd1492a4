"""GPU: what does one gradient all-reduce cost inside a chain of kernels (one-rank RCCL group, the N > 1 code path)?"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
n = int(os.environ.get("N", 170000))
g = torch.zeros(n, device="cuda")
a = torch.zeros(1 << 20, device="cuda")
dist.all_reduce(g); torch.cuda.synchronize()
def loop(with_ar, k=200, work=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        for _ in range(work):
            a.add_(1.0)
        if with_ar == 1:
            dist.all_reduce(g)
        elif with_ar == 2:
            w = dist.all_reduce(g, async_op=True); w.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
for _ in range(2):
    base = loop(0); ar = loop(1); ar2 = loop(2)
print(f"floats {n}: chain of 20 small kernels {base:.1f} us; + all_reduce {ar:.1f} us (+{ar - base:.1f}); async+wait {ar2:.1f} us (+{ar2 - base:.1f})")
dist.destroy_process_group()
