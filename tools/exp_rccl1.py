#!/usr/bin/env python3
"""Lab: fixed cost of an RCCL collective call with ONE rank (no link in it): the floor under the per-sweep all-reduce of the
multi-GPU stop rule.  Device time per call (HIP events around a burst) and host time per call."""
import os, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.zeros(1, dtype=torch.float64, device="cuda")
big = torch.zeros(3 * 80000, dtype=torch.float64, device="cuda")
for name, x in (("all_reduce 8 B", t), ("all_reduce 1.9 MB", big)):
    for _ in range(20): dist.all_reduce(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 500
    h0 = time.perf_counter(); e0.record()
    for _ in range(n): dist.all_reduce(x)
    e1.record(); h1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:18s} device {1e3 * e0.elapsed_time(e1) / n:7.2f} us per call, host issue {1e6 * (h1 - h0) / n:7.2f} us per call", flush=True)
dist.destroy_process_group()
