"""One-rank RCCL probe of the calls bench.py makes at N>1 (init with device_id, barrier, float64 MAX all-reduce)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t0 = time.perf_counter()
dist.barrier()
torch.cuda.synchronize(dev)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
dist.barrier()
print("nccl probe ok: barrier + f64 MAX all-reduce in %.2f s" % (time.perf_counter() - t0))
dist.destroy_process_group()
