"""Does RCCL write to STDOUT when a communicator is created?  (It does: a five-line version banner — which would sit in front of
bench.py's one JSON line.)  Prints nothing itself; run with stdout redirected and count the lines."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1024, device=dev, dtype=torch.bfloat16)
dist.all_reduce(t)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("done", file=sys.stderr)
