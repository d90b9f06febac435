"""A/B of two builds of the library on one GPU box: child processes, alternating, each timing the attention backward at the
SDXL shapes (median of 15 after a warm-up).  The parent never touches the GPU.

    python tools/exp/lib_ab.py old.so new.so [rounds]"""
import os
import subprocess
import sys

CHILD = r'''
import os, statistics, sys, torch
sys.path.insert(0, os.getcwd())
from vision_ft_amd import kernels
dev = torch.device("cuda:0")
def bf(*s): return torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)
out = []
for (B, H, S, D) in ((24, 10, 4096, 64), (24, 20, 1024, 64)):
    q, k, v = (bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(3))
    do = bf(B, S, H, D).permute(0, 2, 1, 3)
    ts = []
    for i in range(18):
        o = kernels.flash_attention(q, k, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o.backward(do); e1.record(); torch.cuda.synchronize()
        q.grad = k.grad = v.grad = None
        if i >= 3: ts.append(e0.elapsed_time(e1))
    out.append(f"S{S}: {statistics.median(ts):.3f} ms")
print("  ".join(out))
'''
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, VFT_LIB_PATH=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=280)
        print(f"{os.path.basename(lib):<20} {res.stdout.strip()} {res.stderr.strip()[-300:] if res.returncode else ''}", flush=True)
