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
from vision_ft_amd import _C
out = []
shapes = [(24, 10, 10, 4096, 64), (24, 20, 20, 1024, 64)]
if os.environ.get("VFT_AB_ALL"): shapes += [(6, 12, 12, 4360, 256), (4, 24, 8, 4352, 96)]
variants = [int(x) for x in os.environ.get("VFT_AB_VARIANTS", "0").split(",")]
if len(variants) > 1:  # the variants' gradients against variant 0's on a ragged, masked, grouped case
    q = bf(2, 1000, 6, 64).permute(0, 2, 1, 3).requires_grad_(True)
    k, v = (bf(2, 900, 2, 64).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(2))
    do = bf(2, 1000, 6, 64).permute(0, 2, 1, 3)
    kvl = torch.tensor([900, 517], device=dev, dtype=torch.int32)
    ref = None
    for var in variants:
        if _C.lib().vft_attn_variant(var) != 0: continue
        o = kernels.flash_attention(q, k, v, kv_len=kvl); o.backward(do); torch.cuda.synchronize()
        g = [t.grad.float().clone() for t in (q, k, v)]
        q.grad = k.grad = v.grad = None
        if ref is None: ref = g
        else: out.append(f"v{var} vs v{variants[0]} rel " + "/".join(f"{((a - b).norm() / b.norm()).item():.1e}" for a, b in zip(g, ref)))
    _C.lib().vft_attn_variant(0)
for (B, H, Hkv, S, D) in shapes:
    q = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
    k, v = (bf(B, S, Hkv, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(2))
    do = bf(B, S, H, D).permute(0, 2, 1, 3)
    for var in (variants if D == 64 else [0]):
        if _C.lib().vft_attn_variant(var) != 0: continue
        tf, tb = [], []
        for i in range(15):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); o = kernels.flash_attention(q, k, v); e[1].record(); o.backward(do); e[2].record(); torch.cuda.synchronize()
            q.grad = k.grad = v.grad = None
            if i >= 3: tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
        out.append(f"D{D} S{S} v{var}: f {statistics.median(tf):.3f} b {statistics.median(tb):.3f}")
    _C.lib().vft_attn_variant(0)
print(" | ".join(out))
'''
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, VFT_LIB_PATH=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=280)
        print(f"{os.path.basename(lib):<20} {res.stdout.strip()} {res.stderr.strip()[-300:] if res.returncode else ''}", flush=True)
