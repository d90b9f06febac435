"""Isolated check (round 4): full reductions to a scalar inside a captured HIP graph, replayed before / after a device-wide
synchronize.  python tools/exp/graph_reduce_bug.py"""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
torch.manual_seed(0)
a = torch.randn(24, 4, 128, 128, device=dev).to(torch.bfloat16)
b = torch.randn(24, 4, 128, 128, device=dev)
flat = torch.randn(10_465_280, device=dev).to(torch.bfloat16)


def ops():
    return {
        "mse_loss": F.mse_loss(a.float(), b, reduction="mean"),
        "pow_mean": (a.float() - b).pow(2).mean(),
        "rows_then_mean": ((a.float() - b) ** 2).view(24 * 64, -1).mean(dim=1).mean(),
        "sum": (a.float() - b).sum(),
        "vector_norm_bf16": torch.linalg.vector_norm(flat, 2.0, dtype=torch.float32),
    }


for _ in range(2):
    want = {k: float(v) for k, v in ops().items()}
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = ops()
for k, v in out.items():
    print(f"{k}: storage bytes {v.untyped_storage().nbytes()}")
for i in range(6):
    if i == 3:
        torch.cuda.synchronize()
        print("-- torch.cuda.synchronize() --")
    g.replay()
    got = {k: float(v) for k, v in out.items()}
    print(i, {k: (round(got[k], 5), "OK" if abs(got[k] - want[k]) <= 1e-4 * abs(want[k]) + 1e-6 else f"WRONG (eager {want[k]:.5f})") for k in got}, flush=True)
