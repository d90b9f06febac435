"""Round 4: which gradients of the SDXL full fine-tune go non-finite / wrong in graph replays after an idle-stream synchronize.
   python tools/exp/graph_full_nan.py [batch]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from vision_ft_amd.trainer.dp import DataParallelGrads  # noqa: E402
from vision_ft_amd.trainer.graph import GraphedTrainStep  # noqa: E402

model = bench.build("sdxl_full", dev)
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
params = [p for _, p in named]
dp = DataParallelGrads(params)
opt = torch.optim.AdamW(params, lr=0.0, fused=True, capturable=True)  # lr 0: the weights stay put, every replay sees the same model
fwd = bench.make_step("sdxl_full", model, dev, B, seed=42)
step = GraphedTrainStep(fwd, opt, dp, warmup=2, split=True)  # graph A = forward + backward; the optimizer graph is not replayed here


def grads():
    return {n: p.grad.detach().float().clone() for n, p in named}


def replay():
    dp.zero_grad()
    step.graph_a.replay()
    torch.cuda.synchronize()
    return float(step.loss), grads()


# the graph's random draws advance per replay: compare replay k pre-sync with replay k of a second, identical process instead?
# simpler: non-finite values and gross outliers
l0, g0 = replay()
print("replay 0 (before any idle sync... the synchronize above IS a device sync after work): loss", l0, flush=True)
bad0 = [n for n, g in g0.items() if not torch.isfinite(g).all()]
print("non-finite grads in replay 0:", len(bad0), bad0[:10], flush=True)
torch.cuda.current_stream().synchronize()
torch.cuda.synchronize()
for k in range(1, 4):
    l, g = replay()
    bad = [n for n, gg in g.items() if not torch.isfinite(gg).all()]
    # relative size against replay 0 (different noise draw, same scale expected)
    ratio = sorted(((float(g[n].norm() / (g0[n].norm() + 1e-30)), n) for n in g if n not in bad), reverse=True)
    print(f"replay {k}: loss {l}; non-finite grads: {len(bad)} {bad[:12]}", flush=True)
    print("    largest |g| / |g0|:", [(round(r, 2), n) for r, n in ratio[:8]], flush=True)
    print("    smallest |g| / |g0|:", [(round(r, 4), n) for r, n in ratio[-5:]], flush=True)
