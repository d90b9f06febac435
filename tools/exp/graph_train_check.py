"""Round 4: does an idle-stream synchronize before graph replays change what the replays COMPUTE (not just the reported loss)?
   python tools/exp/graph_train_check.py eager|graph|graph_sync <out.pt> [B] [workload]     (6 optimizer steps at batch B of a bench workload; default SDXL-LoRA at B=8)"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

mode, out = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
WL = sys.argv[4] if len(sys.argv) > 4 else "sdxl_lora"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from vision_ft_amd.trainer.dp import DataParallelGrads  # noqa: E402
from vision_ft_amd.trainer.graph import GraphedTrainStep  # noqa: E402

model = bench.build(WL, dev)
params = [p for p in model.parameters() if p.requires_grad]
dp = DataParallelGrads(params)
opt = torch.optim.AdamW(params, lr=1e-3, fused=True, capturable=True)
fwd = bench.make_step(WL, model, dev, B, seed=42)
losses = []
if mode == "eager":
    for i in range(6):
        l_ = fwd(); l_.backward(); opt.step(); dp.zero_grad()
        losses.append(float(l_))
else:
    step = GraphedTrainStep(fwd, opt, dp, warmup=2)
    for i in range(6):
        if mode == "graph_sync" and i == 1:
            torch.cuda.current_stream().synchronize()
            torch.cuda.synchronize()
        losses.append(float(step()))
print(mode, "losses", [round(x, 5) for x in losses], flush=True)
torch.save({"losses": losses, "down": torch.cat([p.detach().flatten().float().cpu() for n, p in model.named_parameters() if "lora_down" in n]).to(torch.bfloat16),
            "up": torch.cat([p.detach().flatten().float().cpu() for n, p in model.named_parameters() if "lora_up" in n]).to(torch.bfloat16)}, out)
