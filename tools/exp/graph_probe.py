"""Round 4: which ATen ops give wrong values in graph replays after an idle-stream synchronize (SDXL-LoRA step, B=8)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from vision_ft_amd.modules.loss import diffusion as LD  # noqa: E402
from vision_ft_amd.modules.timestep.sampling import uniform_randint  # noqa: E402
from vision_ft_amd.trainer.dp import DataParallelGrads  # noqa: E402
from vision_ft_amd.trainer.graph import GraphedTrainStep  # noqa: E402

B = 8
model = bench.build("sdxl_lora", dev)
params = [p for p in model.parameters() if p.requires_grad]
dp = DataParallelGrads(params)
opt = torch.optim.AdamW(params, lr=1e-3, fused=True, capturable=True)
g = torch.Generator(device=dev).manual_seed(42)
rn = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
lat, ctx, pooled = rn(B, 4, 128, 128), rn(B, 77, 2048), rn(B, 1280)
size = torch.tensor([[1024, 1024]] * B, device=dev)
crop = torch.zeros(B, 2, dtype=torch.long, device=dev)
probes = {}


def fwd():
    t = uniform_randint(lat.shape, dev)
    noisy, noise = LD.prepare_noised_latents(lat, t)
    pred = model(latents=noisy, timestep=t, encoder_hidden_states=ctx, encoder_pooler_output=pooled,
                 original_size=size, target_size=size, crop_coords_top_left=crop)
    with torch.no_grad():
        pf, nf = pred.detach().float(), noise.float()
        d = pf - nf
        probes["pow2_sum/N (global reduce)"] = d.pow(2).sum() / d.numel()
        probes["rows(1024)_sum_then_sum/N"] = (d * d).reshape(-1, 1024).sum(1).sum() / d.numel()
        probes["mse_loss (no grad)"] = F.mse_loss(pf, nf)
        probes["elementwise d[0]^2"] = (d * d).flatten()[0].clone()
        probes["amax |d|"] = d.abs().amax()
        probes["t[0] (randint)"] = t[0].float().clone()
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    probes["mse_loss (autograd) clone"] = loss.detach().clone()
    return loss


step = GraphedTrainStep(fwd, opt, dp, warmup=2)
for i in range(5):
    if i == 2:
        torch.cuda.current_stream().synchronize()
        print("---- idle stream synchronize ----")
    loss = step()
    print(i, "loss", round(float(loss), 5), {k: round(float(v), 5) for k, v in probes.items()}, flush=True)
