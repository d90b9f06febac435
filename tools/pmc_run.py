"""Tiny driver for rocprofv3 --pmc passes: runs the dominant hand-written kernels at their headline
shapes a few times (SDXL self-attention fwd/bwd at B=24, LoRA rows fwd/bwd, wgrad).
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py"""
import sys

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731

B, H, S, D = 24, 10, 4096, 64
q = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
k = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
v = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
do = bf(B, S, H, D).permute(0, 2, 1, 3)
for _ in range(3):
    kernels.flash_attention(q, k, v).backward(do)

M, K, N, r = 98304, 640, 5120, 4
x, y, dy, dx = bf(M, K), bf(M, N), bf(M, N), bf(M, K)
A, Bw = bf(r, K), bf(N, r)
for _ in range(3):
    t, Bt = kernels._lora_fwd_(x, y, A, Bw, 0.25)
    u = kernels._lora_bwd_dx_(dy, dx, Bt, A, 0.25)
    kernels._lora_wgrad2(x, u, dy, t, A, Bw, 0.25)
torch.cuda.synchronize()
print("algorithmic bytes: attn_fwd q+k+v+o =", 4 * B * H * S * D * 2, " lora_fwd =", 2 * M * K + 4 * M * N,
      " lora_bwd_dx =", 2 * M * N + 4 * M * K, " lora_wgrad2 =", 2 * M * (K + N))

# the projection GEMM (lin8) at the SDXL headline shapes, per-GPU batch 24: the 64x64-latent FeedForward up-projection
# (tokens 98304, 640 -> 5120, rank-4 extension + bias) and the 32x32-latent one (tokens 24576, 1280 -> 10240)
import torch.nn.functional as F  # noqa: E402,F401

# tools/pmc_summarize.py tells the two shapes apart by launch order (3 plain launches each): pin the one-launch 256x256
# tile, or a split entry (cfg 11-13) of the tile table would double the launches per call and skew the attribution
from vision_ft_amd import _C  # noqa: E402

_C.check(_C.lib().vft_linear_variant(1), "vft_linear_variant")
for (M2, N2, K2) in [(98304, 5120, 640), (24576, 10240, 1280)]:
    x2, w2, b2 = bf(M2, K2), bf(N2, K2) * 0.05, bf(N2)
    t2, B2 = bf(M2, 4), bf(N2, 4)
    for _ in range(3):
        kernels.linear_fused(x2, w2, b2, ext_x=t2, ext_w=B2)
    torch.cuda.synchronize()
    print(f"lin8 {M2}x{N2}x{K2} algorithmic bytes (x + w + y + ext) =", 2 * (M2 * K2 + N2 * K2 + M2 * N2 + 4 * (M2 + N2)))
    del x2, w2, b2, t2, B2
_C.check(_C.lib().vft_linear_variant(0), "vft_linear_variant")

# implicit-GEMM convolution (SDXL 320->320 at 128x128, B=16) forward + data gradient
Bc, Ci, Co, Hc = 24, 320, 320, 128
xc = bf(Bc, Ci, Hc, Hc).contiguous(memory_format=torch.channels_last).requires_grad_(True)
wc = (bf(Co, Ci, 3, 3) * 0.02).contiguous(memory_format=torch.channels_last)
bc = bf(Co)
gyc = bf(Bc, Co, Hc, Hc).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    kernels.conv3x3(xc, wc, bc).backward(gyc)
torch.cuda.synchronize()
print("conv3x3 algorithmic bytes (x + y + w) =", 2 * Bc * Hc * Hc * (Ci + Co) + 2 * 9 * Ci * Co)

# weight-gradient kernel (wg8) of the full fine-tune, B = 16: the 1280 x 1280 attention projection over the 32x32-latent
# tokens, and the 3x3 convolution 640 -> 640 at 64x64 (implicit im2col)
Mw, Rw, Cw = 16384, 1280, 1280
dyw, xw = bf(Mw, Rw) * 0.01, bf(Mw, Cw)
for _ in range(3):
    kernels.linear_wgrad(dyw, xw)
Bq, Cq, Hq = 16, 640, 64
xq = bf(Bq, Cq, Hq, Hq).contiguous(memory_format=torch.channels_last)
gq = (bf(Bq, Cq, Hq, Hq) * 0.01).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    kernels.conv3x3_wgrad(gq, xq)
torch.cuda.synchronize()
print("wg8 linear algorithmic bytes (dy + x + dW) =", 2 * (Mw * Rw + Mw * Cw + Rw * Cw),
      " wg8 conv (dy + x + dW) =", 2 * (2 * Bq * Hq * Hq * Cq + 9 * Cq * Cq))

# the NF4 projection GEMM of Lumina2-QLoRA (round 4: the launch dequantises the weight once, then the bf16 main loop): qkv
# projection 2304 -> 3840 over B = 4 x 4352 tokens, rank-4 extension
from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4  # noqa: E402

Mn, out_f, in_f = 17408, 3840, 2304
qn = quantize_nf4(torch.randn(out_f, in_f) * in_f**-0.5, double_quant=True)
layer = NF4Linear(in_f, out_f, bias=False)
layer.load_state_dict({"weight" + k_: v_ for k_, v_ in qn.items()}, strict=True)
layer = layer.to(dev)
xn, tn, Bn = bf(Mn, in_f), bf(Mn, 4), bf(out_f, 4)
_C.check(_C.lib().vft_linear_variant(1), "vft_linear_variant")
for _ in range(3):
    kernels._nf4_gemm(xn, layer.quant_state(), None, False, ext_x=tn, ext_w=Bn)
_C.check(_C.lib().vft_linear_variant(0), "vft_linear_variant")
torch.cuda.synchronize()
print("lin8 NF4 algorithmic bytes (x + y + codes + absmax; the bf16 image written once and read by the main loop comes on top) =",
      2 * Mn * (in_f + out_f) + out_f * in_f // 2 + out_f * in_f // 16)
