"""GPU: the hot-path kernels at BASELINE's full sizes (SDXL 1024² per-GPU batch shapes), checked through
size-independent properties instead of an oracle that would take minutes on the CPU: softmax rows sum to one,
linearity in V, key-permutation invariance; a centre-tap 3x3 convolution is a 1x1 linear and a single off-centre
tap is a shift; GroupNorm output statistics; a zero adapter leaves the base layer untouched and the adapter is
linear in lora_up; the NF4 GEMM equals the bf16 GEMM on the dequantised weight."""
import pytest
import torch
import torch.nn.functional as F

from .helpers import rel_err

pytestmark = pytest.mark.gpu


def _bf(gpu, *s, sc=1.0, seed=0):
    g = torch.Generator(device=gpu).manual_seed(seed)
    return (torch.randn(*s, generator=g, device=gpu, dtype=torch.float32) * sc).to(torch.bfloat16)


def test_attention_properties_at_sdxl_self_attention_size(gpu):
    from vision_ft_amd import kernels

    B, H, S, D = 4, 10, 4096, 64  # the 64x64-latent level of SDXL at 1024²
    q, k = _bf(gpu, B, H, S, D, seed=1), _bf(gpu, B, H, S, D, seed=2)
    v1, v2 = _bf(gpu, B, H, S, D, seed=3), _bf(gpu, B, H, S, D, seed=4)
    ones = torch.ones(B, H, S, D, device=gpu, dtype=torch.bfloat16)
    o1 = kernels.flash_attention(q, k, ones)
    assert (o1.float() - 1).abs().max() < 2e-2  # softmax rows sum to one
    oa, ob = kernels.flash_attention(q, k, v1), kernels.flash_attention(q, k, v2)
    oab = kernels.flash_attention(q, k, (v1.float() + v2.float()).to(torch.bfloat16))
    assert rel_err(oab, oa.float() + ob.float()) < 1e-2  # linear in V
    perm = torch.randperm(S, device=gpu)
    op = kernels.flash_attention(q, k[:, :, perm], v1[:, :, perm])
    assert rel_err(op, oa.float()) < 1e-2  # keys are a set
    # key-padding: masking the last keys == dropping them
    kv_len = torch.full((B,), S - 1000, dtype=torch.int32, device=gpu)
    om = kernels.flash_attention(q, k, v1, kv_len=kv_len)
    od = kernels.flash_attention(q, k[:, :, : S - 1000].contiguous(), v1[:, :, : S - 1000].contiguous())
    assert rel_err(om, od.float()) < 5e-3


def test_attention_backward_properties_at_full_size(gpu):
    from vision_ft_amd import kernels

    B, H, S, D = 2, 20, 1024, 64  # the 32x32-latent level
    q, k, v = (_bf(gpu, B, H, S, D, seed=s).requires_grad_(True) for s in (5, 6, 7))
    g1, g2 = _bf(gpu, B, H, S, D, seed=8), _bf(gpu, B, H, S, D, seed=9)
    o = kernels.flash_attention(q, k, v)
    ga = torch.autograd.grad(o, (q, k, v), g1, retain_graph=True)
    gb = torch.autograd.grad(o, (q, k, v), g2, retain_graph=True)
    gab = torch.autograd.grad(o, (q, k, v), (g1.float() + g2.float()).to(torch.bfloat16), retain_graph=True)
    for x, y, z in zip(ga, gb, gab):
        assert rel_err(z, x.float() + y.float()) < 1.5e-2  # the vector-Jacobian product is linear in dO
    # dV = P^T dO: with dO = ones every row of P sums to one, so sum_keys dV == number of queries per (b, h, d)
    (dv,) = torch.autograd.grad(o, (v,), torch.ones_like(o))
    assert rel_err(dv.float().sum(2), torch.full((B, H, D), float(S), device=gpu)) < 5e-3


def test_attention_backward_forms_agree_at_auraflow_size(gpu):
    """AuraFlow's attention at full size (4 096 patches + 264 text tokens, 12 heads x 256): the stored-dS backward (the dK/dV
    kernel writes dS^T, dQ is a batched GEMM on the weight-gradient kernel) against the dq + dkdv kernel pair on the same
    inputs (vft_attn_variant 3), linearity in dO, and the row-sum identity of dV."""
    from vision_ft_amd import kernels

    B, H, S, D = 2, 12, 4360, 256
    q, k, v = (_bf(gpu, B, S, H, D, seed=s).permute(0, 2, 1, 3).requires_grad_(True) for s in (21, 22, 23))  # (B, S, H, D) memory
    g1, g2 = (_bf(gpu, B, S, H, D, seed=s).permute(0, 2, 1, 3) for s in (24, 25))
    o = kernels.flash_attention(q, k, v)
    ga = torch.autograd.grad(o, (q, k, v), g1, retain_graph=True)
    gb = torch.autograd.grad(o, (q, k, v), g2, retain_graph=True)
    gab = torch.autograd.grad(o, (q, k, v), (g1.float() + g2.float()).to(torch.bfloat16), retain_graph=True)
    for x, y, z in zip(ga, gb, gab):
        assert rel_err(z, x.float() + y.float()) < 1.5e-2
    assert kernels._C.lib().vft_attn_variant(3) == 0
    try:
        gp = torch.autograd.grad(o, (q, k, v), g1, retain_graph=True)
    finally:
        kernels._C.lib().vft_attn_variant(0)
    for x, y in zip(ga, gp):
        assert rel_err(x, y.float()) < 6e-3
    (dv,) = torch.autograd.grad(o, (v,), torch.ones_like(o))
    assert rel_err(dv.float().sum(2), torch.full((B, H, D), float(S), device=gpu)) < 5e-3


def test_conv3x3_properties_at_full_size(gpu):
    from vision_ft_amd import kernels

    B, C, H = 16, 320, 128  # the 128x128-latent level, per-GPU batch of the bench
    x = _bf(gpu, B, C, H, H, seed=10).contiguous(memory_format=torch.channels_last)
    w1 = _bf(gpu, C, C, sc=C**-0.5, seed=11)
    b = _bf(gpu, C, sc=0.1, seed=12)
    w = torch.zeros(C, C, 3, 3, device=gpu, dtype=torch.bfloat16)
    w[:, :, 1, 1] = w1  # only the centre tap: the convolution is a per-pixel linear layer
    w = w.contiguous(memory_format=torch.channels_last)
    y = kernels.conv3x3(x, w, b)
    ref = F.linear(x.permute(0, 2, 3, 1), w1, b).permute(0, 3, 1, 2)
    assert rel_err(y, ref.float()) < 4e-3
    # a single off-centre tap with identity channel mixing shifts the image (zero padding at the border)
    ws = torch.zeros(C, C, 3, 3, device=gpu, dtype=torch.bfloat16)
    ws[:, :, 0, 2] = torch.eye(C, device=gpu, dtype=torch.bfloat16)
    ys = kernels.conv3x3(x, ws.contiguous(memory_format=torch.channels_last), None)
    shifted = torch.zeros_like(x)
    shifted[:, :, 1:, :-1] = x[:, :, :-1, 1:]  # y[oh, ow] = x[oh - 1, ow + 1]
    assert torch.equal(ys, shifted)
    # data gradient of the centre-tap convolution = the transposed linear
    xg = x.detach().clone().requires_grad_(True)
    gy = _bf(gpu, B, C, H, H, seed=13).contiguous(memory_format=torch.channels_last)
    kernels.conv3x3(xg, w, b).backward(gy)
    refg = (gy.permute(0, 2, 3, 1).float() @ w1.float()).permute(0, 3, 1, 2)
    assert rel_err(xg.grad, refg) < 4e-3


def test_groupnorm_statistics_at_full_size(gpu):
    from vision_ft_amd import kernels

    B, C, H = 16, 320, 128
    x = (_bf(gpu, B, C, H, H, seed=14).float() * 3 + 1.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = kernels.group_norm(x, None, None, 32, 1e-5, silu=False)
    yg = y.float().reshape(B, 32, C // 32, H * H).permute(0, 1, 3, 2).reshape(B, 32, -1)
    assert yg.mean(-1).abs().max() < 1e-2 and (yg.var(-1) - 1).abs().max() < 2e-2
    assert y.is_contiguous(memory_format=torch.channels_last)


def test_lora_properties_at_full_size(gpu):
    from vision_ft_amd import kernels

    M, K, N, r = 65536, 640, 5120, 4  # the GEGLU projection of the 64x64 level, B = 16
    lin = torch.nn.Linear(K, N).to(gpu, torch.bfloat16).requires_grad_(False)
    x = _bf(gpu, M, K, seed=15)
    A, B1, B2 = _bf(gpu, r, K, sc=K**-0.5, seed=16), _bf(gpu, N, r, sc=0.3, seed=17), _bf(gpu, N, r, sc=0.3, seed=18)
    base = lin(x)
    y0 = kernels.lora_linear(x, lin, A, torch.zeros_like(B1), 0.25)
    assert torch.equal(y0, base)  # zero-initialised lora_up: the wrapped layer is untouched (reference test_peft.py)
    d1 = kernels.lora_linear(x, lin, A, B1, 0.25).float() - base.float()
    d2 = kernels.lora_linear(x, lin, A, B2, 0.25).float() - base.float()
    d12 = kernels.lora_linear(x, lin, A, (B1.float() + B2.float()).to(torch.bfloat16), 0.25).float() - base.float()
    assert rel_err(d12, d1 + d2) < 3e-2  # (differences of bf16-rounded outputs: looser)


def test_nf4_gemm_equals_bf16_gemm_on_dequantised_weight_at_full_size(gpu):
    from vision_ft_amd import kernels
    from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4

    M, out_f, in_f = 17408, 3840, 2304  # Lumina2 fused qkv projection, B = 4
    w = _bf(gpu, out_f, in_f, sc=0.02, seed=19)
    q = quantize_nf4(w.float().cpu(), double_quant=True)
    layer = NF4Linear(in_f, out_f, bias=False)
    layer.load_state_dict({"weight" + k: v for k, v in q.items()})
    layer = layer.to(gpu)
    wd = layer.dequantize()
    x, gy = _bf(gpu, M, in_f, seed=20), _bf(gpu, M, out_f, seed=21)
    st = layer.quant_state()
    assert rel_err(kernels._nf4_gemm(x, st, None, False), (x.float() @ wd.float().t())) < 4e-3
    assert rel_err(kernels._nf4_gemm(gy, st, None, True), (gy.float() @ wd.float())) < 4e-3


def test_weight_gradient_kernels_at_full_size(gpu):
    """vft_linear_wgrad / vft_conv3x3_wgrad at the full fine-tune's bench sizes (BASELINE config 2), checked without a CPU
    product: (1) every tap block of the convolution's weight gradient is the linear weight gradient of dy against the
    correspondingly shifted activation — the centre tap with the unshifted one; (2) the sum of all entries equals
    sum_m rowsum(dy)[m] * rowsum(x)[m]; (3) linearity in x; (4) bit-identical repeats (ordered partial sums)."""
    from vision_ft_amd import kernels

    # projections: the 64x64-latent attention weights (98 304 tokens, 640 x 640) and the 32x32-latent ones
    for M, R, Cc, seed in ((98304, 640, 640, 20), (24576, 1280, 1280, 21)):
        dy, x1, x2 = _bf(gpu, M, R, sc=M**-0.5, seed=seed), _bf(gpu, M, Cc, seed=seed + 100), _bf(gpu, M, Cc, seed=seed + 200)
        d1, d2 = kernels.linear_wgrad(dy, x1), kernels.linear_wgrad(dy, x2)
        d12 = kernels.linear_wgrad(dy, (x1.float() + x2.float()).to(torch.bfloat16))
        assert rel_err(d12, d1.float() + d2.float()) < 1e-2  # linear in x (bf16 rounding of x1 + x2 and of the outputs)
        total = (dy.float().sum(1).double() * x1.float().sum(1).double()).sum()
        o32 = torch.empty(R, Cc, dtype=torch.float32, device=gpu)
        kernels.linear_wgrad(dy, x1, out=o32)
        assert abs(o32.double().sum() - total) < 2e-3 * (o32.double().abs().sum() ** 0.5 + 1)
        assert torch.equal(d1, kernels.linear_wgrad(dy, x1))
    # 3x3 convolution 320 -> 320 at 128x128, per-GPU batch 16
    B, C, H = 16, 320, 128
    x = _bf(gpu, B, C, H, H, seed=30).contiguous(memory_format=torch.channels_last)
    gy = _bf(gpu, B, C, H, H, sc=(B * H * H) ** -0.5, seed=31).contiguous(memory_format=torch.channels_last)
    dw = kernels.conv3x3_wgrad(gy, x).view(C, 3, 3, C)
    gy2 = gy.permute(0, 2, 3, 1).reshape(-1, C)
    centre = kernels.linear_wgrad(gy2, x.permute(0, 2, 3, 1).reshape(-1, C))
    assert rel_err(dw[:, 1, 1, :], centre.float()) < 4e-3
    shifted = torch.zeros_like(x)
    shifted[:, :, 1:, :-1] = x[:, :, :-1, 1:]  # tap (kh, kw) = (0, 2): x[oh - 1, ow + 1], zero beyond the border
    corner = kernels.linear_wgrad(gy2, shifted.permute(0, 2, 3, 1).reshape(-1, C))
    assert rel_err(dw[:, 0, 2, :], corner.float()) < 4e-3
    assert torch.equal(dw, kernels.conv3x3_wgrad(gy, x).view(C, 3, 3, C))


@pytest.mark.parametrize("hw", [(192, 80), (80, 192)])  # latents of the 640x1536 (w x h) bucket and its transpose
def test_non_square_bucket_at_full_size(gpu, hw):
    """BASELINE config 5's extreme buckets at full size (the round-2 full-size runs were square only): the 3x3 convolution
    (centre tap = per-pixel linear, off-centre tap = shift with zero padding at BOTH unequal borders, data gradient),
    GroupNorm statistics and self-attention over H*W = 15 360 tokens (rows sum to one, keys are a set, backward linear in
    dO) on the rectangular latent."""
    from vision_ft_amd import kernels

    H, W = hw
    B, C = 8, 320
    x = _bf(gpu, B, C, H, W, seed=40).contiguous(memory_format=torch.channels_last)
    w1 = _bf(gpu, C, C, sc=C**-0.5, seed=41)
    b = _bf(gpu, C, sc=0.1, seed=42)
    w = torch.zeros(C, C, 3, 3, device=gpu, dtype=torch.bfloat16)
    w[:, :, 1, 1] = w1
    w = w.contiguous(memory_format=torch.channels_last)
    y = kernels.conv3x3(x, w, b)
    assert y.shape == (B, C, H, W)
    assert rel_err(y, F.linear(x.permute(0, 2, 3, 1), w1, b).permute(0, 3, 1, 2).float()) < 4e-3
    for (kh, kw) in ((0, 2), (2, 0), (1, 0), (2, 1)):  # y[oh, ow] = x[oh + kh - 1, ow + kw - 1], zeros outside
        ws = torch.zeros(C, C, 3, 3, device=gpu, dtype=torch.bfloat16)
        ws[:, :, kh, kw] = torch.eye(C, device=gpu, dtype=torch.bfloat16)
        ys = kernels.conv3x3(x, ws.contiguous(memory_format=torch.channels_last), None)
        shifted = torch.zeros_like(x)
        dh, dw = kh - 1, kw - 1
        shifted[:, :, max(0, -dh):H - max(0, dh), max(0, -dw):W - max(0, dw)] = x[:, :, max(0, dh):H - max(0, -dh), max(0, dw):W - max(0, -dw)]
        assert torch.equal(ys, shifted), (kh, kw)
    xg = x.detach().clone().requires_grad_(True)
    gy = _bf(gpu, B, C, H, W, seed=43).contiguous(memory_format=torch.channels_last)
    kernels.conv3x3(xg, w, b).backward(gy)
    assert rel_err(xg.grad, (gy.permute(0, 2, 3, 1).float() @ w1.float()).permute(0, 3, 1, 2)) < 4e-3
    # weight gradient of the centre tap = the linear weight gradient over all pixels
    dw_ = kernels.conv3x3_wgrad(gy, x)
    ref = gy.permute(0, 2, 3, 1).reshape(-1, C).float().t() @ x.permute(0, 2, 3, 1).reshape(-1, C).float()
    assert dw_.shape == (C, 9 * C)  # rows = [tap][Cin]: the centre tap is block 4
    assert rel_err(dw_[:, 4 * C:5 * C].float(), ref) < 6e-3

    xs = (x.float() * 3 + 1.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    yn = kernels.group_norm(xs, None, None, 32, 1e-5, silu=False)
    yg = yn.float().reshape(B, 32, C // 32, H * W).permute(0, 1, 3, 2).reshape(B, 32, -1)
    assert yg.mean(-1).abs().max() < 1e-2 and (yg.var(-1) - 1).abs().max() < 2e-2

    Bh, Hh, S, D = 2, 5, H * W, 64  # SDXL's first attention level works on the 2x-downsampled latent; full H*W is the harder case
    q, k, v = (_bf(gpu, Bh, Hh, S, D, seed=s).requires_grad_(True) for s in (44, 45, 46))
    ones = torch.ones(Bh, Hh, S, D, device=gpu, dtype=torch.bfloat16)
    assert (kernels.flash_attention(q, k, ones).float() - 1).abs().max() < 2e-2
    o = kernels.flash_attention(q, k, v)
    perm = torch.randperm(S, device=gpu)
    assert rel_err(kernels.flash_attention(q, k[:, :, perm], v[:, :, perm]), o.float()) < 1e-2
    g1, g2 = _bf(gpu, Bh, Hh, S, D, seed=47), _bf(gpu, Bh, Hh, S, D, seed=48)
    ga = torch.autograd.grad(o, (q, k, v), g1, retain_graph=True)
    gb = torch.autograd.grad(o, (q, k, v), g2, retain_graph=True)
    gab = torch.autograd.grad(o, (q, k, v), (g1.float() + g2.float()).to(torch.bfloat16), retain_graph=True)
    for a_, b_, c_ in zip(ga, gb, gab):
        assert rel_err(c_, a_.float() + b_.float()) < 1.5e-2
    (dv,) = torch.autograd.grad(o, (v,), torch.ones_like(o))
    assert rel_err(dv.float().sum(2), torch.full((Bh, Hh, D), float(S), device=gpu)) < 5e-3


def test_round4_kernels_at_bench_size(gpu):
    """Round 4 at the bench's sizes: (a) the NF4 launch that dequantises once equals — bit for bit — the bf16 projection GEMM on
    the image it builds, at Lumina2's M = 17 408 (split launches of the tile table included), forward and data gradient, and is
    bit-reproducible; (b) the head_dim 64 / 96 forward on the LDS-DMA operand path agrees with the round-1 kernel (two
    roundings of the same softmax) at SDXL's and Lumina2's shapes, key mask included, and satisfies the row-sum property."""
    from vision_ft_amd import _C, kernels
    from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4

    M, out_f, in_f = 17408, 3840, 2304
    q = quantize_nf4(torch.randn(out_f, in_f) * in_f**-0.5, double_quant=True)
    layer = NF4Linear(in_f, out_f, bias=False)
    layer.load_state_dict({"weight" + k: v for k, v in q.items()}, strict=True)
    layer = layer.to(gpu)
    qs = layer.quant_state()
    pk = qs.packed.view(-1).long()
    idx = torch.stack([pk >> 4, pk & 15], dim=1).view(-1)
    dense = (qs.code.float()[idx] * qs.absmax_f32().repeat_interleave(64)).to(torch.bfloat16).view(out_f, in_f)
    x, dy = _bf(gpu, M, in_f, seed=21), _bf(gpu, M, out_f, seed=22)
    assert kernels.NF4_DEQUANT_ONCE
    y = kernels._nf4_gemm(x, qs, None, False)
    assert torch.equal(y, kernels.linear_fused(x, dense)) and torch.equal(y, kernels._nf4_gemm(x, qs, None, False))
    dx = kernels._nf4_gemm(dy, qs, None, True)
    assert torch.equal(dx, kernels.linear_fused(dy, dense.t().contiguous()))
    assert not kernels.nf4_barrier_timed_out()

    for (B, Hq, Hkv, S, D, lens) in ((4, 10, 10, 4096, 64, None), (2, 24, 8, 4352, 96, [4352, 4141])):
        qq = _bf(gpu, B, S, Hq, D, seed=31).permute(0, 2, 1, 3)
        kk, vv = _bf(gpu, B, S, Hkv, D, seed=32).permute(0, 2, 1, 3), _bf(gpu, B, S, Hkv, D, seed=33).permute(0, 2, 1, 3)
        kv = torch.tensor(lens, dtype=torch.int32, device=gpu) if lens else None
        try:
            new = kernels.flash_attention(qq, kk, vv, kv_len=kv)
            _C.lib().vft_attn_variant(1)
            old = kernels.flash_attention(qq, kk, vv, kv_len=kv)
        finally:
            _C.lib().vft_attn_variant(0)
        assert rel_err(new.float(), old.float()) < 5e-3, (D, rel_err(new.float(), old.float()))
        ones = torch.ones(B, S, Hkv, D, device=gpu, dtype=torch.bfloat16).permute(0, 2, 1, 3)
        assert (kernels.flash_attention(qq, kk, ones, kv_len=kv).float() - 1).abs().max() < 2e-2
