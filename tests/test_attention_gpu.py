"""GPU parity: HIP flash attention (through the C ABI) vs the CPU oracle."""
import math

import pytest
import torch

from oracle import attention as oattn

from .helpers import rel_err

pytestmark = pytest.mark.gpu

# tolerance for bf16 tensors vs the fp32-accumulated oracle: rtol 1.6e-2, atol 1e-2
RTOL, ATOL = 1.6e-2, 1e-2


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def _run_case(gpu, B, Hq, Hkv, Sq, Sk, D, kv_len=None, layout="bhsd", scale=None, seed=0):
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Hq, Sq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(B, Hkv, Sk, D, generator=g).to(torch.bfloat16)
    v = torch.randn(B, Hkv, Sk, D, generator=g).to(torch.bfloat16)
    do = torch.randn(B, Hq, Sq, D, generator=g).to(torch.bfloat16)
    kvl = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32)

    # oracle (fp32 math on the bf16 operands, autograd on CPU)
    q32, k32, v32 = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = oattn.sdpa(q32, k32, v32, scale=scale, kv_len=kvl, attention_dtype=torch.float32)
    ref.backward(do.float())

    def put(t):
        t = t.to(gpu)
        if layout == "bshd":  # (B,S,H,D) memory viewed as (B,H,S,D)
            t = t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        return t.requires_grad_(True)

    qg, kg, vg = put(q), put(k), put(v)
    out = kernels.flash_attention(qg, kg, vg, kv_len=None if kvl is None else kvl.to(gpu), scale=scale)
    out.backward(do.to(gpu))
    torch.cuda.synchronize()

    res = {
        "out": (out.detach().cpu(), ref.detach()),
        "dq": (qg.grad.cpu(), q32.grad),
        "dk": (kg.grad.cpu(), k32.grad),
        "dv": (vg.grad.cpu(), v32.grad),
    }
    for name, (got, want) in res.items():
        assert got.shape == want.shape, name
        assert torch.isfinite(got.float()).all(), f"{name} has non-finite values"
        rel = _rel(got, want)
        assert rel < 1.2e-2, f"{name}: relative L2 error {rel:.3e}"
        torch.testing.assert_close(got.float(), want.float(), rtol=RTOL, atol=ATOL * max(1.0, want.abs().max().item()))


@pytest.mark.parametrize(
    "B,Hq,Hkv,Sq,Sk,D",
    [
        (2, 10, 10, 1024, 1024, 64),   # SDXL self-attention, 32x32 latent level
        (1, 20, 20, 256, 77, 64),      # SDXL cross-attention (77 text tokens)
        (1, 5, 5, 333, 333, 64),       # ragged, not a tile multiple
        (1, 12, 12, 1160, 1160, 256),  # AuraFlow head_dim 256 (264 text + 896 patches)
        (1, 4, 4, 520, 520, 128),
        (1, 24, 8, 640, 640, 96),      # Lumina2 GQA 24q/8kv x 96
        (1, 2, 2, 1, 1, 64),           # single token
        (1, 1, 1, 130, 65, 256),
    ],
)
def test_attention_parity(gpu, B, Hq, Hkv, Sq, Sk, D):
    _run_case(gpu, B, Hq, Hkv, Sq, Sk, D)


def test_attention_bshd_layout(gpu):
    _run_case(gpu, 2, 10, 10, 512, 512, 64, layout="bshd")


def test_attention_kv_len_mask(gpu):
    # Lumina2 prefix key-padding mask (lumina2/denoiser.py:150-155) as kv_len
    _run_case(gpu, 3, 24, 8, 400, 400, 96, kv_len=[400, 333, 64], scale=math.sqrt(1 / 96))
    _run_case(gpu, 2, 4, 4, 200, 200, 64, kv_len=[1, 129])


def test_attention_large_logits_rescale(gpu):
    """Force the online-softmax rescale path: one key dominates late in the sequence."""
    from vision_ft_amd import kernels

    B, H, S, D = 1, 2, 512, 64
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, H, S, D, generator=g)
    k = torch.randn(B, H, S, D, generator=g)
    v = torch.randn(B, H, S, D, generator=g)
    k[:, :, 400] = q[:, :, 7] * 4.0  # spike in the 7th tile for query 7
    k[:, :, 130] = q[:, :, 300] * 3.0
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    ref = oattn.sdpa(q.float(), k.float(), v.float(), attention_dtype=torch.float32)
    out = kernels.flash_attention(q.to(gpu), k.to(gpu), v.to(gpu)).cpu()
    torch.testing.assert_close(out.float(), ref, rtol=RTOL, atol=2e-2)


def test_attention_rejects_cpu_and_bad_dim(gpu):
    from vision_ft_amd import kernels

    q = torch.randn(1, 1, 8, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        kernels.flash_attention(q, q, q)
    bad = torch.randn(1, 1, 8, 80, dtype=torch.bfloat16, device=gpu)
    with pytest.raises(ValueError):
        kernels.flash_attention(bad, bad, bad)


def test_attention_deterministic(gpu):
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(5)
    q, k, v, do = (torch.randn(1, 4, 700, 64, generator=g).to(torch.bfloat16).to(gpu) for _ in range(4))
    outs = []
    for _ in range(2):
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = kernels.flash_attention(qq, kk, vv)
        o.backward(do)
        outs.append((o.detach(), qq.grad, kk.grad, vv.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize(
    "B,Hq,Hkv,Sq,Sk,kv_len",
    [
        (2, 10, 10, 1024, 1024, None),          # 4 key blocks, few (batch, head) pairs: the key range is sliced, reduce pass
        (32, 8, 8, 300, 700, None),             # 256 pairs: ONE workgroup walks 3 key blocks, running dQ in its scratch plane
        (1, 6, 2, 200, 520, None),              # GQA at head_dim 64: three query heads per kv head share a workgroup
        (2, 4, 4, 333, 390, [390, 131]),        # ragged tiles + key-padding mask ending inside / before the second key block
        (1, 3, 3, 64, 129, None),               # the smallest key length the one-pass kernel takes
        (3, 5, 5, 77, 1030, [1030, 1, 600]),    # fewer queries than one tile, five key blocks, a fully masked tail block
    ],
)
def test_attention_one_pass_backward(gpu, B, Hq, Hkv, Sq, Sk, kv_len):
    """head_dim 64 with more than 128 keys runs the one-pass backward (attn_bwd1_kernel: dQ, dK, dV from one sweep, the
    running dQ kept per workgroup in an fp32 scratch plane): against the oracle, and bit-identical on a second run."""
    from vision_ft_amd import kernels

    _run_case(gpu, B, Hq, Hkv, Sq, Sk, 64, kv_len=kv_len, seed=Sq + Sk)
    g = torch.Generator().manual_seed(9)
    q, do = (torch.randn(B, Hq, Sq, 64, generator=g).to(torch.bfloat16).to(gpu) for _ in range(2))
    k, v = (torch.randn(B, Hkv, Sk, 64, generator=g).to(torch.bfloat16).to(gpu) for _ in range(2))
    kvl = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32, device=gpu)
    runs = []
    for variant in (0, 0, 3):  # 3 = the dq + dkdv pair where the one-pass kernel would run
        assert kernels._C.lib().vft_attn_variant(variant) == 0
        try:
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            kernels.flash_attention(qq, kk, vv, kv_len=kvl).backward(do)
            runs.append((qq.grad, kk.grad, vv.grad))
        finally:
            kernels._C.lib().vft_attn_variant(0)
    for a, b_ in zip(runs[0], runs[1]):
        assert torch.equal(a, b_)  # fixed summation order: reproducible
    for a, c in zip(runs[0], runs[2]):  # the two forms agree to bf16 rounding of differently ordered fp32 sums
        assert _rel(a, c) < 6e-3, _rel(a, c)


@pytest.mark.parametrize(
    "B,Hq,Hkv,Sq,Sk,D,layout",
    [
        (1, 12, 12, 1160, 1160, 256, "bshd"),  # AuraFlow: q / k / v as (B, S, H, D) memory
        (2, 4, 2, 520, 600, 128, "bhsd"),      # GQA (two query heads share K), Sq != Sk, a last tile of 8 queries
        (1, 2, 2, 2056, 512, 256, "bhsd"),     # few key blocks: the dK/dV kernel splits its query tiles, 9 row tiles in the dQ GEMM
    ],
)
def test_attention_stored_ds_backward(gpu, B, Hq, Hkv, Sq, Sk, D, layout):
    """head_dim >= 128 without a key mask (Sq a multiple of 8, >= 512 queries and keys): the dK/dV kernel writes dS^T and dQ is
    one batched GEMM on the weight-gradient kernel (vft_attn_bwd in include/vft_hip.h).  Against the oracle, bit-identical
    on a second run, and against the dq + dkdv pair (variant 3)."""
    from vision_ft_amd import kernels

    _run_case(gpu, B, Hq, Hkv, Sq, Sk, D, layout=layout, seed=Sq + D)
    g = torch.Generator().manual_seed(11)
    q, do = (torch.randn(B, Hq, Sq, D, generator=g).to(torch.bfloat16).to(gpu) for _ in range(2))
    k, v = (torch.randn(B, Hkv, Sk, D, generator=g).to(torch.bfloat16).to(gpu) for _ in range(2))
    runs = []
    for variant in (0, 0, 3):
        assert kernels._C.lib().vft_attn_variant(variant) == 0
        try:
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            kernels.flash_attention(qq, kk, vv).backward(do)
            runs.append((qq.grad, kk.grad, vv.grad))
        finally:
            kernels._C.lib().vft_attn_variant(0)
    for a, b_ in zip(runs[0], runs[1]):
        assert torch.equal(a, b_)
    for a, c in zip(runs[0], runs[2]):
        assert _rel(a, c) < 6e-3, _rel(a, c)


def test_key_padding_mask_with_holes_runs_on_the_kernel(gpu):
    """A boolean key-padding mask whose valid keys are NOT a prefix (the reference passes any mask to SDPA,
    src/modules/attention.py:121-129): served by key compaction + the length-masked kernel; forward and all three gradients
    against the fp32 oracle with the same mask, for the 4-D broadcast form the reference's callers build."""
    from oracle.attention import sdpa
    from vision_ft_amd.modules.attention import scaled_dot_product_attention

    g = torch.Generator().manual_seed(12)
    B, Hq, Hkv, Sq, Sk, D = 2, 6, 2, 150, 333, 96
    q, k, v = (torch.randn(B, h, s, D, generator=g).to(torch.bfloat16) for h, s in ((Hq, Sq), (Hkv, Sk), (Hkv, Sk)))
    do = torch.randn(B, Hq, Sq, D, generator=g).to(torch.bfloat16)
    km = torch.rand(B, Sk, generator=g) > 0.35
    mask4 = km[:, None, None, :].expand(B, Hq, Sq, Sk)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = sdpa(qr, kr, vr, mask=mask4, attention_dtype=torch.float32)
    ref.backward(do.float())
    qg, kg, vg = (t.to(gpu).requires_grad_(True) for t in (q, k, v))
    mask_g = km.to(gpu)[:, None, None, :].expand(B, Hq, Sq, Sk)  # the broadcast view the reference's callers build (lumina2/denoiser.py:150-155)
    out = scaled_dot_product_attention(qg, kg, vg, mask=mask_g)
    out.backward(do.to(gpu))
    assert rel_err(out.cpu(), ref) < 1e-2
    for got, want in ((qg.grad, qr.grad), (kg.grad, kr.grad), (vg.grad, vr.grad)):
        assert rel_err(got.cpu(), want) < 2e-2
    assert float(kg.grad[0][:, ~km[0]].abs().max()) == 0.0  # masked keys receive no gradient
    per_query = mask4.clone()
    per_query[0, 0, 0, 0] = ~per_query[0, 0, 0, 0]
    with pytest.raises(ValueError):
        scaled_dot_product_attention(qg, kg, vg, mask=per_query.to(gpu))
