"""GPU parity of the fused LoRA kernels (C ABI: vft_lora_proj / rank_update / wgrad) vs the oracle."""
import pytest
import torch

from oracle import ops

from .helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["gemm_fused", "adapter_passes"], autouse=True)
def lora_path(request):
    """Every test runs on both compositions: the adapter inside the projection GEMM (vft_linear_fwd with the rank
    extension, the default) and the round-1 one (library GEMM + in-place adapter passes, also the fallback for shapes
    the fused GEMM does not take)."""
    from vision_ft_amd import kernels

    old = kernels.FUSED_LORA_GEMM
    kernels.FUSED_LORA_GEMM = request.param == "gemm_fused"
    yield request.param
    kernels.FUSED_LORA_GEMM = old


def _case(gpu, M, K, N, r, s=0.5, bias=True, seed=0, lead=None):
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(seed)
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    shape = lead + (K,) if lead else (M, K)
    x, W = bf(*shape), bf(N, K, sc=K**-0.5)
    b = bf(N, sc=0.1) if bias else None
    A, B = bf(r, K, sc=K**-0.5), bf(N, r, sc=0.3)
    gy = bf(*(shape[:-1] + (N,)))
    # oracle: fp32 math on the bf16 operands (lora.py:92-104)
    P = {"l.linear.weight": W.float(), "l.lora_down.weight": A.float().requires_grad_(True),
         "l.lora_up.weight": B.float().requires_grad_(True), "l.alpha": torch.tensor(s * r)}
    if bias:
        P["l.linear.bias"] = b.float()
    xo = x.float().requires_grad_(True)
    yo = ops.linear(P, "l", xo)
    yo.backward(gy.float())

    lin = torch.nn.Linear(K, N, bias=bias).to(gpu, torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(W)
        if bias:
            lin.bias.copy_(b)
    lin.requires_grad_(False)
    xg = x.to(gpu).requires_grad_(True)
    Ag, Bg = A.to(gpu).requires_grad_(True), B.to(gpu).requires_grad_(True)
    y = kernels.lora_linear(xg, lin, Ag, Bg, s)
    y.backward(gy.to(gpu))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), yo) < 6e-3, rel_err(y.cpu(), yo)
    assert rel_err(xg.grad.cpu(), xo.grad) < 6e-3
    assert rel_err(Ag.grad.cpu(), P["l.lora_down.weight"].grad) < 1e-2, rel_err(Ag.grad.cpu(), P["l.lora_down.weight"].grad)
    assert rel_err(Bg.grad.cpu(), P["l.lora_up.weight"].grad) < 1e-2, rel_err(Bg.grad.cpu(), P["l.lora_up.weight"].grad)


@pytest.mark.parametrize("M,K,N,r", [
    (1000, 640, 1280, 4),     # SDXL attn projection-ish, ragged M
    (333, 2048, 640, 4),      # SDXL cross-attn to_k (context 2048 -> 640)
    (4096, 640, 5120, 4),     # GEGLU proj
    (777, 2560, 640, 4),      # ff.net.2
    (520, 3072, 8192, 16),    # AuraFlow MLP, rank 16
    (129, 2304, 3840, 8),     # Lumina qkv, rank 8
    (64, 48, 40, 4),          # tiny / non-multiple-of-32 K
    (1, 64, 8, 2),
])
def test_lora_linear_parity(gpu, M, K, N, r):
    _case(gpu, M, K, N, r)


def test_lora_linear_batched_nobias(gpu):
    _case(gpu, 0, 1280, 1280, 4, bias=False, lead=(3, 77))


def test_lora_matches_reference_golden(gpu):
    """The reference LoRALinear's own numbers (tests/golden/ops.safetensors 'lora.*')."""
    from oracle import fixture_weights as fw
    from vision_ft_amd import kernels

    G = load_golden("ops.safetensors")[0]
    sd = fw.synth_state_dict({"lora_down.weight": (4, 48), "lora_up.weight": (40, 4), "alpha": (),
                              "linear.weight": (40, 48), "linear.bias": (40,)}, seed=3)
    lin = torch.nn.Linear(48, 40).to(gpu, torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(sd["linear.weight"])
        lin.bias.copy_(sd["linear.bias"])
    lin.requires_grad_(False)
    A = sd["lora_down.weight"].to(gpu, torch.bfloat16).requires_grad_(True)
    B = sd["lora_up.weight"].to(gpu, torch.bfloat16).requires_grad_(True)
    x = G["lora.x"].to(gpu, torch.bfloat16).requires_grad_(True)
    y = kernels.lora_linear(x, lin, A, B, 1.0 / 4)  # the fixture's synthetic state dict sets alpha = 1
    y.backward(G["lora.gy"].to(gpu, torch.bfloat16))
    # the reference rounds base output, down, up and the scaled branch to bf16 separately (autocast);
    # the fused kernel rounds once, so it sits closer to the fp32 truth than to these numbers
    assert rel_err(y.cpu(), G["lora.y"]) < 1e-2
    assert rel_err(x.grad.cpu(), G["lora.dx"]) < 1e-2
    assert rel_err(A.grad.cpu(), G["lora.d_down"]) < 2e-2
    assert rel_err(B.grad.cpu(), G["lora.d_up"]) < 2e-2


@pytest.mark.parametrize("tag", ["wide", "strided"])
def test_lora_conv2d_matches_reference_golden(gpu, tag):
    """LoRAConv2d on the GPU (reference src/modules/peft/lora.py:167-309) against the reference layer's own numbers and the
    oracle's fp32 truth.  `wide` (64 -> 96, 3x3 / stride 1 / padding 1) runs its frozen base convolution on the implicit-GEMM
    HIP kernel (forward and data gradient); `strided` (8 -> 6, stride 2) is a geometry that kernel does not take."""
    from oracle import fixture_weights as fw
    from oracle import ops
    from oracle.attention import fp32_truth
    from vision_ft_amd import kernels
    from vision_ft_amd.modules.peft.lora import LoRAConfig, LoRAConv2d

    G, _ = load_golden("lora_conv.safetensors")
    cin, cout, k, st, pd, rank = (int(v) for v in G[f"{tag}.geom"])
    sd = fw.synth_state_dict({"alpha": (), "conv2d.bias": (cout,), "conv2d.weight": (cout, cin, k, k),
                              "lora_down.weight": (rank, cin, k, k), "lora_up.weight": (cout, rank, 1, 1)}, seed=5)
    sd["alpha"] = G[f"{tag}.alpha"].clone()
    sd = {k_: v.to(torch.bfloat16) for k_, v in sd.items()}  # the GPU module holds everything in bf16 (autocast's weight cast)
    base = torch.nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pd)
    layer = LoRAConv2d(LoRAConfig(rank=rank, alpha=float(G[f"{tag}.alpha"])), base)
    layer.load_state_dict(sd)
    layer = layer.to(gpu, torch.bfloat16)
    x = G[f"{tag}.x"].to(gpu, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert kernels.conv3x3_eligible(x, layer.conv2d.weight, (st, st), (pd, pd), (1, 1), 1) == (tag == "wide")
    kernels.set_profiling(True)
    y = layer(x)
    y.backward(G[f"{tag}.gy"].to(gpu, torch.bfloat16))
    torch.cuda.synchronize()
    rep = kernels.profile_report()
    kernels.set_profiling(False)
    if tag == "wide":  # the base convolution really went through the HIP kernel, both directions
        assert "conv3x3_fwd" in rep and "conv3x3_dgrad" in rep, sorted(rep)
    # fp32 truth from the oracle on the same bf16-rounded operands
    P = {f"c.{k_}": v.float() for k_, v in sd.items()}
    for k_ in ("c.lora_down.weight", "c.lora_up.weight"):
        P[k_] = P[k_].clone().requires_grad_(True)
    xt = G[f"{tag}.x"].to(torch.bfloat16).float().requires_grad_(True)
    with fp32_truth():
        yt = ops.conv2d(P, "c", xt, stride=st, padding=pd)
    yt.backward(G[f"{tag}.gy"].to(torch.bfloat16).float())
    for name, got, ref, truth in (("y", y, G[f"{tag}.y"], yt), ("dx", x.grad, G[f"{tag}.dx"], xt.grad),
                                  ("d_down", layer.lora_down.weight.grad, G[f"{tag}.d_down"], P["c.lora_down.weight"].grad),
                                  ("d_up", layer.lora_up.weight.grad, G[f"{tag}.d_up"], P["c.lora_up.weight"].grad)):
        e_gpu, e_ref = rel_err(got.float().cpu(), truth.detach()), rel_err(ref, truth.detach())
        assert e_gpu < max(1e-2, 2 * e_ref), (tag, name, e_gpu, e_ref)  # as close to the truth as the reference's own bf16 run
        assert rel_err(got.float().cpu(), ref) < 2e-2, (tag, name)
    layer.set_enabled(False)
    with torch.no_grad():
        assert rel_err(layer(x.detach()).float().cpu(), G[f"{tag}.y_disabled"]) < 1e-2


@pytest.mark.parametrize("M,K,N,r,rp,G", [(5000, 640, 1920, 4, 16, 3), (98304 // 8, 640, 5120, 4, 4, 1), (333, 1280, 1280, 8, 8, 1),
                                          (26160 // 4, 3072, 8192, 16, 16, 1), (777, 192, 320, 4, 4, 1)])
def test_lora_backward_reads_dy_once(gpu, M, K, N, r, rp, G):
    """Round 5: u = s dy B comes out of the weight-gradient pass over dy (vft_lora_bwd_dy) and d(lora_down) out of a pass over x
    (vft_lora_bwd_x) — against the two-kernel form of rounds 2-4 (vft_lora_down(dy) + vft_lora_wgrad2g) on the same operands:
    u equal up to its bf16 rounding, every adapter gradient equal up to the order of the fp32 partial sums; fp32 reference for
    both; twice, bit-identical (ordered sums, no atomics).  G = 3: the block-diagonal form of a fused q|k|v."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(M + r)
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    s = 0.5
    x, dy = bf(M, K).to(gpu), bf(M, N).to(gpu)
    As = [bf(r, K, sc=K**-0.5).to(gpu) for _ in range(G)]
    Bs = [bf(N // G, r, sc=0.3).to(gpu) for _ in range(G)]
    A_cat = torch.zeros(rp, K, dtype=torch.bfloat16, device=gpu)
    Bt = torch.zeros(rp, N, dtype=torch.bfloat16, device=gpu)
    for i in range(G):
        A_cat[i * r:(i + 1) * r] = As[i]
        Bt[i * r:(i + 1) * r, i * (N // G):(i + 1) * (N // G)] = Bs[i].t()
    t = kernels.lora_down(x, A_cat, s)
    params = list(zip(As, Bs))

    def run(one_read: bool):
        old = kernels.LORA_BWD_ONE_READ
        kernels.LORA_BWD_ONE_READ = one_read
        try:
            u, grads = kernels._lora_bwd_rank(x, dy, t, Bt, params, r, rp, s)
            kernels.flush_deferred_wgrads()
        finally:
            kernels.LORA_BWD_ONE_READ = old
        torch.cuda.synchronize()
        return u, grads

    u1, g1 = run(True)
    u0, g0 = run(False)
    u1b, g1b = run(True)
    u_ref = s * (dy.float() @ Bt.float().t())
    assert rel_err(u1.float(), u_ref) < 6e-3 and rel_err(u0.float(), u_ref) < 6e-3
    assert torch.equal(u1, u1b)
    for i in range(G):
        dA_ref = (u_ref[:, i * r:(i + 1) * r].t() @ x.float())
        dB_ref = dy.float()[:, i * (N // G):(i + 1) * (N // G)].t() @ t.float()[:, i * r:(i + 1) * r]
        for got, old_, again, ref in ((g1[2 * i], g0[2 * i], g1b[2 * i], dA_ref), (g1[2 * i + 1], g0[2 * i + 1], g1b[2 * i + 1], dB_ref)):
            assert rel_err(got.float(), ref) < 1.2e-2, rel_err(got.float(), ref)
            assert rel_err(got.float(), old_.float()) < 8e-3
            assert torch.equal(got, again)


def test_lora_wgrad_deterministic(gpu):
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(1)
    x = torch.randn(5000, 640, generator=g).to(torch.bfloat16).to(gpu)
    lin = torch.nn.Linear(640, 640, bias=False).to(gpu, torch.bfloat16).requires_grad_(False)
    A = torch.randn(4, 640, generator=g).to(torch.bfloat16).to(gpu).requires_grad_(True)
    B = torch.randn(640, 4, generator=g).to(torch.bfloat16).to(gpu).requires_grad_(True)
    outs = []
    for _ in range(2):
        A.grad = B.grad = None
        kernels.lora_linear(x, lin, A, B, 0.25).float().square().mean().backward()
        outs.append((A.grad.clone(), B.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("M,K,N,r", [(1000, 640, 640, 4), (333, 2560, 640, 4), (260, 3072, 3072, 16)])
def test_lora_linear_fused_residual_and_norm_skip(gpu, M, K, N, r):
    """`x + lora_linear(norm(x))`-style block with the two skip fusions — the forward add inside the LoRA pass
    (residual=...) and the backward add inside the norm-backward kernel (layer_norm_skip) — equals the plain
    composition in fp32."""
    import torch.nn.functional as F

    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(3)
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    x, W, b = bf(M, K), bf(N, K, sc=K**-0.5), bf(N, sc=0.1)
    A, B = bf(r, K, sc=K**-0.5), bf(N, r, sc=0.3)
    lw, lb = bf(K, sc=0.2) + 1, bf(K, sc=0.1)
    W2 = bf(K, N, sc=N**-0.5)  # maps the block output back to K so that the skip has the input's shape
    gy = bf(M, K)
    s = 0.25

    # fp32 reference: x1 = x + lora_linear2(lora_linear(LN(x)))  (second layer plain, carries the residual)
    xr = x.float().requires_grad_(True)
    Ar, Br = A.float().requires_grad_(True), B.float().requires_grad_(True)
    n = F.layer_norm(xr, (K,), lw.float(), lb.float(), 1e-5)
    h = n @ W.float().t() + b.float() + s * (n @ Ar.t()) @ Br.t()
    yr = xr + h @ W2.float().t()
    yr.backward(gy.float())

    lin = torch.nn.Linear(K, N).to(gpu, torch.bfloat16).requires_grad_(False)
    lin2 = torch.nn.Linear(N, K, bias=False).to(gpu, torch.bfloat16).requires_grad_(False)
    with torch.no_grad():
        lin.weight.copy_(W), lin.bias.copy_(b), lin2.weight.copy_(W2)
    xg = x.to(gpu).requires_grad_(True)
    Ag, Bg = A.to(gpu).requires_grad_(True), B.to(gpu).requires_grad_(True)
    A2 = torch.zeros(r, N, device=gpu, dtype=torch.bfloat16).requires_grad_(True)   # zero adapter on the out layer:
    B2 = torch.zeros(K, r, device=gpu, dtype=torch.bfloat16).requires_grad_(True)   # exercises residual= only
    ng, xs = kernels.layer_norm_skip(xg, lw.to(gpu), lb.to(gpu), 1e-5)
    hg = kernels.lora_linear(ng, lin, Ag, Bg, s)
    yg = kernels.lora_linear(hg, lin2, A2, B2, s, residual=xs)
    yg.backward(gy.to(gpu))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu(), yr) < 8e-3, rel_err(yg.cpu(), yr)
    assert rel_err(xg.grad.cpu(), xr.grad) < 8e-3, rel_err(xg.grad.cpu(), xr.grad)
    assert rel_err(Ag.grad.cpu(), Ar.grad) < 1.5e-2 and rel_err(Bg.grad.cpu(), Br.grad) < 1.5e-2


@pytest.mark.parametrize("M,C,r", [(1000, 640, 4), (333, 1280, 8), (77, 64, 4)])
def test_lora_linear_geglu_fused(gpu, M, C, r):
    """geglu(LoRALinear(x)) through kernels.lora_linear_geglu vs the fp32 composition: output, input gradient and both
    adapter gradients (both LoRA compositions, see the `lora_path` fixture)."""
    import torch.nn.functional as F

    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(5)
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    Fh = 4 * C
    x, W, b = bf(M, C), bf(2 * Fh, C, sc=C**-0.5), bf(2 * Fh, sc=0.1)
    A, B = bf(r, C, sc=C**-0.5), bf(2 * Fh, r, sc=0.3)
    gy = bf(M, Fh)
    s = 0.25
    xr, Ar, Br = x.float().requires_grad_(True), A.float().requires_grad_(True), B.float().requires_grad_(True)
    proj = xr @ W.float().t() + b.float() + s * (xr @ Ar.t()) @ Br.t()
    a, gate = proj.chunk(2, dim=-1)
    yr = a * F.gelu(gate)
    yr.backward(gy.float())

    lin = torch.nn.Linear(C, 2 * Fh).to(gpu, torch.bfloat16).requires_grad_(False)
    with torch.no_grad():
        lin.weight.copy_(W), lin.bias.copy_(b)
    xg, Ag, Bg = x.to(gpu).requires_grad_(True), A.to(gpu).requires_grad_(True), B.to(gpu).requires_grad_(True)
    assert kernels.lora_geglu_eligible(xg, lin, Ag)
    yg = kernels.lora_linear_geglu(xg, lin, Ag, Bg, s)
    yg.backward(gy.to(gpu))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu(), yr) < 8e-3, rel_err(yg.cpu(), yr)
    assert rel_err(xg.grad.cpu(), xr.grad) < 8e-3, rel_err(xg.grad.cpu(), xr.grad)
    assert rel_err(Ag.grad.cpu(), Ar.grad) < 1.5e-2, rel_err(Ag.grad.cpu(), Ar.grad)
    assert rel_err(Bg.grad.cpu(), Br.grad) < 1.5e-2, rel_err(Bg.grad.cpu(), Br.grad)


def test_lora_qkv_fused_projection(gpu):
    """Three LoRA-wrapped bias-free projections of one input through ONE base GEMM per direction (kernels.lora_qkv)
    followed by self-attention: outputs and all gradients equal the three separate lora_linear calls."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(21)
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    B, S, C, H, r, s = 2, 300, 640, 10, 4, 0.25
    x = bf(B, S, C)
    Ws = [bf(C, C, sc=C**-0.5) for _ in range(3)]
    As = [bf(r, C, sc=C**-0.5) for _ in range(3)]
    Bs = [bf(C, r, sc=0.3) for _ in range(3)]
    go = bf(B, H, S, C // H)

    def run(fused: bool):
        xg = x.to(gpu).requires_grad_(True)
        Ag = [a.to(gpu).requires_grad_(True) for a in As]
        Bg = [b.to(gpu).requires_grad_(True) for b in Bs]
        lins = []
        for W in Ws:
            lin = torch.nn.Linear(C, C, bias=False).to(gpu, torch.bfloat16).requires_grad_(False)
            with torch.no_grad():
                lin.weight.copy_(W)
            lins.append(lin)
        if fused:
            wqkv = torch.cat([l.weight for l in lins], 0).contiguous()
            q, k, v = kernels.lora_qkv(xg, wqkv, list(zip(Ag, Bg)), s, H)
        else:
            q, k, v = (kernels.lora_linear(xg, l, a, b, s).view(B, S, H, C // H).permute(0, 2, 1, 3)
                       for l, a, b in zip(lins, Ag, Bg))
        o = kernels.flash_attention(q, k, v)
        o.backward(go.to(gpu))
        return o, xg.grad, [a.grad for a in Ag], [b.grad for b in Bg]

    o1, dx1, dA1, dB1 = run(True)
    o0, dx0, dA0, dB0 = run(False)
    assert rel_err(o1, o0.float()) < 2e-3
    assert rel_err(dx1, dx0.float()) < 1e-2, rel_err(dx1, dx0.float())
    for a1, a0 in zip(dA1 + dB1, dA0 + dB0):
        assert rel_err(a1, a0.float()) < 1e-2, rel_err(a1, a0.float())


@pytest.mark.parametrize("M,C,r", [(1000, 640, 4), (333, 1280, 8)])
def test_lora_geglu_down_fused_feedforward(gpu, M, C, r):
    """x + down(geglu(up(x))) as the SDXL FeedForward calls it (kernels.lora_geglu_down) equals the step-by-step
    composition (lora_linear -> geglu -> lora_linear with residual): output and every gradient."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(31)
    bf = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    Fh, s = 4 * C, 0.25
    x, res = bf(M, C), bf(M, C)
    up = torch.nn.Linear(C, 2 * Fh).to(gpu, torch.bfloat16).requires_grad_(False)
    dn = torch.nn.Linear(Fh, C).to(gpu, torch.bfloat16).requires_grad_(False)
    A1, B1 = bf(r, C, sc=C**-0.5), bf(2 * Fh, r, sc=0.3)
    A2, B2 = bf(r, Fh, sc=Fh**-0.5), bf(C, r, sc=0.3)
    gy = bf(M, C)

    def run(fused: bool):
        xs = x.to(gpu).requires_grad_(True)
        rs = res.to(gpu).requires_grad_(True)
        ps = [t.to(gpu).requires_grad_(True) for t in (A1, B1, A2, B2)]
        if fused:
            y = kernels.lora_geglu_down(xs, up, ps[0], ps[1], s, dn, ps[2], ps[3], s, residual=rs)
        else:
            y = kernels.lora_linear(kernels.geglu(kernels.lora_linear(xs, up, ps[0], ps[1], s)), dn, ps[2], ps[3], s, residual=rs)
        y.backward(gy.to(gpu))
        return [y, xs.grad, rs.grad] + [p.grad for p in ps]

    a, b = run(True), run(False)
    for i, (u, v) in enumerate(zip(a, b)):
        assert rel_err(u, v.float()) < 1.2e-2, (i, rel_err(u, v.float()))


def test_adapter_pack_forms_and_in_place_gradients(gpu, lora_path):
    """One `vft_lora_pack` launch builds the operand forms of every adapter (transposes; the concatenated /
    block-diagonal / zero-padded forms of a fused q|k|v) exactly; with gradient tensors in place the weight-gradient
    kernel adds into them (what AccumulateGrad does with a returned gradient) and autograd gets None."""
    from vision_ft_amd import kernels

    g = torch.Generator(device=gpu).manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g, device=gpu).to(torch.bfloat16)  # noqa: E731
    pack = kernels.AdapterPack(gpu)
    single = [(rn(4, 640).requires_grad_(), rn(1280, 4).requires_grad_())]
    wide = [(rn(6, 320).requires_grad_(), rn(640, 6).requires_grad_())]  # rank 6 -> padded to 8
    qkv = [(rn(4, 640).requires_grad_(), rn(640, 4).requires_grad_()) for _ in range(3)]
    e1, e2, e3 = pack.add(single, 4), pack.add(wide, 8), pack.add(qkv, 16)
    pack.finalize()
    pack.refresh()
    A, B, At, Bt = kernels._pack_forms(e1, single)
    assert A is single[0][0] and B is single[0][1]
    assert torch.equal(At, single[0][0].t()) and torch.equal(Bt, single[0][1].t())
    A, B, At, Bt = kernels._pack_forms(e2, wide)
    pad_a = torch.cat([wide[0][0], wide[0][0].new_zeros(2, 320)], 0)
    pad_b = torch.cat([wide[0][1], wide[0][1].new_zeros(640, 2)], 1)
    assert torch.equal(A, pad_a) and torch.equal(B, pad_b) and torch.equal(At, pad_a.t()) and torch.equal(Bt, pad_b.t())
    A, B, At, Bt = kernels._pack_forms(e3, qkv)
    cat = torch.cat([a for a, _ in qkv] + [qkv[0][0].new_zeros(4, 640)], 0)
    blk = torch.cat([torch.block_diag(*[b for _, b in qkv]), qkv[0][1].new_zeros(1920, 4)], 1)
    assert torch.equal(A, cat) and torch.equal(B, blk) and torch.equal(At, cat.t()) and torch.equal(Bt, blk.t())
    # a parameter update invalidates the entry until the next refresh
    with torch.no_grad():
        single[0][0].add_(1.0)
    assert kernels._pack_forms(e1, single) is None
    pack.refresh()
    assert torch.equal(kernels._pack_forms(e1, single)[2], single[0][0].t())

    # in-place weight gradients == returned gradients added to what the buffers held
    x = rn(2, 512, 640)
    W = rn(1280, 640)
    dy = rn(2, 512, 1280)
    Aq, Bq = single[0]
    base = torch.nn.Linear(640, 1280, bias=False, device=gpu, dtype=torch.bfloat16).requires_grad_(False)
    with torch.no_grad():
        base.weight.copy_(W)
    before = dict(kernels.PACK_STATS)
    y = kernels.lora_linear(x, base, Aq, Bq, 0.25, pack=e1)
    dA_ret, dB_ret = torch.autograd.grad(y, (Aq, Bq), dy, retain_graph=True)  # no .grad tensors yet: returned
    Aq.grad, Bq.grad = rn(4, 640) * 0.1, rn(1280, 4) * 0.1
    held_a, held_b = Aq.grad.clone(), Bq.grad.clone()
    y.backward(dy)
    after = kernels.PACK_STATS
    if lora_path == "gemm_fused":  # (the round-1 composition neither uses the pack nor writes gradients in place)
        assert after["pack_hit"] == before["pack_hit"] + 1 and after["grad_in_place"] == before["grad_in_place"] + 1
    assert rel_err(Aq.grad.float(), held_a.float() + dA_ret.float()) < 4e-3
    assert rel_err(Bq.grad.float(), held_b.float() + dB_ret.float()) < 4e-3
    # fused q|k|v: three adapters' gradients from one grouped launch, in place, against the returned form
    wqkv = rn(1920, 640)
    xq = rn(2, 256, 640)
    outs = kernels.lora_qkv(xq, wqkv, qkv, 0.25, 10, pack=e3)
    douts = [rn(*o.shape) for o in outs]
    flat = [p_ for pair in qkv for p_ in pair]
    ret = torch.autograd.grad(outs, flat, douts, retain_graph=True)
    for p_ in flat:
        p_.grad = torch.zeros_like(p_)
    torch.autograd.backward(outs, douts)
    for p_, r_ in zip(flat, ret):
        assert rel_err(p_.grad.float(), r_.float()) < 4e-3


def test_sibling_projections_chain_their_data_gradients(gpu, lora_path):
    """kernels.lora_siblings: q / k / v style LoRA projections of one input as one autograd node (data gradients chained
    through the GEMM's residual input) against the three layers applied separately."""
    from vision_ft_amd import kernels
    from vision_ft_amd.modules.peft import LoRAConfig, LoRALinear

    torch.manual_seed(4)
    layers = []
    for _ in range(3):
        base = torch.nn.Linear(1024, 1024, bias=False, device=gpu, dtype=torch.bfloat16)
        m = LoRALinear(LoRAConfig(rank=16, alpha=4.0), base).to(gpu)
        with torch.no_grad():
            m.lora_up.weight.normal_(0, 0.05)
        layers.append(m)
    x = torch.randn(2, 700, 1024, device=gpu).to(torch.bfloat16).requires_grad_()
    dys = [torch.randn(2, 700, 1024, device=gpu).to(torch.bfloat16) for _ in range(3)]
    outs = kernels.lora_siblings(x, layers)
    if lora_path != "gemm_fused":
        assert outs is None
        return
    torch.autograd.backward(outs, dys)
    got = [x.grad.clone()] + [p.grad.clone() for m in layers for p in (m.lora_down.weight, m.lora_up.weight)]
    x.grad = None
    for m in layers:
        m.lora_down.weight.grad = m.lora_up.weight.grad = None
    ref_outs = [m(x) for m in layers]
    torch.autograd.backward(ref_outs, dys)
    ref = [x.grad] + [p.grad for m in layers for p in (m.lora_down.weight, m.lora_up.weight)]
    for o, r_ in zip(outs, ref_outs):
        assert torch.equal(o, r_)
    assert rel_err(got[0].float(), ref[0].float()) < 6e-3
    for g_, r_ in zip(got[1:], ref[1:]):
        assert rel_err(g_.float(), r_.float()) < 1e-3


def test_grouped_kv_projection_matches_separate_layers(gpu, lora_path):
    """kernels.lora_group (cross-attention to_k | to_v of the context as one GEMM with a rank-2r extension) against the two
    LoRA layers applied one by one: outputs, the input gradient and the four adapter gradients."""
    from vision_ft_amd import kernels
    from vision_ft_amd.modules.peft import LoRAConfig, LoRALinear

    torch.manual_seed(6)
    layers = []
    for _ in range(2):
        base = torch.nn.Linear(2048, 640, bias=False, device=gpu, dtype=torch.bfloat16)
        m = LoRALinear(LoRAConfig(rank=4, alpha=2.0), base).to(gpu)
        with torch.no_grad():
            m.lora_up.weight.normal_(0, 0.05)
        layers.append(m)
    ctx_ = torch.randn(3, 77, 2048, device=gpu).to(torch.bfloat16).requires_grad_()
    dys = [torch.randn(3, 77, 640, device=gpu).to(torch.bfloat16) for _ in range(2)]
    wkv = torch.cat([m.linear.weight.detach() for m in layers], 0).contiguous()
    outs = kernels.lora_group(ctx_, wkv, [(m.lora_down.weight, m.lora_up.weight) for m in layers], layers[0]._scale())
    if lora_path != "gemm_fused":
        assert outs is None
        return
    torch.autograd.backward(outs, dys)
    got = [ctx_.grad.clone()] + [p.grad.clone() for m in layers for p in (m.lora_down.weight, m.lora_up.weight)]
    ctx_.grad = None
    for m in layers:
        m.lora_down.weight.grad = m.lora_up.weight.grad = None
    ref_outs = [m(ctx_) for m in layers]
    torch.autograd.backward(ref_outs, dys)
    ref = [ctx_.grad] + [p.grad for m in layers for p in (m.lora_down.weight, m.lora_up.weight)]
    for o, r_ in zip(outs, ref_outs):
        assert rel_err(o.float(), r_.float()) < 2e-3
    assert rel_err(got[0].float(), ref[0].float()) < 6e-3
    for g_, r_ in zip(got[1:], ref[1:]):
        assert rel_err(g_.float(), r_.float()) < 2e-3


def test_adapter_applied_twice_in_one_backward_keeps_both_contributions(gpu, lora_path):
    """One LoRA layer called twice before a single backward() (a module reused in a forward pass, or two forward passes
    summed into one loss — the reference Lumina2 low-res + high-res loss, train/lumina2/text_to_image.py:207-216): with
    gradient tensors in place both calls queue a deferred reduction into the SAME `.grad`; the queue must not let the two
    table entries race.  Checked against the two gradients computed one call at a time."""
    from vision_ft_amd import kernels

    g = torch.Generator(device=gpu).manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g, device=gpu).to(torch.bfloat16)  # noqa: E731
    A, B = (rn(4, 640) * 0.04).requires_grad_(), (rn(1280, 4) * 0.3).requires_grad_()
    base = torch.nn.Linear(640, 1280, bias=False, device=gpu, dtype=torch.bfloat16).requires_grad_(False)
    x1, x2 = rn(2048, 640), rn(1536, 640)
    dy1, dy2 = rn(2048, 1280), rn(1536, 1280)
    singles = []
    for x, dy in ((x1, dy1), (x2, dy2)):
        y = kernels.lora_linear(x, base, A, B, 0.25)
        singles.append(torch.autograd.grad(y, (A, B), dy))
    want_a = singles[0][0].float() + singles[1][0].float()
    want_b = singles[0][1].float() + singles[1][1].float()
    for trial in range(3):  # a lost update is a race: give it a few chances
        A.grad, B.grad = torch.zeros_like(A), torch.zeros_like(B)
        ya = kernels.lora_linear(x1, base, A, B, 0.25)
        yb = kernels.lora_linear(x2, base, A, B, 0.25)
        torch.autograd.backward((ya, yb), (dy1, dy2))
        torch.cuda.synchronize()
        assert rel_err(A.grad.float(), want_a) < 6e-3, (trial, rel_err(A.grad.float(), want_a))
        assert rel_err(B.grad.float(), want_b) < 6e-3, (trial, rel_err(B.grad.float(), want_b))


def test_grouped_entry_points_refuse_fp32_adapters(gpu, lora_path):
    """`peft.config.dtype: float32` is schema-valid; the grouped kernels read raw bf16 pointers, so they must fall back
    (lora_siblings / lora_group -> None) or raise (lora_qkv) instead of computing garbage."""
    from types import SimpleNamespace

    from vision_ft_amd import kernels

    rn = lambda *s, dt=torch.bfloat16: torch.randn(*s, device=gpu).to(dt)  # noqa: E731
    x = rn(2, 512, 640)
    wcat = rn(1280, 640)
    ad32 = [(rn(4, 640, dt=torch.float32), rn(640, 4, dt=torch.float32)) for _ in range(2)]
    assert kernels.lora_group(x, wcat, ad32, 0.25) is None
    wqkv = rn(1920, 640)
    with pytest.raises(TypeError):
        kernels.lora_qkv(x, wqkv, ad32 + ad32[:1], 0.25, 10)

    def layer():
        lin = torch.nn.Linear(640, 640, bias=False, device=gpu, dtype=torch.bfloat16).requires_grad_(False)
        return SimpleNamespace(linear=lin, enabled=True, rank=4, training_dropout_active=lambda: False, _scale=lambda: 0.25,
                               lora_down=SimpleNamespace(weight=rn(4, 640, dt=torch.float32)),
                               lora_up=SimpleNamespace(weight=rn(640, 4, dt=torch.float32)))
    assert kernels.lora_siblings(x, (layer(), layer())) is None
