"""GPU parity of the projection GEMM, the convolutions and the NF4-fused linear (C ABI: vft_linear_fwd, vft_conv3x3_*,
vft_linear_nf4, vft_nf4_dequant) vs fp32 PyTorch / the numpy NF4 oracle.  NF4 parity vs bitsandbytes itself is
unpinned (oracle/nf4.py header); integer work (codes, packing, key names) is checked exactly."""
import json

import numpy as np
import pytest
import torch

from oracle import nf4 as onf4

from .helpers import rel_err

pytestmark = pytest.mark.gpu


def _bf(g, *s, sc=1.0):
    return (torch.randn(*s, generator=g) * sc).to(torch.bfloat16)


def test_nf4_odd_reduction_dimension_falls_back_to_a_dense_copy(gpu):
    """A reduction dimension that is no multiple of 64 (the data gradient of a layer with out_features % 64 != 0: none in the
    three models) is not taken by the dequantising GEMM: the wrapper dequantises (vft_nf4_dequant) and multiplies densely."""
    from vision_ft_amd import kernels

    layer, wd, _, _ = _make_layer(gpu, 96, 128, True, bias=False)
    q = layer.quant_state()
    g = torch.Generator().manual_seed(6)
    dy = _bf(g, 50, 96).to(gpu)
    dx = kernels._nf4_gemm(dy, q, None, True)  # K = out_features = 96
    wb = wd.to(torch.bfloat16).float()
    assert rel_err(dx.cpu(), dy.cpu().float() @ wb) < 5e-3
    # ... and keeps the rank extension and the residual of the fused call
    ex, ew, res = _bf(g, 50, 4).to(gpu), _bf(g, 128, 4).to(gpu), _bf(g, 50, 128).to(gpu)
    dx2 = kernels._nf4_gemm(dy, q, None, True, ext_x=ex, ext_w=ew, residual=res)
    want = dy.cpu().float() @ wb + ex.cpu().float() @ ew.cpu().float().t() + res.cpu().float()
    assert rel_err(dx2.cpu(), want) < 5e-3


@pytest.mark.parametrize("double_quant", [False, True])
def test_nf4_quantize_matches_oracle_exactly(double_quant):
    from vision_ft_amd.modules.quant import quantize_nf4

    g = torch.Generator().manual_seed(2)
    w = torch.randn(192, 256, generator=g) * 0.05
    got = quantize_nf4(w, double_quant=double_quant)
    want = onf4.quantize_nf4(w.numpy(), 64, double_quant)
    assert got[""].dtype == torch.uint8 and got[""].shape == (192 * 256 // 2, 1)
    assert np.array_equal(got[""].numpy(), want[""])  # codes + packing: bit-exact
    assert np.array_equal(got[".quant_map"].numpy(), want[".quant_map"])
    if double_quant:
        assert np.array_equal(got[".absmax"].numpy(), want[".absmax"])
        np.testing.assert_allclose(got[".nested_absmax"].numpy(), want[".nested_absmax"], rtol=1e-6)
        np.testing.assert_allclose(got[".nested_quant_map"].numpy(), want[".nested_quant_map"], rtol=1e-6)
    else:
        np.testing.assert_array_equal(got[".absmax"].numpy(), want[".absmax"])
    mg = json.loads(bytes(got[".quant_state.bitsandbytes__nf4"].tolist()).decode())
    mw = json.loads(bytes(want[".quant_state.bitsandbytes__nf4"].tolist()).decode())
    assert mg["quant_type"] == mw["quant_type"] == "nf4" and mg["shape"] == mw["shape"] and mg["blocksize"] == 64



def _make_layer(gpu, out_f, in_f, double_quant, bias=True, seed=3):
    from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4

    g = torch.Generator().manual_seed(seed)
    w = torch.randn(out_f, in_f, generator=g) * in_f**-0.5
    q = quantize_nf4(w, double_quant=double_quant)
    sd = {"weight" + k: v for k, v in q.items()}
    b = None
    if bias:
        b = _bf(g, out_f, sc=0.1)
        sd["bias"] = b
    layer = NF4Linear(in_f, out_f, bias=bias)
    layer.load_state_dict(sd, strict=True)
    layer = layer.to(gpu)
    wd = torch.from_numpy(onf4.dequantize_nf4({k: v.numpy() for k, v in q.items()}))  # fp32 oracle dequant
    return layer, wd, b, sd


@pytest.mark.parametrize("double_quant", [False, True])
def test_nf4_dequant_kernel_matches_oracle(gpu, double_quant):
    layer, wd, _, _ = _make_layer(gpu, 384, 512, double_quant)
    got = layer.dequantize().cpu()
    assert got.shape == wd.shape
    want = wd.to(torch.bfloat16)
    mism = (got != want).float().mean().item()
    assert mism < 1e-3, mism  # bf16 rounding of identical fp32 products (fma contraction may flip a last bit)
    assert rel_err(got, wd) < 4e-3


@pytest.fixture(params=["dequant_once", "dequant_in_loop"])
def nf4_gemm_form(request):
    """Both forms of vft_linear_nf4: the weight dequantised once per launch into the workspace (round 4, default) and
    inside the K-loop by every row tile (round 3; what a call without a workspace runs)."""
    from vision_ft_amd import kernels

    old = kernels.NF4_DEQUANT_ONCE
    kernels.NF4_DEQUANT_ONCE = request.param == "dequant_once"
    yield request.param
    kernels.NF4_DEQUANT_ONCE = old
    assert not kernels.nf4_barrier_timed_out()


@pytest.mark.parametrize("M,out_f,in_f,dq", [(520, 3840, 2304, True), (333, 2304, 9216, True), (77, 256, 128, False),
                                             (1024, 9216, 2304, False), (300, 192, 320, True)])
def test_nf4_linear_forward_backward(gpu, nf4_gemm_form, M, out_f, in_f, dq):
    layer, wd, b, _ = _make_layer(gpu, out_f, in_f, dq)
    g = torch.Generator().manual_seed(4)
    x, gy = _bf(g, 2, M // 2 if M % 2 == 0 else M, in_f), None
    x = x.reshape(-1, in_f)[:M].contiguous()
    gy = _bf(g, M, out_f)
    xg = x.to(gpu).requires_grad_(True)
    y = layer(xg)
    y.backward(gy.to(gpu))
    wb = wd.to(torch.bfloat16).float()  # the kernel multiplies bf16-rounded dequantised weights
    ref = x.float() @ wb.t() + b.float()
    assert rel_err(y.cpu(), ref) < 5e-3, rel_err(y.cpu(), ref)
    assert rel_err(xg.grad.cpu(), gy.float() @ wb) < 5e-3, rel_err(xg.grad.cpu(), gy.float() @ wb)


@pytest.mark.parametrize("M,out_f,in_f,r,dq,res", [(520, 3840, 2304, 4, True, False), (333, 2304, 2304, 8, True, True),
                                                   (1000, 256, 128, 16, False, True)])
def test_qlora_linear_one_launch_per_direction(gpu, nf4_gemm_form, M, out_f, in_f, r, dq, res):
    """NF4 base + LoRA (+ residual) through vft_lora_down + vft_linear_nf4 with the rank extension: output, input
    gradient and adapter gradients vs fp32 math on the dequantised weight."""
    from vision_ft_amd import kernels

    layer, wd, b, _ = _make_layer(gpu, out_f, in_f, dq)
    g = torch.Generator().manual_seed(9)
    x, gy = _bf(g, M, in_f), _bf(g, M, out_f)
    A, B = _bf(g, r, in_f, sc=in_f**-0.5), _bf(g, out_f, r, sc=0.3)
    rs = _bf(g, M, out_f) if res else None
    s = 0.25
    wb = wd.to(torch.bfloat16).float()
    xr, Ar, Br = x.float().requires_grad_(True), A.float().requires_grad_(True), B.float().requires_grad_(True)
    yr = xr @ wb.t() + b.float() + s * (xr @ Ar.t()) @ Br.t()
    if res:
        yr = yr + rs.float()
    yr.backward(gy.float())
    xg, Ag, Bg = x.to(gpu).requires_grad_(True), A.to(gpu).requires_grad_(True), B.to(gpu).requires_grad_(True)
    rg = rs.to(gpu).requires_grad_(True) if res else None
    y = kernels.lora_linear(xg, layer, Ag, Bg, s, residual=rg)
    y.backward(gy.to(gpu))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), yr) < 6e-3, rel_err(y.cpu(), yr)
    assert rel_err(xg.grad.cpu(), xr.grad) < 6e-3, rel_err(xg.grad.cpu(), xr.grad)
    assert rel_err(Ag.grad.cpu(), Ar.grad) < 1.2e-2 and rel_err(Bg.grad.cpu(), Br.grad) < 1.2e-2
    if res:
        assert torch.equal(rg.grad.cpu(), gy)


def test_nf4_dequant_once_equals_the_bf16_gemm_on_the_dequantised_weight_bitwise(gpu):
    """The dequantise-once launch writes exactly vft_nf4_dequant's values (forward orientation and the transposed one of the
    data gradient) and then runs the bf16 main loop: same tile, same order of sums => the SAME BITS as vft_linear_fwd on the
    dequantised weight.  Layers of different sizes alternate on the one shared workspace (its barrier words are at a fixed
    place and advance by the same amount per launch whatever the grid), every call is repeated, and no launch may have timed out at the barrier."""
    from vision_ft_amd import kernels

    assert kernels.NF4_DEQUANT_ONCE
    g = torch.Generator().manual_seed(11)
    layers = [_make_layer(gpu, o, i, dq, bias=False)[0] for o, i, dq in ((2304, 2304, True), (384, 1024, False), (6144, 2304, True), (192, 320, True))]

    def dense_of(q):
        # the weight the launch must build: bf16(code[c] * absmax) with the SAME fp32 absmax the kernel gets (a double-quantised
        # absmax is expanded once on the host side; vft_nf4_dequant expands it itself and may differ in a last bit)
        pk = q.packed.view(-1).long()
        idx = torch.stack([pk >> 4, pk & 15], dim=1).view(-1)
        return (q.code.float()[idx] * q.absmax_f32().repeat_interleave(64)).to(torch.bfloat16).view(q.out_features, q.in_features)

    dense = [dense_of(ly.quant_state()) for ly in layers]
    for ly, w in zip(layers, dense):
        assert (ly.dequantize() != w).float().mean().item() < 1e-3
    for rep in range(2):
        for ly, w in zip(layers, dense):
            q = ly.quant_state()
            for M in (1500, 333):
                x = _bf(g, M, q.in_features).to(gpu)
                dy = _bf(g, M, q.out_features).to(gpu)
                y = kernels._nf4_gemm(x, q, None, False)
                assert torch.equal(y, kernels.linear_fused(x, w)), (rep, w.shape, M)
                dx = kernels._nf4_gemm(dy, q, None, True)
                assert torch.equal(dx, kernels.linear_fused(dy, w.t().contiguous())), (rep, w.shape, M)
    torch.cuda.synchronize()
    assert not kernels.nf4_barrier_timed_out()
    hdr = kernels._nf4_workspace(torch.device(gpu), 1)[:1024].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    launches = int(hdr[0])  # the launch counter; every arrival word stands at 1024 per launch, whatever the grids were
    assert launches >= 2 * 4 * 2 * 2 and all(int(hdr[64 + 16 * c]) == (1024 * launches) % 2**32 for c in range(8)), hdr[:200:16]


def test_nf4_barrier_timeout_is_loud(gpu, monkeypatch):
    """VERDICT r4 weak #5 / ADVICE r4: a dequantising launch whose grid never meets (here: VFT_NF4_DEBUG=4 keeps workgroup 1
    from arriving, what a workgroup held off the chip by a concurrent kernel looks like to the others; 8 = short spin limit)
    must not compute on the incomplete weight image: every output element is NaN, the sticky error word is set,
    `nf4_check_barrier()` — what the Trainer calls at its logging / checkpoint cadence — raises; after a reset the same
    call gives the bits of a run that never failed."""
    from vision_ft_amd import kernels

    assert kernels.NF4_DEQUANT_ONCE
    layer = _make_layer(gpu, 1280, 640, True, bias=False)[0]
    q = layer.quant_state()
    x = torch.randn(2048, 640, device=gpu).to(torch.bfloat16)
    want = kernels._nf4_gemm(x, q, None, False)
    want_t = kernels._nf4_gemm(want, q, None, True)
    torch.cuda.synchronize()
    assert not kernels.nf4_barrier_timed_out()
    kernels.nf4_check_barrier()
    monkeypatch.setenv("VFT_NF4_DEBUG", "12")
    bad = kernels._nf4_gemm(x, q, None, False)
    bad_t = kernels._nf4_gemm(want, q, None, True)
    monkeypatch.delenv("VFT_NF4_DEBUG")
    torch.cuda.synchronize()
    assert torch.isnan(bad).all() and torch.isnan(bad_t).all()
    assert kernels.nf4_barrier_timed_out()
    with pytest.raises(RuntimeError, match="grid barrier"):
        kernels.nf4_check_barrier()
    kernels.nf4_reset_barrier()
    assert not kernels.nf4_barrier_timed_out()
    assert torch.equal(kernels._nf4_gemm(x, q, None, False), want) and torch.equal(kernels._nf4_gemm(want, q, None, True), want_t)
    torch.cuda.synchronize()
    assert not kernels.nf4_barrier_timed_out()


def test_nf4_scratch_is_ordered_across_streams(gpu):
    """one scratch (weight image + barrier words) per device — two NF4 GEMMs issued on different streams with no
    dependency between them must still not overlap: the wrapper orders the second stream behind the first (event), so each
    gives the bits of a serial run and no launch times out."""
    from vision_ft_amd import kernels

    la = _make_layer(gpu, 2304, 2304, True, bias=False)[0].quant_state()
    lb = _make_layer(gpu, 1280, 640, False, bias=False)[0].quant_state()
    xa = torch.randn(4096, 2304, device=gpu).to(torch.bfloat16)
    xb = torch.randn(3000, 640, device=gpu).to(torch.bfloat16)
    want_a, want_b = kernels._nf4_gemm(xa, la, None, False), kernels._nf4_gemm(xb, lb, None, False)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(6):
        with torch.cuda.stream(s1):
            ya = kernels._nf4_gemm(xa, la, None, False)
        with torch.cuda.stream(s2):
            yb = kernels._nf4_gemm(xb, lb, None, False)
        outs.append((ya, yb))
    torch.cuda.synchronize()
    for ya, yb in outs:
        assert torch.equal(ya, want_a) and torch.equal(yb, want_b)
    assert not kernels.nf4_barrier_timed_out()


def test_nf4_dequant_once_replays_from_a_graph(gpu):
    """Capturable: no host-side reset of the barrier words, static workspace address."""
    from vision_ft_amd import kernels

    layer = _make_layer(gpu, 1280, 640, True, bias=False)[0]
    q = layer.quant_state()
    x = torch.randn(2048, 640, device=gpu).to(torch.bfloat16)
    want = kernels._nf4_gemm(x, q, None, False)
    torch.cuda.synchronize()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        kernels._nf4_gemm(x, q, None, False)
    torch.cuda.current_stream().wait_stream(s_)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = kernels._nf4_gemm(x, q, None, False)
        y2 = kernels._nf4_gemm(y[:, :640].contiguous(), q, None, False)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want) and torch.equal(y2, kernels._nf4_gemm(want[:, :640].contiguous(), q, None, False))
    assert not kernels.nf4_barrier_timed_out()


def test_nf4_state_dict_layout_and_roundtrip(gpu):
    """Mirrors the reference's tests/test_modules_quant.py:153-193: key names, uint8 storage, reload
    self-consistency."""
    from vision_ft_amd.modules.quant import NF4Linear, get_quant_type_from_children_dict

    layer, _, _, sd_in = _make_layer(gpu, 256, 128, True)
    sd = layer.state_dict()
    assert sd["weight"].dtype == torch.uint8
    assert "weight.quant_state.bitsandbytes__nf4" in sd and "weight.absmax" in sd and "weight.nested_quant_map" in sd
    assert get_quant_type_from_children_dict({k[len("weight."):]: v for k, v in sd.items() if k.startswith("weight.")}) == "bnb_nf4"
    clone = NF4Linear(128, 256, bias=True)
    clone.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
    clone = clone.to(gpu)
    x = torch.randn(5, 128, device=gpu).to(torch.bfloat16)
    assert torch.equal(layer(x), clone(x))
    with pytest.raises(RuntimeError):
        NF4Linear(128, 256).load_state_dict({"weight": torch.zeros(256, 128), "bias": torch.zeros(256)})  # not pre-quantised


@pytest.fixture(params=["bf16_compute_cache", "in_gemm_dequant", "in_loop_dequant"])
def nf4_mode(request):
    """The compute modes of the NF4 layers: the dequantised bf16 compute copy (labelled option), the dequantising GEMM (default:
    once per launch) and its round-3 form (inside the K-loop)."""
    from vision_ft_amd import kernels
    from vision_ft_amd.modules.quant import nf4

    old, old_once = nf4.NF4_COMPUTE_CACHE, kernels.NF4_DEQUANT_ONCE
    nf4.NF4_COMPUTE_CACHE = request.param == "bf16_compute_cache"
    kernels.NF4_DEQUANT_ONCE = request.param != "in_loop_dequant"
    yield request.param
    nf4.NF4_COMPUTE_CACHE, kernels.NF4_DEQUANT_ONCE = old, old_once
    assert not kernels.nf4_barrier_timed_out()


def test_nf4_compute_cache_is_bit_identical_to_the_dequantising_gemm_input(gpu):
    """The cached copy is exactly what the dequantising kernel feeds its MFMAs (code x absmax rounded to bf16), and a layer
    gives the same forward through either mode up to the two GEMM kernels' accumulation order."""
    from vision_ft_amd.modules.quant import nf4

    layer, _, _, _ = _make_layer(gpu, 1280, 640, True)
    x = torch.randn(2048, 640, device=gpu).to(torch.bfloat16)
    old = nf4.NF4_COMPUTE_CACHE
    try:
        nf4.NF4_COMPUTE_CACHE = True
        assert torch.equal(layer.dense_weight(), layer.dequantize())
        y_cache = layer(x)
        nf4.NF4_COMPUTE_CACHE = False
        assert layer.dense_weight() is None
        y_gemm = layer(x)
    finally:
        nf4.NF4_COMPUTE_CACHE = old
    assert rel_err(y_cache.float(), y_gemm.float()) < 3e-3


def test_qlora_tiny_lumina_step(gpu, nf4_mode):
    """BASELINE config 4 in miniature: NF4 base (checkpoint sniffing -> NF4Linear) + bf16 LoRA on top;
    output close to the bf16 model's, LoRA gradients flow, base stays frozen and 4-bit."""
    from vision_ft_amd.models.lumina2 import Denoiser, DenoiserConfig
    from vision_ft_amd.modules.peft import LoRAConfig, PeftTargetConfig
    from vision_ft_amd.modules.quant import NF4Linear, quantize_state_dict, replace_by_prequantized_weights

    cfg = DenoiserConfig(in_channels=8, out_channels=8, hidden_dim=256, caption_dim=64, timestep_embed_dim=64, depth=2,
                         num_heads=4, num_kv_heads=2, refiner_depth=1, multiple_of=64, axes_dims=[16, 24, 24], axes_lens=[64, 32, 32])
    torch.manual_seed(0)
    ref_model = Denoiser(cfg)
    for blk in list(ref_model.layers) + list(ref_model.noise_refiner):  # zero-init AdaLN would hide everything
        torch.nn.init.normal_(blk.adaLN_modulation[1].weight, std=0.02)
    sd = {k: v.clone() for k, v in ref_model.state_dict().items()}
    qsd = quantize_state_dict({k: v.clone() for k, v in sd.items()}, "bnb_nf4", ["attention.qkv.weight", "attention.out.weight", "feed_forward"],
                              ["noise_refiner", "context_refiner"], double_quant=True)
    qmodel = Denoiser(cfg)
    replace_by_prequantized_weights(qmodel, qsd)
    n_q = sum(isinstance(m, NF4Linear) for m in qmodel.modules())
    assert n_q == 2 * 5  # qkv, out, w1, w2, w3 of the two main layers
    qmodel.load_state_dict(qsd, strict=True)
    ref_model, qmodel = ref_model.to(gpu, torch.bfloat16), qmodel.to(gpu, torch.bfloat16)
    qmodel.requires_grad_(False)
    PeftTargetConfig(include_keys=["attention", "feed_forward"], exclude_keys=["noise_refiner"],
                     config=LoRAConfig(rank=4)).replace_to_peft_layer(qmodel)
    g = torch.Generator(device=gpu).manual_seed(1)
    lat = torch.randn(2, 8, 16, 24, generator=g, device=gpu).to(torch.bfloat16)
    cap = torch.randn(2, 12, 64, generator=g, device=gpu).to(torch.bfloat16)
    mask = torch.arange(12, device=gpu)[None] < torch.tensor([12, 7], device=gpu)[:, None]
    t = torch.tensor([0.2, 0.7], device=gpu)
    qmodel.train()
    o_q = torch.nested.to_padded_tensor(qmodel(latents=lat, caption_features=cap, timestep=t, caption_mask=mask)[0], 0.0)
    with torch.no_grad():
        o_r = torch.nested.to_padded_tensor(ref_model(latents=lat, caption_features=cap, timestep=t, caption_mask=mask)[0], 0.0)
    assert rel_err(o_q, o_r) < 0.15  # 4-bit weights: close to, not equal to, the bf16 model
    o_q.float().square().mean().backward()
    grads = [p.grad for n, p in qmodel.named_parameters() if "lora_" in n]
    assert len(grads) == 2 * 15 and all(gr is not None and torch.isfinite(gr.float()).all() for gr in grads)
    assert any(gr.abs().sum() > 0 for gr in grads)
    assert all(p.grad is None for n, p in qmodel.named_parameters() if "lora_" not in n)


def test_qlora_lumina_matches_the_oracle_on_dequantised_weights(gpu, nf4_mode):
    """BASELINE config 4 against the ORACLE (the only model-level config-4 check compared the product with
    itself at 0.15): the target Linears are quantised by oracle/nf4.py (numpy restatement of the bitsandbytes layout) and
    loaded into the product through the checkpoint-sniffing path; oracle/lumina2.py runs on the oracle's own dequantised
    weights (+ the same LoRA adapters), the GPU model on the 4-bit weights — in both compute modes.  Output at the model
    tolerance (rel-L2 2e-2), loss and three AdamW LoRA steps' losses at rtol 5e-3: the tolerances of the bf16 model tests."""
    from oracle import fixture_weights as fw
    from oracle import lumina2 as olum
    from oracle import ops
    from vision_ft_amd.models.lumina2 import Denoiser, DenoiserConfig
    from vision_ft_amd.modules.loss.flow_match import loss_with_predicted_velocity
    from vision_ft_amd.modules.peft import LoRAConfig, PeftTargetConfig
    from vision_ft_amd.modules.quant import NF4Linear, replace_by_prequantized_weights

    from .helpers import autocast_cpu, load_golden, synth_params

    T, meta = load_golden("lumina_tiny.safetensors")
    cfgd = json.loads(meta["config"])
    with torch.device("meta"):
        shape_mod = Denoiser(DenoiserConfig(**cfgd))
    P = {k: v.to(torch.bfloat16).float() for k, v in synth_params(shape_mod, 2, meta["checksum"]).items()}
    targets = sorted(k[:-len(".weight")] for k, v in P.items() if k.endswith(".weight") and v.dim() == 2
                     and ("attention" in k or "feed_forward" in k) and "noise_refiner" not in k)
    assert len(targets) == 15  # qkv, out, w1, w2, w3 of the two main layers and the context refiner
    # ---- oracle side: quantise (numpy), dequantise (numpy), bf16-round as the kernel's MFMA operand is
    qsd = {k: v for k, v in P.items()}
    Pd = dict(P)
    for n in targets:
        q = onf4.quantize_nf4(P[n + ".weight"].numpy(), double_quant=True)
        for suffix, arr in q.items():
            qsd[n + ".weight" + suffix] = torch.from_numpy(np.ascontiguousarray(arr))
        Pd[n + ".weight"] = torch.from_numpy(onf4.dequantize_nf4(q)).to(torch.bfloat16).float()
    assert 0.02 < rel_err(Pd[targets[0] + ".weight"], P[targets[0] + ".weight"]) < 0.2  # really 4-bit
    lora_names = [n for n in targets if "context_refiner" not in n]  # the LoRA key set of configs/lumina2 + noise_refiner excluded
    shapes = {}
    for n in lora_names:
        w = Pd.pop(n + ".weight")
        Pd[n + ".linear.weight"] = w
        shapes[n + ".lora_down.weight"], shapes[n + ".lora_up.weight"] = (4, w.shape[1]), (w.shape[0], 4)
        Pd[n + ".alpha"] = torch.tensor(1.0)
    synth = {k: (v * (0.2 if "lora_up" in k else 1.0)).to(torch.bfloat16).float() for k, v in fw.synth_state_dict(shapes, seed=23).items()}
    cfg = olum.LuminaCfg(**{k: v for k, v in cfgd.items() if k in olum.LuminaCfg.__dataclass_fields__})
    lora_o = {k: v.clone().to(torch.bfloat16).requires_grad_(True) for k, v in synth.items()}  # adapters are bf16 parameters
    opt_o = torch.optim.AdamW(list(lora_o.values()), lr=1e-3)
    losses_o, out_o = [], None
    for _ in range(3):
        with autocast_cpu():
            outs, _, _ = olum.nextdit_forward({**Pd, **lora_o}, cfg, T["noisy"], T["caption"], T["timestep"], T["caption_mask"])
            o = -torch.stack(outs, 0)
            loss = ops.flow_loss(T["latents"], T["noise"], o)
        if out_o is None:
            out_o = o.detach()
        loss.backward()
        opt_o.step()
        opt_o.zero_grad()
        losses_o.append(loss.detach().float())
    # ---- product side: the oracle-quantised state dict goes through the checkpoint path
    model = Denoiser(DenoiserConfig(**cfgd))
    replace_by_prequantized_weights(model, qsd)
    assert sum(isinstance(m, NF4Linear) for m in model.modules()) == 15
    model.load_state_dict(qsd, strict=True)
    model = model.to(gpu, torch.bfloat16).requires_grad_(False)
    PeftTargetConfig(include_keys=["attention", "feed_forward"], exclude_keys=["noise_refiner", "context_refiner"],
                     config=LoRAConfig(rank=4, alpha=1.0)).replace_to_peft_layer(model)
    sd = model.state_dict()
    assert sorted(k for k in sd if ".lora_" in k) == sorted(synth)
    with torch.no_grad():
        for k, v in synth.items():
            sd[k].copy_(v)
    model.train()
    params = [p_ for p_ in model.parameters() if p_.requires_grad]
    assert len(params) == 2 * len(lora_names)
    opt = torch.optim.AdamW(params, lr=1e-3)
    g = {k: T[k].to(gpu) for k in ("timestep", "caption", "caption_mask", "noise", "latents", "noisy")}
    losses, out = [], None
    for _ in range(3):
        o, _, _ = model(latents=g["noisy"], timestep=g["timestep"], caption_features=g["caption"], caption_mask=g["caption_mask"])
        o = -torch.nested.to_padded_tensor(o, padding=0.0)
        loss = loss_with_predicted_velocity(g["latents"], g["noise"], o.float())
        if out is None:
            out = o.detach().cpu()
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.detach().float().cpu())
    assert rel_err(out, out_o) < 2e-2, (nf4_mode, rel_err(out, out_o))
    torch.testing.assert_close(torch.stack(losses), torch.stack(losses_o), rtol=5e-3, atol=0)
    assert float(losses[0]) != float(losses[2])  # the adapters did train


@pytest.mark.parametrize("M,N,K,r,bias,res", [(4096, 1280, 1280, 4, True, True), (1000, 640, 2048, 8, False, False),
                                              (333, 5120, 640, 16, True, False), (300, 648, 64, 32, True, True),
                                              (777, 1024, 96, 0, True, True), (1, 8, 32, 4, False, False),
                                              (8720, 3072, 3072, 16, True, False), (513, 2304, 320, 0, False, False)])
def test_linear_fused(gpu, M, N, K, r, bias, res):
    """vft_linear_fwd: base GEMM + bias + rank-r extension in the accumulator + residual on the rounded
    output, vs fp32 math (both tile configurations, ragged M / N edges, every supported rank)."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(7)
    x, w = _bf(g, M, K), _bf(g, N, K, sc=K**-0.5)
    b = _bf(g, N, sc=0.2) if bias else None
    rs = _bf(g, M, N) if res else None
    ex = _bf(g, M, r, sc=0.5) if r else None
    ew = _bf(g, N, r, sc=0.5) if r else None
    y = kernels.linear_fused(x.to(gpu), w.to(gpu), None if b is None else b.to(gpu), None if rs is None else rs.to(gpu),
                             None if ex is None else ex.to(gpu), None if ew is None else ew.to(gpu))
    ref = x.float() @ w.float().t()
    if b is not None:
        ref = ref + b.float()
    if r:
        ref = ref + ex.float() @ ew.float().t()
    if rs is not None:
        ref = ref.to(torch.bfloat16).float() + rs.float()
    assert y.shape == (M, N)
    assert rel_err(y.cpu(), ref) < 5e-3, rel_err(y.cpu(), ref)


def test_linear_fused_rejects_bad_arguments(gpu):
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(8)
    x, w = _bf(g, 64, 48).to(gpu), _bf(g, 64, 48).to(gpu)  # K % 32 != 0
    with pytest.raises(ValueError):
        kernels.linear_fused(x, w)
    x, w = _bf(g, 64, 64).to(gpu), _bf(g, 64, 64).to(gpu)
    with pytest.raises(ValueError):
        kernels.linear_fused(x, w, ext_x=_bf(g, 64, 5).to(gpu), ext_w=_bf(g, 64, 5).to(gpu))  # unsupported rank


@pytest.mark.parametrize("B,Cin,Cout,H,W,bias", [(2, 320, 320, 32, 32, True), (1, 640, 320, 17, 23, True),
                                                 (3, 64, 1280, 8, 8, False), (1, 1280, 1280, 16, 16, True),
                                                 (2, 960, 320, 9, 40, True), (1, 32, 32, 1, 1, True),
                                                 (1, 2560, 1280, 6, 6, True)])
def test_conv3x3_implicit_gemm(gpu, B, Cin, Cout, H, W, bias):
    """vft_conv3x3_fwd (forward and data gradient through the same kernel) vs fp32 F.conv2d; weight / bias
    gradients (MIOpen path of the full fine-tune) are checked too."""
    import torch.nn.functional as F

    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(11)
    x = _bf(g, B, Cin, H, W)
    w = _bf(g, Cout, Cin, 3, 3, sc=(9 * Cin) ** -0.5)
    b = _bf(g, Cout, sc=0.2) if bias else None
    gy = _bf(g, B, Cout, H, W)
    xg = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.to(gpu).requires_grad_(True) if bias else None
    assert kernels.conv3x3_eligible(xg, wg, (1, 1), (1, 1), (1, 1), 1)
    y = kernels.conv3x3(xg, wg, bg)
    assert y.shape == (B, Cout, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(gpu))
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    br = b.float().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(gy.float())
    assert rel_err(y.cpu(), yr) < 5e-3, rel_err(y.cpu(), yr)
    assert rel_err(xg.grad.cpu(), xr.grad) < 5e-3, rel_err(xg.grad.cpu(), xr.grad)
    assert rel_err(wg.grad.cpu(), wr.grad) < 8e-3, rel_err(wg.grad.cpu(), wr.grad)
    if bias:
        assert rel_err(bg.grad.cpu(), br.grad) < 8e-3

    # frozen weight (LoRA runs): cached flipped/transposed weight, no weight gradient
    wf = w.to(gpu).contiguous(memory_format=torch.channels_last)
    xg2 = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(2):
        xg2.grad = None
        kernels.conv3x3(xg2, wf, bg.detach() if bias else None).backward(gy.to(gpu))
        assert rel_err(xg2.grad.cpu(), xr.grad) < 5e-3


@pytest.mark.parametrize("B,Cin,Cout,H,W,bias", [(2, 64, 128, 16, 24, True), (3, 320, 320, 32, 32, True), (1, 128, 64, 10, 6, False),
                                                  (2, 640, 640, 64, 64, True)])
def test_conv3x3_stride2_forward_and_gradients(gpu, B, Cin, Cout, H, W, bias):
    """The SDXL Downsample (src/models/sdxl/denoiser.py:371-408: 3x3, stride 2, padding 1) on the implicit-GEMM kernel's
    stride-2 gather, its data / weight / bias gradients on the stride-1 kernels over the zero-upsampled output gradient —
    against an fp32 F.conv2d on the same bf16-rounded operands; twice, bit-identical (no atomics)."""
    import torch.nn.functional as F

    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(B * 1000 + Cin + H)
    x = _bf(g, B, Cin, H, W)
    w = _bf(g, Cout, Cin, 3, 3, sc=(9 * Cin) ** -0.5)
    bvec = _bf(g, Cout, sc=0.3) if bias else None
    gy = _bf(g, B, Cout, H // 2, W // 2)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    br = bvec.float().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    yr.backward(gy.float())
    outs = []
    for _ in range(2):
        xg = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wg = w.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bg = bvec.to(gpu).requires_grad_(True) if bias else None
        assert kernels.conv3x3_s2_eligible(xg, wg, (2, 2), (1, 1), (1, 1), 1)
        y = kernels.conv3x3_s2(xg, wg, bg)
        y.backward(gy.to(gpu))
        outs.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone(), bg.grad.clone() if bias else None))
    y, dx, dw, db = outs[0]
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert rel_err(y.cpu(), yr) < 5e-3, rel_err(y.cpu(), yr)
    assert rel_err(dx.cpu(), xr.grad) < 5e-3, rel_err(dx.cpu(), xr.grad)
    assert rel_err(dw.cpu(), wr.grad) < 1e-2, rel_err(dw.cpu(), wr.grad)
    if bias:
        assert rel_err(db.cpu(), br.grad) < 1e-2
    # bit-reproducible: output, data and bias gradient always; the weight gradient wherever vft_conv3x3_wgrad takes the shape
    # (4 096 pixels and up — every SDXL bucket; below that the library's kernel, which sums with atomics, is the fallback)
    for name, a, b_ in zip(("y", "dx", "dw", "db"), outs[0], outs[1]):
        if a is not None and (name != "dw" or kernels.conv3x3_wgrad_eligible(B, H, W, Cin, Cout)):
            assert torch.equal(a, b_), name


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 4, 64, 16, 24), (2, 64, 4, 16, 24), (1, 4, 320, 32, 32), (1, 320, 4, 32, 32), (2, 8, 24, 8, 8)])
def test_conv3x3_narrow_channels_forward_and_gradients(gpu, B, Cin, Cout, H, W):
    """The 4-channel stem and head of the UNet (src/models/sdxl/denoiser.py:1114 and :1111-1115) on the implicit-GEMM kernel
    with zero-padded channels: forward, data / weight / bias gradients against fp32 F.conv2d; reproducible bit for bit."""
    import torch.nn.functional as F

    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    x = _bf(g, B, Cin, H, W)
    w = _bf(g, Cout, Cin, 3, 3, sc=(9 * Cin) ** -0.5)
    bvec = _bf(g, Cout, sc=0.3)
    gy = _bf(g, B, Cout, H, W)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), bvec.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(gy.float())
    outs = []
    for _ in range(2):
        xg = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wg, bg = w.to(gpu).requires_grad_(True), bvec.to(gpu).requires_grad_(True)
        assert kernels.conv3x3_narrow_eligible(xg, wg, (1, 1), (1, 1), (1, 1), 1)
        y = kernels.conv3x3_narrow(xg, wg, bg)
        y.backward(gy.to(gpu))
        outs.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone(), bg.grad.clone()))
    y, dx, dw, db = outs[0]
    assert y.shape == yr.shape
    assert rel_err(y.cpu(), yr) < 5e-3 and rel_err(dx.cpu(), xr.grad) < 5e-3
    assert rel_err(dw.cpu(), wr.grad) < 1e-2 and rel_err(db.cpu(), br.grad) < 1e-2
    for name, a, b_ in zip(("y", "dx", "dw", "db"), outs[0], outs[1]):
        if name != "dw" or kernels.conv3x3_wgrad_eligible(B, H, W, (Cin + 63) // 64 * 64, (Cout + 63) // 64 * 64):
            assert torch.equal(a, b_), name


def test_every_tile_variant_gives_the_same_bits(gpu):
    """vft_linear_variant 1 / 2 / 3 (tile sizes) and 11 / 12 / 13 (whole rounds on that tile + the remaining rows as a second
    launch on a smaller tile): the K order per output element does not depend on the tile, so the results are bit-identical —
    for the GEMM with every epilogue input and for the implicit-GEMM convolution, whose first part must still read the halo
    rows that belong to the second part (sizes with several rounds of 256 tile slots so that the split really splits)."""
    from vision_ft_amd import _C, kernels

    g = torch.Generator().manual_seed(21)
    M, N, K = 256 * 70 + 40, 3072, 640   # 71 x 12 = 852 tiles of 256x256: 3 whole rounds + a ragged one
    x, w, b = _bf(g, M, K).to(gpu), _bf(g, N, K, sc=K**-0.5).to(gpu), _bf(g, N, sc=0.2).to(gpu)
    rs, ex, ew = _bf(g, M, N).to(gpu), _bf(g, M, 4, sc=0.5).to(gpu), _bf(g, N, 4, sc=0.5).to(gpu)
    xc = _bf(g, 3, 320, 96, 80).to(gpu).contiguous(memory_format=torch.channels_last)   # 23040 pixels x 640 outputs
    wc = _bf(g, 640, 320, 3, 3, sc=(9 * 320) ** -0.5).to(gpu).contiguous(memory_format=torch.channels_last)
    bc = _bf(g, 640, sc=0.2).to(gpu)
    outs = {}
    try:
        for v in (1, 2, 3, 11, 12, 13):
            assert _C.lib().vft_linear_variant(v) == 0
            outs[v] = (kernels.linear_fused(x, w, b, rs, ex, ew), kernels._conv3x3_call(xc, wc, bc, 3, 96, 80, 320, 640, "conv3x3_fwd"))
    finally:
        _C.lib().vft_linear_variant(0)
    for v in (2, 3, 11, 12, 13):
        assert torch.equal(outs[v][0], outs[1][0]), v
        assert torch.equal(outs[v][1], outs[1][1]), v
    ref = (x.float() @ w.float().t() + b.float() + ex.float() @ ew.float().t()).to(torch.bfloat16).float() + rs.float()
    assert rel_err(outs[11][0].float(), ref) < 5e-3
    import torch.nn.functional as F
    refc = F.conv2d(xc.float(), wc.float(), bc.float(), padding=1)
    assert rel_err(outs[11][1].float(), refc) < 5e-3


@pytest.mark.parametrize("M,R,Cc", [(1000, 640, 640), (16384, 1280, 1280), (4097, 320, 2560), (70, 8, 8), (5000, 264, 136),
                                    (24576, 2560, 640), (40, 72, 264), (129, 1280, 8)])
def test_linear_wgrad_matches_fp32(gpu, M, R, Cc):
    """vft_linear_wgrad (dW = dyᵀ·x, the trainable projection's weight gradient: transposed LDS reads, split reduction
    with ordered partial sums) against the fp32 product; tolerance: bf16 rounding of the output (rel L2 < 5e-3)."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(M + R)
    dy, x = _bf(g, M, R, sc=M**-0.5), _bf(g, M, Cc)
    ref = dy.float().t() @ x.float()
    assert kernels.linear_wgrad_eligible(dy.to(gpu), x.to(gpu))
    dw = kernels.linear_wgrad(dy.to(gpu), x.to(gpu))
    assert dw.shape == (R, Cc) and dw.dtype == torch.bfloat16
    assert rel_err(dw.cpu(), ref) < 5e-3, rel_err(dw.cpu(), ref)


def test_linear_wgrad_views_accumulate_fp32_and_reproducible(gpu):
    """Strided operand views (a column block of a fused projection's gradient / input), accumulation into a view of a flat
    gradient buffer, fp32 output, and bit-identical results run to run."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(5)
    M, R, Cc = 9000, 640, 1280
    dbig, xbig = _bf(g, M, 3 * R, sc=0.01).to(gpu), _bf(g, M, Cc + 64).to(gpu)
    dy, x = dbig[:, R:2 * R], xbig[:, 64:]
    ref = dy.float().t() @ x.float()
    flat = torch.zeros(R * Cc + 128, dtype=torch.bfloat16, device=gpu)
    view = flat[128:].view(R, Cc)
    prev = _bf(g, R, Cc, sc=0.5).to(gpu)
    view.copy_(prev)
    kernels.linear_wgrad(dy, x, out=view, accumulate=True)
    assert rel_err(view.float().cpu(), (ref + prev.float()).cpu()) < 5e-3
    assert torch.equal(flat[:128], torch.zeros(128, dtype=torch.bfloat16, device=gpu))
    o32 = torch.empty(R, Cc, dtype=torch.float32, device=gpu)
    kernels.linear_wgrad(dy, x, out=o32)
    assert rel_err(o32.cpu(), ref.cpu()) < 2e-5, rel_err(o32.cpu(), ref.cpu())
    a, b = kernels.linear_wgrad(dy, x), kernels.linear_wgrad(dy, x)
    assert torch.equal(a, b)


def test_linear_trained_gradients_and_in_place_accumulation(gpu):
    """A trainable nn.Linear through kernels.linear_trained: outputs and input / weight / bias gradients against fp32
    autograd, and the weight gradient accumulated straight into an existing `.grad` (the trainer's flat-buffer view)."""
    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(11)
    B, S, Cin, Cout = 2, 2500, 640, 1280
    x, w, b = _bf(g, B, S, Cin), _bf(g, Cout, Cin, sc=Cin**-0.5), _bf(g, Cout, sc=0.1)
    gy = _bf(g, B, S, Cout, sc=0.02)
    xr, wr, br = (t.float().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.linear(xr, wr, br).backward(gy.float())
    xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    assert kernels.trained_linear_eligible(xg, wg)
    y = kernels.linear_trained(xg, wg, bg)
    assert rel_err(y.float().cpu(), torch.nn.functional.linear(x.float(), w.float(), b.float())) < 5e-3
    y.backward(gy.to(gpu))
    for got, ref in ((xg.grad, xr.grad), (wg.grad, wr.grad), (bg.grad, br.grad)):
        assert rel_err(got.float().cpu(), ref) < 6e-3, rel_err(got.float().cpu(), ref)
    # second pass: .grad exists now -> accumulated in place by the kernel (autograd gets None for the weight)
    before, ptr = wg.grad.clone(), wg.grad.data_ptr()
    kernels.linear_trained(xg, wg, bg).backward(gy.to(gpu))
    assert wg.grad.data_ptr() == ptr
    assert rel_err(wg.grad.float().cpu(), 2 * wr.grad) < 8e-3
    assert not torch.equal(before, wg.grad)
    assert rel_err(bg.grad.float().cpu(), 2 * br.grad) < 8e-3  # vft_colsum accumulated into the existing bias gradient
    # the skip connection in the forward GEMM's epilogue (round 3: all three GEMMs of a trainable layer on our kernels;
    # FeedForward-sized output = more than 30 weight-gradient tiles, which round 2 left to the library)
    Cff = 5120
    w2, res = _bf(g, Cin, Cff, sc=Cff**-0.5), _bf(g, B, S, Cin)
    x2, gy2 = _bf(g, B, S, Cff), _bf(g, B, S, Cin, sc=0.02)
    x2r, w2r, rr = (t.float().requires_grad_(True) for t in (x2, w2, res))
    (rr + torch.nn.functional.linear(x2r, w2r)).backward(gy2.float())
    x2g, w2g, rg = (t.to(gpu).requires_grad_(True) for t in (x2, w2, res))
    y2 = kernels.linear_trained(x2g, w2g, None, rg)
    assert rel_err(y2.float().cpu(), res.float() + torch.nn.functional.linear(x2.float(), w2.float())) < 5e-3
    y2.backward(gy2.to(gpu))
    for got, ref in ((x2g.grad, x2r.grad), (w2g.grad, w2r.grad), (rg.grad, rr.grad)):
        assert rel_err(got.float().cpu(), ref) < 6e-3, rel_err(got.float().cpu(), ref)
    wide = _bf(g, Cff * 2, Cin, sc=Cin**-0.5).to(gpu).requires_grad_(True)  # 640 -> 10240: 120 weight-gradient tiles
    xw = x.to(gpu).requires_grad_(True)
    gyw = _bf(g, B, S, Cff * 2, sc=0.02)
    kernels.linear_trained(xw, wide, None).backward(gyw.to(gpu))
    ref_w = gyw.float().reshape(-1, Cff * 2).t() @ x.float().reshape(-1, Cin)
    assert rel_err(wide.grad.float().cpu(), ref_w) < 6e-3
    assert rel_err(xw.grad.float().cpu(), gyw.float() @ wide.detach().float().cpu()) < 6e-3


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 320, 320, 64, 64), (1, 640, 320, 41, 104), (3, 192, 1280, 48, 32),
                                            (5, 64, 72, 30, 40), (1, 1280, 640, 64, 64)])
def test_conv3x3_weight_gradient_kernel(gpu, B, Cin, Cout, H, W):
    """vft_conv3x3_wgrad (implicit im2col: shifted pixel rows, image borders masked, images back to back in one
    reduction) against fp32 autograd of F.conv2d; also through kernels.conv3x3's backward with the gradient
    accumulated into an existing channels-last `.grad`."""
    import torch.nn.functional as F

    from vision_ft_amd import kernels

    g = torch.Generator().manual_seed(B * 1000 + Cin)
    x, gy = _bf(g, B, Cin, H, W), _bf(g, B, Cout, H, W, sc=(H * W) ** -0.5)
    w = _bf(g, Cout, Cin, 3, 3, sc=(9 * Cin) ** -0.5)
    wr = w.float().requires_grad_(True)
    F.conv2d(x.float(), wr, None, padding=1).backward(gy.float())
    assert kernels.conv3x3_wgrad_eligible(B, H, W, Cin, Cout)
    xg = x.to(gpu).contiguous(memory_format=torch.channels_last)
    gg = gy.to(gpu).contiguous(memory_format=torch.channels_last)
    dw = kernels.conv3x3_wgrad(gg, xg).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    assert rel_err(dw.float().cpu(), wr.grad) < 6e-3, rel_err(dw.float().cpu(), wr.grad)
    # through autograd, twice: the second pass accumulates in place into the channels-last gradient
    wg = w.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = torch.zeros(Cout, dtype=torch.bfloat16, device=gpu, requires_grad=True)
    kernels.conv3x3(xg, wg, bg).backward(gg)
    assert rel_err(wg.grad.float().cpu(), wr.grad) < 6e-3
    bref = gy.float().sum((0, 2, 3))
    assert rel_err(bg.grad.float().cpu(), bref) < 6e-3
    wg.grad = wg.grad.contiguous(memory_format=torch.channels_last)
    ptr = wg.grad.data_ptr()
    kernels.conv3x3(xg, wg, bg).backward(gg)
    assert wg.grad.data_ptr() == ptr and rel_err(wg.grad.float().cpu(), 2 * wr.grad) < 8e-3
    assert rel_err(bg.grad.float().cpu(), 2 * bref) < 8e-3
