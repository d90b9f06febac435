"""Pins the CPU oracle (oracle/) to golden vectors produced by the reference itself
(oracle/gen_golden.py, run next to /root/reference).  CPU only."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import auraflow as oaura
from oracle import index_ops, ops
from oracle import lumina2 as olum
from oracle import sdxl as osdxl
from oracle.attention import sdpa

from .helpers import add_lora_params, autocast_cpu, load_golden, load_index, rel_err, synth_params


@pytest.fixture(scope="module")
def G():
    return load_golden("ops.safetensors")[0]


@pytest.fixture(scope="module")
def J():
    return load_index()


# ------------------------------------------------------------------------------------------
# L0 ops
# ------------------------------------------------------------------------------------------
def test_attention_matches_reference(G):
    out = sdpa(G["attn.q"], G["attn.k"], G["attn.v"])
    assert out.dtype == torch.bfloat16  # fp32 in -> attention dtype out (attention.py:113-118)
    torch.testing.assert_close(out.float(), G["attn.out"].float(), rtol=1.6e-2, atol=1e-2)
    mk = sdpa(G["attn.q"], G["attn.k"], G["attn.v"], scale=0.3, kv_len=G["attn.kv_len"])
    torch.testing.assert_close(mk.float(), G["attn.out_masked"].float(), rtol=1.6e-2, atol=1e-2)
    o2 = sdpa(G["attn.q"], G["attn.k"][:, :, :37], G["attn.v"][:, :, :37])
    torch.testing.assert_close(o2.float(), G["attn.out_qkv"].float(), rtol=1.6e-2, atol=1e-2)


def test_timestep_embeddings_exact(G):
    assert torch.equal(ops.timestep_embedding(G["temb.t_int"], 320, True, 0.0), G["temb.sdxl320"])
    assert torch.equal(ops.timestep_embedding(G["temb.t_int"], 33), G["temb.default33"])
    assert torch.equal(ops.timestep_embedding(G["temb.t_float"], 256, True, 0), G["temb.lumina256"])
    assert torch.equal(ops.aura_timestep_embedding(G["temb.t_float"], 256), G["temb.aura256"])


def test_rope_exact(G):
    assert torch.equal(ops.image_position_indices(16, 24, 3), G["rope.pos"])
    pos = torch.cat([torch.zeros(5, 3), G["rope.pos"]], 0)
    freqs = ops.rope_frequencies(pos, [8, 12, 12], 10000)
    assert torch.equal(freqs, G["rope.freqs"])
    assert torch.equal(ops.apply_rope(G["rope.x"], freqs), G["rope.out"])
    assert torch.equal(ops.apply_rope(G["rope.x"].to(torch.bfloat16), freqs), G["rope.out_bf16"])


def test_lumina_rope_exact(G):
    table = ops.lumina_freqs_table([8, 12, 12], [40, 16, 16], 10000.0)
    fc = ops.lumina_freqs_cis(G["lrope.ids"], table)
    assert torch.equal(torch.view_as_real(fc), G["lrope.freqs_real"])
    assert torch.equal(ops.lumina_apply_rope(G["lrope.x"], fc), G["lrope.out"])
    assert torch.equal(ops.lumina_apply_rope(G["lrope.x"].to(torch.bfloat16), fc), G["lrope.out_bf16"])


def test_norms_exact(G):
    x = G["norm.x"].to(torch.bfloat16)
    assert torch.equal(ops.fp32_layer_norm(x, 64, eps=1e-6), G["norm.fp32ln_noaffine"])
    assert torch.equal(ops.fp32_layer_norm(x, 64, eps=1e-5), G["norm.aura_ln"])


def test_noising_and_losses_exact(G):
    noisy = ops.ddpm_noise(G["ddpm.latents"], G["ddpm.t"], G["ddpm.noise"])
    assert torch.equal(noisy, G["ddpm.noisy"])
    assert torch.equal(ops.ddpm_loss(G["ddpm.noise"], G["ddpm.pred"]), G["ddpm.loss"])
    assert torch.equal(ops.flow_noise(G["ddpm.latents"], G["flow.t"], G["flow.noise"]), G["flow.noisy"])
    assert torch.equal(ops.flow_loss(G["ddpm.latents"], G["flow.noise"], G["ddpm.pred"]), G["flow.loss"])


def test_lora_linear_matches_reference(G):
    from oracle import fixture_weights as fw

    shapes = {"lora_down.weight": (4, 48), "lora_up.weight": (40, 4), "alpha": (), "linear.weight": (40, 48), "linear.bias": (40,)}
    sd = fw.synth_state_dict(shapes, seed=3)
    # the reference layer keeps lora params in bf16 and the base in fp32
    P = {f"l.{k}": (v.to(torch.bfloat16) if k.startswith("lora") or k == "alpha" else v) for k, v in sd.items()}
    assert abs(fw.checksum({k[2:]: v for k, v in P.items()}) - float(G["lora.checksum"])) < 1e-3
    for k in ("l.lora_down.weight", "l.lora_up.weight"):
        P[k] = P[k].clone().requires_grad_(True)
    x = G["lora.x"].clone().requires_grad_(True)
    with autocast_cpu():
        y = ops.linear(P, "l", x)
    y.backward(G["lora.gy"].to(y.dtype))
    assert torch.equal(y, G["lora.y"])
    assert torch.equal(x.grad, G["lora.dx"])
    assert torch.equal(P["l.lora_down.weight"].grad, G["lora.d_down"])
    assert torch.equal(P["l.lora_up.weight"].grad, G["lora.d_up"])


def _lora_conv_case(tag):
    from oracle import fixture_weights as fw

    G, meta = load_golden("lora_conv.safetensors")
    cin, cout, k, st, pd, rank = (int(v) for v in G[f"{tag}.geom"])
    shapes = {"alpha": (), "conv2d.bias": (cout,), "conv2d.weight": (cout, cin, k, k),
              "lora_down.weight": (rank, cin, k, k), "lora_up.weight": (cout, rank, 1, 1)}
    assert sorted(shapes) == json.loads(meta["state_dict_keys"])  # the reference layer's own state-dict keys
    sd = fw.synth_state_dict(shapes, seed=5)
    sd["alpha"] = G[f"{tag}.alpha"].clone()
    # the reference layer keeps the adapter in bf16 (LoRAConfig.dtype default) and the base convolution in fp32
    sd = {k_: (v.to(torch.bfloat16) if k_.startswith("lora") or k_ == "alpha" else v) for k_, v in sd.items()}
    return G, sd, (cin, cout, k, st, pd, rank)


@pytest.mark.parametrize("tag", ["wide", "strided"])
def test_lora_conv2d_oracle_matches_reference(tag):
    """oracle/ops.py::conv2d's LoRAConv2d branch against the reference layer's own forward / backward numbers
    (src/modules/peft/lora.py:167-309; tests/golden/lora_conv.safetensors, written by oracle/gen_golden.py::gen_lora_conv)."""
    G, sd, (cin, cout, k, st, pd, rank) = _lora_conv_case(tag)
    P = {f"c.{k_}": v for k_, v in sd.items()}
    for k_ in ("c.lora_down.weight", "c.lora_up.weight"):
        P[k_] = P[k_].clone().requires_grad_(True)
    x = G[f"{tag}.x"].clone().requires_grad_(True)
    with autocast_cpu():
        y = ops.conv2d(P, "c", x, stride=st, padding=pd)
    y.backward(G[f"{tag}.gy"].to(y.dtype))
    assert torch.equal(y.float(), G[f"{tag}.y"])
    assert torch.equal(x.grad, G[f"{tag}.dx"])
    assert torch.equal(P["c.lora_down.weight"].grad.float(), G[f"{tag}.d_down"])
    assert torch.equal(P["c.lora_up.weight"].grad.float(), G[f"{tag}.d_up"])


@pytest.mark.parametrize("tag", ["wide", "strided"])
def test_lora_conv2d_module_matches_reference_on_cpu(tag):
    """The product's LoRAConv2d (vision_ft_amd/modules/peft/lora.py) — constructor, state-dict keys, frozen base, forward,
    disabled forward, every adapter gradient, from_weights round trip — against the same golden numbers on the CPU."""
    from vision_ft_amd.modules.peft.lora import LoRAConfig, LoRAConv2d

    G, sd, (cin, cout, k, st, pd, rank) = _lora_conv_case(tag)
    base = torch.nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pd)
    layer = LoRAConv2d(LoRAConfig(rank=rank, alpha=float(G[f"{tag}.alpha"])), base)
    assert sorted(layer.state_dict()) == sorted(sd)
    assert torch.count_nonzero(layer.lora_up.weight) == 0  # zero-init up: the adapter starts as a no-op (reference :235-238)
    layer.load_state_dict(sd)
    assert not base.weight.requires_grad and not base.bias.requires_grad and not layer.alpha.requires_grad
    assert layer.lora_down.weight.requires_grad and layer.lora_up.weight.requires_grad
    x = G[f"{tag}.x"].clone().requires_grad_(True)
    with autocast_cpu():
        y = layer(x)
    y.backward(G[f"{tag}.gy"].to(y.dtype))
    assert torch.equal(y.float(), G[f"{tag}.y"]) and torch.equal(x.grad, G[f"{tag}.dx"])
    assert torch.equal(layer.lora_down.weight.grad.float(), G[f"{tag}.d_down"])
    assert torch.equal(layer.lora_up.weight.grad.float(), G[f"{tag}.d_up"])
    layer.set_enabled(False)
    with autocast_cpu(), torch.no_grad():
        assert torch.equal(layer(x.detach()).float(), G[f"{tag}.y_disabled"])
    layer.set_enabled(True)
    clone = LoRAConv2d.from_weights({k_: v for k_, v in layer.state_dict().items() if not k_.startswith("conv2d.")}, base)
    with autocast_cpu(), torch.no_grad():
        assert torch.equal(clone(x.detach()).float(), G[f"{tag}.y"])
    layer.train()
    assert not layer.conv2d.training
    layer.requires_grad_(True)
    assert not base.weight.requires_grad


# ------------------------------------------------------------------------------------------
# integer / string work: bit-exact
# ------------------------------------------------------------------------------------------
def test_generate_buckets_exact(J):
    for name, rec in J["buckets"].items():
        got = index_ops.generate_buckets(**rec["kwargs"])
        assert got.tolist() == rec["rows"], name
    assert len(J["buckets"]["b1024_s64_m384"]["rows"]) == 420  # duplicates included
    assert len(J["buckets"]["b1024_s128_m384"]["rows"]) == 115


def test_find_nearest_exact(J):
    for key in ("find_nearest", "find_nearest_128"):
        rec = J[key]
        buckets = np.array(J["buckets"][rec["bucket_set"]]["rows"])
        for w, h, want in rec["cases"]:
            assert index_ops.find_nearest(buckets, w, h) == want, (key, w, h)
    b = np.array(J["buckets"]["b1024_s64_m384"]["rows"])
    assert b[index_ops.find_nearest(b, 1920, 1080)].tolist() == [1280, 832]  # early-break quirk


def test_bucket_batches_exact(J):
    for key, want in J["bucket_batches"].items():
        n, bs, rep = (int(v) for v in key.split("_"))
        assert index_ops.bucket_batch_indices(n, bs, rep) == want, key


def test_target_keys_exact(J):
    keys = J["target_keys"]["keys"]
    for case in J["target_keys"]["cases"]:
        got = sorted(index_ops.get_target_keys(case["include"], case["exclude"], keys))
        assert got == case["result"], case


def test_position_ids_and_patchify_exact(J):
    for rec in J["position_ids"]:
        assert olum.position_ids(*rec["args"]).tolist() == rec["ids"]
    cfg = oaura.AuraCfg(pos_embed_max_size=256)
    for rec in J["aura_pe_index"]:
        assert oaura.pe_index(cfg, *rec["hw"]).tolist() == rec["idx"]
    x = torch.arange(2 * 4 * 4 * 6, dtype=torch.float32).view(2, 4, 4, 6)
    p = oaura.patchify(x, 2)
    assert p.tolist() == J["aura_patchify"]
    assert oaura.unpatchify(p, 2, 3, 2, 4).tolist() == J["aura_unpatchify"]
    xl = torch.arange(8 * 4 * 6, dtype=torch.float32).view(8, 4, 6)
    pl = olum.patchify(xl, 2)
    assert pl.tolist() == J["lumina_patchify"]
    assert olum.unpatchify(pl, 4, 6, 2, 8).tolist() == J["lumina_unpatchify"]
    assert torch.equal(olum.unpatchify(pl, 4, 6, 2, 8), xl)  # round trip


# ------------------------------------------------------------------------------------------
# denoisers: forward, backward, LoRA training steps
# ------------------------------------------------------------------------------------------
def _leafify(P, names):
    for n in names:
        P[n] = P[n].clone().requires_grad_(True)
    return P


def _train_lora(P, forward_loss, steps=3):
    params = [v for k, v in P.items() if ".lora_" in k]
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.AdamW(params, lr=1e-3)
    losses = []
    for _ in range(steps):
        with autocast_cpu():
            loss = forward_loss(P)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.detach().float())
    return torch.stack(losses)


@pytest.mark.parametrize("fixture", ["sdxl_tiny", "sdxl_tiny_h16w24", "sdxl_tiny_h24w16"])
def test_sdxl_oracle_matches_reference(fixture):
    """Square bucket and the two non-square ones (BASELINE config 5: mixed aspect-ratio buckets)."""
    from vision_ft_amd.models.sdxl import Denoiser, DenoiserConfig

    T, meta = load_golden(fixture + ".safetensors")
    cfgd = json.loads(meta["config"])
    with torch.device("meta"):
        mod = Denoiser(DenoiserConfig(**cfgd))
    P = synth_params(mod, 0, meta["checksum"])
    cfg = osdxl.SDXLCfg(**cfgd)
    grads = [k[5:] for k in T if k.startswith("grad.")]
    _leafify(P, grads)
    x = T["noisy"].clone().requires_grad_(True)

    def fwd(P, x):
        return osdxl.unet_forward(P, cfg, x, T["timestep"], T["ehs"], T["pooled"], T["original_size"],
                                  T["target_size"], T["crop"])

    with autocast_cpu():
        out = fwd(P, x)
        loss = ops.ddpm_loss(T["noise"], out)
    loss.backward()
    assert rel_err(out, T["out"]) < 2e-3
    torch.testing.assert_close(loss.float(), T["loss"], rtol=2e-3, atol=0)
    assert rel_err(x.grad, T["d_noisy"]) < 1e-2
    for g in grads:
        assert rel_err(P[g].grad, T["grad." + g]) < 2e-2, g
    # LoRA: 3 AdamW steps, per-step loss
    Pl = add_lora_params({k: v.detach() for k, v in P.items()}, json.loads(meta["lora_keys"]))
    assert sorted(k for k in Pl if ".lora_" in k or k.endswith(".alpha")) == json.loads(meta["lora_keys"])
    losses = _train_lora(Pl, lambda P_: ops.ddpm_loss(T["noise"], fwd(P_, T["noisy"])))
    torch.testing.assert_close(losses, T["lora.losses"], rtol=2e-3, atol=0)


def test_sdxl_rope_flow_match_oracle_matches_reference():
    """DenoiserWithRoPE (2-D RoPE on attn1/attn2, origin "center", non-square latent) under the
    flow-matching step of train/sdxl/flow_match.py — tables and position ids exact, rotation exact, forward / loss /
    gradients / LoRA losses within the CPU-vs-CPU tolerances of the plain SDXL test."""
    from vision_ft_amd.models.sdxl.adapter.rope import DenoiserConfigWithRoPE, DenoiserWithRoPE, RoPEEmbedder
    from vision_ft_amd.modules.loss import flow_match as FM

    T, meta = load_golden("sdxl_rope_tiny.safetensors")
    cfgd, rope = json.loads(meta["config"]), json.loads(meta["rope"])
    assert rope == dict(rope_dims=[32, 32], rope_theta=10000.0, origin_position="center")  # DenoiserConfigWithRoPE defaults
    h, w = T["latents"].shape[-2:]
    # index work and tables: exact, oracle and product (the product builds its tables on the host like the reference)
    ids = ops.sdxl_image_position_ids(h, w, "center")
    assert torch.equal(ids, T["ids.image"])
    assert torch.equal(ops.sdxl_image_position_ids(5, 7, "top_left"), T["ids.image_top_left"])
    f_img = ops.sdxl_rope_freqs(ids, rope["rope_dims"], rope["rope_theta"])
    f_ctx = ops.sdxl_rope_freqs(ops.sdxl_context_position_ids(T["ehs"].shape[1]), rope["rope_dims"], rope["rope_theta"])
    assert torch.equal(torch.view_as_real(f_img), T["freqs.image"])
    assert torch.equal(torch.view_as_real(f_ctx), T["freqs.context"])
    emb = RoPEEmbedder(**rope)
    assert torch.equal(emb.get_image_position_ids(1, h, w)[0], T["ids.image"])
    assert torch.equal(RoPEEmbedder([32, 32]).get_image_position_ids(1, 5, 7)[0], T["ids.image_top_left"])
    assert torch.equal(torch.view_as_real(emb.get_image_freqs(1, h, w, torch.device("cpu"))[0]), T["freqs.image"])
    assert torch.equal(torch.view_as_real(emb.get_context_freqs(1, 11, torch.device("cpu"))[0]), T["freqs.context"])
    cos, sin = emb.image_table(h, w, torch.device("cpu"))
    assert torch.equal(cos, T["freqs.image"][..., 0]) and torch.equal(sin, T["freqs.image"][..., 1])
    assert torch.equal(ops.sdxl_apply_rope(T["apply.in"], f_img), T["apply.out"])
    # noising and the x0 -> velocity conversion: exact (product functions and oracle restatement)
    assert torch.equal(ops.flow_scaled_noise(T["latents"], T["t"], T["noise"]), T["noisy"])
    assert torch.equal(ops.flow_scaled_noise(T["latents"], T["t"], T["noise"], clean_at_zero=True), T["noisy_c0"])
    torch.manual_seed(123)
    n2, e2 = FM.prepare_scaled_noised_latents(T["latents"], T["t"], noise_scale=1.0, clean_at_zero=False)
    assert torch.equal(n2, T["noisy"]) and torch.equal(e2, T["noise"])
    assert torch.equal(ops.x0_to_velocity(T["latents"], T["noisy"], T["t"]), T["x0_velocity_target"])
    assert torch.equal(FM.convert_x0_to_velocity(T["latents"], T["noisy"], T["t"]), T["x0_velocity_target"])

    with torch.device("meta"):
        mod = DenoiserWithRoPE(DenoiserConfigWithRoPE(**cfgd))
    P = synth_params(mod, 0, meta["checksum"])  # no RoPE parameters: the plain SDXL state dict
    cfg = osdxl.SDXLCfg(**cfgd)
    grads = [k[5:] for k in T if k.startswith("grad.")]
    _leafify(P, grads)
    x = T["noisy"].clone().requires_grad_(True)

    def fwd(P, x, rope=rope):
        return osdxl.unet_forward(P, cfg, x, T["t"] * 1000, T["ehs"], T["pooled"], T["original_size"],
                                  T["target_size"], T["crop"], rope=rope)

    with autocast_cpu():
        out = fwd(P, x)
        loss = ops.flow_loss(T["latents"], T["noise"], out)
        loss_img = F.mse_loss(ops.x0_to_velocity(out, T["noisy"], T["t"]), ops.x0_to_velocity(T["latents"], T["noisy"], T["t"]))
        out_off = fwd(P, T["noisy"], rope=None)
    loss.backward()
    assert rel_err(out, T["out"]) < 2e-3
    assert rel_err(out_off, T["out_rope_off"]) < 2e-3
    assert rel_err(out, T["out_rope_off"]) > 2e-2  # the rotation matters at this size: a path that ignores it fails
    torch.testing.assert_close(loss.float(), T["loss"], rtol=2e-3, atol=0)
    torch.testing.assert_close(loss_img.float(), T["loss_image_pred"], rtol=2e-3, atol=0)
    assert rel_err(x.grad, T["d_noisy"]) < 1e-2
    for g in grads:
        assert rel_err(P[g].grad, T["grad." + g]) < 2e-2, g
    Pl = add_lora_params({k: v.detach() for k, v in P.items()}, json.loads(meta["lora_keys"]))
    losses = _train_lora(Pl, lambda P_: ops.flow_loss(T["latents"], T["noise"], fwd(P_, T["noisy"])))
    torch.testing.assert_close(losses, T["lora.losses"], rtol=2e-3, atol=0)


@pytest.mark.parametrize("tag", ["", "_rope", "_d256", "_d256_rope"])
def test_aura_oracle_matches_reference(tag):
    from vision_ft_amd.models.auraflow import Denoiser, DenoiserConfig

    T, meta = load_golden(f"aura_tiny{tag}.safetensors")
    cfgd = json.loads(meta["config"])
    with torch.device("meta"):
        mod = Denoiser(DenoiserConfig(**cfgd))
    P = synth_params(mod, 1, meta["checksum"])
    cfg = oaura.AuraCfg(**{k: v for k, v in cfgd.items() if k in oaura.AuraCfg.__dataclass_fields__})
    grads = [k[5:] for k in T if k.startswith("grad.")]
    _leafify(P, grads)
    x = T["noisy"].clone().requires_grad_(True)
    with autocast_cpu():
        out = oaura.mmdit_forward(P, cfg, x, T["ehs"], T["timestep"])
        loss = ops.flow_loss(T["latents"], T["noise"], out)
    loss.backward()
    assert rel_err(out, T["out"]) < 2e-3
    torch.testing.assert_close(loss.float(), T["loss"], rtol=2e-3, atol=0)
    assert rel_err(x.grad, T["d_noisy"]) < 1e-2
    for g in grads:
        assert rel_err(P[g].grad, T["grad." + g]) < 2e-2, g
    Pl = add_lora_params({k: v.detach() for k, v in P.items()}, json.loads(meta["lora_keys"]))
    losses = _train_lora(Pl, lambda P_: ops.flow_loss(T["latents"], T["noise"],
                                                      oaura.mmdit_forward(P_, cfg, T["noisy"], T["ehs"], T["timestep"])))
    torch.testing.assert_close(losses, T["lora.losses"], rtol=2e-3, atol=0)
    # BASELINE config 3 trains rank 16
    Pl = add_lora_params({k: v.detach() for k, v in P.items()}, json.loads(meta["lora_keys"]), rank=16)
    losses = _train_lora(Pl, lambda P_: ops.flow_loss(T["latents"], T["noise"],
                                                      oaura.mmdit_forward(P_, cfg, T["noisy"], T["ehs"], T["timestep"])))
    torch.testing.assert_close(losses, T["lora16.losses"], rtol=2e-3, atol=0)


@pytest.mark.parametrize("fixture", ["lumina_tiny", "lumina_tiny_d96"])
def test_lumina_oracle_matches_reference(fixture):
    from vision_ft_amd.models.lumina2 import Denoiser, DenoiserConfig

    T, meta = load_golden(fixture + ".safetensors")
    cfgd = json.loads(meta["config"])
    with torch.device("meta"):
        mod = Denoiser(DenoiserConfig(**cfgd))
    P = synth_params(mod, 2, meta["checksum"])
    cfg = olum.LuminaCfg(**{k: v for k, v in cfgd.items() if k in olum.LuminaCfg.__dataclass_fields__})
    grads = [k[5:] for k in T if k.startswith("grad.")]
    _leafify(P, grads)
    x = T["noisy"].clone().requires_grad_(True)

    def fwd(P_, x_):
        outs, _, _ = olum.nextdit_forward(P_, cfg, x_, T["caption"], T["timestep"], T["caption_mask"])
        return -torch.stack(outs, 0)

    with autocast_cpu():
        out = fwd(P, x)
        loss = ops.flow_loss(T["latents"], T["noise"], out)
    loss.backward()
    assert rel_err(out, T["out"]) < 2e-3
    torch.testing.assert_close(loss.float(), T["loss"], rtol=2e-3, atol=0)
    assert rel_err(x.grad, T["d_noisy"]) < 1e-2
    for g in grads:
        assert rel_err(P[g].grad, T["grad." + g]) < 2e-2, g
    # ragged batch (two resolutions)
    with autocast_cpu(), torch.no_grad():
        outs, _, capf = olum.nextdit_forward(P, cfg, [T["ragged.in0"], T["ragged.in1"]], T["caption"], T["timestep"],
                                             T["caption_mask"])
    assert rel_err(outs[0], T["ragged.out0"]) < 2e-3 and rel_err(outs[1], T["ragged.out1"]) < 2e-3
    assert rel_err(capf, T["ragged.caption_features"]) < 2e-3
    Pl = add_lora_params({k: v.detach() for k, v in P.items()}, json.loads(meta["lora_keys"]))
    losses = _train_lora(Pl, lambda P_: ops.flow_loss(T["latents"], T["noise"], fwd(P_, T["noisy"])))
    torch.testing.assert_close(losses, T["lora.losses"], rtol=2e-3, atol=0)
