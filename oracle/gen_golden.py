"""Generate tests/golden/* by running the REFERENCE implementation (imported from /root/reference).

Run in the build container only:  ``python oracle/gen_golden.py``.  The reference's source never
enters this repository or the GPU box — only the inputs and outputs written here do.  Weights are
not stored: they are rebuilt from ``oracle/fixture_weights.py`` (checksum recorded).

Import recipe: the model packages' ``__init__`` pull diffusers via pipeline/vae, so
bare package stubs are registered for ``src.models.{sdxl,auraflow,lumina2}`` and the denoiser/config
modules are imported directly; torchvision/imagesize are stubbed for the bucket code (numpy only).
"""
from __future__ import annotations

import contextlib
import importlib
import io
import json
import os
import sys
import types

import numpy as np
import torch
from safetensors.torch import save_file

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import fixture_weights as fw  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    for fam in ("sdxl", "auraflow", "lumina2"):
        name = f"src.models.{fam}"
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, "src", "models", fam)]
        sys.modules[name] = mod
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)
    import transformers.activations  # noqa: F401
    # src.dataset/__init__ imports the whole dataset zoo; load the two numpy-only modules by path
    ds = types.ModuleType("src.dataset")
    ds.__path__ = [os.path.join(REF, "src", "dataset")]
    sys.modules["src.dataset"] = ds


def autocast():
    return torch.autocast("cpu", dtype=torch.bfloat16)


def shapes_of(module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed=0):
    sd = fw.synth_state_dict(shapes_of(module), seed)
    cur = module.state_dict()
    sd = {k: v.to(cur[k].dtype) for k, v in sd.items()}
    module.load_state_dict(sd, strict=True)
    return fw.checksum(module.state_dict())


def t2c(d):
    return {k: v.detach().contiguous().clone() for k, v in d.items()}


# ----------------------------------------------------------------------------------------------
def gen_ops():
    from src.modules import attention as A
    from src.modules.loss import diffusion as LD
    from src.modules.loss import flow_match as LF
    from src.modules.norm import FP32LayerNorm
    from src.modules.positional_encoding import rope as R
    from src.modules.timestep import embedding as TE

    aura = importlib.import_module("src.models.auraflow.denoiser")
    lum = importlib.import_module("src.models.lumina2.denoiser")

    g = torch.Generator().manual_seed(1234)
    T = {}
    # attention (fp32 in -> bf16 out), with and without a bool mask, GQA-free
    q = torch.randn(2, 3, 37, 32, generator=g)
    k = torch.randn(2, 3, 45, 32, generator=g)
    v = torch.randn(2, 3, 45, 32, generator=g)
    T["attn.q"], T["attn.k"], T["attn.v"] = q, k, v
    T["attn.out"] = A.scaled_dot_product_attention(q, k, v)
    lens = torch.tensor([45, 17])
    mask = (torch.arange(45)[None, :] < lens[:, None]).view(2, 1, 1, 45).expand(-1, 3, 37, -1)
    T["attn.kv_len"] = lens.to(torch.int32)
    T["attn.out_masked"] = A.scaled_dot_product_attention(q, k, v, mask=mask, scale=0.3)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        T["attn.out_qkv"] = A.scaled_qkv_attention(q, k[:, :, :37], v[:, :, :37], scale=None, use_flash=False)

    # timestep embeddings
    ts = torch.tensor([0, 1, 50, 999, 512], dtype=torch.int32)
    T["temb.t_int"] = ts
    T["temb.sdxl320"] = TE.get_timestep_embedding(ts, 320, flip_sin_to_cos=True, downscale_freq_shift=0.0)
    T["temb.default33"] = TE.get_timestep_embedding(ts, 33)
    tf = torch.tensor([0.0, 0.25, 0.5, 0.999, 1.0])
    T["temb.t_float"] = tf
    T["temb.lumina256"] = TE.get_timestep_embedding(tf, 256, flip_sin_to_cos=True, downscale_freq_shift=0)
    T["temb.aura256"] = aura.TimestepEmbedder.timestep_embedding(tf, 256)

    # flux-style rope (AuraFlow use_rope)
    pos = R.image_position_indices(16, 24, 3)
    T["rope.pos"] = pos
    allpos = torch.cat([torch.zeros(5, 3), pos], dim=0)
    freqs = R.get_rope_frequencies(allpos, [8, 12, 12], 10000)
    T["rope.freqs"] = freqs
    x = torch.randn(2, 3, allpos.shape[0], 32, generator=g)
    T["rope.x"] = x
    with contextlib.redirect_stdout(io.StringIO()):
        T["rope.out"] = R.applye_rope_frequencies(x, freqs)
        T["rope.out_bf16"] = R.applye_rope_frequencies(x.to(torch.bfloat16), freqs)

    # lumina complex rope
    emb = lum.RoPEEmbedder(theta=10000.0, axes_dims=[8, 12, 12], axes_lens=[40, 16, 16])
    ids = torch.stack([torch.randint(0, 40, (2, 50), generator=g), torch.randint(0, 16, (2, 50), generator=g),
                       torch.randint(0, 16, (2, 50), generator=g)], dim=-1).to(torch.int32)
    fc = emb(ids)
    T["lrope.ids"] = ids
    T["lrope.freqs_real"] = torch.view_as_real(fc).contiguous()
    xl = torch.randn(2, 50, 4, 32, generator=g)
    T["lrope.x"] = xl
    T["lrope.out"] = lum.SelfAttention.apply_rope(None, xl, fc)
    T["lrope.out_bf16"] = lum.SelfAttention.apply_rope(None, xl.to(torch.bfloat16), fc)

    # norms
    xn = torch.randn(2, 7, 64, generator=g) * 3 + 1
    T["norm.x"] = xn
    T["norm.fp32ln_noaffine"] = FP32LayerNorm(64, elementwise_affine=False, eps=1e-6)(xn.to(torch.bfloat16))
    T["norm.aura_ln"] = aura.Fp32LayerNorm(64, bias=False, elementwise_affine=False)(xn.to(torch.bfloat16))

    # DDPM noising + loss (noise stored)
    lat = torch.randn(3, 4, 8, 8, generator=g)
    tt = torch.tensor([0, 500, 999], dtype=torch.int32)
    torch.manual_seed(7)
    noisy, noise = LD.prepare_noised_latents(lat, tt)
    T["ddpm.latents"], T["ddpm.t"], T["ddpm.noise"], T["ddpm.noisy"] = lat, tt, noise, noisy
    pred = torch.randn(3, 4, 8, 8, generator=g)
    T["ddpm.pred"] = pred
    T["ddpm.loss"] = LD.loss_with_predicted_noise(lat, noise, pred)
    # flow matching
    tfm = torch.tensor([0.1, 0.5, 0.93])
    torch.manual_seed(8)
    fnoisy, fnoise = LF.prepare_noised_latents(lat, tfm)
    T["flow.t"], T["flow.noise"], T["flow.noisy"] = tfm, fnoise, fnoisy
    T["flow.loss"] = LF.loss_with_predicted_velocity(lat, fnoise, pred)

    # LoRA linear forward/backward (reference layer, synthetic weights, autocast bf16)
    from src.modules.peft.lora import LoRAConfig, LoRALinear

    base = torch.nn.Linear(48, 40, bias=True)
    layer = LoRALinear(LoRAConfig(rank=4, alpha=2.0), base)
    T["lora.checksum"] = torch.tensor(load_synth(layer, seed=3))
    layer.requires_grad_(True)
    xin = torch.randn(2, 9, 48, generator=g).requires_grad_(True)
    with autocast():
        y = layer(xin)
    gy = torch.randn(2, 9, 40, generator=g)
    y.backward(gy.to(y.dtype))
    T["lora.x"], T["lora.y"], T["lora.gy"] = xin.detach(), y.detach(), gy
    T["lora.dx"] = xin.grad
    T["lora.d_down"] = layer.lora_down.weight.grad
    T["lora.d_up"] = layer.lora_up.weight.grad
    save_file(t2c(T), os.path.join(OUT, "ops.safetensors"))
    print("ops:", len(T), "tensors")


def gen_lora_conv():
    """LoRAConv2d (src/modules/peft/lora.py:167-309): the reference layer's forward / backward on two geometries — a
    3x3 / stride 1 / padding 1 convolution wide enough for the implicit-GEMM kernel (64 -> 96 channels) and a strided,
    narrow one (8 -> 6, stride 2) — with a NON-zero up-projection so that every adapter gradient is exercised."""
    from src.modules.peft.lora import LoRAConfig, LoRAConv2d

    g = torch.Generator().manual_seed(77)
    T = {}
    for tag, (cin, cout, k, st, pd, hw, rank, alpha) in {"wide": (64, 96, 3, 1, 1, (12, 10), 4, 2.0),
                                                         "strided": (8, 6, 3, 2, 1, (9, 14), 2, 1.0)}.items():
        base = torch.nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pd)
        layer = LoRAConv2d(LoRAConfig(rank=rank, alpha=alpha), base)
        sd = fw.synth_state_dict(shapes_of(layer), seed=5)
        sd["alpha"] = torch.tensor(alpha)
        cur = layer.state_dict()
        layer.load_state_dict({k_: v.to(cur[k_].dtype) for k_, v in sd.items()}, strict=True)
        layer.requires_grad_(True)
        xin = torch.randn(2, cin, *hw, generator=g).requires_grad_(True)
        with autocast():
            y = layer(xin)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.to(y.dtype))
        T[f"{tag}.x"], T[f"{tag}.y"], T[f"{tag}.gy"] = xin.detach(), y.detach().float(), gy
        T[f"{tag}.dx"] = xin.grad
        T[f"{tag}.d_down"] = layer.lora_down.weight.grad.float()
        T[f"{tag}.d_up"] = layer.lora_up.weight.grad.float()
        T[f"{tag}.geom"] = torch.tensor([cin, cout, k, st, pd, rank])
        T[f"{tag}.alpha"] = torch.tensor(alpha)
        layer.set_enabled(False)
        with autocast(), torch.no_grad():
            T[f"{tag}.y_disabled"] = layer(xin.detach()).float()
        assert layer.lora_down.weight.dtype == torch.bfloat16 and base.weight.requires_grad is False
    T["keys"] = torch.tensor(len(layer.state_dict()))
    save_file(t2c(T), os.path.join(OUT, "lora_conv.safetensors"),
              metadata={"state_dict_keys": json.dumps(sorted(layer.state_dict().keys()))})
    print("lora_conv:", len(T), "tensors", sorted(layer.state_dict().keys()))


# ----------------------------------------------------------------------------------------------
TINY_SDXL = dict(in_channels=4, out_channels=4, hidden_dim=64, num_head_channels=64, context_dim=64,
                 block_out_channels=[64, 128, 128],  # global_cond_dim / additional_condition_dim are NOT
                 # forwarded by the reference's Denoiser.__init__ (sdxl/denoiser.py:1239-1256): always 2816 / 256
                 num_transformers_per_block=[1, 1, 2], layers_per_block=2)  # skip bookkeeping assumes 2 (denoiser.py:1080-1092)
TINY_AURA = dict(in_channels=4, out_channels=4, patch_size=2, caption_projection_dim=128, num_double_layers=1,
                 num_single_layers=2, num_attention_heads=2, attention_head_dim=64, joint_attention_dim=32,
                 pos_embed_max_size=16 * 16, num_register_tokens=8, use_flash_attn=False)
TINY_LUMINA = dict(in_channels=8, out_channels=8, hidden_dim=256, caption_dim=48, timestep_embed_dim=64,
                   depth=2, num_heads=4, num_kv_heads=2, refiner_depth=1, multiple_of=64,
                   axes_dims=[16, 24, 24], axes_lens=[64, 32, 32], vae_channels=8)
# The head geometries of the REAL models: AuraFlow's 256-wide heads over more than 512 tokens (the
# stored-dS backward and attn_fwd2_kernel<256> only run there) and Lumina2's 96-wide heads with grouped K/V (6 query heads
# on 2 key/value heads, as 24 / 8) and ragged caption lengths (the key-padding mask inside the head_dim 96 kernels).
TINY_AURA_D256 = dict(TINY_AURA, caption_projection_dim=512, attention_head_dim=256, pos_embed_max_size=32 * 32)
TINY_LUMINA_D96 = dict(TINY_LUMINA, hidden_dim=576, num_heads=6, num_kv_heads=2, axes_dims=[32, 32, 32],
                       axes_lens=[64, 32, 32])


def lora_train_steps(model, call, peft_include, rank, target_fn, steps=3):
    """Reference LoRA replacement + `steps` AdamW steps; returns losses and the LoRA key list."""
    from src.modules.peft import PeftTargetConfig
    from src.modules.peft.lora import LoRAConfig

    model.requires_grad_(False)
    PeftTargetConfig(include_keys=peft_include, config=LoRAConfig(rank=rank, alpha=1.0)).replace_to_peft_layer(model)
    # deterministic LoRA init (the reference's kaiming/zeros init depends on the torch RNG stream)
    sd = model.state_dict()
    lora_shapes = {k: tuple(v.shape) for k, v in sd.items() if "lora_" in k}
    synth = fw.synth_state_dict(lora_shapes, seed=11)
    for k, v in synth.items():
        if "lora_up" in k:
            v = torch.zeros_like(v)  # reference init: zeros(up)  (src/modules/peft/lora.py:74-77)
        sd[k] = v.to(sd[k].dtype)
    model.load_state_dict(sd)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3)
    losses = []
    for _ in range(steps):
        with autocast():
            loss = target_fn(call(model))
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.detach().float())
    keys = sorted(k for k in model.state_dict() if "lora_" in k or k.endswith(".alpha"))
    return torch.stack(losses), keys


def gen_sdxl(hw=(16, 16), fname="sdxl_tiny.safetensors", seed=42, sizes=None):
    """hw = latent (height, width).  The default writes the square fixture of round 1 unchanged; `gen_sdxl_rect`
    writes two non-square buckets (BASELINE config 5: mixed aspect-ratio buckets)."""
    cfgm = importlib.import_module("src.models.sdxl.config")
    den = importlib.import_module("src.models.sdxl.denoiser")
    from src.modules.loss import diffusion as LD

    cfg = cfgm.DenoiserConfig(**TINY_SDXL)
    model = den.Denoiser(cfg)
    cs = load_synth(model, seed=0)
    g = torch.Generator().manual_seed(seed)
    B = 2
    sizes = sizes or dict(original_size=[[512, 512], [768, 640]], target_size=[[512, 512], [704, 640]], crop=[[0, 0], [32, 0]])
    T = {
        "latents": torch.randn(B, 4, *hw, generator=g),
        "noise": torch.randn(B, 4, *hw, generator=g),
        "timestep": torch.tensor([17, 803], dtype=torch.int32),
        "ehs": torch.randn(B, 11, 64, generator=g),
        "pooled": torch.randn(B, 1280, generator=g),
        "original_size": torch.tensor(sizes["original_size"]),
        "target_size": torch.tensor(sizes["target_size"]),
        "crop": torch.tensor(sizes["crop"]),
    }
    noisy = LD.prepare_noised_latents(T["latents"], T["timestep"])  # only for the alpha table; redo with stored noise
    ac = (1.0 - torch.linspace(0.00085**0.5, 0.012**0.5, 1000, dtype=torch.float32) ** 2).cumprod(0)
    a = ac[T["timestep"].long()].sqrt().view(B, 1, 1, 1)
    s = (1 - ac[T["timestep"].long()]).sqrt().view(B, 1, 1, 1)
    T["noisy"] = a * T["latents"] + s * T["noise"]
    del noisy

    def call(m, x=None):
        return m(latents=T["noisy"] if x is None else x, timestep=T["timestep"], encoder_hidden_states=T["ehs"],
                 encoder_pooler_output=T["pooled"], original_size=T["original_size"],
                 target_size=T["target_size"], crop_coords_top_left=T["crop"])

    model.requires_grad_(True)
    x = T["noisy"].clone().requires_grad_(True)
    with autocast():
        out = call(model, x)
        loss = LD.loss_with_predicted_noise(T["latents"], T["noise"], out)
    loss.backward()
    T["out"] = out.detach().float()
    T["loss"] = loss.detach().float()
    T["d_noisy"] = x.grad
    for name in ("out.2.weight", "input_blocks.blocks.0.0.weight", "middle_block.blocks.1.transformer_blocks.0.attn1.to_q.weight",
                 "output_blocks.blocks.1.1.transformer_blocks.0.ff.net.0.proj.weight", "time_embed.0.weight"):
        T["grad." + name] = dict(model.named_parameters())[name].grad
    model.zero_grad()
    losses, keys = lora_train_steps(
        model, call, ["attn1", "attn2", ".ff."], 4,
        lambda o: LD.loss_with_predicted_noise(T["latents"], T["noise"], o))
    T["lora.losses"] = losses
    meta = {"config": json.dumps(TINY_SDXL), "checksum": repr(cs), "lora_keys": json.dumps(keys),
            "n_state_keys": str(len(shapes_of(den.Denoiser(cfg))))}
    save_file(t2c(T), os.path.join(OUT, fname), metadata=meta)
    print(fname, "loss", float(T["loss"]), "lora losses", losses.tolist(), "n lora keys", len(keys))


def gen_sdxl_rect():
    """Two non-square aspect-ratio buckets of the tiny UNet (latent 16x24 = "tall 128x192 px", 24x16 = its transpose):
    the (h, w) order of every reshape / conv / up- and down-sampling is pinned against the reference."""
    gen_sdxl(hw=(16, 24), fname="sdxl_tiny_h16w24.safetensors", seed=52,
             sizes=dict(original_size=[[128, 192], [200, 260]], target_size=[[128, 192], [128, 192]], crop=[[0, 0], [8, 16]]))
    gen_sdxl(hw=(24, 16), fname="sdxl_tiny_h24w16.safetensors", seed=53,
             sizes=dict(original_size=[[192, 128], [300, 210]], target_size=[[192, 128], [192, 128]], crop=[[0, 0], [0, 24]]))


def gen_sdxl_rope():
    """SDXL with the 2-D RoPE attention adapter trained with the flow-matching objective:
    reference `DenoiserWithRoPE` (src/models/sdxl/adapter/rope.py) on a non-square latent, origin "center" pushed into
    the embedders the way `SDXLWithRoPEModel.apply_rope_config` does, and the step of train/sdxl/flow_match.py:109-174
    (`prepare_scaled_noised_latents`, timestep * 1000 into the denoiser, velocity loss; plus the image-prediction
    branch of `treat_loss`).  The adapter module imports `..pipeline` (VAE -> diffusers, absent here) only for the
    model-holder class, so that one name is stubbed."""
    cfgm = importlib.import_module("src.models.sdxl.config")
    importlib.import_module("src.models.sdxl.denoiser")
    pipe = types.ModuleType("src.models.sdxl.pipeline")
    pipe.SDXLModel = type("SDXLModel", (torch.nn.Module,), {})
    sys.modules["src.models.sdxl.pipeline"] = pipe
    rope = importlib.import_module("src.models.sdxl.adapter.rope")
    from src.modules.loss import flow_match as FM

    hw = (16, 24)
    cfg = rope.DenoiserConfigWithRoPE(**TINY_SDXL)  # rope_dims [32, 32], theta 1e4, origin "center"
    model = rope.DenoiserWithRoPE(cfg)
    for m in model.modules():  # SDXLWithRoPEModel.apply_rope_config, rope.py:553-560
        if isinstance(m, rope.RoPEEmbedder):
            m.rope_dims, m.rope_theta, m.origin_position = cfg.rope_dims, cfg.rope_theta, cfg.origin_position
    model.train()
    cs = load_synth(model, seed=0)
    g = torch.Generator().manual_seed(77)
    B = 2
    T = {
        "latents": torch.randn(B, 4, *hw, generator=g),
        "t": torch.tensor([0.183, 0.764], dtype=torch.float32),  # sample_timestep output, 0..1
        "ehs": torch.randn(B, 11, 64, generator=g),
        "pooled": torch.randn(B, 1280, generator=g),
        "original_size": torch.tensor([[128, 192], [200, 260]]),
        "target_size": torch.tensor([[128, 192], [128, 192]]),
        "crop": torch.tensor([[0, 0], [8, 16]]),
    }
    torch.manual_seed(123)
    noisy, noise = FM.prepare_scaled_noised_latents(T["latents"], T["t"], noise_scale=1.0, clean_at_zero=False)
    T["noisy"], T["noise"] = noisy, noise
    torch.manual_seed(123)
    T["noisy_c0"] = FM.prepare_scaled_noised_latents(T["latents"], T["t"], noise_scale=1.0, clean_at_zero=True)[0]
    emb = next(m for m in model.modules() if isinstance(m, rope.RoPEEmbedder))
    T["freqs.image"] = torch.view_as_real(emb.get_image_freqs(1, hw[0], hw[1], torch.device("cpu"))[0])
    T["freqs.context"] = torch.view_as_real(emb.get_context_freqs(1, 11, torch.device("cpu"))[0])
    T["ids.image"] = emb.get_image_position_ids(1, hw[0], hw[1])[0]
    tl = rope.RoPEEmbedder([32, 32], origin_position="top_left")
    T["ids.image_top_left"] = tl.get_image_position_ids(1, 5, 7)[0]
    qk = torch.randn(2, 2, hw[0] * hw[1], 64, generator=g)
    T["apply.in"] = qk
    T["apply.out"] = rope.apply_rope(qk, emb.get_image_freqs(2, hw[0], hw[1], torch.device("cpu")))

    def call(m, x=None):
        return m(latents=T["noisy"] if x is None else x, timestep=T["t"] * 1000, encoder_hidden_states=T["ehs"],
                 encoder_pooler_output=T["pooled"], original_size=T["original_size"],
                 target_size=T["target_size"], crop_coords_top_left=T["crop"])

    model.requires_grad_(True)
    x = T["noisy"].clone().requires_grad_(True)
    with autocast():
        out = call(model, x)
        loss = FM.loss_with_predicted_velocity(T["latents"], T["noise"], out)
        # treat_loss, model_prediction "image" / loss_type "velocity" (flow_match.py:68-87)
        v_t = FM.convert_x0_to_velocity(T["latents"], T["noisy"], T["t"], eps=1e-5, clean_at_zero=False)
        v_p = FM.convert_x0_to_velocity(out, T["noisy"], T["t"], eps=1e-5, clean_at_zero=False)
        loss_img = torch.nn.functional.mse_loss(v_p, v_t, reduction="mean")
    loss.backward()
    T["out"], T["loss"], T["loss_image_pred"], T["d_noisy"] = out.detach().float(), loss.detach().float(), loss_img.detach().float(), x.grad
    T["x0_velocity_target"] = v_t.detach().float()
    for name in ("out.2.weight", "middle_block.blocks.1.transformer_blocks.0.attn1.to_q.weight",
                 "middle_block.blocks.1.transformer_blocks.0.attn2.to_k.weight",
                 "output_blocks.blocks.1.1.transformer_blocks.0.attn1.to_k.weight"):
        T["grad." + name] = dict(model.named_parameters())[name].grad
    model.zero_grad()
    with torch.no_grad(), autocast():  # rope switched off = the plain denoiser (set_rope_enabled, rope.py:472-478)
        model.set_rope_enabled(False)
        T["out_rope_off"] = call(model).float()
        model.set_rope_enabled(True)
    losses, keys = lora_train_steps(model, call, ["attn1", "attn2", ".ff."], 4,
                                    lambda o: FM.loss_with_predicted_velocity(T["latents"], T["noise"], o))
    T["lora.losses"] = losses
    meta = {"config": json.dumps(TINY_SDXL), "checksum": repr(cs), "lora_keys": json.dumps(keys),
            "rope": json.dumps(dict(rope_dims=cfg.rope_dims, rope_theta=cfg.rope_theta, origin_position=cfg.origin_position))}
    save_file(t2c(T), os.path.join(OUT, "sdxl_rope_tiny.safetensors"), metadata=meta)
    print("sdxl_rope_tiny loss", float(T["loss"]), "image-pred loss", float(T["loss_image_pred"]), "lora losses", losses.tolist(),
          "rope on/off output distance", float((T["out"] - T["out_rope_off"]).norm() / T["out_rope_off"].norm()))


def gen_aura(only=None):
    cfgm = importlib.import_module("src.models.auraflow.config")
    den = importlib.import_module("src.models.auraflow.denoiser")
    from src.modules.loss import flow_match as LF

    for tag, base, extra, hw in (("", TINY_AURA, {}, (16, 24)),
                                 ("_rope", TINY_AURA, {"use_rope": True, "rope_dim_sizes": [16, 24, 24]}, (16, 24)),
                                 ("_d256", TINY_AURA_D256, {}, (48, 56)),  # 672 patches + 9 text + 8 registers = 689 tokens
                                 ("_d256_rope", TINY_AURA_D256, {"use_rope": True, "rope_dim_sizes": [32, 112, 112]}, (48, 56))):
        if only is not None and tag not in only:
            continue
        kw = dict(base, **extra)
        cfg = cfgm.DenoiserConfig(**kw)
        model = den.Denoiser(cfg)
        cs = load_synth(model, seed=1)
        g = torch.Generator().manual_seed(43)
        B = 2
        T = {
            "latents": torch.randn(B, 4, *hw, generator=g),
            "noise": torch.randn(B, 4, *hw, generator=g),
            "timestep": torch.tensor([0.3, 0.81]),
            "ehs": torch.randn(B, 9, 32, generator=g),
        }
        t4 = T["timestep"].view(B, 1, 1, 1)
        T["noisy"] = (1 - t4) * T["latents"] + t4 * T["noise"]

        def call(m, x=None):
            with contextlib.redirect_stdout(io.StringIO()):
                import warnings

                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    return m(latent=T["noisy"] if x is None else x, encoder_hidden_states=T["ehs"],
                             timestep=T["timestep"])

        model.requires_grad_(True)
        x = T["noisy"].clone().requires_grad_(True)
        with autocast():
            out = call(model, x)
            loss = LF.loss_with_predicted_velocity(T["latents"], T["noise"], out)
        loss.backward()
        T["out"] = out.detach().float()
        T["loss"] = loss.detach().float()
        T["d_noisy"] = x.grad
        big = "d256" in tag  # keep the wide fixtures small: the MLP / modulation weight gradients are 3-6 MB each there
        for name in ("final_linear.weight", "init_x_linear.weight", "double_layers.0.attn.w2q.weight",
                     "single_layers.1.attn.w1k.weight", "register_tokens") if big else (
                     "final_linear.weight", "init_x_linear.weight", "double_layers.0.attn.w2q.weight",
                     "single_layers.1.mlp.c_fc1.weight", "single_layers.0.modCX.1.weight", "register_tokens"):
            T["grad." + name] = dict(model.named_parameters())[name].grad
        model.zero_grad()
        losses, keys = lora_train_steps(
            model, call, [".attn.", ".mlp"], 4,
            lambda o: LF.loss_with_predicted_velocity(T["latents"], T["noise"], o))
        T["lora.losses"] = losses
        # BASELINE config 3 trains rank 16: same model, fresh instance, rank-16 adapters
        model16 = den.Denoiser(cfg)
        load_synth(model16, seed=1)
        losses16, keys16 = lora_train_steps(
            model16, call, [".attn.", ".mlp"], 16,
            lambda o: LF.loss_with_predicted_velocity(T["latents"], T["noise"], o))
        T["lora16.losses"] = losses16
        meta = {"config": json.dumps(kw), "checksum": repr(cs), "lora_keys": json.dumps(keys)}
        save_file(t2c(T), os.path.join(OUT, f"aura_tiny{tag}.safetensors"), metadata=meta)
        print(f"aura{tag}: loss", float(T["loss"]), "lora losses", losses.tolist(), len(keys), "rank16", losses16.tolist())


def gen_lumina(kw=None, fname="lumina_tiny.safetensors", hw=(16, 24), cap=(12, (12, 7))):
    cfgm = importlib.import_module("src.models.lumina2.config")
    den = importlib.import_module("src.models.lumina2.denoiser")
    from src.modules.loss import flow_match as LF

    kw = TINY_LUMINA if kw is None else kw
    cfg = cfgm.DenoiserConfig(**kw)
    model = den.Denoiser(cfg)
    cs = load_synth(model, seed=2)
    g = torch.Generator().manual_seed(44)
    B = 2
    T = {
        "latents": torch.randn(B, 8, *hw, generator=g),
        "noise": torch.randn(B, 8, *hw, generator=g),
        "timestep": torch.tensor([0.2, 0.77]),
        "caption": torch.randn(B, cap[0], 48, generator=g),
        "caption_mask": (torch.arange(cap[0])[None, :] < torch.tensor(list(cap[1]))[:, None]),
    }
    t4 = (1 - T["timestep"]).view(B, 1, 1, 1)  # train/lumina2/text_to_image.py:109-121 inverts time
    T["noisy"] = (1 - t4) * T["latents"] + t4 * T["noise"]

    def call(m, x=None):
        out, _, _ = m(latents=T["noisy"] if x is None else x, timestep=T["timestep"],
                      caption_features=T["caption"], caption_mask=T["caption_mask"])
        return -torch.nested.to_padded_tensor(out, padding=0.0)

    model.requires_grad_(True)
    model.train()
    x = T["noisy"].clone().requires_grad_(True)
    with autocast():
        out = call(model, x)
        loss = LF.loss_with_predicted_velocity(T["latents"], T["noise"], out)
    loss.backward()
    T["out"] = out.detach().float()
    T["loss"] = loss.detach().float()
    T["d_noisy"] = x.grad
    for name in ("final_layer.linear.weight", "x_embedder.weight", "layers.0.attention.qkv.weight",
                 "layers.1.feed_forward.w2.weight", "noise_refiner.0.adaLN_modulation.1.bias",
                 "context_refiner.0.attention.q_norm.weight", "cap_embedder.1.weight"):
        if kw is not TINY_LUMINA and "feed_forward" in name:
            continue  # 3.5 MB at the wide geometry; the attention projections are what that fixture is for
        T["grad." + name] = dict(model.named_parameters())[name].grad
    model.zero_grad()

    # ragged batch: two different resolutions, list input (forward only)
    rag = [torch.randn(8, *hw, generator=g), torch.randn(8, hw[0] // 2, hw[1] - 8, generator=g)]
    with autocast(), torch.no_grad():
        o2, _, capf = model(latents=rag, timestep=T["timestep"], caption_features=T["caption"],
                            caption_mask=T["caption_mask"])
    T["ragged.in0"], T["ragged.in1"] = rag
    T["ragged.out0"], T["ragged.out1"] = [t.float().contiguous() for t in o2.unbind()]
    T["ragged.caption_features"] = capf.float()

    losses, keys = lora_train_steps(
        model, call, ["attention", "feed_forward"], 4,
        lambda o: LF.loss_with_predicted_velocity(T["latents"], T["noise"], o))
    T["lora.losses"] = losses
    meta = {"config": json.dumps(kw), "checksum": repr(cs), "lora_keys": json.dumps(keys)}
    save_file(t2c(T), os.path.join(OUT, fname), metadata=meta)
    print(fname, "loss", float(T["loss"]), "lora losses", losses.tolist(), len(keys))


# ----------------------------------------------------------------------------------------------
def gen_samplers():
    """Train-time timestep samplers (src/modules/timestep/sampling.py, src/models/lumina2/scheduler.py:48-72).  The
    raw draws are stored next to the outputs (never rely on RNG stream equality): the reference functions
    call torch.randn / torch.rand / torch.randint / np.random.choice exactly once, so re-seeding and making the same call
    reproduces the draw they consumed."""
    from src.modules.timestep import sampling as S

    sched = importlib.import_module("src.models.lumina2.scheduler")
    dev = torch.device("cpu")
    T = {}
    B = 16
    shapes = {"sq": (B, 16, 128, 128), "tall": (B, 16, 192, 80), "small": (B, 4, 32, 48)}

    def draw(kind, seed):
        torch.manual_seed(seed)
        return {"randn": lambda: torch.randn(B, device=dev), "rand": lambda: torch.rand(B, device=dev),
                "randint": lambda: torch.randint(low=0, high=1000, size=(B,), device=dev, dtype=torch.int)}[kind]()

    def run(name, kind, seed, fn):
        T[f"{name}.draw"] = draw(kind, seed)
        torch.manual_seed(seed)
        T[f"{name}.out"] = fn()

    run("sigmoid_randn", "randn", 1, lambda: S.sigmoid_randn(torch.Size(shapes["sq"]), dev))
    run("sigmoid_randn_s2", "randn", 2, lambda: S.sigmoid_randn(torch.Size(shapes["sq"]), dev, sigmoid_scale=2.0))
    run("shift_sigmoid_randn", "randn", 3, lambda: S.shift_sigmoid_randn(torch.Size(shapes["sq"]), dev))
    run("flux_shift_randn_sq", "randn", 4, lambda: S.flux_shift_randn(torch.Size(shapes["sq"]), dev))
    run("flux_shift_randn_small", "randn", 5, lambda: S.flux_shift_randn(torch.Size(shapes["small"]), dev))
    run("scale_shift_sigmoid_randn", "randn", 6, lambda: S.scale_shift_sigmoid_randn(torch.Size(shapes["sq"]), dev))
    run("uniform_rand", "rand", 7, lambda: S.uniform_rand(torch.Size(shapes["sq"]), dev))
    run("shift_uniform_rand", "rand", 8, lambda: S.shift_uniform_rand(torch.Size(shapes["sq"]), dev))
    run("uniform_randint", "randint", 9, lambda: S.uniform_randint(torch.Size(shapes["sq"]), dev))
    for tag in ("sq", "tall", "small"):
        run(f"lumina_lognorm_{tag}", "randn", 10, lambda: sched.Scheduler().sample_sigmoid_randn(torch.Size(shapes[tag]), dev, patch_size=2))
    for name, typ, kw in [("dispatch_sigmoid", "sigmoid", {}), ("dispatch_shift", "shift_sigmoid", {"discrete_flow_shift": 2.0}),
                          ("dispatch_flux", "flux_shift", {})]:
        run(name, "randn", 11, lambda: S.sample_timestep(torch.Size(shapes["sq"]), dev, typ, **kw))
    # fraction samplers: numpy global RNG (sampling.py:137); the candidate table and the picks are stored
    div = [20, 25, 30, 32]
    T["fraction.table"] = torch.from_numpy(S._create_fraction(div).copy())
    np.random.seed(12)
    picks = np.random.choice(S._create_fraction(div), size=B, replace=True)
    T["fraction.draw"] = torch.from_numpy(picks.copy())
    np.random.seed(12)
    T["fraction.out"] = S.fraction_uniform_rand(torch.Size(shapes["sq"]), dev, divisible=div)
    np.random.seed(12)
    T["shift_fraction.out"] = S.shift_fraction_uniform_rand(torch.Size(shapes["sq"]), dev, shift=6.0, divisible=div)
    T["fraction.table_default"] = torch.from_numpy(S._create_fraction(list(range(20, 30))).copy())
    save_file(t2c(T), os.path.join(OUT, "samplers.safetensors"),
              metadata={"shapes": json.dumps(shapes), "divisible": json.dumps(div), "batch": str(B)})
    print("samplers:", len(T), "tensors")


# ----------------------------------------------------------------------------------------------
def gen_checkpoint_keys():
    """Checkpoint key renaming of the three pipelines (src/models/sdxl/util.py, lumina2/util.py, auraflow/pipeline.py:35-54):
    original single-file names <-> module-tree names, VAE / text-encoder prefixes included, and the ComfyUI adapter names.
    The original-side key lists below are data (names found in sgm / ComfyUI checkpoints, the ones the reference's
    tests/models/test_sdxl.py:21-151 uses plus a few per family); every expected value comes from the reference."""
    import importlib.util

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    su = by_path("_ref_sdxl_util", "src/models/sdxl/util.py")
    lu = by_path("_ref_lumina_util", "src/models/lumina2/util.py")
    sdxl_orig = [
        "model.diffusion_model.input_blocks.5.1.transformer_blocks.1.attn1.to_out.0.weight",
        "model.diffusion_model.input_blocks.3.0.op.bias",
        "model.diffusion_model.input_blocks.0.0.weight",
        "model.diffusion_model.output_blocks.7.0.in_layers.0.weight",
        "model.diffusion_model.output_blocks.8.0.skip_connection.weight",
        "model.diffusion_model.output_blocks.2.2.conv.weight",
        "model.diffusion_model.middle_block.1.transformer_blocks.0.attn1.to_k.weight",
        "model.diffusion_model.middle_block.0.emb_layers.1.bias",
        "model.diffusion_model.time_embed.0.weight",
        "model.diffusion_model.label_emb.0.2.bias",
        "model.diffusion_model.out.2.weight",
        "diffusion_model.input_blocks.4.1.proj_in.weight",
        "conditioner.embedders.0.transformer.text_model.embeddings.position_embedding.weight",
        "conditioner.embedders.1.model.transformer.resblocks.0.attn.in_proj_bias",
        "conditioner.embedders.1.model.text_projection",
        "first_stage_model.encoder.down.1.block.0.conv1.weight",
        "first_stage_model.encoder.down.2.block.0.nin_shortcut.weight",
        "first_stage_model.encoder.down.0.downsample.conv.weight",
        "first_stage_model.decoder.mid.attn_1.q.weight",
        "first_stage_model.decoder.mid.attn_1.proj_out.bias",
        "first_stage_model.decoder.mid.block_2.norm1.weight",
        "first_stage_model.decoder.up.0.block.0.nin_shortcut.weight",
        "first_stage_model.decoder.up.3.upsample.conv.weight",
        "first_stage_model.post_quant_conv",
        "first_stage_model.encoder.norm_out.weight",
        "first_stage_model.decoder.conv_in.weight",
    ]
    unet_orig = ["input_blocks.5.1.transformer_blocks.1.attn1.to_out.0.weight", "input_blocks.3.0.op.bias",
                 "output_blocks.7.0.in_layers.0.weight", "output_blocks.8.0.skip_connection.weight",
                 "middle_block.1.transformer_blocks.0.attn1.to_k.weight", "out.2.bias", "time_embed.2.weight"]
    vae_orig = ["encoder.down.1.block.0.conv1.weight", "encoder.down.2.block.0.nin_shortcut.weight", "decoder.mid.attn_1.q.weight",
                "decoder.mid.attn_1.k.bias", "decoder.mid.attn_1.norm.weight", "decoder.mid.block_1.conv2.bias",
                "decoder.up.0.block.0.nin_shortcut.weight", "decoder.up.2.block.1.norm2.weight", "post_quant_conv",
                "encoder.norm_out.weight", "decoder.conv_in.weight", "decoder.up.1.upsample.conv.bias",
                "encoder.down.3.block.1.conv2.weight"]
    J = {"sdxl": {
        "from_original": [[k, su.convert_from_original_key(k)] for k in sdxl_orig],
        "to_original": [[su.convert_from_original_key(k), su.convert_to_original_key(su.convert_from_original_key(k))] for k in sdxl_orig],
        "unet_from_original": [[k, su.unet_block_convert_from_original_key(k)] for k in unet_orig],
        "unet_to_original": [[su.unet_block_convert_from_original_key(k),
                              su.unet_block_convert_to_original_key(su.unet_block_convert_from_original_key(k))] for k in unet_orig],
        "vae_from_original": [[k, su.vae_convert_from_original_key(k)] for k in vae_orig],
        "vae_to_original": [[su.vae_convert_from_original_key(k),
                             su.vae_convert_to_original_key(su.vae_convert_from_original_key(k))] for k in vae_orig],
        "comfy": [[k, su.convert_to_comfy_key(k)] for k in [su.convert_from_original_key(k) for k in sdxl_orig]],
    }}
    lumina_orig = ["model.diffusion_model.layers.3.attention.qkv.weight", "model.diffusion_model.noise_refiner.0.adaLN_modulation.1.bias",
                   "model.diffusion_model.cap_embedder.1.weight", "diffusion_model.final_layer.linear.weight",
                   "model.diffusion_model.layers.0.attention.qkv.weight.quant_state.bitsandbytes__nf4",
                   "text_encoders.gemma2_2b.transformer.model.layers.0.mlp.up_proj.weight",
                   "vae.decoder.mid.attn_1.q.weight", "vae.encoder.down.0.block.1.conv1.weight", "vae.decoder.up.1.upsample.conv.bias"]
    J["lumina2"] = {
        "from_original": [[k, lu.convert_from_original_key(k)] for k in lumina_orig],
        "to_original": [[lu.convert_from_original_key(k), lu.convert_to_original_key(lu.convert_from_original_key(k))] for k in lumina_orig],
        "comfy": [[k, lu.convert_to_comfy_key(k)] for k in [lu.convert_from_original_key(k) for k in lumina_orig]],
    }
    # AuraFlow's converters live in pipeline.py (which imports diffusers); they are three module-level functions over
    # three prefix constants, evaluated here in a namespace that holds only those constants
    src = open(os.path.join(REF, "src/models/auraflow/pipeline.py")).read()
    import ast

    tree = ast.parse(src)
    ns = {"DENOISER_TENSOR_PREFIX": "model.", "VAE_TENSOR_PREFIX": "vae.",
          "TEXT_ENCODER_TENSOR_PREFIX": "text_encoders.pile_t5xl.transformer."}
    for const, rel in (("DENOISER_TENSOR_PREFIX", "src/models/auraflow/denoiser.py"), ("VAE_TENSOR_PREFIX", "src/models/auraflow/vae.py"),
                       ("TEXT_ENCODER_TENSOR_PREFIX", "src/models/auraflow/text_encoder.py")):
        for node in ast.parse(open(os.path.join(REF, rel)).read()).body:
            if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == const:
                assert ast.literal_eval(node.value) == ns[const], (const, ast.literal_eval(node.value))
    funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in
             ("convert_to_original_key", "convert_to_comfy_key", "convert_from_original_key")]
    exec(compile(ast.Module(body=funcs, type_ignores=[]), "auraflow_pipeline_converters", "exec"), ns)
    aura_orig = ["model.double_layers.0.attn.w2q.weight", "model.single_layers.11.mlp.c_fc1.weight", "model.register_tokens",
                 "model.t_embedder.mlp.0.weight", "diffusion_model.single_layers.3.modCX.1.weight", "model.final_linear.weight",
                 "vae.decoder.conv_in.weight", "text_encoders.pile_t5xl.transformer.encoder.block.0.layer.0.SelfAttention.q.weight"]
    J["auraflow"] = {
        "from_original": [[k, ns["convert_from_original_key"](k)] for k in aura_orig],
        "to_original": [[ns["convert_from_original_key"](k), ns["convert_to_original_key"](ns["convert_from_original_key"](k))] for k in aura_orig],
        "comfy": [[k, ns["convert_to_comfy_key"](k)] for k in [ns["convert_from_original_key"](k) for k in aura_orig]],
        "prefixes": {k: v for k, v in ns.items() if k.endswith("_PREFIX")},
    }
    with open(os.path.join(OUT, "checkpoint_keys.json"), "w") as f:
        json.dump(J, f, indent=0)
    print("checkpoint_keys:", {k: {kk: len(vv) for kk, vv in v.items()} for k, v in J.items()})


# ----------------------------------------------------------------------------------------------
def gen_index():
    import importlib.util

    def load_by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    # bucket.py needs PIL (present) + torchvision stubs; util.py needs `datasets` (present)
    import datasets  # noqa: F401
    from src.modules.peft import PeftTargetConfig, get_adapter_parameters  # noqa: F401 (before stubbing)

    for stub in ("torchvision", "torchvision.transforms", "torchvision.transforms.v2",
                 "torchvision.transforms.functional", "imagesize"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].v2 = sys.modules["torchvision.transforms.v2"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    load_by_path("src.dataset.util", "src/dataset/util.py")
    load_by_path("src.dataset.bucket", "src/dataset/bucket.py")
    arb = load_by_path("src.dataset.aspect_ratio_bucket", "src/dataset/aspect_ratio_bucket.py")
    bucket_mod = sys.modules["src.dataset.bucket"]
    from src.utils.state_dict import RegexMatch, get_target_keys

    J = {}
    J["buckets"] = {}
    for name, kw in {
        "b1024_s64_m384": dict(target_area=1024 * 1024, start_size=1024, step=64, min_size=384),
        "b1024_s128_m384": dict(target_area=1024 * 1024, start_size=1024, step=128, min_size=384),
        "b512_s64_m256": dict(target_area=512 * 512, start_size=512, step=64, min_size=256),
        "default": dict(),
    }.items():
        J["buckets"][name] = {"kwargs": kw, "rows": arb.generate_buckets(**kw).tolist()}
    mgr = arb.AspectRatioBucketManager(arb.generate_buckets(1024 * 1024, 1024, 64, 384))
    sizes = [(1920, 1080), (1080, 1920), (1024, 1024), (4000, 3000), (1500, 1500), (832, 1216), (1216, 832),
             (2048, 512), (640, 1536), (1536, 640), (1111, 999), (3000, 1000), (1000, 3000), (1024, 768)]
    J["find_nearest"] = {"bucket_set": "b1024_s64_m384",
                         "cases": [[w, h, int(mgr.find_nearest(w, h))] for w, h in sizes]}
    mgr2 = arb.AspectRatioBucketManager(arb.generate_buckets(1024 * 1024, 1024, 128, 384))
    J["find_nearest_128"] = {"bucket_set": "b1024_s128_m384",
                             "cases": [[w, h, int(mgr2.find_nearest(w, h))] for w, h in sizes]}

    class _B(bucket_mod.Bucket):
        pass

    batches = {}
    for n, bs, rep in [(10, 4, 1), (3, 8, 1), (7, 2, 3), (16, 4, 1)]:
        b = _B(np.arange(n), batch_size=bs, num_repeats=rep)
        ds = bucket_mod.BucketDataset(b)
        batches[f"{n}_{bs}_{rep}"] = [np.asarray(ds[i]).tolist() for i in range(len(ds))]
    J["bucket_batches"] = batches

    # collate (dataloader.concatnate_collate_fn): tensors concatenated on dim 0, lists summed
    from src.dataloader import concatnate_collate_fn

    col = concatnate_collate_fn([{"a": torch.arange(6).view(2, 3), "c": ["x", "y"]},
                                 {"a": torch.arange(6, 9).view(1, 3), "c": ["z"]}])
    J["collate"] = {"a": col["a"].tolist(), "c": col["c"]}

    keys = ["a.attn1.to_q", "a.attn2.to_k", "b.ff.net.0.proj", "b.ffx", "single_layers.3.attn.w1q",
            "single_layers.3.w1q", "double_layers.0.attn.w2q", "noise_refiner.0.attention.qkv", "layers.1.attention.out",
            "layers.1.feed_forward.w1", "x.norm"]
    cases = []
    for inc, exc in [
        (["attn1", "attn2", ".ff."], []),
        (["attention", "feed_forward"], ["noise_refiner"]),
        ([{"regex": r".*single_layers\.\d+\.w1[qkvo]"}], []),
        ([{"regex": r".*single_layers\.\d+\.attn\.w1[qkvo]"}, "double"], [{"regex": r".*w2q"}]),
        (["a"], ["attn2", {"regex": r"layers\.1\..*w1$"}]),
    ]:
        conv = lambda L: [RegexMatch(regex=p["regex"]) if isinstance(p, dict) else p for p in L]  # noqa: E731
        cases.append({"include": inc, "exclude": exc, "result": sorted(get_target_keys(conv(inc), conv(exc), keys))})
    J["target_keys"] = {"keys": keys, "cases": cases}

    # lumina position ids + aura PE index
    lum = importlib.import_module("src.models.lumina2.denoiser")
    lcfg = importlib.import_module("src.models.lumina2.config")
    m = lum.Denoiser(lcfg.DenoiserConfig(**TINY_LUMINA))
    J["position_ids"] = [{"args": [c, h, w], "ids": m.get_position_ids(c, h, w).tolist()}
                         for c, h, w in [(0, 2, 3), (5, 4, 4), (12, 8, 12), (1, 1, 1)]]
    aura = importlib.import_module("src.models.auraflow.denoiser")
    acfg = importlib.import_module("src.models.auraflow.config")
    am = aura.Denoiser(acfg.DenoiserConfig(**TINY_AURA))
    J["aura_pe_index"] = [{"hw": [h, w], "idx": am.pe_selection_index_based_on_dim(h, w).tolist()}
                          for h, w in [(16, 24), (32, 32), (8, 8), (30, 14)]]
    x = torch.arange(2 * 4 * 4 * 6, dtype=torch.float32).view(2, 4, 4, 6)
    J["aura_patchify"] = am.patchify(x).tolist()
    J["aura_unpatchify"] = am.unpatchify(am.patchify(x), 2, 3).tolist()
    xl = torch.arange(8 * 4 * 6, dtype=torch.float32).view(8, 4, 6)
    J["lumina_patchify"] = m.patchify(xl).tolist()
    J["lumina_unpatchify"] = m.unpatchify(m.patchify(xl), 4, 6, 2, 8).tolist()

    # full-size LoRA target lists on meta + adapter key names
    from src.modules.peft import PeftTargetConfig, get_adapter_parameters
    from src.modules.peft.lora import LoRAConfig

    def targets(model, include, exclude, rank):
        names = [n for n, _ in model.named_modules()]
        conv = lambda L: [RegexMatch(regex=p["regex"]) if isinstance(p, dict) else p for p in L]  # noqa: E731
        PeftTargetConfig(include_keys=conv(include), exclude_keys=conv(exclude),
                         config=LoRAConfig(rank=rank)).replace_to_peft_layer(model)
        wrapped = sorted(n for n, mod in model.named_modules() if hasattr(mod, "lora_down"))
        n_lora = sum(p.numel() for n, p in model.named_parameters() if "lora_" in n)
        return {"include": include, "exclude": exclude, "rank": rank, "n_modules": len(names),
                "wrapped": wrapped, "lora_params": n_lora,
                "adapter_keys_head": sorted(get_adapter_parameters(model).keys())[:12]}

    full = {}
    with torch.device("meta"):
        sd = importlib.import_module("src.models.sdxl.denoiser")
        sc = importlib.import_module("src.models.sdxl.config")
        full["sdxl_r4"] = targets(sd.Denoiser(sc.DenoiserConfig()), ["attn1", "attn2", ".ff."], [], 4)
        # configs/auraflow/lora.yml:19-24 (the single_layers regex never matches: no `.attn.`)
        full["aura_yaml_r16"] = targets(
            aura.Denoiser(acfg.DenoiserConfig()),
            ["attn.w2", {"regex": r".*single_layers\.\d+\.w1[qkvo]"}, ".mlp.", ".mlpX."],
            ["text_encoder", "vae", "t_embedder", "final_linear", ".mod"], 16)
        # tests/assets/debug_dataset.yml:17-25
        full["aura_debug_r16"] = targets(
            aura.Denoiser(acfg.DenoiserConfig()),
            [{"regex": r".*\.attn\.w2[qkvo]"}, {"regex": r".*\.mlp[X]?\."},
             {"regex": r".*single_layers\.\d+\.attn\.w1[qkvo]"}],
            ["text_encoder", "vae", "t_embedder", "final_linear", {"regex": r"\.mod[CX]{1,2}"}], 16)
        full["lumina_r4"] = targets(lum.Denoiser(lcfg.DenoiserConfig()), ["attention", "feed_forward"],
                                    ["text_encoder", "vae", "noise_refiner"], 4)
    J["full_size_targets"] = full

    # SDXL checkpoint / ComfyUI key renaming (src/models/sdxl/util.py)
    u = importlib.import_module("src.models.sdxl.util")
    keys = ["denoiser.input_blocks.blocks.4.1.transformer_blocks.0.attn1.to_q.lora_down.weight",
            "denoiser.middle_block.blocks.1.transformer_blocks.9.ff.net.0.proj.lora_up.weight",
            "denoiser.output_blocks.blocks.3.1.proj_in.alpha", "denoiser.time_embed.0.weight",
            "text_encoder.text_encoder_1.x.y", "text_encoder.text_encoder_2.a", "vae.decoder.z"]
    orig = ["input_blocks.1.0.in_layers.0.weight", "middle_block.1.norm.bias", "output_blocks.8.0.skip_connection.weight",
            "out.2.bias"]
    J["sdxl_keys"] = {
        "comfy": [[k, u.convert_to_comfy_key(k)] for k in keys],
        "from_original": [[k, u.denoiser_convert_from_original_key(k)] for k in orig],
        "to_original": [[u.denoiser_convert_from_original_key(k),
                         u.denoiser_convert_to_original_key(u.denoiser_convert_from_original_key(k))] for k in orig],
    }
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(J, f)
    print("index: sdxl wrapped", len(full["sdxl_r4"]["wrapped"]), full["sdxl_r4"]["lora_params"],
          "aura", len(full["aura_yaml_r16"]["wrapped"]), full["aura_yaml_r16"]["lora_params"],
          "aura-debug", len(full["aura_debug_r16"]["wrapped"]), full["aura_debug_r16"]["lora_params"],
          "lumina", len(full["lumina_r4"]["wrapped"]), full["lumina_r4"]["lora_params"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    import_reference()
    which = sys.argv[1:] or ["ops", "index", "sdxl", "sdxl_rect", "sdxl_rope", "aura", "lumina", "lumina_d96", "lora_conv", "samplers", "checkpoint_keys"]
    torch.set_num_threads(8)
    for w in which:
        {"ops": gen_ops, "index": gen_index, "sdxl": gen_sdxl, "sdxl_rect": gen_sdxl_rect, "sdxl_rope": gen_sdxl_rope, "aura": gen_aura, "lumina": gen_lumina, "lora_conv": gen_lora_conv,
         "aura_d256": lambda: gen_aura(only=("_d256", "_d256_rope")),
         "lumina_d96": lambda: gen_lumina(TINY_LUMINA_D96, "lumina_tiny_d96.safetensors", hw=(32, 48), cap=(40, (40, 23))),
         "samplers": gen_samplers, "checkpoint_keys": gen_checkpoint_keys}[w]()
