#!/usr/bin/env python
"""bench.py — training images/sec of the denoiser step on synthetic bucketed latents (MI355X).

  python bench.py --gpus N --steps K --warmup W [--workload sdxl_lora|sdxl_full|aura_lora|lumina_lora] [--batch B]

For N>1 either launch one rank per GPU with torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the
environment) or just run `python bench.py --gpus N`: with WORLD_SIZE unset the script starts its own N ranks as CHILD
processes (torch.distributed.run, rendezvous on 127.0.0.1) before anything touches the GPU and exits with their code.
Gradients are exchanged with RCCL on a side stream (vision_ft_amd/trainer/dp.py).
One "step" = noising + denoiser forward + loss + backward + gradient exchange + AdamW update on one
batch of B synthetic latents per GPU (weak scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# NB: MIOpen's default (exhaustive) find is kept on purpose: it costs ~1-2 min in the first warm-up step on
# a fresh box, but MIOPEN_FIND_MODE=FAST picks convolution kernels that make the whole SDXL step 1.34x slower.

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM_GBS = 8000.0

# algorithmic train TFLOP per image at 1024^2 (recompute not counted)
F_ALG = {"sdxl_lora": 13.5, "sdxl_full": 20.3, "aura_lora": 88.0, "lumina_lora": 47.6, "lumina_qlora": 47.6,
         "sdxl_buckets": 13.5}


def build(workload: str, device: torch.device, use_rope: bool = False, aura_keys: str = "debug252"):
    from vision_ft_amd.modules.peft import LoRAConfig, PeftTargetConfig, RegexMatch

    torch.manual_seed(42)
    with torch.device(device):
        if workload.startswith("sdxl"):
            from vision_ft_amd.models.sdxl import Denoiser, DenoiserConfig

            model = Denoiser(DenoiserConfig())
            peft = PeftTargetConfig(include_keys=["attn1", "attn2", ".ff."], config=LoRAConfig(rank=4, alpha=1.0))
        elif workload.startswith("aura"):
            from vision_ft_amd.models.auraflow import Denoiser, DenoiserConfig

            # learned positional embedding, and a second run with use_rope (rope_dim_sizes [32,112,112])
            model = Denoiser(DenoiserConfig(use_rope=True, rope_dim_sizes=[32, 112, 112]) if use_rope else DenoiserConfig())
            if aura_keys == "yaml124":
                # the reference's shipped configs/auraflow/lora.yml:19-24 (its single_layers regex never matches: 124 Linears)
                peft = PeftTargetConfig(
                    include_keys=["attn.w2", RegexMatch(regex=r".*single_layers\.\d+\.w1[qkvo]"), ".mlp.", ".mlpX."],
                    exclude_keys=["text_encoder", "vae", "t_embedder", "final_linear", ".mod"],
                    config=LoRAConfig(rank=16, alpha=1.0))
            else:
                # tests/assets/debug_dataset.yml:17-25 key set of the reference (252 Linears, rank 16)
                peft = PeftTargetConfig(
                    include_keys=[RegexMatch(regex=r".*\.attn\.w2[qkvo]"), RegexMatch(regex=r".*\.mlp[X]?\."),
                                  RegexMatch(regex=r".*single_layers\.\d+\.attn\.w1[qkvo]")],
                    exclude_keys=["text_encoder", "vae", "t_embedder", "final_linear", RegexMatch(regex=r"\.mod[CX]{1,2}")],
                    config=LoRAConfig(rank=16, alpha=1.0))
            with torch.no_grad():  # the reference zero-inits these; re-init so outputs are non-trivial
                for n, p in model.named_parameters():
                    if ".mod" in n or n in ("final_linear.weight", "cond_seq_linear.weight"):
                        p.normal_(0, 0.02)
        else:
            from vision_ft_amd.models.lumina2 import Denoiser, DenoiserConfig

            model = Denoiser(DenoiserConfig())
            peft = PeftTargetConfig(include_keys=["attention", "feed_forward"],
                                    exclude_keys=["text_encoder", "vae", "noise_refiner"], config=LoRAConfig(rank=4, alpha=1.0))
    model = model.to(torch.bfloat16)
    if workload == "lumina_qlora":
        # BASELINE config 4: the LoRA target Linears' base weights stored NF4 (blocksize 64, double-quantised
        # absmax, bitsandbytes key layout), dequantised inside the GEMM kernel; LoRA wraps the NF4 layers
        from vision_ft_amd.modules.quant.functional import quantize_state_dict, replace_by_prequantized_weights

        sd = {k: v for k, v in model.state_dict().items()}
        keys = [k for k in sd if k.endswith(".weight") and sd[k].dim() == 2 and ("attention" in k or "feed_forward" in k)
                and "noise_refiner" not in k]
        q = quantize_state_dict({k: sd[k] for k in keys}, "bnb_nf4", include_keys=keys, double_quant=True)
        replace_by_prequantized_weights(model, q)
        model.load_state_dict({**{k: v for k, v in sd.items() if k not in keys}, **q}, strict=True)
        model = model.to(device)
    if workload.endswith("lora") or workload == "sdxl_buckets":
        model.requires_grad_(False)
        peft.replace_to_peft_layer(model)
    model.train()
    return model


def make_step(workload: str, model, device, batch: int, seed: int, wh: tuple[int, int] = (1024, 1024),
              drawn_lengths: bool = False):
    from vision_ft_amd.modules.loss import diffusion as LD
    from vision_ft_amd.modules.loss import flow_match as LF
    from vision_ft_amd.modules.timestep.sampling import sigmoid_randn, uniform_rand, uniform_randint

    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
    if workload.startswith("sdxl"):
        lat, ctx, pooled = rn(batch, 4, wh[1] // 8, wh[0] // 8), rn(batch, 77, 2048), rn(batch, 1280)
        size = torch.tensor([[wh[1], wh[0]]] * batch, device=device)
        crop = torch.zeros(batch, 2, dtype=torch.long, device=device)

        def fwd():
            t = uniform_randint(lat.shape, device)
            noisy, noise = LD.prepare_noised_latents(lat, t)
            pred = model(latents=noisy, timestep=t, encoder_hidden_states=ctx, encoder_pooler_output=pooled,
                         original_size=size, target_size=size, crop_coords_top_left=crop)
            return LD.loss_with_predicted_noise(lat, noise, pred.float())
    elif workload.startswith("aura"):
        lat, ctx = rn(batch, 4, 128, 128), rn(batch, 256, 2048)

        def fwd():
            t = sigmoid_randn(lat.shape, device)
            noisy, noise = LF.prepare_noised_latents(lat, t.to(lat.dtype))
            pred = model(latent=noisy, encoder_hidden_states=ctx, timestep=t)
            return LF.loss_with_predicted_velocity(lat.float(), noise.float(), pred.float())
    else:
        lat, cap = rn(batch, 16, 128, 128), rn(batch, 256, 2304)
        full = torch.ones(batch, 256, dtype=torch.bool, device=device)
        pos = torch.arange(256, device=device)[None, :]

        def fwd():
            t = uniform_rand(lat.shape, device)
            noisy, noise = LF.prepare_noised_latents(lat, (1 - t).to(lat.dtype))
            # caption lengths as host integers (a data loader knows them before the batch goes to the device): without them the
            # model reads them back from the mask — the step's only host synchronisation, which also rules out graph replay
            if drawn_lengths:
                # mask lengths U{32..256}, a new ragged layout every step.  Round 4: drawn ON THE DEVICE
                # (graph-safe Philox stream of the default generator) and fed to the shape-static forward — the lengths never
                # reach the host, one HIP graph serves every combination (rounds 2-3: drawn on the host, eager launches)
                lens_d = torch.randint(32, 257, (batch,), device=device)
                mask = pos < lens_d[:, None]
                out, _, _ = model(latents=noisy, timestep=t, caption_features=cap, caption_mask=mask, static_layout=True)
            else:
                lens, mask = [256] * batch, full
                out, _, _ = model(latents=noisy, timestep=t, caption_features=cap, caption_mask=mask, caption_lengths=lens,
                                  padded_output=True)
            pred = -out
            return LF.loss_with_predicted_velocity(lat.float(), noise.float(), pred.float())
    return fwd


def _nf4_cache_on() -> bool:
    from vision_ft_amd.modules.quant import nf4

    return bool(nf4.NF4_COMPUTE_CACHE)


def physical_cores() -> int:
    """Physical host cores available to this process (the CPU leg uses all of them)."""
    try:
        import psutil

        n = psutil.cpu_count(logical=False) or 0
    except Exception:
        n = 0
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(n or avail, avail))


def cpu_baseline(threads: int) -> dict:
    from oracle.bench_cpu import time_sdxl_lora_step
    from vision_ft_amd.models.sdxl import Denoiser, DenoiserConfig
    from vision_ft_amd.utils.state_dict import get_target_keys

    with torch.device("meta"):
        m = Denoiser(DenoiserConfig())
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    linears = [n for n, mod in m.named_modules() if isinstance(mod, torch.nn.Linear)]
    targets = sorted(set(get_target_keys(["attn1", "attn2", ".ff."], [], [n for n, _ in m.named_modules()])) & set(linears))
    # best of a small thread sweep (a B = 1 step over 1 024 tokens does not scale to every core of the host: on 128 threads it
    # ran 23.4 s/step in round 4 against 10.8 s on the survey container's 8) — the count that won is the one reported
    sweep = sorted({n for n in (16, 32, 64, threads) if n <= threads}) or [threads]
    r = time_sdxl_lora_step(shapes, targets, threads=sweep, warmup=1, steps=2)
    return {"value": round(r["images_per_s"], 5), "unit": "images/s", "cores": r["threads"], "kind": "port",
            "sample": f"oracle restatement of BASELINE config 1 (SDXL UNet LoRA r4 x{len(targets)} Linears, B=1, 512^2, AdamW, "
                      f"autocast cpu bf16): 1 warm-up step, 1 timed step at each of {sweep} threads ({threads} physical cores), 1 more "
                      f"at the fastest ({r['threads']}): {r['s_per_step']:.2f} s/step; sweep s/step {r['sweep_s_per_step']}"}


def per_call_roofline(prof: dict) -> dict:
    """Algorithmic work / measured time of every hand-written C-ABI call of the step, against the roofline that
    bounds it: flops vs the dense bf16 MFMA peak for the matrix kernels, bytes vs the HBM peak for the streaming ones
    (vision_ft_amd/kernels.py passes each call's algorithmic work to its timer)."""
    out = {}
    for k, v in prof.items():
        if v["ms"] <= 0:
            continue
        rate = v["work"] / (v["ms"] * 1e-3)
        mfma = k.startswith(("attn_", "conv3x3", "gemm_", "linear_", "lib_gemm"))
        if mfma:
            out[k] = {"bound": "mfma", "achieved": round(rate / 1e12, 1), "unit": "TFLOP/s", "frac": round(rate / 1e12 / PEAK_BF16_MFMA_TFLOPS, 3)}
        else:
            out[k] = {"bound": "hbm", "achieved": round(rate / 1e9), "unit": "GB/s", "frac": round(rate / 1e9 / PEAK_HBM_GBS, 3)}
    return out


# hand-written kernels by family: (profile-key prefixes, HIP kernel name, roofline that bounds it)
FAMILIES = {
    "linear_gemm": (("linear_fwd",), "lin8_kernel (vft_linear_fwd: base GEMM + bias + LoRA rank extension + skip, forward and data gradient)", "mfma"),
    "attention_bwd": (("attn_bwd",), "attn_bwd1_kernel (one pass: head_dim 64 self-attention) / attn_bwd_dkdv3_kernel<.., true> + batched wg8_kernel (stored dS: "
                                     "head_dim >= 128) / attn_bwd_dq3_kernel + attn_bwd_dkdv3_kernel (short key sequences) / attn_bwd_dq_kernel + "
                                     "attn_bwd_dkdv4_kernel (head_dim 96)", "mfma"),
    "attention_fwd": (("attn_fwd",), "attn_fwd3_kernel (head_dim 64 / 96) / attn_fwd2_kernel (head_dim >= 128)", "mfma"),
    "conv3x3": (("conv3x3_fwd", "conv3x3_dgrad"), "lin8_kernel, implicit-GEMM mode (vft_conv3x3_fwd: forward and data gradient)", "mfma"),
    "weight_grad": (("linear_wgrad", "conv3x3_wgrad"), "wg8_kernel (vft_linear_wgrad / vft_conv3x3_wgrad: weight gradients of trainable projections and 3x3 convolutions)", "mfma"),
    "gemm_nf4": (("gemm_nf4",), "lin8_kernel<.., NF4> (vft_linear_nf4: the launch dequantises the 4-bit weight once, then the bf16 main loop)", "mfma"),
    # NOT ours: the library GEMMs that remain (full fine-tune: trainable projections), timed like the rest so that the
    # dominant kernel is named honestly when it is theirs
    "library_gemm": (("lib_gemm",), "hipBLASLt Cijk_* (torch F.linear / matmul of trainable weights: forward, data gradient, large weight gradients)", "mfma"),
    "lora_rank": (("lora_",), "lora_rows_kernel / lora_wgrad kernels", "hbm"),
    "norms_activations": (("norm_", "groupnorm_", "geglu_", "swiglu_", "gate_residual", "qk_norm_rope", "colsum"), "norm / GroupNorm / GEGLU / SwiGLU / column-sum kernels", "hbm"),
}


# short names for the ONE printed line (the long descriptions above go to the detail file)
SHORT_KERNEL = {
    "linear_gemm": "lin8_kernel (vft_linear_fwd)", "attention_bwd": "attn_bwd* kernels (vft_attn_bwd)", "attention_fwd": "attn_fwd* kernels (vft_attn_fwd)",
    "conv3x3": "lin8_kernel, implicit-GEMM mode (vft_conv3x3_fwd)", "weight_grad": "wg8_kernel (vft_linear_wgrad / vft_conv3x3_wgrad)",
    "gemm_nf4": "NF4 projection GEMM (vft_linear_nf4)", "library_gemm": "hipBLASLt Cijk_* (not ours)", "lora_rank": "lora_down32 / lora_wgrad kernels",
    "norms_activations": "norm / GroupNorm / GEGLU / SwiGLU kernels",
}


def compact_roofline(r: dict) -> dict:
    """The roofline object of the printed line: the dominant family's figures + one [ms/step, fraction of its roofline] pair per
    family.  Everything else (per-call table, kernel descriptions) is in the detail file."""
    if not r:
        return {}
    keep = ("bound", "family", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step", "share_of_step", "step_mfma_frac")
    out = {k: r.get(k) for k in keep}
    out["kernel"] = SHORT_KERNEL.get(r.get("family"), r.get("kernel"))
    note = r.get("traffic_note")
    out["traffic_note"] = None if not note else note[:160]
    out["families_ms_frac"] = {k: [v["ms_per_step"], v["frac"]] for k, v in r.get("families", {}).items()}
    out["hand_written_ms_per_step"] = r.get("hand_written_ms_per_step")
    out["library_and_aten_ms_per_step"] = r.get("library_and_aten_ms_per_step")
    out["measured"] = r.get("measured")
    return out


def compact_config(c: dict) -> dict:
    keep = ("workload", "resolution", "per_gpu_batch", "global_batch", "parallelism", "optimizer", "hip_graph", "trainable_params", "loss",
            "peak_mem_gib", "algorithmic_tflop_per_image", "use_rope", "lora_key_set", "caption_lengths", "nf4_barrier_timed_out")
    out = {k: c[k] for k in keep if k in c}
    if "base_weights" in c:
        out["base_weights"] = "NF4 in HBM, dequantised in the GEMM kernel" if "only copy" in c["base_weights"] else "NF4 state, bf16 compute cache (labelled option)"
    if "buckets" in c:
        out["buckets"] = {"grad_accum": c["buckets"]["grad_accum"], "n_buckets": len(c["buckets"]["buckets_wh"])}
    return out


def family_table(prof: dict, steps: int) -> dict:
    out = {}
    for fam, (prefixes, kname, bound) in FAMILIES.items():
        keys = [k for k in prof if k.startswith(prefixes)]
        ms = sum(prof[k]["ms"] for k in keys)
        if ms <= 0:
            continue
        work = sum(prof[k]["work"] for k in keys)
        n = sum(prof[k]["launches"] for k in keys)
        rate = work / (ms * 1e-3)
        peak = PEAK_BF16_MFMA_TFLOPS * 1e12 if bound == "mfma" else PEAK_HBM_GBS * 1e9
        out[fam] = {"kernel": kname, "bound": bound, "ms_per_step": round(ms / steps, 2), "launches_per_step": n // max(steps, 1),
                    "achieved": round(rate / (1e12 if bound == "mfma" else 1e9), 1), "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                    "frac": round(rate / peak, 4), "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
                    "work_per_launch": round(work / max(n, 1) / 1e9, 3)}
    return out


def pmc_traffic(family: str):
    """HBM bytes per launch of the family's kernel from the committed PMC passes (profiles/r05_a_pmc_hbm_traffic.csv, falling
    back to earlier files: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs of tools/pmc_run.py,
    gfx950 x2 fetch correction applied) — measured at the shape named in the note, not inside this run."""
    import csv

    want = {"linear_gemm": "lin8_kernel", "conv3x3": "lin8_conv", "attention_fwd": "attn_fwd3_kernel", "attention_bwd": "attn_bwd1_kernel", "gemm_nf4": "lin8_nf4",
            "weight_grad": "wg8_kernel<false>"}.get(family)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("r05_a_pmc_hbm_traffic.csv", "r04_b_pmc_hbm_traffic.csv", "r04_a_pmc_hbm_traffic.csv", "r03_d_pmc_hbm_traffic.csv", "r03_c_pmc_hbm_traffic.csv", "r03_b_pmc_hbm_traffic.csv", "r03_a_pmc_hbm_traffic.csv", "r02_e_pmc_hbm_traffic.csv", "r02_c_pmc_hbm_traffic.csv", "r02_b_pmc_hbm_traffic.csv", "r02_pmc_hbm_traffic.csv", "r01_pmc_hbm_traffic.csv"):
        try:
            for row in csv.DictReader(open(os.path.join(root, name))):
                if want and row["kernel"] == want:
                    return (round(float(row["traffic_MB"]) * 1e6),
                            f"{name}: bytes/launch at {row.get('shape', 'the headline shape of tools/pmc_run.py')}; algorithmic {row['algorithmic_MB']} MB")
        except OSError:
            pass
    return None, None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process and return
    its exit code.  Called before this process has made any GPU call (never exec from a process that has, and the children
    initialise the GPU themselves)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL over dmabuf IPC (the host driver has no legacy IPC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def claim_stdout():
    """Keep STDOUT for the result line alone: returns a handle on the real stdout and points file descriptor 1 at stderr for
    everything else in this process — native libraries included.  RCCL writes a five-line version banner to STDOUT when the
    first communicator is created (`RCCL version : ... Librccl path : ...`, measured on the MI355X box with
    tools/exp/rccl_banner.py; only NCCL_DEBUG=NONE silences it), in EVERY rank, and torch.distributed.run forwards every rank's
    stdout — the driver reads ONE JSON line there."""
    sys.stdout.flush()
    keep = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sdxl_lora", choices=list(F_ALG))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single", action="store_true", help="time only --workload (a default run adds an AuraFlow-LoRA leg)")
    ap.add_argument("--nf4-cache", action="store_true", help="lumina_qlora: compute on a dequantised bf16 copy (labelled option)")
    ap.add_argument("--drawn-lengths", action="store_true", help="lumina_*: caption lengths U{32..256} per step (eager)")
    ap.add_argument("--use-rope", action="store_true", help="aura_lora: use_rope=True, rope_dim_sizes [32,112,112]")
    ap.add_argument("--aura-keys", default="debug252", choices=["debug252", "yaml124"], help="aura_lora: LoRA target key set")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-gemm-table", action="store_true", help="library GEMMs use their default heuristic, not the committed table")
    ap.add_argument("--tune-gemms", action="store_true", help="time library GEMM solutions for missing shapes and update the table")
    ap.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the ranks ourselves, as child processes, BEFORE any GPU call (this process never
    #      touches the GPU; it only relays the children's exit code — rank 0's JSON line goes straight to our stdout)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    result_out = claim_stdout()  # from here on fd 1 is stderr; rank 0 writes its one line to `result_out`
    if args.launch_selftest:
        # host-only rehearsal of the launch path (tests/test_host_logic.py, no GPU): process group, barrier, the MAX
        # reduction the timing uses, one JSON line from rank 0
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            result_out.write(json.dumps({"selftest": "launch", "n_gpus": world, "max_over_ranks": float(t.item())}) + "\n")
            result_out.flush()
        dist.destroy_process_group()
        return
    # (test hooks: VFT_FORCE_DEVICE / VFT_DIST_BACKEND=gloo let two ranks share one GPU to rehearse the N>1 path)
    device = torch.device("cuda", int(os.environ.get("VFT_FORCE_DEVICE", local)))
    torch.cuda.set_device(device)
    one_rank_group = world == 1 and os.environ.get("VFT_DP_ALWAYS_REDUCE", "0") == "1"  # rehearsal hook, trainer/dp.py
    if world > 1 or one_rank_group:
        backend = os.environ.get("VFT_DIST_BACKEND", "nccl")
        kw = {}
        if one_rank_group:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            kw = {"rank": 0, "world_size": 1}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)

    from vision_ft_amd import kernels, tuning
    from vision_ft_amd.trainer.dp import DataParallelGrads

    gemm_table = False if args.no_gemm_table else tuning.enable(tune=args.tune_gemms and rank == 0)
    if os.environ.get("VFT_LORA_WGRAD_TARGET"):  # A/B knob: workgroups the LoRA weight-gradient row blocking aims at (0 = automatic)
        from vision_ft_amd import _C as _c

        _c.lib().vft_lora_wgrad_target(int(os.environ["VFT_LORA_WGRAD_TARGET"]))
    default_batch = {"sdxl_lora": 24, "sdxl_full": 24, "aura_lora": 6, "lumina_lora": 4, "lumina_qlora": 4, "sdxl_buckets": 16}

    def fence():
        if world > 1 or one_rank_group:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(workload: str, batch: int, steps: int, warmup: int, nf4_cache: bool = False, drawn_lengths: bool = False,
                     use_rope: bool = False, aura_keys: str = "debug252") -> dict:
        """Build, warm up, time `steps` steps (barrier + synchronize on both sides, max over ranks), then an eager pass with
        HIP events around every hand-written launch.  Returns the result dict of this workload.
        nf4_cache: Lumina2-QLoRA computes on a dequantised bf16 copy (an MI355X memory-for-speed option) instead of the
        fused in-kernel dequantisation BASELINE config 4 names; drawn_lengths: caption lengths U{32..256} per step (eager:
        every step has a new ragged layout); use_rope / aura_keys: the AuraFlow variants of SURVEY §8d config 3."""
        from vision_ft_amd.modules.quant import nf4 as _nf4

        _nf4.NF4_COMPUTE_CACHE = bool(nf4_cache)
        model = build(workload, device, use_rope=use_rope, aura_keys=aura_keys)
        params = [p for p in model.parameters() if p.requires_grad]
        dp = DataParallelGrads(params)
        use_graph = not args.no_graph
        # (full fine-tune: 1e-3 — the reference's default AdamW rate, src/config.py:22-26, meant for adapters — on 2.6 B randomly
        #  initialised weights diverges within three steps and the line would report loss = nan; throughput does not depend on it)
        lr = 1e-5 if workload == "sdxl_full" else 1e-3
        opt = torch.optim.AdamW(params, lr=lr, fused=True, capturable=use_graph)
        fwd = make_step(workload, model, device, batch, seed=42 + rank, drawn_lengths=drawn_lengths)

        def eager_step():
            loss = fwd()
            loss.backward()
            dp.reduce_async()
            dp.wait()
            opt.step()
            dp.zero_grad()
            return loss

        imgs_per_step = batch
        bucket_note = None
        graphs = []
        if workload == "sdxl_buckets":
            # BASELINE config 5: mixed aspect-ratio buckets {640x1536 ... 1536x640}, one bucket per micro-batch drawn
            # per rank from a seeded uniform multinomial, 4 micro-batches accumulated per optimizer step.  One
            # forward/backward graph per bucket shape (shared memory pool), one shared optimizer graph.
            from vision_ft_amd.dataset.synthetic import headline_buckets
            from vision_ft_amd.trainer.graph import GraphedTrainStep

            ga = 4
            buckets = headline_buckets()
            fwds = [make_step(workload, model, device, batch, seed=42 + rank, wh=wh) for wh in buckets]
            opt_graph, pool = None, None
            for f in fwds:
                g_ = GraphedTrainStep(f, opt, dp, warmup=1, split=True, opt_graph=opt_graph, pool=pool, loss_scale=1.0 / ga)
                opt_graph, pool = g_.graph_b, g_.pool()
                graphs.append(g_)
            draw = torch.Generator().manual_seed(1234 + rank)
            counts = [0] * len(buckets)

            def step():
                for m in range(ga):
                    b = int(torch.randint(len(buckets), (1,), generator=draw))
                    counts[b] += 1
                    loss_ = graphs[b](boundary=m == ga - 1)
                return loss_

            imgs_per_step = batch * ga
            bucket_note = {"buckets_wh": buckets, "grad_accum": ga, "draws": counts}
            fwd = fwds[buckets.index((1024, 1024))]
        elif use_graph:
            from vision_ft_amd.trainer.graph import GraphedTrainStep

            step = GraphedTrainStep(fwd, opt, dp, warmup=2)
        else:
            step = eager_step

        for _ in range(warmup):
            wl = step()
            if os.environ.get("VFT_BENCH_DEBUG_LOSS"):  # diagnostic only (synchronises every step)
                print(f"[debug] {workload} warm-up loss {float(wl.detach().float()):.6f}", file=sys.stderr, flush=True)
        kernels.set_profiling(rank == 0 and not use_graph)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
            if os.environ.get("VFT_BENCH_DEBUG_LOSS"):  # diagnostic only (synchronises every step)
                print(f"[debug] {workload} loss {float(loss.detach().float()):.6f}", file=sys.stderr, flush=True)
        fence()
        elapsed = time.perf_counter() - t0
        # read the last step's loss NOW: under graph replay it is a static tensor inside the graph's private pool, which is
        # released below (round 3 printed whatever the eager pass had written there since: negative / nan "losses")
        last_loss = float(loss.detach().float())
        prof_steps = steps
        peak_timed = torch.cuda.max_memory_allocated(device) / 2**30
        eager_ms = None
        if use_graph:
            # release the graphs' private pool before the eager timing pass allocates its own activations
            step = None
            graphs.clear()
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            # per-kernel HIP-event timing cannot be recorded inside a graph replay: time the same step
            # eagerly (same process, same tensors) right after the timed region, outside of it.  Every rank
            # runs it (the step contains the gradient all-reduce); only rank 0 records.
            prof_steps = 2
            kernels.set_profiling(rank == 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(prof_steps):
                eager_step()
            e1.record()
            torch.cuda.synchronize()
            eager_ms = e0.elapsed_time(e1) / prof_steps
        prof = kernels.profile_report() if rank == 0 else {}
        kernels.set_profiling(False)
        if world > 1 or one_rank_group:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        res = {}
        if rank == 0:
            images_s = world * imgs_per_step * steps / elapsed
            ms_per_step = 1e3 * elapsed / steps
            fam = family_table(prof, prof_steps)
            ours_ms = sum(v["ms_per_step"] for k, v in fam.items() if k != "library_gemm")
            if "library_gemm" in fam and use_graph:
                # HIP events around a LIBRARY call in the eager pass also span the host's dispatch work for it (the pass is
                # host-bound: measured 1 926 ms of "library GEMM" in a 652 ms step).  What the graph-replayed step leaves
                # after our own launches bounds it from above (ATen elementwise and MIOpen included): use that.
                lib = fam["library_gemm"]
                lib_ms = max(min(lib["ms_per_step"], ms_per_step - ours_ms), 1e-6)
                rate = lib["work_per_launch"] * 1e9 * lib["launches_per_step"] / (lib_ms * 1e-3)
                lib.update({"ms_per_step": round(lib_ms, 2), "achieved": round(rate / 1e12, 1), "frac": round(rate / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                            "avg_launch_us": round(1e3 * lib_ms / max(lib["launches_per_step"], 1), 2),
                            "measured": "upper bound: graph-replayed step time minus the HIP-event time of our own launches (includes ATen / MIOpen)"})
            # the kernel family with the largest share of the step — ours or the library's (measured, not assumed)
            dom = max(fam, key=lambda k: fam[k]["ms_per_step"]) if fam else None
            traffic, traffic_note = pmc_traffic(dom) if dom else (None, None)
            roofline = {"bound": fam[dom]["bound"], "kernel": fam[dom]["kernel"], "family": dom, "achieved": fam[dom]["achieved"],
                        "peak": PEAK_BF16_MFMA_TFLOPS if fam[dom]["bound"] == "mfma" else PEAK_HBM_GBS, "unit": fam[dom]["unit"],
                        "frac": fam[dom]["frac"], "traffic": traffic, "traffic_note": traffic_note,
                        "avg_launch_us": fam[dom]["avg_launch_us"], "launches_per_step": fam[dom]["launches_per_step"],
                        "share_of_step": round(fam[dom]["ms_per_step"] / ms_per_step, 3)} if dom else {}
            roofline.update({
                "families": fam,
                "hand_written_ms_per_step": round(ours_ms, 1),
                # what is NOT one of our C-ABI launches in the eager pass: library GEMMs (the timed ones are also the
                # "library_gemm" family above) / MIOpen convolutions that remain (time-embedding MLPs, 4-channel stem / head,
                # stride-2 convolutions), ATen elementwise, the optimizer
                "library_and_aten_ms_per_step": round(max(ms_per_step - ours_ms, 0.0), 1),
                "eager_pass_ms_per_step": None if eager_ms is None else round(eager_ms, 1),
                "kernel_ms_per_step": {k: round(v["ms"] / prof_steps, 3) for k, v in prof.items()},
                "per_call": per_call_roofline(prof),
                "measured": "HIP events around each launch, " + ("eager pass after the graph-replayed timed region" if use_graph else "inside the timed region"),
                "step_mfma_frac": round(images_s / world * F_ALG[workload] / PEAK_BF16_MFMA_TFLOPS, 4)})
            res = {
                "value": round(images_s, 3), "ms_per_step": round(ms_per_step, 2),
                "config": {"workload": workload, "resolution": "mixed buckets ~1 MP" if bucket_note else "1024x1024", "per_gpu_batch": batch,
                           "global_batch": batch * world, "parallelism": f"dp{world}" + (" (rehearsal: one-rank process group, gradients through the backend's all-reduce)" if one_rank_group else ""), "optimizer": f"AdamW(fused, lr={lr:g})", "hip_graph": use_graph,
                           "gemm_table": bool(gemm_table), "lora_in_gemm": bool(kernels.FUSED_LORA_GEMM),
                           "trainable_params": sum(p.numel() for p in params), "loss": round(last_loss, 5),
                           "peak_mem_gib": round(peak_timed, 1), "algorithmic_tflop_per_image": F_ALG[workload],
                           **({"buckets": bucket_note} if bucket_note else {}),
                           **({"base_weights": "NF4 (blocksize 64, double-quantised absmax): " + (
                               "module state / checkpoint format; compute on a dequantised bf16 cache (VFT_NF4_CACHE=1: NOT the "
                               "config as BASELINE states it — a labelled memory-for-speed option)"
                               if _nf4_cache_on() else "the only copy in HBM, dequantised inside the GEMM kernel (VFT_NF4_CACHE=0, default)"),
                               "caption_lengths": "U{32..256} drawn per step on the device, shape-static forward" if drawn_lengths else "256 (fixed)",
                               # (the dequantising launch's grid barrier has a bounded wait: did any launch give up on it?)
                               "nf4_barrier_timed_out": bool(kernels.nf4_barrier_timed_out())}
                              if workload.startswith("lumina") else {}),
                           **({"use_rope": use_rope, "lora_key_set": "configs/auraflow/lora.yml (124 Linears)" if aura_keys == "yaml124"
                               else "tests/assets/debug_dataset.yml (252 Linears)"} if workload.startswith("aura") else {})},
                "roofline": roofline,
            }
        # free everything before the next leg
        del model, params, dp, opt, fwd
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(device)
        return res

    main_res = run_workload(args.workload, args.batch or default_batch[args.workload], args.steps, args.warmup,
                            nf4_cache=args.nf4_cache, drawn_lengths=args.drawn_lengths, use_rope=args.use_rope, aura_keys=args.aura_keys)
    # The metric names two models ("SDXL-LoRA, AuraFlow-MMDiT"): a default run also times AuraFlow-LoRA (BASELINE config 3,
    # rank 16, 252 Linears) with its own roofline, reported under "workloads".  `value` stays the named workload's.
    legs = {}
    if args.workload == "sdxl_lora" and not args.single:
        short = max(3, min(args.steps, 5))
        legs["aura_lora"] = run_workload("aura_lora", default_batch["aura_lora"], short, 2)
        if world == 1:  # BASELINE configs[1] (SDXL full fine-tune, 1 GPU): a third, single-GPU leg of the default run
            legs["sdxl_full"] = run_workload("sdxl_full", default_batch["sdxl_full"], short, 2)
            # BASELINE configs[3] (Lumina-Image-2.0 NextDiT QLoRA: NF4 base + bf16 LoRA) AS WRITTEN: the 4-bit weights are the
            # only copy in HBM and are dequantised inside the GEMM kernel
            legs["lumina_qlora"] = run_workload("lumina_qlora", default_batch["lumina_qlora"], short, 2)
            # ... and its variants, labelled: caption lengths drawn U{32..256}, and the bf16 compute cache
            legs["lumina_qlora_drawn_lengths"] = run_workload("lumina_qlora", default_batch["lumina_qlora"], 3, 1, drawn_lengths=True)
            legs["lumina_qlora_bf16_cache"] = run_workload("lumina_qlora", default_batch["lumina_qlora"], 3, 2, nf4_cache=True)
            # SURVEY §8d config 3, the other two AuraFlow runs: use_rope=True, and the shipped YAML's 124-Linear key set
            # BASELINE configs[4]: SDXL mixed aspect-ratio buckets {640x1536 ... 1536x640}, 4 micro-batches accumulated per
            # optimizer step, one forward/backward graph per bucket (a step = 4 x 16 images)
            legs["sdxl_buckets"] = run_workload("sdxl_buckets", default_batch["sdxl_buckets"], 3, 1)
            legs["aura_lora_rope"] = run_workload("aura_lora", default_batch["aura_lora"], 3, 2, use_rope=True)
            legs["aura_lora_yaml124"] = run_workload("aura_lora", default_batch["aura_lora"], 3, 2, aura_keys="yaml124")

    if rank == 0:
        base = {
            "metric": "train images/sec at 1024^2", "value": main_res["value"], "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        }
        detail = {**base, "config": main_res["config"], "roofline": main_res["roofline"]}
        if legs:
            detail["workloads"] = {k: {"metric": "train images/sec at 1024^2", "unit": "images/s", **v} for k, v in legs.items()}
        cpu = cpu_baseline(threads=physical_cores()) if world == 1 and not args.no_cpu_baseline else None
        if cpu:
            detail["cpu_baseline"] = cpu
        # the full record (per-call table, every leg's family table and config) goes to a file and to stderr; stdout carries ONE
        # compact JSON line (round 3's 34 KB line was longer than the driver's capture and could not be parsed)
        detail_path = None
        for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
            try:
                os.makedirs(d, exist_ok=True)
                detail_path = os.path.join(d, f"bench_detail_n{world}.json")
                with open(detail_path, "w") as f:
                    json.dump(detail, f, indent=1)
                break
            except OSError:
                detail_path = None
        print("bench detail: " + json.dumps(detail), file=sys.stderr, flush=True)
        out = {**base, "config": compact_config(main_res["config"]), "roofline": compact_roofline(main_res["roofline"])}
        if legs:
            out["workloads"] = {k: {"value": v["value"], "ms_per_step": v["ms_per_step"], "per_gpu_batch": v["config"]["per_gpu_batch"],
                                    "step_mfma_frac": v["roofline"].get("step_mfma_frac"), "dominant": v["roofline"].get("family"),
                                    "dominant_frac": v["roofline"].get("frac"), "dominant_share": v["roofline"].get("share_of_step")}
                                for k, v in legs.items()}
        if cpu:
            out["cpu_baseline"] = cpu
        out["detail_file"] = os.path.relpath(detail_path, ROOT) if detail_path else "stderr"
        sys.stdout.flush()
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if world > 1 or one_rank_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
