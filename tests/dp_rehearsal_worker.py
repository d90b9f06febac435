"""Rank body of tests/test_dp_gpu.py: one data-parallel rank of the tiny SDXL-LoRA trainer.  Imported by the ranks that
the fork server creates (a module of its own so that they can find it by name)."""
from __future__ import annotations

import os


def config(tmp_path: str, hip_graph: bool, accum: int = 1, clip: float | None = 0.05):
    from vision_ft_amd.config import TrainConfig

    return TrainConfig.model_validate({
        "model": {"checkpoint_path": "random", "dtype": "bfloat16", "max_token_length": 75,
                  "denoiser": {"hidden_dim": 64, "block_out_channels": [64, 128, 128], "context_dim": 64,
                               "num_transformers_per_block": [1, 1, 2]}},
        "peft": {"config": {"type": "lora", "rank": 4, "alpha": 1.0}, "include_keys": ["attn1", "attn2", ".ff."],
                 "exclude_keys": ["text_encoder", "vae"]},
        "dataset": {"batch_size": 2, "bucket_base_size": 256, "step": 64, "min_size": 128, "samples_per_bucket": 6,
                    "buckets_wh": [[256, 256], [192, 320]], "num_workers": 0, "shuffle": False},
        "optimizer": {"name": "torch.optim.AdamW", "args": {"lr": 1e-3}},
        "saving": None,
        "trainer": {"hip_graph": hip_graph, "gradient_accumulation_steps": accum, "clip_grad_norm": clip, "tuned_gemms": False},
        "num_train_epochs": 1,
    })


def run_rank(rank: int, world: int, port: int, out_dir: str, hip_graph: bool) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      VFT_FORCE_DEVICE="0", VFT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys

    import torch
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from train.sdxl.text_to_image import SDXLForTextToImageTraining
    from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig
    from vision_ft_amd.trainer.common import Trainer

    try:
        trainer = Trainer(config(out_dir, hip_graph))
        trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
        trainer.register_model_class(SDXLForTextToImageTraining)
        trainer.train()
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu() for k, v in trainer.raw_model.model.state_dict().items() if ".lora_" in k}
        torch.save({"lora": sd, "losses": list(trainer.losses), "steps_per_epoch": len(trainer.train_dataloader),
                    "world": trainer.accelerator.world, "side_stream": trainer.dp.stream is not None}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def run_bench_in_one_rank_rccl_group(out_path: str) -> None:
    """Body of tests/test_dp_gpu.py::test_bench_line_is_alone_on_stdout_with_rccl: bench.py inside a one-rank `nccl` group
    (VFT_DP_ALWAYS_REDUCE=1), stdout and stderr captured separately."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NCCL_DEBUG")}
    env.update(VFT_DP_ALWAYS_REDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(34500 + os.getpid() % 2000))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--single", "--batch", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-gemm-table"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    with open(out_path, "w") as f:
        json.dump({"rc": r.returncode, "stdout": r.stdout[-20000:], "stderr": r.stderr[-20000:]}, f)


def run_self_launched_bench(out_path: str) -> None:
    """Body of tests/test_dp_gpu.py::test_bench_gpus_2_self_launch: runs in a child of the clean fork server (this process
    has never touched the GPU), so starting bench.py — which starts its own two ranks — is an ordinary child process."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VFT_FORCE_DEVICE="0", VFT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single", "--batch", "1", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-gemm-table"], env=env, cwd=root, capture_output=True, text=True,
                       timeout=900)
    with open(out_path, "w") as f:
        json.dump({"rc": r.returncode, "stdout": r.stdout[-20000:], "stderr": r.stderr[-4000:]}, f)


def run_one_rank_backend_group(out_dir: str, hip_graph: bool, backend: str, always_reduce: bool) -> None:
    """Body of tests/test_dp_gpu.py::test_one_rank_rccl_group_*: the tiny trainer on cuda:0, either plainly (no process
    group) or inside a ONE-rank process group of `backend` with VFT_DP_ALWAYS_REDUCE=1 — every optimizer step then sends the
    flat gradient buffer through the backend's all-reduce on the side stream (the identity for one rank) and, in graph mode,
    takes the split forward/backward | exchange | clip | optimizer form of the N > 1 step."""
    env = dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(33500 + os.getpid() % 2000),
               HSA_ENABLE_IPC_MODE_LEGACY="0", VFT_DIST_BACKEND=backend)
    if always_reduce:
        env["VFT_DP_ALWAYS_REDUCE"] = "1"
    os.environ.update(env)
    import sys

    import torch
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from train.sdxl.text_to_image import SDXLForTextToImageTraining
    from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig
    from vision_ft_amd.trainer.common import Trainer

    try:
        trainer = Trainer(config(out_dir, hip_graph))
        trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
        trainer.register_model_class(SDXLForTextToImageTraining)
        trainer.train()
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu() for k, v in trainer.raw_model.model.state_dict().items() if ".lora_" in k}
        identity = None
        if always_reduce:
            # the exchange alone, on known numbers (Adam's update does not see a wrong gradient scale; this does): one rank's
            # mean over one rank must hand back every bit
            from vision_ft_amd.trainer.dp import DataParallelGrads

            ps = [torch.nn.Parameter(torch.zeros(n, device="cuda", dtype=torch.bfloat16)) for n in (1000, 77, 4096)]
            dp = DataParallelGrads(ps, bucket_bytes=4096)  # several all-reduce calls per buffer
            for b in dp.flat.buffers.values():
                b.copy_(torch.randn(b.shape, device="cuda").to(b.dtype))
            before = {k: b.clone() for k, b in dp.flat.buffers.items()}
            dp.reduce_async()
            dp.wait()
            torch.cuda.synchronize()
            identity = dp.collective and all(torch.equal(before[k], b) for k, b in dp.flat.buffers.items())
        torch.save({"lora": sd, "losses": list(trainer.losses), "collective": bool(trainer.dp.collective),
                    "group": dist.is_initialized() and dist.get_backend(), "steps": len(trainer.losses), "identity": identity},
                   os.path.join(out_dir, "result.pt"))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
