"""GPU: the Trainer end to end on a tiny SDXL config (LoRA, synthetic buckets), eager and HIP-graph
replayed; adapter checkpoint written with ComfyUI key names."""
import pytest
import torch
from safetensors import safe_open

pytestmark = pytest.mark.gpu


def _config(tmp_path, hip_graph: bool):
    from vision_ft_amd.config import TrainConfig

    return TrainConfig.model_validate({
        "model": {"checkpoint_path": "random", "dtype": "bfloat16", "max_token_length": 75,
                  "denoiser": {"hidden_dim": 64, "block_out_channels": [64, 128, 128], "context_dim": 64,
                               "num_transformers_per_block": [1, 1, 2]}},
        "peft": {"config": {"type": "lora", "rank": 4, "alpha": 1.0}, "include_keys": ["attn1", "attn2", ".ff."],
                 "exclude_keys": ["text_encoder", "vae"]},
        "dataset": {"batch_size": 2, "bucket_base_size": 256, "step": 64, "min_size": 128, "samples_per_bucket": 4,
                    "buckets_wh": [[256, 256], [192, 320]], "num_workers": 0},
        "optimizer": {"name": "torch.optim.AdamW", "args": {"lr": 1e-3}},
        "saving": {"strategy": {"per_epochs": 1}, "callbacks": [{"type": "safetensors", "name": "tiny", "save_dir": str(tmp_path)}]},
        "trainer": {"hip_graph": hip_graph},
        "num_train_epochs": 2,
    })


@pytest.mark.parametrize("hip_graph", [False, True])
def test_trainer_runs_and_saves_adapter(gpu, tmp_path, hip_graph):
    import sys

    sys.path.insert(0, ".")
    from train.sdxl.text_to_image import SDXLForTextToImageTraining
    from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig
    from vision_ft_amd.trainer.common import Trainer

    trainer = Trainer(_config(tmp_path, hip_graph))
    trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
    trainer.register_model_class(SDXLForTextToImageTraining)
    trainer.train()
    assert len(trainer.losses) == 2 * 4  # 2 buckets x 2 batches x 2 epochs
    assert all(torch.isfinite(torch.tensor(trainer.losses)))
    assert 0.5 < trainer.losses[0] < 2.0  # eps-prediction MSE of a random-init net is ~1
    files = sorted(tmp_path.glob("tiny_*.safetensors"))
    assert [f.name for f in files] == ["tiny_00001e_000004s.safetensors", "tiny_00002e_000008s.safetensors"]
    with safe_open(files[-1], "pt") as f:
        keys = list(f.keys())
        assert all(k.startswith("diffusion_model.") for k in keys)
        assert not any(".blocks." in k for k in keys)  # original (sgm/ComfyUI) block indexing
        assert any(k.endswith("attn1.to_q.lora_down.weight") for k in keys) and any(k.endswith(".alpha") for k in keys)
        up = f.get_tensor(next(k for k in keys if k.endswith("attn1.to_q.lora_up.weight")))
        assert up.abs().sum() > 0  # the adapter actually trained (zero-init up moved)
