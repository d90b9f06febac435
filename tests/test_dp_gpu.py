"""GPU rehearsal of the N>1 training path on a one-GPU box: two ranks share cuda:0 and exchange gradients over gloo
(VFT_FORCE_DEVICE / VFT_DIST_BACKEND hooks of trainer.init_distributed), which exercises everything of the data-parallel
step except the RCCL transport itself: sharding with equal step counts, the flat gradient buffer, the all-reduce on the
side stream, clipping after the exchange, and — in graph mode — the split forward/backward | clip | optimizer graphs.
The ranks are forked from the clean fork server that tests/conftest.py starts before any GPU call."""
import multiprocessing
import os

import pytest
import torch

from .dp_rehearsal_worker import config, run_rank

pytestmark = pytest.mark.gpu


def _spawn(tmp_path, hip_graph: bool, tag: str):
    out = tmp_path / tag
    out.mkdir()
    ctx = multiprocessing.get_context("forkserver")
    port = 32500 + (os.getpid() % 2000) + (7 if hip_graph else 0)
    procs = [ctx.Process(target=run_rank, args=(r, 2, port, str(out), hip_graph)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [torch.load(out / f"rank{r}.pt") for r in range(2)]


def _reference_mean_gradient_run(gpu, tmp_path):
    """One process, same data: per optimizer step the two ranks' micro-batches (dataset items r, r+2, ...) with the loss
    halved, i.e. the mean gradient; each micro-batch restarts from the RNG state of that step, as both ranks (seeded alike)
    draw the same noise / timesteps for their different data; clip, AdamW."""
    import sys

    sys.path.insert(0, ".")
    from train.sdxl.text_to_image import SDXLForTextToImageTraining
    from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig
    from vision_ft_amd.trainer.common import Trainer

    tr = Trainer(config(str(tmp_path), hip_graph=False))
    tr.register_train_dataset_class(SyntheticLatentDatasetConfig)
    tr.register_model_class(SDXLForTextToImageTraining)
    tr.before_train()
    tr.raw_model.sanity_check()
    tr.raw_model.before_train_epoch()
    batches = list(tr.train_dataloader)
    assert len(batches) == 6
    params = [p for p in tr.model.parameters() if p.requires_grad]
    losses = [[], []]
    for j in range(3):
        state = torch.cuda.get_rng_state()
        for r in range(2):
            torch.cuda.set_rng_state(state)
            loss = tr.model(batches[2 * j + r])
            (loss / 2).backward()
            losses[r].append(float(loss))
        torch.nn.utils.clip_grad_norm_(params, 0.05)
        tr.optimizer.step()
        tr.dp.zero_grad()
    sd = {k: v.detach().cpu() for k, v in tr.raw_model.model.state_dict().items() if ".lora_" in k}
    return sd, losses


def _dist(a, b):
    num = sum(((a[k].float() - b[k].float()) ** 2).sum() for k in a) ** 0.5
    den = sum((b[k].float() ** 2).sum() for k in a) ** 0.5
    return float(num / den)


def test_two_ranks_on_one_gpu_match_the_mean_gradient_run(gpu, tmp_path):
    eager = _spawn(tmp_path, hip_graph=False, tag="eager")
    assert all(r["world"] == 2 and r["steps_per_epoch"] == 3 and r["side_stream"] and len(r["losses"]) == 3 for r in eager)
    for k in eager[0]["lora"]:  # replicas stay bit-identical
        assert torch.equal(eager[0]["lora"][k], eager[1]["lora"][k]), k
    assert eager[0]["losses"] != eager[1]["losses"]  # the ranks really saw different data
    ref, ref_losses = _reference_mean_gradient_run(gpu, tmp_path)
    # first-step losses do not depend on any update: the same kernels on the same inputs
    assert abs(eager[0]["losses"][0] - ref_losses[0][0]) < 2e-3 * abs(ref_losses[0][0])
    assert abs(eager[1]["losses"][0] - ref_losses[1][0]) < 2e-3 * abs(ref_losses[1][0])
    moved = _dist({k: torch.zeros_like(v) for k, v in ref.items() if "lora_up" in k}, {k: v for k, v in ref.items() if "lora_up" in k})
    assert moved == 1.0 and any(v.abs().sum() > 0 for k, v in ref.items() if "lora_up" in k)
    # Adam normalises the update, so bf16 run-to-run noise of the reductions shows in the weights: bounded, small
    assert _dist(eager[0]["lora"], ref) < 3e-2, _dist(eager[0]["lora"], ref)
    torch.testing.assert_close(torch.tensor(eager[0]["losses"]), torch.tensor(ref_losses[0]), rtol=5e-3, atol=1e-4)
    torch.testing.assert_close(torch.tensor(eager[1]["losses"]), torch.tensor(ref_losses[1]), rtol=5e-3, atol=1e-4)

    # HIP-graph mode: forward/backward graph | gradient exchange on the side stream | clip graph | optimizer graph
    graph = _spawn(tmp_path, hip_graph=True, tag="graph")
    for k in graph[0]["lora"]:
        assert torch.equal(graph[0]["lora"][k], graph[1]["lora"][k]), k
    for r in range(2):
        assert len(graph[r]["losses"]) == 3
        assert abs(graph[r]["losses"][0] - eager[r]["losses"][0]) < 5e-3 * abs(eager[r]["losses"][0])
        assert all(torch.isfinite(torch.tensor(graph[r]["losses"])))


def _one_rank(tmp_path, tag, hip_graph, always_reduce):
    from .dp_rehearsal_worker import run_one_rank_backend_group

    out = tmp_path / tag
    out.mkdir()
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=run_one_rank_backend_group, args=(str(out), hip_graph, "nccl", always_reduce))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0, p.exitcode
    return torch.load(out / "result.pt")


@pytest.mark.parametrize("hip_graph", [False, True])
def test_one_rank_rccl_group_takes_the_same_steps(gpu, tmp_path, hip_graph):
    """The RCCL call path of the data-parallel step on the one GPU there is: a one-rank `nccl` process group with
    VFT_DP_ALWAYS_REDUCE=1 sends every step's flat gradient buffer through RCCL's all-reduce on the side stream (pre-scaled
    by 1 / 1; the identity) and, in graph mode, runs the split-graph form of the N > 1 step.  Same data, same seeds as the
    plain single-GPU run => the same losses and weights (within the run-to-run noise of the tiny model's library GEMMs), and
    a buffer of known numbers comes back from the exchange bit for bit."""
    plain = _one_rank(tmp_path, "plain", hip_graph, always_reduce=False)
    rccl = _one_rank(tmp_path, "rccl", hip_graph, always_reduce=True)
    assert not plain["collective"] and plain["group"] is False
    assert rccl["collective"] and rccl["group"] == "nccl" and rccl["steps"] == plain["steps"] == 6
    assert rccl["identity"] is True  # known numbers through the side-stream all-reduce: every bit back
    # (the tiny model's conditioning GEMMs are library calls whose reductions are not run-to-run reproducible, and Adam
    #  amplifies their last bits: equal within that noise, as in the two-rank rehearsal above)
    assert rccl["losses"][0] == plain["losses"][0]  # no update yet: the same kernels on the same inputs
    torch.testing.assert_close(torch.tensor(rccl["losses"]), torch.tensor(plain["losses"]), rtol=5e-3, atol=1e-4)
    assert _dist(rccl["lora"], plain["lora"]) < 3e-2, _dist(rccl["lora"], plain["lora"])
    assert any(v.abs().sum() > 0 for k, v in rccl["lora"].items() if "lora_up" in k)


def test_bench_line_is_alone_on_stdout_with_rccl(tmp_path):
    """RCCL prints a version banner on STDOUT when its first communicator is created (five lines per rank); bench.py keeps
    stdout for its one JSON line (`claim_stdout`).  Run inside a one-rank `nccl` group: the banner must have gone to stderr,
    stdout must be exactly the line, and the line must say that the gradients went through the backend."""
    import json

    from .dp_rehearsal_worker import run_bench_in_one_rank_rccl_group

    out = tmp_path / "bench_rccl.json"
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=run_bench_in_one_rank_rccl_group, args=(str(out),))
    p.start()
    p.join(timeout=1000)
    assert p.exitcode == 0
    res = json.loads(out.read_text())
    assert res["rc"] == 0, res["stderr"][-3000:]
    lines = [ln for ln in res["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1, res["stdout"][:2000]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and "one-rank process group" in line["config"]["parallelism"] and line["value"] > 0


def test_bench_gpus_2_self_launch(tmp_path):
    """`python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) starts its two ranks itself and prints ONE JSON line with
    n_gpus = 2 and the whole-job rate.  Rehearsed on the one GPU (both ranks on cuda:0, gloo transport): everything of the
    driver's N > 1 command line except RCCL itself."""
    import json

    from .dp_rehearsal_worker import run_self_launched_bench

    out = tmp_path / "bench2.json"
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=run_self_launched_bench, args=(str(out),))
    p.start()
    p.join(timeout=1000)
    assert p.exitcode == 0
    res = json.loads(out.read_text())
    assert res["rc"] == 0, res["stderr"]
    lines = [ln for ln in res["stdout"].splitlines() if ln.strip()]  # nothing but the line on stdout
    assert len(lines) == 1, res["stdout"][-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 2 and line["value"] > 0
    assert abs(line["value"] - 2 * 1 / (line["ms_per_step"] * 1e-3)) < 0.02 * line["value"]
