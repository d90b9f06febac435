"""Data-parallel gradient exchange on CPU: 2 ranks over gloo (the N>1 path of bench.py / Trainer).
Checks that the flat-buffer all-reduce yields the mean gradient, that ranks stay in lock-step, and
that sharding gives each rank a disjoint, covering slice of the bucket batches."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank: int, world: int, port: int, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vision_ft_amd.trainer.dp import DataParallelGrads, shard_indices

        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
        model[0].weight.requires_grad_(False)  # frozen parameter: must not enter the buffer
        params = [p for p in model.parameters() if p.requires_grad]
        dp = DataParallelGrads(params, bucket_bytes=64)  # tiny buckets -> several all-reduces
        opt = torch.optim.SGD(params, lr=0.1)
        losses = []
        for step in range(3):
            g = torch.Generator().manual_seed(100 * step + rank)  # different data per rank
            x, y = torch.randn(5, 8, generator=g), torch.randn(5, 4, generator=g)
            loss = torch.nn.functional.mse_loss(model(x), y)
            loss.backward()
            dp.reduce_async()
            dp.wait()
            opt.step()
            dp.zero_grad()
            losses.append(loss.item())
        flat = torch.cat([p.detach().flatten() for p in model.parameters()])
        out[rank] = (flat, losses, shard_indices(10, rank, world), dp.flat.nbytes())
    finally:
        dist.destroy_process_group()


def _reference_run():
    """Single process computing the mean gradient of both ranks' batches explicitly."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    model[0].weight.requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.1)
    for step in range(3):
        grads = [torch.zeros_like(p) for p in params]
        for rank in range(2):
            g = torch.Generator().manual_seed(100 * step + rank)
            x, y = torch.randn(5, 8, generator=g), torch.randn(5, 4, generator=g)
            loss = torch.nn.functional.mse_loss(model(x), y)
            for acc, gr in zip(grads, torch.autograd.grad(loss, params)):
                acc += gr / 2
        for p, gr in zip(params, grads):
            p.grad = gr
        opt.step()
    return torch.cat([p.detach().flatten() for p in model.parameters()])


def test_flat_buffer_allreduce_two_ranks():
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (p0, l0, s0, nb0), (p1, l1, s1, nb1) = out[0], out[1]
    assert torch.equal(p0, p1)  # replicas stay identical
    torch.testing.assert_close(p0, _reference_run(), rtol=1e-5, atol=1e-6)  # == mean-gradient SGD
    assert l0 != l1  # ranks really saw different data
    assert sorted(s0 + s1) == list(range(10)) and not set(s0) & set(s1)
    assert nb0 == nb1 > 0


def test_grad_views_alias_flat_buffer():
    from vision_ft_amd.trainer.dp import FlatGradBuffer

    lin = torch.nn.Linear(4, 3)
    fb = FlatGradBuffer(list(lin.parameters()))
    lin(torch.ones(2, 4)).sum().backward()
    buf = fb.buffers[torch.float32]
    assert buf.abs().sum() > 0  # autograd accumulated straight into the flat buffer
    assert lin.weight.grad.data_ptr() == buf.data_ptr()
    fb.zero()
    assert lin.weight.grad.abs().sum() == 0


# ---------------------------------------------------------------------------------------------------------
# sharding when the world size does not divide the number of bucket batches, and the whole Trainer loop over gloo
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,world", [(36, 8), (10, 4), (3, 2), (8, 8), (5, 8), (17, 3)])
def test_shard_indices_equal_counts_with_wraparound(n, world):
    """Every rank takes ceil(n/W) batches (the last round wraps to the first batches): unequal shards would leave
    the longer ranks in an all-reduce while the others wait in the end-of-epoch barrier."""
    from vision_ft_amd.trainer.dp import shard_indices

    shards = [shard_indices(n, r, world) for r in range(world)]
    assert len({len(s) for s in shards}) == 1 and len(shards[0]) == -(-n // world)
    assert set().union(*map(set, shards)) == set(range(n))
    flat = [shards[r][i] for i in range(len(shards[0])) for r in range(world)]
    assert flat[:n] == list(range(n))                    # round-robin order r, r+W, ...
    assert flat[n:] == [i % n for i in range(n, len(flat))]  # the wrap re-uses the first batches in order
    try:  # accelerate (the reference's data-parallel layer) shards a batch-per-item loader the same way
        from accelerate.data_loader import BatchSamplerShard
        from torch.utils.data import BatchSampler, SequentialSampler
    except Exception:
        return
    for r in range(world):
        bs = BatchSamplerShard(BatchSampler(SequentialSampler(range(n)), batch_size=1, drop_last=False), num_processes=world,
                               process_index=r, even_batches=True)
        assert [b[0] for b in bs] == shards[r], (r, [b[0] for b in bs], shards[r])


def _toy_trainer(clip, n_batches: int = 3):
    """Trainer over a CPU toy task: `n_batches` bucket batches (3 for 2 ranks, 11 for 8: neither divides), clipping on."""
    import torch.utils.data as data

    from vision_ft_amd.config import TrainConfig
    from vision_ft_amd.dataset.util import DatasetConfig
    from vision_ft_amd.models.for_training import ModelForTraining
    from vision_ft_amd.trainer.common import Trainer
    from pydantic import BaseModel

    class ToyModelConfig(BaseModel):
        width: int = 6

    class ToyDataset(data.Dataset):
        def __len__(self):
            return n_batches

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(1000 + i)
            return {"x": torch.randn(4, 6, generator=g) * 3, "y": torch.randn(4, 2, generator=g) * 3, "idx": [i]}

    class ToyDatasetConfig(DatasetConfig):
        shuffle: bool = False

        def get_dataset(self):
            return ToyDataset()

    class ToyTask(ModelForTraining):
        model_config_class = ToyModelConfig
        seen: list

        def setup_model(self):
            torch.manual_seed(0)
            self.model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
            self.seen = []

        def sanity_check(self):
            pass

        def train_step(self, batch):
            self.seen += batch["idx"]
            return torch.nn.functional.mse_loss(self.model(batch["x"]), batch["y"])

    cfg = TrainConfig(model={}, dataset={}, optimizer={"name": "torch.optim.SGD", "args": {"lr": 0.05}}, saving=None,
                      trainer={"clip_grad_norm": clip, "hip_graph": False, "tuned_gemms": False}, num_train_epochs=2)
    tr = Trainer(cfg)
    tr.register_train_dataset_class(ToyDatasetConfig)
    tr.register_model_class(ToyTask)
    return tr


def _trainer_worker(rank: int, world: int, port: int, out, n_batches: int = 3):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr = _toy_trainer(clip=0.5, n_batches=n_batches)
        tr.train()
        flat = torch.cat([p.detach().flatten() for p in tr.raw_model.model.parameters()])
        out[rank] = (flat, list(tr.raw_model.seen), len(tr.train_dataloader), len(tr.losses))
    finally:
        dist.destroy_process_group()


def test_trainer_two_ranks_uneven_batches_clip_after_exchange():
    """3 batches / 2 ranks: both ranks take 2 steps per epoch (rank 1 wraps to batch 0); the clipped update is that of
    the AVERAGED gradient (reference: DDP reduces inside backward, clipping follows — src/trainer/common.py:355-359,
    src/models/for_training.py:98-109), so replicas stay identical and equal a one-process mean-gradient run."""
    mgr = mp.Manager()
    out = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_trainer_worker, args=(2, port, out), nprocs=2, join=True)
    (p0, seen0, n0, l0), (p1, seen1, n1, l1) = out[0], out[1]
    assert n0 == n1 == 2 and l0 == l1 == 4
    assert seen0 == [0, 2, 0, 2] and seen1 == [1, 0, 1, 0]
    assert torch.equal(p0, p1)
    # one process: mean of the two ranks' gradients, clipped once, SGD
    tr = _toy_trainer(clip=0.5)
    tr.before_train()
    model = tr.raw_model.model
    ds = tr.dataset_config.get_dataset()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    for _epoch in range(2):
        for pair in ([0, 1], [2, 0]):
            grads = [torch.zeros_like(p) for p in model.parameters()]
            for i in pair:
                b = ds[i]
                loss = torch.nn.functional.mse_loss(model(b["x"]), b["y"])
                for a, g in zip(grads, torch.autograd.grad(loss, list(model.parameters()))):
                    a += g / 2
            for p, g in zip(model.parameters(), grads):
                p.grad = g
            torch.nn.utils.clip_grad_norm_(list(model.parameters()), 0.5)
            opt.step()
    want = torch.cat([p.detach().flatten() for p in model.parameters()])
    torch.testing.assert_close(p0, want, rtol=1e-5, atol=1e-6)
    # clipping really was active, and clipping local gradients before the exchange would have given something else
    assert not torch.allclose(p0, _unclipped_reference(), rtol=1e-4, atol=1e-5)


def _unclipped_reference():
    tr = _toy_trainer(clip=None)
    tr.before_train()
    model = tr.raw_model.model
    ds = tr.dataset_config.get_dataset()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    for _epoch in range(2):
        for pair in ([0, 1], [2, 0]):
            opt.zero_grad()
            for i in pair:
                b = ds[i]
                (torch.nn.functional.mse_loss(model(b["x"]), b["y"]) / 2).backward()
            opt.step()
    return torch.cat([p.detach().flatten() for p in model.parameters()])


def test_trainer_eight_ranks_uneven_batches():
    """The driver's first real N = 8 run is the round-end scaling bench: rehearse the Trainer loop at world size 8 over gloo
    with a batch count 8 does not divide (11 -> two steps per rank and epoch, the second round wraps to batches 0..4).  Every
    rank must take the same number of steps (no rank left alone in an all-reduce), the replicas must stay bit-identical, and the
    weights must equal a one-process run that averages the eight ranks' gradients, clips once and steps
    (reference: src/trainer/common.py:287-365 under accelerate's DDP + even_batches sharding)."""
    from vision_ft_amd.trainer.dp import shard_indices

    world, n = 8, 11
    mgr = mp.Manager()
    out = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_trainer_worker, args=(world, port, out, n), nprocs=world, join=True)
    shards = [shard_indices(n, r, world) for r in range(world)]
    assert all(len(s) == 2 for s in shards)
    for r in range(world):
        flat, seen, nloader, nloss = out[r]
        assert nloader == 2 and nloss == 4 and seen == shards[r] * 2, (r, seen)
        assert torch.equal(flat, out[0][0])
    tr = _toy_trainer(clip=0.5, n_batches=n)
    tr.before_train()
    model = tr.raw_model.model
    ds = tr.dataset_config.get_dataset()
    params = list(model.parameters())
    opt = torch.optim.SGD(params, lr=0.05)
    for _epoch in range(2):
        for k in range(2):
            grads = [torch.zeros_like(p) for p in params]
            for r in range(world):
                b = ds[shards[r][k]]
                for a, g in zip(grads, torch.autograd.grad(torch.nn.functional.mse_loss(model(b["x"]), b["y"]), params)):
                    a += g / world
            for p, g in zip(params, grads):
                p.grad = g
            torch.nn.utils.clip_grad_norm_(params, 0.5)
            opt.step()
    torch.testing.assert_close(out[0][0], torch.cat([p.detach().flatten() for p in params]), rtol=1e-5, atol=1e-6)
