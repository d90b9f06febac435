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
