"""Train-time timestep samplers against fixtures generated from the reference
(oracle/gen_golden.py::gen_samplers): the stored raw draw is fed to the product functions by patching the
RNG call they make, so the comparison does not depend on RNG stream equality across torch builds/devices.
Value transforms are float32 elementwise math: bit-exact on CPU."""
import json

import numpy as np
import pytest
import torch

from oracle import ops

from .helpers import load_golden


@pytest.fixture(scope="module")
def S():
    T, meta = load_golden("samplers.safetensors")
    return T, {k: tuple(v) for k, v in json.loads(meta["shapes"]).items()}, json.loads(meta["divisible"])


CASES = [  # fixture prefix, product function name, rng call patched, shape tag, kwargs
    ("sigmoid_randn", "sigmoid_randn", "randn", "sq", {}),
    ("sigmoid_randn_s2", "sigmoid_randn", "randn", "sq", {"sigmoid_scale": 2.0}),
    ("shift_sigmoid_randn", "shift_sigmoid_randn", "randn", "sq", {}),
    ("flux_shift_randn_sq", "flux_shift_randn", "randn", "sq", {}),
    ("flux_shift_randn_small", "flux_shift_randn", "randn", "small", {}),
    ("scale_shift_sigmoid_randn", "scale_shift_sigmoid_randn", "randn", "sq", {}),
    ("uniform_rand", "uniform_rand", "rand", "sq", {}),
    ("shift_uniform_rand", "shift_uniform_rand", "rand", "sq", {}),
]


def _patch(monkeypatch, kind, draw):
    def fake(*a, **k):
        dev = k.get("device", None)
        return draw.clone().to(dev) if dev is not None else draw.clone()

    monkeypatch.setattr(torch, kind, fake)


@pytest.mark.parametrize("prefix,fn,kind,tag,kw", CASES)
def test_sampler_value_transforms_exact(S, monkeypatch, prefix, fn, kind, tag, kw):
    from vision_ft_amd.modules.timestep import sampling

    T, shapes, _ = S
    want = T[prefix + ".out"]
    # the oracle's restatement first ...
    got_o = ops.sampler_transform(fn, T[prefix + ".draw"], shapes[tag], **kw)
    assert torch.equal(got_o, want), prefix
    # ... then the product with the stored draw in place of its RNG call
    _patch(monkeypatch, kind, T[prefix + ".draw"])
    got = getattr(sampling, fn)(torch.Size(shapes[tag]), torch.device("cpu"), **kw)
    assert got.dtype == want.dtype and torch.equal(got, want), prefix


def test_uniform_randint_exact(S, monkeypatch):
    from vision_ft_amd.modules.timestep import sampling

    T, shapes, _ = S
    with monkeypatch.context() as mp:
        mp.setattr(torch, "randint", lambda *a, **k: T["uniform_randint.draw"].clone())
        got = sampling.uniform_randint(torch.Size(shapes["sq"]), torch.device("cpu"))
    assert got.dtype == torch.int32 and torch.equal(got, T["uniform_randint.out"])
    assert torch.equal(ops.sampler_transform("uniform_randint", T["uniform_randint.draw"]), T["uniform_randint.out"])
    # the range / dtype contract of the real call (reference :191-207: [min, max) int32 of shape (B,))
    real = sampling.uniform_randint(torch.Size((64, 4, 8, 8)), torch.device("cpu"), 10, 20)
    assert real.shape == (64,) and real.dtype == torch.int32 and int(real.min()) >= 10 and int(real.max()) < 20


@pytest.mark.parametrize("tag", ["sq", "tall", "small"])
def test_lumina_lognorm_sampler_exact(S, monkeypatch, tag):
    """Scheduler.sample_sigmoid_randn (reference src/models/lumina2/scheduler.py:48-72)."""
    from vision_ft_amd.models.lumina2 import Scheduler

    T, shapes, _ = S
    key = f"lumina_lognorm_{tag}"
    assert torch.equal(ops.sampler_transform("lumina_lognorm", T[key + ".draw"], shapes[tag], patch_size=2), T[key + ".out"])
    _patch(monkeypatch, "randn", T[key + ".draw"])
    got = Scheduler().sample_sigmoid_randn(torch.Size(shapes[tag]), torch.device("cpu"), patch_size=2)
    assert torch.equal(got, T[key + ".out"])


@pytest.mark.parametrize("prefix,typ,kw", [("dispatch_sigmoid", "sigmoid", {}), ("dispatch_shift", "shift_sigmoid", {"discrete_flow_shift": 2.0}),
                                           ("dispatch_flux", "flux_shift", {})])
def test_sample_timestep_dispatch(S, monkeypatch, prefix, typ, kw):
    from vision_ft_amd.modules.timestep import sampling

    T, shapes, _ = S
    _patch(monkeypatch, "randn", T[prefix + ".draw"])
    got = sampling.sample_timestep(torch.Size(shapes["sq"]), torch.device("cpu"), typ, **kw)
    assert torch.equal(got, T[prefix + ".out"])
    with pytest.raises(ValueError):
        sampling.sample_timestep(torch.Size(shapes["sq"]), torch.device("cpu"), "nope")


def test_fraction_samplers_exact(S, monkeypatch):
    """fraction_uniform_rand / shift_fraction_uniform_rand: candidate table (a set of floats, so its ORDER is part of
    the behaviour under a seeded numpy RNG) and the transform."""
    from vision_ft_amd.modules.timestep import sampling

    T, shapes, div = S
    assert np.array_equal(sampling._create_fraction(div), T["fraction.table"].numpy())
    assert np.array_equal(ops.create_fraction(div), T["fraction.table"].numpy())
    assert np.array_equal(sampling._create_fraction(list(range(20, 30))), T["fraction.table_default"].numpy())
    # same seeded numpy stream -> same picks (numpy's legacy global RNG is stable across versions) ...
    np.random.seed(12)
    got = sampling.fraction_uniform_rand(torch.Size(shapes["sq"]), torch.device("cpu"), divisible=div)
    assert torch.equal(got, T["fraction.out"])
    # ... and, independent of the stream, the transform on the stored picks
    monkeypatch.setattr(np.random, "choice", lambda *a, **k: T["fraction.draw"].numpy().copy())
    got = sampling.shift_fraction_uniform_rand(torch.Size(shapes["sq"]), torch.device("cpu"), shift=6.0, divisible=div)
    assert torch.equal(got, T["shift_fraction.out"])
    assert torch.equal(ops.sampler_transform("shift_fraction_uniform_rand", T["fraction.draw"], shift=6.0), T["shift_fraction.out"])


def test_lumina_task_timestep_sampling_switch(S, monkeypatch):
    """The `timestep_sampling` field of the Lumina2 task config (reference train/lumina2/text_to_image.py:32-37,144-171)."""
    import importlib.util
    import os
    import types

    T, shapes, div = S
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train", "lumina2", "text_to_image.py")
    spec = importlib.util.spec_from_file_location("vft_train_lumina2_t2i", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Cfg, Task = mod.Lumina2ForTextToImageTrainingConfig, mod.Lumina2ForTextToImageTraining
    assert Cfg(checkpoint_path="random").timestep_sampling == "uniform"
    with pytest.raises(Exception):
        Cfg(checkpoint_path="random", timestep_sampling="bogus")
    task = Task.__new__(Task)
    task.accelerator = types.SimpleNamespace(device=torch.device("cpu"))
    shape = torch.Size(shapes["tall"])
    task.model_config = Cfg(checkpoint_path="random", timestep_sampling="lognorm")
    _patch(monkeypatch, "randn", T["lumina_lognorm_tall.draw"])
    assert torch.equal(task.sample_timesteps(shape), T["lumina_lognorm_tall.out"])
    task.model_config = Cfg(checkpoint_path="random", timestep_sampling="uniform")
    _patch(monkeypatch, "rand", T["uniform_rand.draw"])
    assert torch.equal(task.sample_timesteps(shape), T["uniform_rand.out"])
    task.model_config = Cfg(checkpoint_path="random", timestep_sampling="shift_fraction_uniform", timestep_fraction_divisible=div)
    monkeypatch.setattr(np.random, "choice", lambda *a, **k: T["fraction.draw"].numpy().copy())
    assert torch.equal(task.sample_timesteps(shape), 1 - T["shift_fraction.out"])


@pytest.mark.gpu
def test_samplers_on_device(S, monkeypatch, gpu):
    """Same transforms on the MI355X (device sigmoid / exp: 2 ulp)."""
    from vision_ft_amd.models.lumina2 import Scheduler
    from vision_ft_amd.modules.timestep import sampling

    T, shapes, _ = S
    for prefix, fn, kind, tag, kw in CASES:
        _patch(monkeypatch, kind, T[prefix + ".draw"])
        got = getattr(sampling, fn)(torch.Size(shapes[tag]), gpu, **kw)
        assert got.is_cuda
        torch.testing.assert_close(got.cpu(), T[prefix + ".out"], rtol=3e-7, atol=3e-7)
    _patch(monkeypatch, "randn", T["lumina_lognorm_tall.draw"])
    got = Scheduler().sample_sigmoid_randn(torch.Size(shapes["tall"]), gpu)
    torch.testing.assert_close(got.cpu(), T["lumina_lognorm_tall.out"], rtol=3e-7, atol=3e-7)
    monkeypatch.undo()
    t = sampling.uniform_randint(torch.Size((4096, 4, 8, 8)), gpu)
    assert t.dtype == torch.int32 and 0 <= int(t.min()) and int(t.max()) < 1000 and t.float().mean().item() == pytest.approx(499.5, abs=30)
