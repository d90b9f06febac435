"""CPU: the NF4 host logic (quantiser vs the numpy oracle bit-exact, bitsandbytes key layout, checkpoint-key sniffing —
the checks the reference's tests/test_modules_quant.py:21-70,153-193 make) and the channels-last gradient-buffer views
of the data-parallel engine.  No GPU, no compute kernels."""
import json

import numpy as np
import pytest
import torch

from oracle import nf4 as onf4


@pytest.mark.parametrize("double_quant", [False, True])
def test_quantize_nf4_bit_exact_vs_oracle(double_quant):
    from vision_ft_amd.modules.quant import quantize_nf4

    g = torch.Generator().manual_seed(2)
    w = torch.randn(192, 256, generator=g) * 0.05
    got = quantize_nf4(w, double_quant=double_quant)
    want = onf4.quantize_nf4(w.numpy(), 64, double_quant)
    assert got[""].dtype == torch.uint8 and got[""].shape == (192 * 256 // 2, 1)
    assert np.array_equal(got[""].numpy(), want[""])  # codes + packing
    assert np.array_equal(got[".quant_map"].numpy(), want[".quant_map"])
    assert np.array_equal(got[".absmax"].numpy(), want[".absmax"])
    meta = json.loads(bytes(got[".quant_state.bitsandbytes__nf4"].tolist()).decode())
    assert meta["quant_type"] == "nf4" and meta["blocksize"] == 64 and meta["shape"] == [192, 256]
    # dequantising the oracle's way reproduces the weight to NF4 resolution
    back = onf4.dequantize_nf4({k: v.numpy() for k, v in got.items()})
    assert np.abs(back - w.numpy()).max() <= np.abs(w.numpy()).reshape(-1, 64).max(1).max() * 0.2


def test_quant_key_layout_and_sniffing():
    from vision_ft_amd.modules.quant import NF4Linear, get_quant_type_from_children_dict, quantize_nf4
    from vision_ft_amd.modules.quant.functional import quantize_state_dict, replace_by_prequantized_weights

    lin = torch.nn.Sequential(torch.nn.Linear(128, 256), torch.nn.Linear(256, 64))
    sd = {k: v.detach().clone() for k, v in lin.state_dict().items()}
    q = quantize_state_dict(dict(sd), "bnb_nf4", include_keys=["0.weight"], double_quant=True)
    assert q["0.weight"].dtype == torch.uint8 and q["1.weight"].dtype == torch.float32  # only the targeted layer
    suffixes = sorted(k[len("0.weight"):] for k in q if k.startswith("0.weight"))
    assert suffixes == ["", ".absmax", ".nested_absmax", ".nested_quant_map", ".quant_map", ".quant_state.bitsandbytes__nf4"]
    children = {k[len("0.weight."):]: v for k, v in q.items() if k.startswith("0.weight.")}
    assert get_quant_type_from_children_dict(children) == "bnb_nf4"
    with pytest.raises(ValueError):  # as the reference: unknown key sets are an error, not 'unquantised'
        get_quant_type_from_children_dict({})
    replace_by_prequantized_weights(lin, q)  # checkpoint sniffing swaps exactly the quantised layer
    assert isinstance(lin[0], NF4Linear) and type(lin[1]) is torch.nn.Linear
    lin.load_state_dict(q, strict=True)
    out = lin[0].state_dict()
    assert set(out) == {"weight", "bias", "weight.absmax", "weight.quant_map", "weight.nested_absmax",
                        "weight.nested_quant_map", "weight.quant_state.bitsandbytes__nf4"}
    assert torch.equal(out["weight"], q["0.weight"])  # payload round-trips bit-exactly
    with pytest.raises(RuntimeError):  # a float checkpoint is not silently accepted by an NF4 layer
        NF4Linear(128, 256).load_state_dict({"weight": torch.zeros(256, 128), "bias": torch.zeros(256)})
    with pytest.raises(NotImplementedError):
        quantize_state_dict(dict(sd), "bnb_int8", include_keys=["0.weight"])
    assert set(quantize_nf4(sd["0.weight"], double_quant=False)) == {"", ".absmax", ".quant_map", ".quant_state.bitsandbytes__nf4"}


def test_flat_gradient_buffer_follows_parameter_layout():
    """Gradient views into the flat buffer take the parameter's dense layout (contiguous or channels-last), so fused
    optimizers see matching strides after UNet.to_channels_last(); the re-striding keeps accumulated values."""
    from vision_ft_amd.models.sdxl.denoiser import UNet
    from vision_ft_amd.trainer.dp import FlatGradBuffer

    conv = torch.nn.Conv2d(8, 16, 3, padding=1)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    lin = torch.nn.Linear(8, 8)
    buf = FlatGradBuffer([conv.weight, conv.bias, lin.weight])
    assert conv.weight.grad.stride() == conv.weight.stride() and lin.weight.grad.is_contiguous()
    flat = next(iter(buf.buffers.values()))
    conv.weight.grad.fill_(2.0)
    assert float(flat.sum()) == 2.0 * conv.weight.numel()  # views alias the buffer
    # a parameter converted after the buffer was built (UNet.to_channels_last): values survive the re-striding
    net = torch.nn.Module()
    net.c = torch.nn.Conv2d(4, 8, 3, padding=1)
    fb = FlatGradBuffer([net.c.weight])
    net.c.weight.grad.copy_(torch.arange(net.c.weight.numel(), dtype=torch.float32).view_as(net.c.weight))
    before = net.c.weight.grad.clone()
    UNet.to_channels_last(net)
    assert net.c.weight.is_contiguous(memory_format=torch.channels_last)
    assert net.c.weight.grad.stride() == net.c.weight.stride() and torch.equal(net.c.weight.grad, before)
    assert net.c.weight.grad.untyped_storage().data_ptr() == next(iter(fb.buffers.values())).untyped_storage().data_ptr()


def test_nf4_linear_dtype_only_conversion_keeps_the_payload_where_it_is():
    """`.to(dtype)` / `.float()` on a model holding NF4 layers must not move the packed weight (it used to be probed
    with a CPU tensor, which dragged a GPU-resident payload to the CPU); device moves still move everything."""
    import torch

    from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4

    w = torch.randn(64, 128)
    layer = NF4Linear(128, 64, bias=True)
    layer.load_state_dict({**{"weight" + k: v for k, v in quantize_nf4(w, double_quant=True).items()}, "bias": torch.zeros(64)})
    packed = layer.weight.clone()
    layer.to(torch.float32)
    assert layer.weight.dtype == torch.uint8 and torch.equal(layer.weight, packed) and layer.bias.dtype == torch.float32
    layer.to(torch.bfloat16)
    assert layer.weight.device.type == "cpu" and layer.absmax.device.type == "cpu" and layer.bias.dtype == torch.bfloat16
    m = layer.to("meta")
    assert m.weight.device.type == "meta" and m.absmax.device.type == "meta" and m.quant_map.device.type == "meta"
    m.to(torch.float32)  # dtype-only conversion of a meta-resident layer: nothing is copied out of meta
    assert m.weight.device.type == "meta" and m.weight.dtype == torch.uint8
