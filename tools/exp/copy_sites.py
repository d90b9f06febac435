"""Which Python call sites make layout copies (.contiguous() / reshape of non-contiguous tensors) during one step?"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from vision_ft_amd.trainer.dp import DataParallelGrads  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "aura_lora"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model = bench.build(wl, dev)
params = [p for p in model.parameters() if p.requires_grad]
dp = DataParallelGrads(params)
fwd = bench.make_step(wl, model, dev, batch, seed=42)
fwd().backward()
sites = collections.Counter()
orig_c, orig_r, orig_to = torch.Tensor.contiguous, torch.Tensor.reshape, torch.Tensor.to


def site():
    fr = [f for f in traceback.extract_stack()[:-2] if "vision_ft_amd" in f.filename or "bench.py" in f.filename]
    return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])


def contiguous(self, *a, **k):
    if self.numel() > 1 << 20 and not self.is_contiguous(*a, **k):
        sites[("contiguous", tuple(self.shape), site())] += 1
    return orig_c(self, *a, **k)


def reshape(self, *a, **k):
    out = orig_r(self, *a, **k)
    if self.numel() > 1 << 20 and out.data_ptr() != self.data_ptr():
        sites[("reshape-copy", tuple(self.shape), site())] += 1
    return out


def to(self, *a, **k):
    out = orig_to(self, *a, **k)
    if self.numel() > 1 << 20 and out.data_ptr() != self.data_ptr():
        sites[("to-copy", tuple(self.shape), site())] += 1
    return out


torch.Tensor.contiguous, torch.Tensor.reshape, torch.Tensor.to = contiguous, reshape, to
fwd().backward()
torch.cuda.synchronize()
for (kind, shape, s), n in sites.most_common(40):
    print(f"{n:4d} {kind:14s} {str(shape):28s} {s}")
