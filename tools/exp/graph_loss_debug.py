"""Diagnostic: the static loss tensor of a graph-replayed SDXL-LoRA step at bench size (round 4: it read negative).
   python tools/exp/graph_loss_debug.py --batch 24 [--mode full|fwd|fwdbwd] [--set kernels.NAME=VALUE ...] [--attn-variant N]"""
import argparse
import importlib
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--mode", default="full")
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--attn-variant", type=int, default=0)
ap.add_argument("--lin-variant", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from vision_ft_amd import _C, kernels  # noqa: E402
from vision_ft_amd.trainer.dp import DataParallelGrads  # noqa: E402
from vision_ft_amd.trainer.graph import GraphedTrainStep  # noqa: E402

for kv in args.set:
    name, val = kv.split("=")
    mod, attr = name.rsplit(".", 1)
    m = importlib.import_module("vision_ft_amd." + mod)
    setattr(m, attr, type(getattr(m, attr))(int(val)))
if args.attn_variant:
    _C.lib().vft_attn_variant(args.attn_variant)
if args.lin_variant:
    _C.lib().vft_linear_variant(args.lin_variant)
model = bench.build("sdxl_lora", dev)
params = [p for p in model.parameters() if p.requires_grad]
dp = DataParallelGrads(params)
opt = torch.optim.AdamW(params, lr=1e-3, fused=True, capturable=True)
fwd = bench.make_step("sdxl_lora", model, dev, args.batch, seed=42)
tag = f"B={args.batch} mode={args.mode} set={args.set} attn={args.attn_variant} lin={args.lin_variant}"
def owners(ptr):
    """Which long-lived buffers we know of contain address `ptr`?"""
    out = []
    def chk(name, t):
        if t is not None and torch.is_tensor(t) and t.is_cuda and t.numel():
            a = t.untyped_storage().data_ptr(); b = a + t.untyped_storage().nbytes()
            if a <= ptr < b:
                out.append(f"{name} [{a:#x},{b:#x}) +{ptr - a}")
    for i, p_ in enumerate(params):
        chk(f"param{i}", p_); chk(f"grad{i}", p_.grad)
        for k, v in opt.state.get(p_, {}).items():
            chk(f"opt.{k}{i}", v)
    for n_, p_ in model.named_parameters():
        chk("w:" + n_, p_)
    for n_, b_ in model.named_buffers():
        chk("buf:" + n_, b_)
    for q in kernels._WgradQueue.queues.values():
        chk("wgrad.table", q.table); chk("wgrad.arena", q.arena)
        for j, r_ in enumerate(q.retired):
            chk(f"wgrad.retired{j}", r_)
    for dt, b_ in dp.flat.buffers.items():
        chk(f"flat.{dt}", b_)
    pk = getattr(model, "_adapter_pack", None) or getattr(model, "adapter_pack", None)
    if pk is not None:
        chk("pack.flat", getattr(pk, "_flat", None)); chk("pack.table", getattr(pk, "_table", None))
    return out


if args.mode in ("full", "full_split"):
    torch.cuda.memory._record_memory_history(max_entries=2000000)
    step = GraphedTrainStep(fwd, opt, dp, warmup=2, split=(args.mode == "full_split"))
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    base_ = step._raw_loss.data_ptr()
    hits = []
    for tr in snap.get("device_traces", []):
        for ev in tr:
            a, sz = ev.get("addr", 0), ev.get("size", 0)
            if a <= base_ < a + sz:
                fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in ev.get("frames", []) if "vision_ft_amd" in f["filename"] or "bench" in f["filename"] or "torch/nn/functional" in f["filename"]][:4]
                hits.append((ev["action"], a - base_, sz, ev.get("stream"), fr))
    L = step._raw_loss
    print(f"[{tag}] loss: storage nbytes {L.untyped_storage().nbytes()} offset {L.storage_offset()} storage ptr {L.untyped_storage().data_ptr():#x} data_ptr {L.data_ptr():#x} is_view {L._is_view()} base {None if L._base is None else (tuple(L._base.shape), L._base.dtype)}", flush=True)
    n_ev = sum(len(tr) for tr in snap.get("device_traces", []))
    print(f"[{tag}] trace entries {n_ev}", flush=True)
    for tr in snap.get("device_traces", []):
        for ev in tr:
            if ev.get("size", 0) <= 512 and ev["action"] == "alloc" and any("mse_loss" in f["name"] for f in ev.get("frames", [])):
                print(f"[{tag}]   small alloc in mse_loss: addr {ev['addr']:#x} size {ev['size']} stream {ev.get('stream')}", flush=True)
    for seg in snap.get("segments", []):
        if seg["address"] <= base_ < seg["address"] + seg["total_size"]:
            print(f"[{tag}] segment {seg['address']:#x} size {seg['total_size']} pool {seg.get('segment_pool_id')} type {seg.get('segment_type')} stream {seg.get('stream')}", flush=True)
            off = seg["address"]
            for b in seg["blocks"]:
                if off <= base_ < off + b["size"] or abs(off - base_) < (8 << 20):
                    print(f"[{tag}]    block +{off - base_} size {b['size']} state {b['state']} req {b.get('requested_size')}", flush=True)
                off += b["size"]
    print(f"[{tag}] allocator events covering the loss address ({len(hits)}):", flush=True)
    for h in hits[-6:]:
        print(f"[{tag}]    {h}", flush=True)
    print(f"[{tag}] loss tensor at {step.loss.data_ptr():#x}; owners: {owners(step.loss.data_ptr())}", flush=True)
    for i in range(2):
        loss = step()
        print(f"[{tag}] replay {i}: loss {float(loss):.5f}", flush=True)
    import time
    t0 = time.perf_counter()
    loss = step(); print(f"[{tag}] after perf_counter: loss {float(loss):.5f}", flush=True)
    kernels.set_profiling(False)
    loss = step(); print(f"[{tag}] after set_profiling(False): loss {float(loss):.5f}", flush=True)
    time.sleep(2.0)
    loss = step(); print(f"[{tag}] after sleep(2): loss {float(loss):.5f} raw {float(step._raw_loss):.5f}", flush=True)
    torch.cuda.current_stream().synchronize()
    loss = step(); print(f"[{tag}] after stream synchronize: loss {float(loss):.5f} raw {float(step._raw_loss):.5f}", flush=True)
    torch.cuda.synchronize()
    loss = step(); print(f"[{tag}] after synchronize: loss {float(loss):.5f} raw {float(step._raw_loss):.5f}", flush=True)
    loss = step(); print(f"[{tag}] next: loss {float(loss):.5f} raw {float(step._raw_loss):.5f}", flush=True)
    import ctypes, numpy as np
    hip = ctypes.CDLL("libamdhip64.so")
    def peek(addr, nbytes):
        buf = (ctypes.c_ubyte * nbytes)()
        rc = hip.hipMemcpy(buf, ctypes.c_void_p(addr), ctypes.c_size_t(nbytes), 2)
        return rc, np.frombuffer(bytes(buf), dtype=np.float32)
    torch.cuda.synchronize()
    base = step._raw_loss.data_ptr()
    rc, v = peek(base - 0x3000, 0x3000 + 0x2000)
    nz = np.flatnonzero(v)
    print(f"[{tag}] peek rc={rc}; nonzero float32 words in [base-0x3000, base+0x2000): {len(nz)}", flush=True)
    for off in range(0, len(v), 128):  # one line per 512-byte block that has content
        blk = v[off:off + 128]
        if np.any(blk):
            print(f"[{tag}]   +{off * 4 - 0x3000:#07x}: first8 {np.array2string(blk[:8], precision=4)} nonzero {int(np.count_nonzero(blk))}", flush=True)
    for i in range(3):
        loss = step()
    print(f"[{tag}] three replays without a read in between, then: loss {float(loss):.5f}", flush=True)
elif args.mode == "fwdbwdopt":
    for _ in range(2):
        l_ = fwd(); l_.backward(); opt.step(); dp.zero_grad()
    torch.cuda.synchronize()
    import gc
    gc.collect(); torch.cuda.empty_cache()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = fwd()
        mid = loss.detach().clone()
        loss.backward()
        mid2 = loss.detach().clone()
        opt.step()
        mid3 = loss.detach().clone()
        dp.zero_grad()
        out = loss.detach()
    print(f"[{tag}] loss tensor at {out.data_ptr():#x}; owners: {owners(out.data_ptr())}", flush=True)
    for i in range(6):
        g.replay()
        if i >= 3:
            torch.cuda.synchronize()
        print(f"[{tag}] replay {i}{' (device sync first)' if i >= 3 else ''}: loss {float(out):.5f}  after fwd {float(mid):.5f}  after bwd {float(mid2):.5f}  after opt {float(mid3):.5f}", flush=True)
else:
    for _ in range(2):
        l_ = fwd()
        if args.mode == "fwdbwd":
            l_.backward()
    torch.cuda.synchronize()
    print(f"[{tag}] eager loss {float(l_):.5f}", flush=True)
    import gc
    gc.collect(); torch.cuda.empty_cache()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = fwd()
        mid = loss.detach().clone()  # captured right behind the loss computation
        if args.mode == "fwdbwd":
            loss.backward()
        out = loss.detach()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print(f"[{tag}] replay {i}: loss {float(out):.5f}  (copy taken right after the forward: {float(mid):.5f})", flush=True)
