"""CPU: real-image front half of train_step — vision_ft_amd.dataset.{transform,text_to_image} against
oracle/image_ops.py.  Integer work (cover size, bucket choice, crop window, SDXL size conditions) is exact; the
antialiased bicubic resample must agree with the numpy restatement to one fp16 ulp of the -1..1 range (4.9e-4) —
the resampling numerics are "parity unpinned" (torchvision absent, see oracle/image_ops.py)."""
import json
import warnings

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import image_ops as oimg
from vision_ft_amd.dataset.aspect_ratio_bucket import AspectRatioBucketManager
from vision_ft_amd.dataset.text_to_image import TextToImageDatasetConfig
from vision_ft_amd.dataset.transform import ObjectCoverResize, pil_to_tensor, random_crop_params, to_normalised_half

FP16_ULP = 4.9e-4


def _rand_u8(rng, h, w):
    # smooth + noise, so the resample is tested on structure as well as on the Nyquist band
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 100 * np.sin(xx / 7.0)[None] * np.cos(yy / 5.0)[None] * np.array([1.0, 0.6, -0.8])[:, None, None]
    return np.clip(base + rng.normal(0, 20, (3, h, w)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("hw,target", [((97, 141), (64, 48)), ((300, 200), (128, 192)), ((64, 64), (64, 64)), ((131, 517), (256, 128))])
def test_cover_resize_matches_restatement(hw, target):
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    u8 = _rand_u8(rng, *hw)
    x = to_normalised_half(torch.from_numpy(u8))
    assert np.array_equal(x.numpy(), oimg.normalise_half(u8))  # exact: same fp16 roundings
    tw, th = target
    out = ObjectCoverResize(tw, th)(x)
    sh, sw = oimg.cover_size(hw[1], hw[0], tw, th)
    assert out.shape == (3, sh, sw) and out.dtype == torch.float16
    assert sh >= th and sw >= tw and (sh == th or sw == tw)  # covers the bucket, one side tight
    ref = oimg.resize_bicubic_aa(oimg.normalise_half(u8), sh, sw).astype(np.float16).astype(np.float32)
    assert np.abs(out.float().numpy() - ref).max() <= FP16_ULP


def test_cover_resize_rejects_small_images_unless_upscaling():
    x = torch.zeros(3, 40, 200, dtype=torch.float16)
    with pytest.raises(ValueError, match="too small"):
        ObjectCoverResize(64, 64)(x)
    out = ObjectCoverResize(64, 64, do_upscale=True)(x)
    assert out.shape == (3, 64, 320)


def test_random_crop_params_draw_order_and_bounds():
    img = torch.zeros(3, 100, 160)
    torch.manual_seed(5)
    top, left, h, w = random_crop_params(img, (64, 96))
    torch.manual_seed(5)
    assert top == int(torch.randint(0, 37, (1,))) and left == int(torch.randint(0, 65, (1,)))  # top first, then left
    assert (h, w) == (64, 96)
    assert random_crop_params(torch.zeros(3, 64, 96), (64, 96)) == (0, 0, 64, 96)  # exact fit: no draw
    with pytest.raises(ValueError):
        random_crop_params(torch.zeros(3, 10, 10), (64, 96))


@pytest.fixture()
def image_folder(tmp_path):
    rng = np.random.default_rng(3)
    sizes = {"a": (300, 260), "b": (512, 300), "c": (280, 520), "d": (700, 700), "e": (100, 90), "f": (320, 256)}  # (w, h)
    for name, (w, h) in sizes.items():
        Image.fromarray(_rand_u8(rng, h, w).transpose(1, 2, 0)).save(tmp_path / f"{name}.png")
        if name == "b":
            (tmp_path / "b.json").write_text(json.dumps({"caption": "from metadata"}))
        elif name == "f":
            (tmp_path / "f.json").write_text(json.dumps({"caption": "skipped", "skip": True}))
        else:
            (tmp_path / f"{name}.txt").write_text(f"caption of {name}")
    return tmp_path, sizes


def test_text_to_image_buckets_batches_and_size_conditions(image_folder):
    folder, sizes = image_folder
    cfg = TextToImageDatasetConfig(folder=str(folder), batch_size=2, bucket_base_size=256, step=64, min_size=128, num_workers=0)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        buckets = cfg.generate_buckets()
    assert any("too small" in str(w.message) for w in caught)  # the 100x90 image fits no bucket
    manager = AspectRatioBucketManager(cfg.buckets)
    expect = {}
    for name, (w, h) in sizes.items():
        if name in ("e", "f"):  # too small / "skip": true
            continue
        bw, bh = (int(v) for v in cfg.buckets[manager.find_nearest(w, h)])
        expect.setdefault((bw, bh), []).append(name)
    got = {(b.width, b.height): sorted(row["image"].rsplit("/", 1)[1][0] for row in b.items) for b in buckets}
    assert got == {k: sorted(v) for k, v in expect.items()}

    ds = cfg.get_dataset()
    torch.manual_seed(11)
    first = [ds[i] for i in range(len(ds))]
    torch.manual_seed(11)
    again = [ds[i] for i in range(len(ds))]
    for b1, b2 in zip(first, again):
        assert torch.equal(b1["image"], b2["image"]) and torch.equal(b1["crop_coords_top_left"], b2["crop_coords_top_left"])
    seen = set()
    for batch in first:
        img = batch["image"]
        assert img.dtype == torch.float16 and img.shape[0] == 2 and len(batch["caption"]) == 2  # always full (wrap-around)
        bh, bw = img.shape[-2:]
        assert (bw, bh) in expect
        for i in range(2):
            w, h = batch["width"][i], batch["height"][i]
            sh, sw = oimg.cover_size(w, h, bw, bh)
            top, left = (int(v) for v in batch["crop_coords_top_left"][i])
            want = oimg.size_conditions((sh, sw), top, left, (bh, bw))
            assert batch["original_size"][i].tolist() == want["original_size"]   # the RESIZED size, (h, w)
            assert batch["target_size"][i].tolist() == want["target_size"]
            assert 0 <= top <= sh - bh and 0 <= left <= sw - bw
            # the crop is the window of the resampled image, against the restatement
            name = next(n for n, s in sizes.items() if s == (w, h))
            u8 = pil_to_tensor(Image.open(folder / f"{name}.png")).numpy()
            ref = oimg.resize_bicubic_aa(oimg.normalise_half(u8), sh, sw).astype(np.float16).astype(np.float32)
            assert np.abs(img[i].float().numpy() - ref[:, top:top + bh, left:left + bw]).max() <= FP16_ULP
            seen.add(batch["caption"][i])
    assert seen == {"caption of a", "from metadata", "caption of c", "caption of d"}


def test_missing_caption_raises(tmp_path):
    Image.new("RGB", (300, 300)).save(tmp_path / "x.png")
    cfg = TextToImageDatasetConfig(folder=str(tmp_path), batch_size=1, bucket_base_size=256, step=64, min_size=128)
    with pytest.raises(FileNotFoundError):
        cfg.generate_buckets()


def _stand_in_encoders():
    def image_encoder(px):  # stand-in for vae.encode(px).sample() * scaling_factor: (B,3,H,W) -> (B,4,H/8,W/8)
        lat = torch.nn.functional.avg_pool2d(px.float(), 8)
        return torch.cat([lat, lat[:, :1]], dim=1)

    def prompt_encoder(captions):  # deterministic per caption text
        outs = []
        for c in captions:
            g = torch.Generator().manual_seed(sum(c.encode()))
            outs.append({"ehs": torch.randn(77, 16, generator=g), "pooled": torch.randn(24, generator=g)})
        return {k: torch.stack([o[k] for o in outs]) for k in outs[0]}

    return image_encoder, prompt_encoder


def test_latent_cache_reproduces_the_front_half(image_folder, tmp_path_factory):
    """SURVEY §8 f3, the latent-cache alternative: the frozen encoders run ONCE over TextToImageDatasetConfig
    (dataset/latent_cache.py); training then reads the same bucket batches from the cache.  Against the front-half
    fixtures: every cached entry equals encoder(crop window of the cover-resized image) for its recorded crop, the size
    conditions are the ones the restatement gives for that window, and the cached dataset emits full, wrap-around batches
    of one bucket each with the keys the tasks consume."""
    from vision_ft_amd.dataset.latent_cache import (CachedLatentDatasetConfig, LatentCacheWriter, entry_name,
                                                    split_text_conditions)

    folder, sizes = image_folder
    cache = tmp_path_factory.mktemp("latent_cache")
    cfg = TextToImageDatasetConfig(folder=str(folder), batch_size=2, bucket_base_size=256, step=64, min_size=128, num_workers=0)
    image_encoder, prompt_encoder = _stand_in_encoders()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        index = LatentCacheWriter(str(cache), image_encoder, prompt_encoder, crops_per_image=3, seed=7).build(cfg)
        again = LatentCacheWriter(str(cache), image_encoder, prompt_encoder, crops_per_image=3, seed=7).build(cfg)
    assert index == again  # seeded per file: rebuilding the cache draws the same windows
    files = {e["file"] for e in index["entries"]}
    assert files == {"a.png", "b.png", "c.png", "d.png"}  # e: too small, f: "skip": true — as in the image dataset
    manager = AspectRatioBucketManager(cfg.buckets)
    from safetensors.torch import load_file

    for e in index["entries"]:
        w, h = sizes[e["file"][0]]
        bw, bh = (int(v) for v in cfg.buckets[manager.find_nearest(w, h)])
        assert e["bucket"] == [bw, bh] and (e["width"], e["height"]) == (w, h)
        sh, sw = oimg.cover_size(w, h, bw, bh)
        top, left = e["crop"]
        want = oimg.size_conditions((sh, sw), top, left, (bh, bw))
        assert e["original_size"] == want["original_size"] and e["target_size"] == want["target_size"]
        assert 0 <= top <= sh - bh and 0 <= left <= sw - bw
        assert e["entry"] == entry_name(e["file"], bw, bh, top, left)   # keyed by file + crop
        t = load_file(str(cache / e["entry"]))
        u8 = pil_to_tensor(Image.open(folder / e["file"])).numpy()
        ref = oimg.resize_bicubic_aa(oimg.normalise_half(u8), sh, sw).astype(np.float16).astype(np.float32)
        window = torch.from_numpy(ref[:, top:top + bh, left:left + bw])
        assert t["latents"].shape == (4, bh // 8, bw // 8) and t["latents"].dtype == torch.bfloat16
        # bf16 storage (8 bits) on top of the one-fp16-ulp resample agreement
        assert (t["latents"].float() - image_encoder(window[None])[0]).abs().max() <= 2 ** -8 + FP16_ULP
        text = prompt_encoder([e["caption"]])
        assert torch.equal(t["text.ehs"], text["ehs"][0]) and torch.equal(t["text.pooled"], text["pooled"][0])

    ccfg = CachedLatentDatasetConfig(cache_folder=str(cache), batch_size=2, num_workers=0)
    ds = ccfg.get_dataset()
    torch.manual_seed(3)
    batches = [ds[i] for i in range(len(ds))]
    by_entry = {(e["file"], tuple(e["crop"])): e for e in index["entries"]}
    captions = set()
    for b in batches:
        assert b["latents"].shape[0] == 2 and len(b["caption"]) == 2   # always full (wrap-around), one bucket per batch
        bh, bw = b["latents"].shape[-2] * 8, b["latents"].shape[-1] * 8
        text = split_text_conditions(b)
        assert set(text) == {"ehs", "pooled"} and text["ehs"].shape == (2, 77, 16)
        for i in range(2):
            name = next(n for n, s in sizes.items() if s == (b["width"][i], b["height"][i])) + ".png"
            e = by_entry[(name, tuple(int(v) for v in b["crop_coords_top_left"][i]))]
            assert e["bucket"] == [bw, bh] and b["original_size"][i].tolist() == e["original_size"]
            assert b["target_size"][i].tolist() == [bh, bw] and b["caption"][i] == e["caption"]
            assert torch.equal(b["latents"][i], load_file(str(cache / e["entry"]))["latents"])
            captions.add(b["caption"][i])
    assert captions == {"caption of a", "from metadata", "caption of c", "caption of d"}
    # the bucket collate of the loader (one item = one batch) leaves the batch as it is
    from vision_ft_amd.dataloader import concatnate_collate_fn

    merged = concatnate_collate_fn([batches[0]])
    assert torch.equal(merged["latents"], batches[0]["latents"]) and merged["caption"] == batches[0]["caption"]
