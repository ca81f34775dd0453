"""Dataset adapters (tdvc_amd/data.py) on synthetic PNG trees: the tree layout, lambda -> QP map, GOP enumeration and
return tuples of main/dataloader/dataset.py:16-190; the Vimeo sample rule (:211-240); the shared augmentation draw."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from tdvc_amd import data, synth


def _png(path, arr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def _frames(seed, T, H, W):
    g = synth.make_gop(seed, T, H, W)
    return [(f.permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8) for f in g]


def _tree(root, seqs, nframes, gop, qp, H=48, W=64):
    for s, name in enumerate(seqs):
        fr = _frames(40 + s, nframes, H, W)
        for i, f in enumerate(fr):
            _png(os.path.join(root, "ori_img", name, f"im{i + 1:03d}.png"), f)
        for k in range(nframes // gop):
            stem = os.path.join(root, "compress_img_bpg", name, str(qp), f"im{k * gop + 1:03d}_{qp}")
            _png(stem + ".png", np.clip(fr[k * gop].astype(np.int16) + 3, 0, 255).astype(np.uint8))
            with open(stem + ".txt", "w") as f:
                f.write(f"{0.1 * (k + 1) + s:.4f}\nsecond line is ignored\n")
    return root


def test_uvg_tree_and_item(tmp_path):
    root = _tree(str(tmp_path), ["Beauty", "Jockey10", "Jockey2"], nframes=16, gop=8, qp=27)
    ds = data.UVGDataSet(root, 2048, 8, testfull=True, isTrain=False, compute_ref_metrics=False)
    assert ds.qp == 27 and len(ds) == 3 * 2
    assert [os.path.basename(os.path.dirname(os.path.dirname(p))) for p in ds.ref[::2]] == ["Beauty", "Jockey2", "Jockey10"]      # natural order
    assert ds.ref[1].endswith(os.path.join("Beauty", "27", "im009_27.png")) and ds.refbpp[1] == pytest.approx(0.2)
    assert [os.path.basename(p) for p in ds.input[1]] == [f"im{i:03d}.png" for i in range(9, 17)]
    inp, ref, bpp, psnr, msssim, names, raw = ds[1]
    assert inp.shape == (7, 3, 48, 64) and raw.shape == (8, 3, 48, 64) and ref.shape == (3, 48, 64) and names == ds.input[1]
    assert inp.dtype == np.float32 and float(raw.max()) <= 1.0
    want = np.asarray(Image.open(ds.input[1][3]).convert("RGB"), dtype=np.float32).transpose(2, 0, 1) / 255.0
    np.testing.assert_array_equal(raw[3], want)
    np.testing.assert_array_equal(inp[2], raw[3])                      # P-frame inputs = raw frames 1..7
    assert 30.0 < psnr < 45.0                                          # the "+3" I-frame
    # not testfull: always 8 GOPs per sequence are listed (dataset.py:44-45) -- here that runs past the files
    with pytest.raises(FileNotFoundError):
        data.UVGDataSet(root, 2048, 8, testfull=False, isTrain=False, compute_ref_metrics=False)
    with pytest.raises(ValueError):
        data.UVGDataSet(root, 777, 8)
    assert {l: data.LAMBDA_TO_QP[l] for l in (512, 1024, 2048, 4096, 16, 32, 64, 128)} == {512: 37, 1024: 32, 2048: 27, 4096: 22, 16: 37, 32: 32, 64: 27, 128: 22}


def test_uvg_train_mode_resizes(tmp_path):
    root = _tree(str(tmp_path), ["Bosphorus"], nframes=8, gop=8, qp=32)
    ds = data.UVGDataSet(root, 1024, 8, testfull=True, isTrain=True, compute_ref_metrics=False)
    inp, ref, *_ = ds[0]
    assert inp.shape == (7, 3, 256, 256) and ref.shape == (3, 256, 256)


def test_hevc_class_filter(tmp_path):
    root = _tree(str(tmp_path), ["BasketballPass_416x240_50", "RaceHorses_416x240_30", "RaceHorses_832x480_30", "Cactus_1920x1080_50"],
                 nframes=10, gop=10, qp=27)
    d = data.HEVCDataSet(root, 2048, 10, "D", testfull=True, isTrain=False, compute_ref_metrics=False)
    assert sorted(os.path.basename(os.path.dirname(os.path.dirname(p))) for p in d.ref) == ["BasketballPass_416x240_50", "RaceHorses_416x240_30"]
    c = data.HEVCDataSet(root, 2048, 10, "C", testfull=True, isTrain=False, compute_ref_metrics=False)
    assert [os.path.basename(os.path.dirname(os.path.dirname(p))) for p in c.ref] == ["RaceHorses_832x480_30"]
    item = d[0]
    assert item[5] == d.ref[0] and item[0].shape[0] == 9                # HEVC returns the I-frame name (dataset.py:190)
    with pytest.raises(ValueError):
        data.HEVCDataSet(root, 2048, 10, "Z")


def test_vimeo_sample_rule_and_augmentation(tmp_path):
    root = str(tmp_path)
    for d, clip in (("00001", "0001"), ("00001", "0002"), ("00010", "0001"), ("00002", "0005")):
        for i, f in enumerate(_frames(hash((d, clip)) % 1000, 7, 72, 96)):
            _png(os.path.join(root, d, clip, f"im{i + 1}.png"), f)
    ds = data.DataSet(root, 64, seed=3)
    assert len(ds) == 4 * 7
    idx = lambda paths: [int(os.path.basename(p)[2:-4]) for p in paths]
    first = [(idx(ds.image_ref_list[k]), idx([ds.image_input_list[k]])[0]) for k in range(7)]
    assert first == [([1, 1, 1, 1], 2), ([1, 1, 2, 2], 3), ([1, 1, 2, 3], 4), ([1, 2, 3, 4], 5), ([1, 3, 4, 5], 6), ([1, 4, 5, 6], 7),
                     ([1, 1, 3, 5], 7)]                                          # dataset.py:215-240
    assert [os.path.basename(os.path.dirname(os.path.dirname(p))) for p in ds.image_input_list[::14]] == ["00001", "00002"]
    x, refs = ds[5]
    assert x.shape == (3, 64, 64) and refs.shape == (4, 3, 64, 64) and x.dtype == torch.float32
    assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    # one draw is shared by the input and its references: identical frames stay identical, both crop branches
    f = np.asarray(Image.open(ds.image_input_list[0]).convert("RGB"))
    seen = set()
    for s in range(12):
        a, r = data.augment_clip(f, [f, f], (64, 64), np.random.default_rng(s))
        assert torch.equal(a, r[0]) and torch.equal(a, r[1]) and a.shape == (3, 64, 64)
        b, _ = data.augment_clip(f, [f, f], (64, 64), np.random.default_rng(s))
        assert torch.equal(a, b)                                                 # seeded draws reproduce
        seen.add(round(float(a.mean()), 4))
    assert len(seen) > 6
