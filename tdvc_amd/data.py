"""Dataset adapters of the reference's evaluation and training loops (host-side, PIL for PNG):

* `UVGDataSet`, `HEVCDataSet` — `main/dataloader/dataset.py:16-99,101-190`: tree layout
  `<root>/ori_img/<seq>/imNNN.png`, `<root>/compress_img_bpg/<seq>/<qp>/imNNN_<qp>.{png,txt}`, the lambda -> QP map,
  GOP enumeration (`framerange = len // GOP` with `testfull`, else 8) and the 7-tuple `__getitem__` returns;
* `DataSet` — the Vimeo-septuplet sampler (`dataset.py:193-258`): per clip six (input, 4 refs) samples from the
  original frames plus the `[1, 1, 3, 5] -> 7` sample, augmented like `augmentation.imgauglist2`
  (`main/dataloader/augmentation.py:29-77`).

The file listing, QP map and sample rule are deterministic and tested against the rule stated in the reference; the
augmentation draws (albumentations / torchvision, neither present here) are restated by distribution, not bit-pinned:
flips p = 0.5 / 0.4, one of {RGB shift +-20, brightness / contrast +-0.2} with p = 0.5, then either a random
size x size crop or a random-resized crop (area 0.5 .. 1, aspect 3/4 .. 4/3, bilinear) with p = 0.5 each, one draw
shared by the input and its references.  cv2.resize's INTER_LINEAR (training-mode resize to 256 x 256) is PIL's BILINEAR
here."""
from __future__ import annotations

import glob
import math
import os
import re

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

LAMBDA_TO_QP = {512: 37, 16: 37, 1024: 32, 32: 32, 2048: 27, 64: 27, 4096: 22, 128: 22}      # dataset.py:25-36
HEVC_CLASSES = {                                                                            # dataset.py:109-124
    "A": ("2560x1600", ["Traffic", "PeopleOnStreet"]),
    "B": ("1920x1080", ["ParkScene", "Kimono1", "Cactus", "BasketballDrive", "BQTerrace"]),
    "C": ("832x480", ["BasketballDrill", "BQMall", "PartyScene", "RaceHorses"]),
    "D": ("416x240", ["BasketballPass", "BQSquare", "BlowingBubbles", "RaceHorses"]),
    "E": ("1280x720", ["vidyo1", "vidyo3", "vidyo4"]),
}


def natsorted(items):
    """natural order (digit runs compare as numbers), what `natsort.natsorted` gives on these file names"""
    return sorted(items, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)])


def read_rgb(path: str, size: tuple[int, int] | None = None) -> np.ndarray:
    """(3, H, W) float32 in [0, 1] (dataset.py:67-72: imread, BGR->RGB, optional resize, /255)"""
    im = Image.open(path).convert("RGB")
    if size is not None:
        im = im.resize(size, Image.BILINEAR)
    return np.asarray(im, dtype=np.uint8).transpose(2, 0, 1).astype(np.float32) / 255.0


def calc_psnr(a: np.ndarray, b: np.ndarray) -> float:
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * math.log10(1.0 / mse) if mse > 0 else 100.0


class _GopDataSet(torch.utils.data.Dataset):
    """common part of UVGDataSet / HEVCDataSet: one item = one GOP (BPG I-frame + GOP_size original frames)"""

    def __init__(self, root, train_lambda, GOP_size, testfull, isTrain, compute_ref_metrics):
        self.inputPath = os.path.join(root, "ori_img")
        self.refPath = os.path.join(root, "compress_img_bpg")
        self.isTrain, self.compute_ref_metrics = isTrain, compute_ref_metrics
        self.ref, self.refbpp, self.input = [], [], []
        if int(train_lambda) not in LAMBDA_TO_QP:
            raise ValueError(f"train_lambda {train_lambda} has no QP (known: {sorted(LAMBDA_TO_QP)})")
        self.qp = LAMBDA_TO_QP[int(train_lambda)]
        self.GOP_size, self.testfull = int(GOP_size), bool(testfull)

    def _add_sequence(self, seq):
        imglist = natsorted(glob.glob(os.path.join(self.inputPath, seq, "*.png")))
        framerange = len(imglist) // self.GOP_size if self.testfull else 8
        qp = str(self.qp)
        for i in range(framerange):
            stem = "im" + str(i * self.GOP_size + 1).zfill(3) + "_" + qp
            with open(os.path.join(self.refPath, seq, qp, stem + ".txt"), "r", encoding="utf-8") as f:
                rbpp = f.read().splitlines()[0]
            self.ref.append(os.path.join(self.refPath, seq, qp, stem + ".png"))
            self.refbpp.append(float(rbpp))
            self.input.append([os.path.join(self.inputPath, seq, "im" + str(i * self.GOP_size + j + 1).zfill(3) + ".png")
                               for j in range(self.GOP_size)])

    def __len__(self):
        return len(self.ref)

    def _item(self, index):
        size = (256, 256) if self.isTrain else None
        ref_image = read_rgb(self.ref[index], size)
        h, w = ref_image.shape[1:]
        input_images, raw_video = [], []
        refpsnr = refmsssim = None
        for filename in self.input[index]:
            img = read_rgb(filename, size)[:, :h, :w]
            if refpsnr is None:                       # the first frame is the I-frame: its quality, not a P-frame input
                refpsnr = calc_psnr(img, ref_image)
                if self.compute_ref_metrics:
                    from . import metrics                # HIP kernels: needs the GPU (no CPU fallback)
                    refmsssim = np.asarray(float(metrics.ms_ssim(torch.from_numpy(img[None]).cuda(), torch.from_numpy(ref_image[None]).cuda(),
                                                                 data_range=1.0)), dtype=np.float32)
                else:
                    refmsssim = np.float32("nan")
            else:
                input_images.append(img)
            raw_video.append(img)
        return np.array(input_images), ref_image, self.refbpp[index], refpsnr, refmsssim, np.array(raw_video)


class UVGDataSet(_GopDataSet):
    def __init__(self, root, train_lambda, GOP_size, testfull=False, isTrain=True, compute_ref_metrics=True):
        super().__init__(root, train_lambda, GOP_size, testfull, isTrain, compute_ref_metrics)
        for folder in natsorted(os.listdir(self.inputPath)):
            self._add_sequence(folder.rstrip())

    def __getitem__(self, index):
        inp, ref, bpp, psnr, msssim, raw = self._item(index)
        return inp, ref, bpp, psnr, msssim, self.input[index], raw             # dataset.py:99


class HEVCDataSet(_GopDataSet):
    def __init__(self, root, train_lambda, GOP_size, cls, testfull=False, isTrain=True, compute_ref_metrics=True):
        super().__init__(root, train_lambda, GOP_size, testfull, isTrain, compute_ref_metrics)
        if cls not in HEVC_CLASSES:
            raise ValueError(f"HEVC class {cls!r} (known: {sorted(HEVC_CLASSES)})")
        resolution, names = HEVC_CLASSES[cls]
        for folder in os.listdir(self.inputPath):                                # folder = <Name>_<WxH>_<fps>
            seq = folder.rstrip()
            parts = seq.split("_")
            if len(parts) >= 2 and parts[0] in names and parts[1] == resolution:
                self._add_sequence(seq)

    def __getitem__(self, index):
        inp, ref, bpp, psnr, msssim, raw = self._item(index)
        return inp, ref, bpp, psnr, msssim, self.ref[index], raw               # dataset.py:190


def vimeo_samples(clip_dir: str, n_frames: int):
    """dataset.py:211-240 for one clip directory: ([ref paths x4], input path) pairs"""
    im = lambda i: os.path.join(clip_dir, f"im{i}.png")
    refs, inputs = [], []
    start = 1
    while start + 1 <= n_frames:
        tmp = [im(1)] + [im(i) for i in range(max(start + 1 - 3, 1), start + 1)]
        tmp += [tmp[-1]] * (4 - len(tmp))
        refs.append(tmp)
        inputs.append(im(start + 1))
        start += 1
    refs.append([im(1), im(1), im(3), im(5)])
    inputs.append(im(7))
    return refs, inputs


def augment_clip(input_image: np.ndarray, ref_images: list, size: tuple[int, int], rng: np.random.Generator):
    """augmentation.imgauglist2 restated by distribution (module docstring).  uint8 HWC in -> float (3,h,w), (R,3,h,w)"""
    frames = [input_image] + list(ref_images)
    if rng.random() < 0.5:
        frames = [f[:, ::-1] for f in frames]
    if rng.random() < 0.4:
        frames = [f[::-1] for f in frames]
    x = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).float()          # (T, H, W, 3), 0..255
    if rng.random() < 0.5:
        if rng.random() < 0.5:
            x = x + torch.tensor(rng.uniform(-20, 20, size=3), dtype=torch.float32)
        else:
            x = x * float(1.0 + rng.uniform(-0.2, 0.2)) + 255.0 * float(rng.uniform(-0.2, 0.2))
        x = x.clamp(0, 255).round()
    x = x.permute(0, 3, 1, 2) / 255.0
    H, W = x.shape[-2:]
    th, tw = size
    if rng.random() < 0.5:                       # RandomSizedCrop([s, s], s, s): a plain random crop
        if H < th or W < tw:
            raise ValueError(f"frames {H}x{W} smaller than the crop {th}x{tw}")
        y0, x0 = int(rng.integers(0, H - th + 1)), int(rng.integers(0, W - tw + 1))
        x = x[..., y0:y0 + th, x0:x0 + tw]
    else:                                        # RandomResizedCrop(size, scale=(0.5, 1.0)), default aspect range
        ch, cw, y0, x0 = H, W, 0, 0
        for _ in range(10):
            area = H * W * rng.uniform(0.5, 1.0)
            ar = math.exp(rng.uniform(math.log(3 / 4), math.log(4 / 3)))
            cw_, ch_ = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
            if 0 < cw_ <= W and 0 < ch_ <= H:
                ch, cw = ch_, cw_
                y0, x0 = int(rng.integers(0, H - ch + 1)), int(rng.integers(0, W - cw + 1))
                break
        x = F.interpolate(x[..., y0:y0 + ch, x0:x0 + cw], size=(th, tw), mode="bilinear", align_corners=False)
    x = x.contiguous()
    return x[0], x[1:]


class DataSet(torch.utils.data.Dataset):
    """Vimeo septuplets: `<root>/<dir>/<clip>/im{1..7}.png` (dataset.py:193-258)"""

    def __init__(self, dataset_path, resize_size, seed: int | None = None):
        self.image_input_list, self.image_ref_list = self.get_vimeo(dataset_path)
        self.size = [resize_size, resize_size]
        self.im_height, self.im_width = self.size
        self.rng = np.random.default_rng(seed)

    @staticmethod
    def get_vimeo(dataset_path):
        inputs, refs = [], []
        for d in natsorted(os.listdir(dataset_path)):
            for clip in natsorted(os.listdir(os.path.join(dataset_path, d))):
                clip_dir = os.path.join(dataset_path, d, clip)
                r, i = vimeo_samples(clip_dir, len(glob.glob(os.path.join(clip_dir, "*.png"))))
                refs += r
                inputs += i
        return inputs, refs

    def __len__(self):
        return len(self.image_input_list)

    def __getitem__(self, index):
        rd = lambda p: np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8)
        return augment_clip(rd(self.image_input_list[index]), [rd(p) for p in self.image_ref_list[index]], tuple(self.size), self.rng)
