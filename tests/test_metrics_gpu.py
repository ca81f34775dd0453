"""MS-SSIM / SSIM / PSNR on the GPU (csrc/metrics.hip through the C-ABI) against the reference's golden outputs and the
CPU oracle; error behaviour of the reference's interface."""
import math
import os

import numpy as np
import pytest
import torch

from oracle.tdvc_ref import metrics as ref
from tests.helpers_metrics import CASES, make_pair

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "msssim.npz"))
TOL = 2e-5          # fp32 sums in a different order (tiles, then a double-precision final sum)


def test_ms_ssim_matches_reference_golden(report):
    from tdvc_amd import metrics
    worst = 0.0
    for i, case in enumerate(CASES):
        X, Y = make_pair(*case)
        Xd, Yd = X.cuda(), Y.cuda()
        ms = metrics.ms_ssim(Xd, Yd, data_range=1.0, size_average=False).cpu().numpy()
        s, cs = metrics.ssim(Xd, Yd, data_range=1.0, size_average=False, full=True)
        for got, key in ((ms, f"ms_{i}"), (s.cpu().numpy(), f"ssim_{i}"), (cs.cpu().numpy(), f"cs_{i}")):
            worst = max(worst, float(np.abs(got - GOLD[key]).max()))
            np.testing.assert_allclose(got, GOLD[key], rtol=TOL, atol=TOL)
        ms255 = float(metrics.ms_ssim(Xd * 255, Yd * 255, data_range=255))
        assert abs(ms255 - float(GOLD[f"ms255_{i}"])) < 5e-5
        assert abs(float(metrics.ms_ssim(Xd, Yd, data_range=1.0)) - float(ref.ms_ssim(X, Y, data_range=1.0))) < TOL
    report(f"ms_ssim / ssim / cs on the GPU vs the reference's outputs: max abs diff {worst:.2e} over {len(CASES)} cases")


def test_ms_ssim_1080p_properties(report):
    from tdvc_amd import metrics, synth
    gop = synth.make_gop(1234, 2, 1080, 1920).float().cuda()
    X, Y = gop[0:1], gop[1:2]
    one = float(metrics.ms_ssim(X, X, data_range=1.0))
    assert abs(one - 1.0) < 1e-6
    a, b = float(metrics.ms_ssim(X, Y, data_range=1.0)), float(metrics.ms_ssim(Y, X, data_range=1.0))
    assert abs(a - b) < 1e-6 and 0.0 < a < 1.0                         # symmetric
    want = float(ref.ms_ssim(X.cpu(), Y.cpu(), data_range=1.0))
    assert abs(a - want) < TOL
    p = metrics.psnr(Y, X)
    mse = float(((Y - X) ** 2).mean())
    assert abs(p - 10 * math.log10(1 / mse)) < 1e-9
    report(f"1080p frame pair: ms_ssim {a:.6f} (oracle {want:.6f}), psnr {p:.3f} dB")


def test_interface_errors():
    from tdvc_amd import metrics
    X = torch.rand(1, 3, 64, 64, device="cuda")
    with pytest.raises(ValueError):
        metrics.ms_ssim(X[0], X[0])
    with pytest.raises(ValueError):
        metrics.ms_ssim(X, X[:, :, :32])
    with pytest.raises(ValueError):
        metrics.ms_ssim(X, X, win_size=10)
    with pytest.raises(ValueError):
        metrics.ms_ssim(X, X)                      # 64 / 16 = 4 < 11-tap window at the last level
    with pytest.raises(RuntimeError):
        metrics.ms_ssim(X.cpu(), X.cpu())


def test_predict_tool_on_a_dataset_tree(tmp_path, report):
    """python -m tdvc_amd.tools.predict over an HEVC-class-D style tree (tdvc_amd.data adapters): the BPG I-frame's own
    bpp / PSNR / MS-SSIM and the nine P-frames of the GOP enter the averages (tools/predict.py:43-108)"""
    import json
    import subprocess
    import sys

    from PIL import Image

    from tdvc_amd import synth
    root = str(tmp_path)
    name, qp = "BQSquare_416x240_60", 27
    frames = [(f.permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8) for f in synth.make_gop(77, 10, 240, 416)]
    for i, f in enumerate(frames):
        os.makedirs(os.path.join(root, "ori_img", name), exist_ok=True)
        Image.fromarray(f).save(os.path.join(root, "ori_img", name, f"im{i + 1:03d}.png"))
    d = os.path.join(root, "compress_img_bpg", name, str(qp))
    os.makedirs(d)
    Image.fromarray(np.clip(frames[0].astype(np.int16) + 2, 0, 255).astype(np.uint8)).save(os.path.join(d, f"im001_{qp}.png"))
    with open(os.path.join(d, f"im001_{qp}.txt"), "w") as f:
        f.write("0.4321\n")
    out = os.path.join(root, "res.txt")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "tdvc_amd.tools.predict", "--dataset-root", root, "--val-dataset", "HEVC", "--cls", "D",
                        "--train-lambda", "2048", "--out", out], cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["frames"] == 10 and 0.0 < res["msssim"] <= 1.0 and res["psnr"] > 10.0 and res["bpp"] > 0.0
    txt = open(out).read()
    assert "msssim : " in txt and "bpp : " in txt and "psnr : " in txt
    report(f"predict tool on a class-D tree: {res['frames']} frames, bpp {res['bpp']:.4f}, psnr {res['psnr']:.3f}, ms-ssim {res['msssim']:.5f}")


def test_rd_sweep_tool(report):
    """BASELINE configs[4] driver: lambda list x GOP shards -> one BPP / PSNR / MS-SSIM table (single rank here; the rank
    partition and the gather are covered on gloo in tests/test_dist_cpu.py)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "tdvc_amd.tools.rd_sweep", "--lambdas", "256", "2048", "--gops", "2", "--gop-size", "3",
                          "--height", "192", "--width", "256"], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.splitlines()[0])
    rows = res["curve"]
    report(f"rd_sweep: {rows}")
    assert [r["lambda"] for r in rows] == [256, 2048] and all(r["frames"] == 4 for r in rows)
    assert all(r["bpp"] > 0 and 5.0 < r["psnr"] < 60.0 and 0.0 < r["msssim"] <= 1.0 for r in rows)
    assert rows[0]["bpp"] == rows[1]["bpp"]            # no checkpoints: the same filler weights code both lambdas
