"""The oracle's AMP-emulation switch (oracle/tdvc_ref/blocks.py, codec.py): what the reference computes on a GPU with
`enable_amp: True` (main/model/pnet.py:27-78, cfg/predict.yaml:7), restated on the CPU.  CPU-only tests of the restatement's
own contract: off = the fp32 path the golden vectors pin, bit for bit; on = fp16 convs with torch's dtype promotion in the three
autocast regions, fp32 coders."""
import math

import torch
import torch.nn.functional as F

from oracle.tdvc_ref import VideoCompressor as Ref
from oracle.tdvc_ref import blocks
from tdvc_amd.synth import fill_parameters, make_gop, ref_list


def test_conv_follows_autocast_fp16_policy():
    torch.manual_seed(0)
    c = blocks.Conv2d(8, 16, 3, 1, 1)
    x = torch.randn(1, 8, 12, 12)
    y32 = c(x)
    assert y32.dtype == torch.float32 and torch.equal(y32, F.conv2d(x, c.weight, c.bias, 1, 1))
    with blocks.amp_region(True):
        y16 = c(x)
        y16_from_half = c(x.half())
    h = lambda t: t.half().float()
    want = F.conv2d(h(x), h(c.weight), h(c.bias), 1, 1).half()
    assert y16.dtype == torch.float16 and torch.equal(y16, want) and torch.equal(y16_from_half, want)
    assert not blocks._Amp.on                      # the region restores the flag
    c3 = blocks.Conv3d(4, 4, (1, 3, 3), padding=(0, 1, 1))
    x3 = torch.randn(1, 4, 2, 6, 6)
    with blocks.amp_region(True):
        y3 = c3(x3)
    assert y3.dtype == torch.float16 and torch.equal(y3, F.conv3d(h(x3), h(c3.weight), h(c3.bias), 1, (0, 1, 1)).half())


def test_amp_emulation_switch():
    ref = Ref().eval()
    fill_parameters(ref)
    g = make_gop(1234, 2, 64, 64)
    refs = ref_list([g[0:1]])
    with torch.no_grad():
        r0, b0, m0 = ref(g[1:2], refs, False)
        r0a, b0a, m0a = ref(g[1:2], refs, True)             # switch off: `enabled_amp` is a no-op, as autocast is on the CPU
        ref.amp_emulation = True
        r1f, b1f, m1f = ref(g[1:2], refs, False)            # switch on but enabled_amp=False: still the fp32 path
        tr = {}
        r1, b1, m1 = ref(g[1:2], refs, True, trace=tr)
    assert torch.equal(r0, r0a) and torch.equal(r0, r1f) and float(b0) == float(b1f) and float(m0) == float(m1f)
    # dtypes as torch's promotion gives them on the GPU: fp16 features, fp32 coder outputs, fp32 sum of the two
    assert tr["f_cur"].dtype == tr["estmv"].dtype == tr["pred"].dtype == tr["resid"].dtype == torch.float16
    assert tr["mv_x_hat"].dtype == tr["res_x_hat"].dtype == tr["recon_f"].dtype == torch.float32
    assert r1.dtype == torch.float32 and torch.equal(r1, r1.half().float())       # the picture the reference returns is an fp16 tensor
    agree = 10 * math.log10(1.0 / float(((r1 - r0) ** 2).mean()))
    assert agree > 40.0 and not torch.equal(r1, r0)      # a different rounding of the same function, not another function
    assert abs(float(b1 + m1) - float(b0 + m0)) < 0.02 * float(b0 + m0)
