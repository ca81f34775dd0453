"""Index tables for `tdvc_pack_conv_weights_indexed`: fp32 master weights on the device -> fp16 MFMA fragment order.

One packed element is addressed by (row r, channel c, tap t) of the conv THE KERNEL RUNS; its value is
`w_flat[row_off[r] + chan_off[c] + tap_off[t]]`, or zero when `row_off[r] < 0`, `chan_off[c] < 0` or
`tap_mask[t] & (row_mask[r] | chan_mask[c])`.  That is enough for every layer form on the TDVC path:

* forward: plain / masked tap lists, PixelShuffle row order, concatenation channel order, zero-padded channels,
  Conv3d holders, and the space-to-depth form of the 3x3 stride-2 convs (`ops._s2d_weights`);
* data gradient (dX = conv(dY, W^T mirrored)) of stride-1 convs, of sub-pixel convs, and of stride-2 convs
  (a 4-phase sub-pixel conv over dY: `dX[2Y+py] = sum_oy W[py - 2*oy + pad] * dY[Y+oy]`).

All tables are built once per layer on the host (a few hundred ints) and live on the device; re-packing after an
optimizer step is one kernel launch per layer and form.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import os

import numpy as np
import torch


@dataclass
class WeightLayout:
    """How the logical (cout, cin, kh, kw) weight of a layer sits in its parameter's flat fp32 storage."""
    cout: int
    cin: int
    kh: int
    kw: int
    row_src: np.ndarray          # [cout]  element offset of output channel o
    chan_src: np.ndarray         # [cin]   element offset of input channel i
    tap_src: np.ndarray          # [kh*kw] element offset of tap (dy, dx), index dy*kw + dx

    @staticmethod
    def dense(cout, cin, kh, kw) -> "WeightLayout":
        return WeightLayout(cout, cin, kh, kw, np.arange(cout, dtype=np.int64) * cin * kh * kw,
                            np.arange(cin, dtype=np.int64) * kh * kw, np.arange(kh * kw, dtype=np.int64))

    @staticmethod
    def conv3d_temporal(cout, cin, kt) -> "WeightLayout":
        """Conv3d (kt,1,1) weight (cout, cin, kt, 1, 1) seen as a 1x1 conv over kt*cin channels, t-major."""
        chan = np.array([c * kt + t for t in range(kt) for c in range(cin)], dtype=np.int64)
        return WeightLayout(cout, kt * cin, 1, 1, np.arange(cout, dtype=np.int64) * cin * kt, chan, np.zeros(1, dtype=np.int64))


@dataclass
class PackTables:
    """geometry of the conv the kernel runs + device index tables"""
    cout: int
    cin: int                     # packed (padded) input channels
    kh: int
    kw: int
    pad: int
    taps: list
    ck: int
    shuffle: bool
    row_off: torch.Tensor
    chan_off: torch.Tensor
    tap_off: torch.Tensor
    row_mask: torch.Tensor
    chan_mask: torch.Tensor
    tap_mask: torch.Tensor
    tap_lin: torch.Tensor = field(default=None)      # forward only: dy*kw + dx of every tap (wgrad scatter)


def _dev(a, dtype, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(device)


def _tables(device, cout, cin, kh, kw, pad, taps, ck, shuffle, row_off, chan_off, tap_off, row_mask=None, chan_mask=None,
            tap_mask=None, tap_lin=None) -> PackTables:
    z = lambda n: np.zeros(n, dtype=np.uint8)
    return PackTables(cout, cin, kh, kw, pad, list(taps), ck, shuffle,
                      _dev(row_off, torch.int32, device), _dev(chan_off, torch.int32, device), _dev(tap_off, torch.int32, device),
                      _dev(z(cout) if row_mask is None else row_mask, torch.uint8, device),
                      _dev(z(cin) if chan_mask is None else chan_mask, torch.uint8, device),
                      _dev(z(len(taps)) if tap_mask is None else tap_mask, torch.uint8, device),
                      None if tap_lin is None else _dev(tap_lin, torch.int32, device))


def shuffle_perm(cout: int) -> np.ndarray:
    """packed row (i*2+j)*cq + c  <-  original row c*4 + i*2 + j  (PixelShuffle(2) store order)"""
    cq = cout // 4
    return np.arange(cout).reshape(cq, 4).T.reshape(-1)


def forward_tables(lay: WeightLayout, *, cin_pad: int, taps, pad: int, ck: int, shuffle=False, cin_perm=None, device="cuda") -> PackTables:
    rows = shuffle_perm(lay.cout) if shuffle else np.arange(lay.cout)
    chan = np.full(cin_pad, -1, dtype=np.int64)
    src = np.arange(lay.cin) if cin_perm is None else np.asarray(cin_perm)
    chan[:len(src)] = lay.chan_src[src]
    tap_lin = np.array([dy * lay.kw + dx for dy, dx in taps], dtype=np.int64)
    return _tables(device, lay.cout, cin_pad, lay.kh, lay.kw, pad, taps, ck, shuffle, lay.row_src[rows], chan, lay.tap_src[tap_lin],
                   tap_lin=tap_lin)


def forward_tables_s2d(lay: WeightLayout, *, ck: int, device="cuda") -> PackTables:
    """3x3 stride-2 pad-1 conv as a 2x2 conv over the space-to-depth view: virtual channel q*C + c with parity
    (py, px) = (q>>1, q&1); virtual tap (dy, dx) covers kernel row ky = 2*dy + py - 1 (invalid at ky = -1)."""
    assert (lay.kh, lay.kw) == (3, 3)
    C_ = lay.cin
    chan = np.empty(4 * C_, dtype=np.int64)
    cmask = np.empty(4 * C_, dtype=np.uint8)
    for q in range(4):
        py, px = q >> 1, q & 1
        chan[q * C_:(q + 1) * C_] = lay.chan_src + lay.tap_src[py * 3 + px]
        cmask[q * C_:(q + 1) * C_] = (1 if py == 0 else 0) | (2 if px == 0 else 0)
    taps = [(0, 0), (0, 1), (1, 0), (1, 1)]
    # tap_src is affine for every layout built here: offset(ky, kx) = ky * s_kh + kx * s_kw
    s_kh, s_kw = int(lay.tap_src[3] - lay.tap_src[0]), int(lay.tap_src[1] - lay.tap_src[0])
    tap_off = np.array([(2 * dy - 1) * s_kh + (2 * dx - 1) * s_kw for dy, dx in taps], dtype=np.int64)
    tmask = np.array([(1 if dy == 0 else 0) | (2 if dx == 0 else 0) for dy, dx in taps], dtype=np.uint8)
    return _tables(device, lay.cout, 4 * C_, 2, 2, 1, taps, ck, False, lay.row_src, chan, tap_off, chan_mask=cmask, tap_mask=tmask)


def dgrad_tables(lay: WeightLayout, *, g_channels: int, x_channels: int, taps, pad: int, stride: int, ck: int, shuffle=False, cin_perm=None,
                 device="cuda") -> PackTables:
    """Tables of the conv that maps dY (g_channels wide, in the forward conv's OUTPUT order) to dX (x_channels wide).

    stride 1: window mirrored (tap (dy,dx) -> (kh-1-dy, kw-1-dx)), pad' = k-1-pad, rows = input channels, channels = output rows.
    stride 2 (k in {1, 3}): 4-phase sub-pixel conv, rows q*x_channels + c, taps (1+oy, 1+ox) of a 3x3 pad-1 window,
    kernel row ky = py - 2*oy + pad; stored through the PixelShuffle epilogue."""
    kh, kw = lay.kh, lay.kw
    src = np.arange(lay.cin) if cin_perm is None else np.asarray(cin_perm)
    xrow = np.full(x_channels, -1, dtype=np.int64)          # dX channel -> weight input-channel offset
    xrow[:len(src)] = lay.chan_src[src]
    rows = shuffle_perm(lay.cout) if shuffle else np.arange(lay.cout)
    gchan = np.full(g_channels, -1, dtype=np.int64)         # dY channel -> weight output-row offset
    gchan[:lay.cout] = lay.row_src[rows]
    if stride == 1:
        assert kh - 1 - pad >= 0 and kh == kw
        mtaps = [(kh - 1 - dy, kw - 1 - dx) for dy, dx in taps]
        tap_off = np.array([lay.tap_src[dy * kw + dx] for dy, dx in taps], dtype=np.int64)
        # mirrored taps in raster order: the dense-window kernels (conv_mfma_v10 / v11 / n16: tap t = (t / kw, t % kw)) then take the
        # data gradients too (round 3: with the mirrored list they fell back to conv_mfma_v3, 122 launches of a training step; the
        # step time itself did not move measurably: 31-45 ms from process to process, the side stream's weight gradients bound it)
        order = sorted(range(len(mtaps)), key=lambda i: mtaps[i])
        mtaps = [mtaps[i] for i in order]
        tap_off = tap_off[order]
        return _tables(device, x_channels, g_channels, kh, kw, kh - 1 - pad, mtaps, ck, False, xrow, gchan, tap_off)
    assert stride == 2 and kh == kw and kh in (1, 3) and len(taps) == kh * kw
    s_kh = int(lay.tap_src[kw] - lay.tap_src[0]) if kh > 1 else 0
    s_kw = int(lay.tap_src[1] - lay.tap_src[0]) if kw > 1 else 0
    row_off = np.full(4 * x_channels, -1, dtype=np.int64)
    rmask = np.zeros(4 * x_channels, dtype=np.uint8)
    if kh == 3:
        assert pad == 1
        # ky = py - 2*oy + 1: (py,oy) = (0,0)->1, (1,0)->2, (1,1)->0, (0,1) invalid  =>  row part py+1, tap part -2*oy
        for q in range(4):
            py, px = q >> 1, q & 1
            valid = xrow >= 0
            row_off[q * x_channels:(q + 1) * x_channels] = np.where(valid, xrow + (py + 1) * s_kh + (px + 1) * s_kw + lay.tap_src[0], -1)
            rmask[q * x_channels:(q + 1) * x_channels] = (1 if py == 0 else 0) | (2 if px == 0 else 0)
        vt = [(0, 0), (0, 1), (1, 0), (1, 1)]
        tap_off = np.array([-2 * oy * s_kh - 2 * ox * s_kw for oy, ox in vt], dtype=np.int64)
        tmask = np.array([(1 if oy == 1 else 0) | (2 if ox == 1 else 0) for oy, ox in vt], dtype=np.uint8)
        wtaps = [(1 + oy, 1 + ox) for oy, ox in vt]
        return _tables(device, 4 * x_channels, g_channels, 3, 3, 1, wtaps, ck, True, row_off, gchan, tap_off, row_mask=rmask, tap_mask=tmask)
    assert pad == 0                                          # 1x1 stride 2: only phase (0,0) receives a gradient
    row_off[:x_channels] = np.where(xrow >= 0, xrow + lay.tap_src[0], -1)
    return _tables(device, 4 * x_channels, g_channels, 1, 1, 0, [(0, 0)], ck, True, row_off, gchan, np.zeros(1, dtype=np.int64))
