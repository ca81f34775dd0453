"""CPU: the C-ABI library loads, exports every symbol declared in include/tdvc_hip.h, and its
host-side entry points (packing, plan, range coder, argument validation) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tdvc_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tdvc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tdvc_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/tdvc_hip.h but not exported"
        assert s in L.SIGNATURES, f"{s} has no ctypes signature in tdvc_amd/_lib.py"
    assert set(L.SIGNATURES) == set(syms)
    assert lib.tdvc_abi_version() >= 1


def test_struct_sizes_match_header_layout():
    assert C.sizeof(L.FMapDesc) == 40
    # ConvDesc: 2 fmaps, 2 ptrs, 2 ints, 98 bytes of taps (+2 pad), 7 ints, fmap, int, float, int, 2 fmaps, int
    assert C.sizeof(L.ConvDesc) % 8 == 0 and C.sizeof(L.ConvDesc) >= 5 * 40 + 16 + 98


def test_conv_plan_and_errors():
    lib = L.lib()
    assert lib.tdvc_conv_plan(64, 3, 3, 1) == 64
    assert lib.tdvc_conv_plan(8, 7, 7, 1) == 8
    assert lib.tdvc_conv_plan(64, 7, 7, 1) == 32
    assert lib.tdvc_conv_plan(64, 3, 3, 2) == 16
    assert lib.tdvc_conv_plan(432, 1, 1, 1) == 64
    assert lib.tdvc_conv_plan(64, 9, 9, 1) < 0 and b"unsupported" in lib.tdvc_last_error()
    assert lib.tdvc_conv_plan(63, 3, 3, 1) < 0
    assert lib.tdvc_conv2d(None, None) == L.lib().tdvc_conv2d(None, None) != 0      # null descriptor -> error code, no crash
    assert lib.tdvc_dcn_v2_forward_f32(*([None] * 6), *([1] * 14), None) != 0


def ref_pack(w, cin_pad, taps, ck):
    """independent python restatement of the fragment order documented in conv_mfma.hip"""
    cout, cin, kh, kw = w.shape
    ck8 = ck // 8
    nchunks = (cin_pad + ck - 1) // ck
    steps = (len(taps) * ck8 + 1) // 2
    tiles = 1 if cout <= 32 else 2 * ((cout + 63) // 64)
    out = np.zeros((tiles, nchunks, steps, 64, 8), dtype=np.float16)
    for t in range(tiles):
        for ch in range(nchunks):
            for s in range(steps):
                for lane in range(64):
                    r, h = lane & 31, lane >> 5
                    co, kc = t * 32 + r, 2 * s + h
                    if kc >= len(taps) * ck8 or co >= cout:
                        continue
                    tap, c8 = divmod(kc, ck8)
                    for j in range(8):
                        ci = ch * ck + c8 * 8 + j
                        if ci < cin:
                            out[t, ch, s, lane, j] = w[co, ci, taps[tap][0], taps[tap][1]]
    return out


@pytest.mark.parametrize("cout,cin,k,ck", [(64, 64, 3, 64), (2, 16, 7, 16), (40, 3, 3, 8), (426, 512, 1, 64), (128, 72, 3, 32)])
def test_pack_conv_weights_fragment_order(cout, cin, k, ck):
    lib = L.lib()
    rng = np.random.default_rng(0)
    w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
    cin_pad = (cin + 7) // 8 * 8
    taps = [(dy, dx) for dy in range(k) for dx in range(k)]
    nbytes = lib.tdvc_conv_packed_bytes(cout, cin_pad, len(taps), ck)
    dst = np.zeros(nbytes // 2, dtype=np.uint16)
    dy = np.array([t[0] for t in taps], dtype=np.int8)
    dx = np.array([t[1] for t in taps], dtype=np.int8)
    rc = lib.tdvc_pack_conv_weights(w.ctypes.data, cout, cin, cin_pad, k, k, len(taps), dy.ctypes.data, dx.ctypes.data, ck, dst.ctypes.data)
    assert rc == 0
    want = ref_pack(w, cin_pad, taps, ck)
    assert np.array_equal(dst.view(np.float16), want.reshape(-1))


def test_rans_c_matches_oracle_bitstream():
    """bit-identical range-coder output: product C++ coder vs the oracle's python restatement"""
    from oracle.tdvc_ref import coder as oc
    lib = L.lib()
    rng = np.random.default_rng(5)
    ntab, width = 6, 40
    cdfs = np.zeros((ntab, width), dtype=np.int32)
    sizes = np.zeros(ntab, dtype=np.int32)
    offsets = -rng.integers(1, 12, ntab).astype(np.int32)
    for i in range(ntab):
        n = int(rng.integers(3, width - 2))
        p = rng.random(n) ** 3 + 1e-9
        c = oc.pmf_to_quantized_cdf((p / p.sum()).tolist(), 16)
        cdfs[i, : len(c)] = c
        sizes[i] = len(c)
    N = 20000
    idx = rng.integers(0, ntab, N).astype(np.int32)
    syms = np.array([int(rng.integers(offsets[i] - (30 if rng.random() < 0.03 else 0),
                                      offsets[i] + sizes[i] - 2 + (30 if rng.random() < 0.03 else 0))) for i in idx], dtype=np.int32)
    want = oc.rans_encode(syms.tolist(), idx.tolist(), cdfs.tolist(), sizes.tolist(), offsets.tolist())
    out = np.zeros(4 * N + 64, dtype=np.uint8)
    n = lib.tdvc_rans_encode(syms.ctypes.data, idx.ctypes.data, N, cdfs.ctypes.data, width, sizes.ctypes.data,
                             offsets.ctypes.data, out.ctypes.data, out.size)
    assert n == len(want)
    assert out[:n].tobytes() == want
    dec = np.zeros(N, dtype=np.int32)
    rc = lib.tdvc_rans_decode(out.ctypes.data, n, idx.ctypes.data, N, cdfs.ctypes.data, width, sizes.ctypes.data,
                              offsets.ctypes.data, dec.ctypes.data)
    assert rc == 0 and np.array_equal(dec, syms)
    # too-small output buffer and corrupt stream are reported, not crashed on
    assert lib.tdvc_rans_encode(syms.ctypes.data, idx.ctypes.data, N, cdfs.ctypes.data, width, sizes.ctypes.data,
                                offsets.ctypes.data, out.ctypes.data, 16) < 0
    assert lib.tdvc_rans_decode(out.ctypes.data, 4, idx.ctypes.data, N, cdfs.ctypes.data, width, sizes.ctypes.data,
                                offsets.ctypes.data, dec.ctypes.data) != 0


def test_empty_stream_roundtrip():
    lib = L.lib()
    z = np.zeros(1, dtype=np.int32)
    out = np.zeros(16, dtype=np.uint8)
    n = lib.tdvc_rans_encode(z.ctypes.data, z.ctypes.data, 0, z.ctypes.data, 1, z.ctypes.data, z.ctypes.data, out.ctypes.data, 16)
    assert n == 8
    assert lib.tdvc_rans_decode(out.ctypes.data, 8, z.ctypes.data, 0, z.ctypes.data, 1, z.ctypes.data, z.ctypes.data, z.ctypes.data) == 0


def test_product_has_no_cpu_fallback():
    from tdvc_amd.model import VideoCompressor
    m = VideoCompressor().eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 4, 3, 64, 64), True)
    import _ext
    with pytest.raises(RuntimeError):
        _ext.dcn_v2_forward(torch.zeros(1, 2, 4, 4), torch.zeros(2, 2, 3, 3), torch.zeros(2), torch.zeros(1, 18, 4, 4),
                            torch.zeros(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)


def test_no_packed_fp32_instructions_in_the_device_code(tmp_path):
    """The library must not contain v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: on MI355X a packed-FP32 instruction with an
    operand swizzle returned wrong values in lanes 48..63 while another stream ran MFMA kernels (tools/race_pk_min.py,
    DESIGN.md section 4).  The Makefile switches the instruction class off; this guards the switch."""
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    lib = tmp_path / "libtdvc_hip.so"
    shutil.copy(L.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)      # extracts the code objects
    objs = sorted(tmp_path.glob("libtdvc_hip.so.*gfx950*"))
    assert objs, "no gfx950 code object found in the library"
    pat = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
    hits, mfma = 0, 0
    for o in objs:
        text = subprocess.run([objdump, "-d", str(o)], check=True, capture_output=True, text=True).stdout
        hits += len(pat.findall(text))
        mfma += text.count("v_mfma_f32_32x32x16")
    assert mfma > 1000, "the disassembly does not look like the library's device code"
    assert hits == 0, f"{hits} packed-FP32 instructions in the device code: build with csrc/Makefile's NOPK flag"
