"""Which of torch's OWN gfx950 kernels contain the packed-FP32 instruction form that returned wrong values next to a concurrent MFMA stream
(`v_pk_{add,mul,fma}_f32 ... op_sel:[0,1]`: DESIGN.md section 4, profiles/r03_packed_fp32_race.txt)?  libtdvc_hip.so is built without packed
FP32 (csrc/Makefile), torch's libtorch_hip.so is not, and a training step launches a handful of torch element-wise kernels inside the backward
sweep, next to the side stream's MFMA weight-gradient kernels.

Steps (CPU only, ~4 minutes on 6 cores): copy the `.hip_fatbin` section out of libtorch_hip.so, cut it into its compressed offload bundles
(`CCOB` records), unbundle the gfx950 code object of each (clang-offload-bundler), disassemble (llvm-objdump) and count per kernel.
Prints the totals and, for the kernel families a training step launches (profiles/r03_train_last_step.txt), whether any instantiation has the form.

python tools/scan_torch_pk.py [workdir]"""
import mmap
import os
import struct
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import torch

LLVM = "/opt/rocm/lib/llvm/bin"
# kernel families of the training step (torch kernels in profiles/r03_train_last_step.txt) -> substring of the demangled name
FAMILIES = {
    "fill (mirrors, buckets)": "FillFunctor",
    "add (accumulate)": "CUDAFunctor_add<float>",
    "mul (scale)": "MulFunctor<float>",
    "copy / cast": "direct_copy_kernel_cuda",
    "2-norm reduce (clipping)": "NormTwoOps<float",
    "fused Adam": "FusedAdamMathFunctor",
    "softplus (entropy bottleneck chain)": "softplus_kernel",
    "tanh (entropy bottleneck chain)": "::tanh_kernel_cuda",
    "sigmoid": "sigmoid_kernel",
}


def scan(co):
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode("latin1")
    res, k = {}, None
    for ln in out.splitlines():
        if ln.endswith(">:") and "<" in ln:
            k = ln[ln.index("<") + 1:-2]
        elif "v_pk_add_f32" in ln or "v_pk_mul_f32" in ln or "v_pk_fma_f32" in ln:
            r = res.setdefault(k, [0, 0])
            r[0] += 1
            r[1] += "op_sel:[0,1]" in ln
    return res


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/tdvc_torch_scan"
    os.makedirs(work, exist_ok=True)
    lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_hip.so")
    fat = os.path.join(work, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    cos = []
    with open(fat, "rb") as f:
        m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        pos, k = 0, 0
        while True:
            i = m.find(b"CCOB", pos)
            if i < 0:
                break
            ver, meth = struct.unpack_from("<HH", m, i + 4)
            tot = struct.unpack_from("<I", m, i + 8)[0]
            if ver == 2 and meth == 1 and m[i + 24:i + 28] == b"\x28\xb5\x2f\xfd":          # version 2, zstd
                piece = os.path.join(work, f"b{k:03d}.hipfb")
                open(piece, "wb").write(m[i:i + tot])
                co = os.path.join(work, f"b{k:03d}.co")
                subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={piece}", f"--output={co}"], stderr=subprocess.DEVNULL)
                os.remove(piece)
                if os.path.exists(co) and os.path.getsize(co) > 0:
                    cos.append(co)
                k += 1
                pos = i + tot
            else:
                pos = i + 4
    print(f"torch {torch.__version__}: {k} offload bundles, {len(cos)} with a gfx950 code object")
    allk = {}
    with ThreadPoolExecutor(6) as ex:
        for r in ex.map(scan, cos):
            allk.update(r)
    names = list(allk)
    dem = subprocess.run(["c++filt"], input="\n".join(names).encode(), stdout=subprocess.PIPE).stdout.decode().splitlines()
    bad = [d for n, d in zip(names, dem) if allk[n][1]]
    print(f"kernels with packed-FP32 instructions: {len(names)}; with the op_sel:[0,1] form: {len(bad)}")
    for fam, sub in FAMILIES.items():
        hit = [d for d in bad if sub in d]
        any_pk = sum(1 for d in dem if sub in d)
        print(f"  {fam:38s} instantiations with packed FP32: {any_pk:4d}; with op_sel:[0,1]: {len(hit):3d}" + (f"   e.g. {hit[0][:110]}" if hit else ""))


if __name__ == "__main__":
    main()
