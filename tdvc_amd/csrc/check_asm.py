#!/usr/bin/env python3
"""Build-time guard for hand-counted `s_waitcnt vmcnt(N)`: count an instruction inside one kernel of a gfx950 assembly
listing (hipcc -S --cuda-device-only) and fail the build unless the count is the one the source assumes.

conv_mfma_v7's top-of-stage wait `vmcnt(8)` is correct only while the full-tile epilogue compiles to exactly 8 global
stores per thread (`epilogue_simple_rows`, 2 rows x 4 x 16 B); fewer, wider stores would let the barrier release MFMA
reads of LDS-DMA pieces still in flight.

usage: check_asm.py listing.s <kernel-symbol-substring> <instruction-prefix> <expected-count>
       check_asm.py listing.s <kernel-symbol-substring> --no-scratch

--no-scratch (conv_row): EVERY kernel whose symbol contains the substring must use no private (scratch) memory.  A loader wave's
counted `s_waitcnt vmcnt(N)` counts its LDS-DMA instructions only; a register spill is a vector-memory instruction the count
does not know about, and the wait would then release a barrier over pieces still in flight.
"""
import re
import sys


def count(path, kernel_sub, prefix):
    lines = open(path).read().splitlines()
    inside, n, seen = False, 0, []
    for ln in lines:
        m = re.match(r"^([A-Za-z_][\w$.]*):", ln)
        if m and not m.group(1).startswith(".L"):          # a function label (local labels are .LBB*)
            inside = kernel_sub in m.group(1)
            if inside:
                seen.append(m.group(1))
        elif inside and ln.strip().startswith(prefix):
            n += 1
    return n, seen


def no_scratch(path, kernel_sub):
    lines = open(path).read().splitlines()
    cur, desc, found, bad = None, None, 0, {}
    for ln in lines:
        m = re.match(r"^([A-Za-z_][\w$.]*):", ln)
        if m and not m.group(1).startswith(".L"):          # a function label: the code that follows belongs to it
            cur = m.group(1)
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            desc = m.group(1)                              # the kernel descriptor block names its kernel itself
        m = re.match(r"^\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", ln)
        if m and desc and kernel_sub in desc:
            found += 1
            if int(m.group(1)) != 0:
                bad[desc] = f"{m.group(1)} bytes of private memory per lane"
        if cur and kernel_sub in cur and re.match(r"^\s*scratch_", ln):
            bad.setdefault(cur, "scratch instructions")
    if not found:
        sys.exit(f"check_asm: no kernel matching '{kernel_sub}' in {path}")
    if bad:
        sys.exit("check_asm: kernels with private memory (register spills) -- their counted s_waitcnt vmcnt is no longer exact:\n  " +
                 "\n  ".join(f"{k}: {v}" for k, v in sorted(bad.items())))
    print(f"check_asm: {found} kernels matching '{kernel_sub}': no private memory (the counted waits see every vector-memory instruction)")


def main():
    if len(sys.argv) == 4 and sys.argv[3] == "--no-scratch":
        return no_scratch(sys.argv[1], sys.argv[2])
    path, kernel_sub, prefix, want = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    n, seen = count(path, kernel_sub, prefix)
    if len(seen) != 1:
        sys.exit(f"check_asm: expected exactly one kernel matching '{kernel_sub}' in {path}, found {seen}")
    if n != want:
        sys.exit(f"check_asm: {seen[0]} has {n} x {prefix}, the source's counted s_waitcnt assumes {want} -- "
                 f"re-derive the wait in the kernel (or fall back to vmcnt(0)) before shipping this build")
    print(f"check_asm: {seen[0]}: {n} x {prefix} (as the counted wait assumes)")


if __name__ == "__main__":
    main()
