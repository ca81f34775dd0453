"""Container of one coded P-frame: the record layout the reference sketches in `tools/utils/encoder.py:61-68` /
`tools/utils/decoder.py:29-46` — per string a big-endian `>4I` shape, a uint16 byte count, the payload.

The reference writes three records against modules that are not in its tree; here a frame has four strings in the
order [mv_y, mv_z, res_y, res_z] (both coders' Gaussian-conditional and factorised-prior streams), each with the shape
(batch index, latent channels, z height, z width).  Extension: a payload longer than 65 534 bytes writes the count
0xFFFF followed by a big-endian uint32 length (untrained weights at 1080p exceed the reference's 16-bit field).
"""
from __future__ import annotations

import struct

import numpy as np


def write_records(f, strings, shapes) -> int:
    """-> bytes written"""
    n = 0
    for s, shp in zip(strings, shapes):
        assert len(shp) == 4
        f.write(struct.pack(">4I", *[int(v) for v in shp]))
        if len(s) < 0xFFFF:
            f.write(np.array(len(s), dtype=np.uint16).tobytes())
            n += 2
        else:
            f.write(np.array(0xFFFF, dtype=np.uint16).tobytes() + struct.pack(">I", len(s)))
            n += 6
        f.write(s)
        n += 16 + len(s)
    return n


def read_records(f, count: int):
    """-> (strings, shapes)"""
    strings, shapes = [], []
    for _ in range(count):
        shapes.append(struct.unpack(">4I", f.read(16)))
        ln = int(np.frombuffer(f.read(2), dtype=np.uint16)[0])
        if ln == 0xFFFF:
            ln = struct.unpack(">I", f.read(4))[0]
        s = f.read(ln)
        if len(s) != ln:
            raise ValueError("truncated bitstream")
        strings.append(s)
    return strings, shapes
