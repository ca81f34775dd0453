"""tdvc_amd — MI355X-native (gfx950) implementation of TDVC's per-P-frame encode/reconstruct path.

`tdvc_amd.model.pnet.VideoCompressor` is the drop-in for `main.model.pnet.VideoCompressor`;
`_ext` (repo root) is the drop-in for the reference's DCNv2 extension module.
"""
__version__ = "0.1.0"
