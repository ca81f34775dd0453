from .pnet import VideoCompressor  # noqa: F401
