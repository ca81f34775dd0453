from .codec import VideoCompressor  # noqa: F401
