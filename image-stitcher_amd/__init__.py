"""MI355X-native registration-and-fusion hot path of the image stitcher.

The compute lives in ``csrc/`` (HIP kernels for gfx950 behind a C ABI,
``include/mistitch.h``); the modules here mirror the reference's Python
interfaces for this path and hand device pointers to that library.
"""
__all__ = ["synth"]
