#!/usr/bin/env python3
"""Diagnostic: throughput of the chunk layout kernel on the benchmark's mosaic (4 x 37070 x 37071 uint16,
chunks of 2048 x 2048) and on its first pyramid level (float64)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_stitcher_amd import omezarr

dev = torch.device("cuda", 0)
for dtype, shape in ((torch.uint16, (4, 37070, 37071)), (torch.float64, (4, 18535, 18535))):
    img = torch.empty(shape, dtype=torch.int16 if dtype == torch.uint16 else dtype, device=dev)
    img = img.view(dtype) if dtype == torch.uint16 else img
    out = omezarr.chunk_layout(img, 2048, 2048)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = omezarr.chunk_layout(img, 2048, 2048)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    moved = img.numel() * img.element_size() + out.numel() * out.element_size()
    print(f"{dtype} {shape}: {1e3 * dt:.2f} ms, {moved / dt / 1e9:.0f} GB/s read + written (output allocation included)", flush=True)
    del img, out
