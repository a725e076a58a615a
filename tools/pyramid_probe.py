#!/usr/bin/env python3
"""Time the pyramid of a config-3 sized mosaic (4 x 37070 x 37071 uint16, chunks of one tile): row-major levels, their
layout into chunks level by level, and the levels born in chunk order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_stitcher_amd import fusion, omezarr

dev = torch.device("cuda", 0)
ch_n, h, w, t = 4, 37070, 37071, 2048
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
img = torch.randint(0, 30000, (ch_n, h, w), dtype=torch.int16, device=dev).view(torch.uint16)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


ms_plain, plain = timed(lambda: fusion.generate_pyramid(img, levels))
print(f"row-major levels 1..{levels - 1}: {ms_plain:.2f} ms")
total = 0.0
for l in range(1, levels):
    ms, _ = timed(lambda: omezarr.chunk_layout(plain[l], t, t))
    total += ms
    print(f"  layout of level {l} ({tuple(plain[l].shape)}): {ms:.2f} ms")
print(f"row-major levels + their layout: {ms_plain + total:.2f} ms")
del plain
ms_born, born = timed(lambda: fusion.generate_pyramid(img, levels, level_chunks=(t, t)))
print(f"levels born in chunk order: {ms_born:.2f} ms")
