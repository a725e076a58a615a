#!/usr/bin/env python3
"""What does a plain device-to-device copy reach on this GPU?  (context for the fusion roofline)"""
import torch

dev = torch.device("cuda", 0)
for gib in (1, 4, 10):
    n = gib * (1 << 30) // 2
    a = torch.empty(n, dtype=torch.int16, device=dev).random_(0, 30000)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"contiguous copy {gib} GiB: {ms:.3f} ms  {2 * n * 2 / ms / 1e6:.0f} GB/s (read+write)", flush=True)
    del a, b
# 2-D strided: 2048x2048 uint16 tiles pasted into a wide mosaic (what fusion does without overlaps)
H = W = 2048
g = 18
mos = torch.empty((g * H, g * W + 7), dtype=torch.int16, device=dev)
tiles = torch.empty((g * g, H, W), dtype=torch.int16, device=dev).random_(0, 30000)
def paste():
    for r in range(g):
        for c in range(g):
            mos[r * H:(r + 1) * H, c * W:(c + 1) * W].copy_(tiles[r * g + c])
paste(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); paste(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"torch per-tile paste {g}x{g}: {ms:.3f} ms  {2 * g * g * H * W * 2 / ms / 1e6:.0f} GB/s", flush=True)
