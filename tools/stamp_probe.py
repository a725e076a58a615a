#!/usr/bin/env python3
"""Diagnostic: phase durations of the FFT kernels from in-kernel wall-clock stamps (needs tools/variants/stamps.so, built by tools/build_reg_variant.sh stamps -DMS_STAMPS)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_stitcher_amd import _lib, synth
from image_stitcher_amd import registration as reg

g = synth.make_grid(1, 6, 6, 2048, 2048)
tiles = reg.to_device_u16(g.tiles)
for _ in range(3):
    out, st = reg.register_pairs(tiles, g.pair_idx, g.pair_dir, 256)
torch.cuda.synchronize()
buf = np.zeros((3, 4096, 8), dtype=np.uint64)
_lib.lib.mistitch_debug_stamps.argtypes = [C.c_void_p]
assert _lib.lib.mistitch_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
names = {0: ["tw+load", "fft", "store"], 1: ["tw+load", "fft fwd", "cps", "fft inv", "store"], 2: ["load", "fft inv", "pcm/topk"]}
for k in range(3):
    b = buf[k].astype(np.int64)
    used = b[:, 0] > 0
    d = np.diff(b[used][:, : len(names[k]) + 1], axis=1) / 100.0  # 100 MHz wall clock -> microseconds
    print("kernel", k, "blocks", int(used.sum()), {n: round(float(np.median(d[:, i])), 2) for i, n in enumerate(names[k])},
          "total", round(float(np.median(d.sum(axis=1))), 2), "us")
