#!/usr/bin/env python3
"""Diagnostic: per-kernel durations of one registration step on the 20x20 grid of 2048^2 tiles (one channel, no
flatfield), for library variants (MISTITCH_LIB=tools/variants/<name>.so).  Prints how many planted shifts came back,
so a variant that breaks results on purpose (a store left out, ...) is still timed.
usage: reg_probe.py [edge_width] [f64|f32] [steps] [extra flags, e.g. 22 = consensus + ten peaks + subpixel]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_stitcher_amd import _lib, synth
from image_stitcher_amd._lib import check, lib, ptr

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fft = sys.argv[2] if len(sys.argv) > 2 else "f64"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = cols = 20
tile = 2048
device = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream(device).cuda_stream)
true_yx, _, _ = synth.planted_positions(3, rows, cols, tile)
pair_idx, pair_dir = synth.grid_pairs(rows, cols)
tiles, _, _, _ = synth.make_device_acquisition(rows, cols, tile, 1, 3, device)
n = len(pair_idx)
d_idx = torch.as_tensor(pair_idx, dtype=torch.int32, device=device)
d_dir = torch.as_tensor(pair_dir, dtype=torch.uint8, device=device)
out = torch.empty((n, 3), dtype=torch.float64, device=device)
st = torch.empty((n,), dtype=torch.int8, device=device)
flag = (_lib.FFT_F64 if fft == "f64" else _lib.FFT_F32) | (int(sys.argv[4]) if len(sys.argv) > 4 else 0)


def run():
    check(lib.mistitch_register_pairs(ctx.handle, ptr(tiles[0]), rows * cols, tile, tile, tile * tile, ptr(d_idx), ptr(d_dir),
                                      n, edge, flag, ptr(out), ptr(st)))


for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
expected = true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]]
good = int(((st.cpu().numpy() == 0) & (np.rint(out.cpu().numpy()[:, 1:]).astype(np.int64) == expected).all(axis=1)).sum())
ctx.profile(True)
for _ in range(2):
    run()
prof = {k: round(v[0] / 2, 3) for k, v in ctx.profile_collect().items()}
ctx.profile(False)
print(f"{os.environ.get('MISTITCH_LIB', 'libmistitch.so')}: edge {edge} {fft}: {ms:.3f} ms/step, {n / ms * 1e3:.0f} pairs/s, "
      f"{good}/{n} shifts right; kernels {prof}", flush=True)
