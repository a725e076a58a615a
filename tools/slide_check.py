#!/usr/bin/env python3
"""Whole-slide scale check (one channel of config 5: 50 x 50 tiles of 2048^2, 4900 pairs, a
92 k x 92 k mosaic): register every pair, solve the positions on the host, fuse at the solved
positions in both modes and compare every mosaic pixel with the scene the tiles were cut from.
Everything stays on the device; exercises 64-bit offsets (10.5 G tile pixels, 8.5 G mosaic pixels)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50)
    ap.add_argument("--cols", type=int, default=50)
    ap.add_argument("--tile", type=int, default=2048)
    args = ap.parse_args()
    import torch
    from image_stitcher_amd import fusion, registration, solve, synth
    dev = torch.device("cuda", 0)
    rows, cols, t, jitter = args.rows, args.cols, args.tile, 3
    pitch = t - int(round(t * 0.10))
    sh, sw = (rows - 1) * pitch + t + 2 * jitter, (cols - 1) * pitch + t + 2 * jitter
    rng = np.random.default_rng(50)
    jit = rng.integers(-jitter, jitter + 1, size=(rows * cols, 2))
    jit[0] = 0
    true_yx = np.array([[jitter + r * pitch + jit[r * cols + c, 0], jitter + c * pitch + jit[r * cols + c, 1]]
                        for r in range(rows) for c in range(cols)], dtype=np.int64)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    scene = torch.empty((sh, sw), dtype=torch.int16, device=dev)
    band = 4096
    for y in range(0, sh, band):   # in bands: the float scene never exists whole
        h = min(band, sh - y)
        scene[y:y + h] = (1000.0 + 30000.0 * torch.rand((h, sw), generator=gen, device=dev)).to(torch.int16)
    tiles = torch.empty((rows * cols, t, t), dtype=torch.int16, device=dev)
    for n, (y0, x0) in enumerate(true_yx):
        tiles[n] = scene[y0:y0 + t, x0:x0 + t]
    tiles = tiles.view(torch.uint16)
    pair_idx, pair_dir = synth.grid_pairs(rows, cols)
    print(f"{rows * cols} tiles ({tiles.numel() / 1e9:.2f} G px), {len(pair_idx)} pairs", flush=True)

    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        out, status = registration.register_pairs(tiles, pair_idx, pair_dir, 256)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out, status = out.cpu().numpy(), status.cpu().numpy()
    exp = true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]]
    shifts_ok = bool((status == 0).all() and np.array_equal(out[:, 1:].astype(np.int64), exp))
    print(f"registration {dt * 1e3:.1f} ms = {len(pair_idx) / dt:.0f} pairs/s, planted shifts recovered: {shifts_ok}", flush=True)

    row, col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
    t0 = time.perf_counter()
    res = solve.solve_positions(solve.grid_from_pairs(row, col, pair_idx, pair_dir, out, status), t, t)
    yx = np.stack([res.y_pos, res.x_pos], axis=1).astype(np.int32)
    pos_ok = bool(np.array_equal(yx, true_yx - true_yx.min(axis=0)))
    print(f"solve {1e3 * (time.perf_counter() - t0):.0f} ms, positions are the planted ones: {pos_ok}", flush=True)

    mh, mw = int(yx[:, 0].max()) + t, int(yx[:, 1].max()) + t
    y0, x0 = (int(v) for v in true_yx.min(axis=0))
    want = scene[y0:y0 + mh, x0:x0 + mw]
    covered = torch.zeros((mh, mw), dtype=torch.bool, device=dev)
    for y, x in yx:
        covered[y:y + t, x:x + t] = True
    ok = shifts_ok and pos_ok
    plane = np.zeros(rows * cols, dtype=np.int32)
    for mode in ("overwrite", "feather"):
        mosaic = None
        for rep in range(2):
            del mosaic
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mosaic = fusion.fuse_tiles(tiles, yx, plane, 1, mh, mw, mode=mode)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        m = mosaic[0].view(torch.int16)
        bad = int(((m != want) & covered).sum().item()) + int(((m != 0) & ~covered).sum().item())
        print(f"fuse {mode}: {dt * 1e3:.1f} ms = {mh * mw / dt / 1e6:.0f} MPix/s, mosaic {mh} x {mw}, "
              f"pixels that differ from the scene: {bad}", flush=True)
        ok = ok and bad == 0
        del m
    print("SLIDE CHECK", "OK" if ok else "FAILED", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
