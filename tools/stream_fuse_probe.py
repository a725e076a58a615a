#!/usr/bin/env python3
"""Band-streamed fusion at the default bench size (20 x 20 tiles of 2048^2, 4 channels +
flatfield): time of upload -> fuse -> download with the pinned buffers filled / drained at no
cost (PCIe and the pipeline alone), next to the PCIe bound and the resident fusion time."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--band-rows", type=int, default=3686)
    ap.add_argument("--mode", default="feather")
    args = ap.parse_args()
    import torch
    from image_stitcher_amd import ingest, synth
    cfg = bench.WORKLOADS["c3"]
    rows, cols, t, ch = cfg["rows"], cfg["cols"], cfg["tile"], cfg["channels"]
    pitch = t - int(round(t * 0.10))
    rng = np.random.default_rng(1)
    yx1 = np.array([[r * pitch + rng.integers(0, 7), c * pitch + rng.integers(0, 7)] for r in range(rows) for c in range(cols)],
                   dtype=np.int32)
    yx1 -= yx1.min(axis=0)
    yx = np.tile(yx1, (ch, 1))
    plane = np.repeat(np.arange(ch), rows * cols).astype(np.int32)
    # placement order must ascend with the tile index; channels interleave freely, so sort by row then channel
    mh, mw = int(yx[:, 0].max()) + t, int(yx[:, 1].max()) + t
    ffs = np.stack([synth.make_flatfield(t, t, amplitude=0.2 - 0.03 * c) for c in range(ch)])
    moved = {"up": 0, "down": 0}

    def fill(idx, dst):
        moved["up"] += dst.nbytes

    def drain(r0, r1, src):
        moved["down"] += src.nbytes

    ws = {}
    for rep in range(3):
        moved["up"] = moved["down"] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = ingest.fuse_streaming(fill, drain, yx, plane, ch, t, t, mh, mw, band_rows=args.band_rows, flatfields=ffs,
                                  tile_ff=plane, mode=args.mode, workspace=ws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"run {rep}: {n} bands of {args.band_rows} rows, {dt * 1e3:.0f} ms; up {moved['up'] / 1e9:.1f} GB, "
              f"down {moved['down'] / 1e9:.1f} GB -> {moved['up'] / dt / 1e9:.1f} + {moved['down'] / dt / 1e9:.1f} GB/s",
              flush=True)


if __name__ == "__main__":
    main()
