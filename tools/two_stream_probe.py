#!/usr/bin/env python3
"""Would two passes in flight at once help?  Registers the pairs of the default bench workload
(a) in one call on one stream, (b) as two half calls on two streams with a context each."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from image_stitcher_amd import _lib
    from image_stitcher_amd._lib import check, lib, ptr
    dev = torch.device("cuda", 0)
    cfg = dict(bench.WORKLOADS["c3"], channels=1)
    tiles, true_yx, pair_idx, pair_dir = bench.build_device_acquisition(cfg, 3, dev)
    tiles = tiles[0]
    n_tiles, tile = tiles.shape[0], cfg["tile"]
    ff = torch.as_tensor(bench.synth.make_flatfield(tile, tile), device=dev) if hasattr(bench, "synth") else None
    n = len(pair_idx)
    d_idx = torch.as_tensor(pair_idx, dtype=torch.int32, device=dev)
    d_dir = torch.as_tensor(pair_dir, dtype=torch.uint8, device=dev)
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    st = torch.empty((n,), dtype=torch.int8, device=dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    ctx1 = _lib.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    ctxs = [_lib.Context(0, s.cuda_stream) for s in streams]
    half = n // 2

    def call(ctx, lo, hi):
        check(lib.mistitch_register_pairs(ctx.handle, ptr(tiles), n_tiles, tile, tile, tile * tile, ptr(d_idx[lo:hi]),
                                          ptr(d_dir[lo:hi]), hi - lo, 256, _lib.FFT_F64, ptr(out[lo:hi]), ptr(st[lo:hi])))

    def one():
        call(ctx1, 0, n)

    def two():
        call(ctxs[0], 0, half)
        call(ctxs[1], half, n)

    for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(f"{name:12s} {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
    exp = true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]]
    print("shifts ok:", bool(np.array_equal(out.cpu().numpy()[:, 1:].astype(np.int64), exp)))


if __name__ == "__main__":
    main()
