#!/usr/bin/env python3
"""Can a registration + fusion step be captured into a HIP graph and replayed?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from image_stitcher_amd import _lib, fusion, registration, synth
    dev = torch.device("cuda", 0)
    g = synth.make_grid(7, 6, 6, 512, 512, overlap=0.1, jitter=3)
    tiles = registration.to_device_u16(g.tiles)
    s = torch.cuda.Stream(dev)
    ctx = _lib.Context(0, s.cuda_stream)
    n = len(g.pair_idx)
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    status = torch.empty((n,), dtype=torch.int8, device=dev)
    yx = (g.true_yx - g.true_yx.min(axis=0)).astype(np.int32)
    mh, mw = int(yx[:, 0].max()) + 512, int(yx[:, 1].max()) + 512
    d_yx = torch.as_tensor(yx, device=dev)
    d_plane = torch.zeros(len(yx), dtype=torch.int32, device=dev)
    d_idx = torch.as_tensor(g.pair_idx, dtype=torch.int32, device=dev)
    d_dir = torch.as_tensor(g.pair_dir, dtype=torch.uint8, device=dev)
    mosaic = torch.empty((1, mh, mw), dtype=torch.int16, device=dev).view(torch.uint16)

    def step():
        registration.register_pairs(tiles, d_idx, d_dir, 64, ctx=ctx, out=out, status=status)
        fusion.fuse_tiles(tiles, d_yx, d_plane, 1, mh, mw, mode="feather", out=mosaic, ctx=ctx)

    with torch.cuda.stream(s):
        step()          # sizes the workspaces
    torch.cuda.synchronize()
    want_out, want_mosaic = out.clone(), mosaic.clone()
    out.zero_()
    mosaic.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    print("replay equals the eager step:", bool(torch.equal(out, want_out) and torch.equal(mosaic, want_mosaic)))
    for name, fn in (("eager", lambda: step()), ("graph", graph.replay)):
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")


if __name__ == "__main__":
    main()
