#!/usr/bin/env python3
"""Does a kernel's duration depend on how long the GPU has been busy?  Runs the feather + flatfield fusion of the
benchmark geometry N times back to back with an event pair around every call and prints the durations along the run."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from image_stitcher_amd import _lib, synth
    from image_stitcher_amd._lib import check, lib, ptr

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    jitter = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev = torch.device("cuda", 0)
    t, ch, rows, cols = 2048, 4, 20, 20
    n = rows * cols
    pitch = t - int(round(t * 0.1))
    rng = np.random.default_rng(0)
    yx = np.array([[r * pitch + rng.integers(0, jitter), c * pitch + rng.integers(0, jitter)]
                   for r in range(rows) for c in range(cols)], dtype=np.int32)
    mh, mw = int(yx[:, 0].max()) + t, int(yx[:, 1].max()) + t
    tiles = torch.randint(0, 30000, (ch * n, t, t), dtype=torch.int16, device=dev).view(torch.uint16)
    d_yx = torch.as_tensor(np.tile(yx, (ch, 1)), dtype=torch.int32, device=dev)
    d_plane = torch.as_tensor(np.repeat(np.arange(ch), n), dtype=torch.int32, device=dev)
    mosaic = torch.empty((ch, mh, mw), dtype=torch.int16, device=dev).view(torch.uint16)
    ffs = torch.as_tensor(np.stack([synth.make_flatfield(t, t) for _ in range(ch)]), device=dev)
    ctx = _lib.Context(0, torch.cuda.current_stream(dev).cuda_stream)

    def run():
        check(lib.mistitch_fuse(ctx.handle, ptr(tiles), ch * n, t, t, t * t, ptr(d_yx), ptr(d_plane), ptr(ffs),
                                _lib.FF_F32, ptr(d_plane), ptr(mosaic), ch, mh, mw, 0, mh, _lib.FUSE_FEATHER))
    run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    for i in sorted(set([0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 300, 500, 800, reps - 1])):
        if i < reps:
            print(f"call {i:4d}: {ms[i]:.3f} ms", flush=True)
    print(f"min {min(ms):.3f}  median {sorted(ms)[reps // 2]:.3f}  max {max(ms):.3f}")


if __name__ == "__main__":
    main()
