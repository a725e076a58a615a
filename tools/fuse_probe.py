#!/usr/bin/env python3
"""Time the fusion kernel variants (mode x flatfield) on a synthetic grid; prints ms and GB/s."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20)
    ap.add_argument("--cols", type=int, default=20)
    ap.add_argument("--tile", type=int, default=2048)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--overlap", type=float, default=0.1)
    ap.add_argument("--jitter", type=int, default=7)
    ap.add_argument("--only", default="", help="run one variant only, e.g. feather:f32")
    ap.add_argument("--kernels", action="store_true", help="per-kernel milliseconds of one more call")
    ap.add_argument("--align", type=int, default=1, help="round tile columns down to a multiple of this")
    ap.add_argument("--chunks", action="store_true",
                    help="also time the mosaic born in chunk order (chunks of one tile) and the chunk layout of the plain one")
    args = ap.parse_args()
    import torch
    from image_stitcher_amd import _lib, synth
    from image_stitcher_amd._lib import check, lib, ptr

    dev = torch.device("cuda", 0)
    t, ch, n = args.tile, args.channels, args.rows * args.cols
    pitch = t - int(round(t * args.overlap))
    rng = np.random.default_rng(0)
    yx = np.array([[r * pitch + rng.integers(0, args.jitter), c * pitch + rng.integers(0, args.jitter)]
                   for r in range(args.rows) for c in range(args.cols)], dtype=np.int32)
    yx[:, 1] -= yx[:, 1] % args.align
    mh, mw = int(yx[:, 0].max()) + t, int(yx[:, 1].max()) + t
    tiles = torch.randint(0, 30000, (ch * n, t, t), dtype=torch.int16, device=dev).view(torch.uint16)
    d_yx = torch.as_tensor(np.tile(yx, (ch, 1)), dtype=torch.int32, device=dev)
    d_plane = torch.as_tensor(np.repeat(np.arange(ch), n), dtype=torch.int32, device=dev)
    mosaic = torch.empty((ch, mh, mw), dtype=torch.int16, device=dev).view(torch.uint16)
    ffs32 = torch.as_tensor(np.stack([synth.make_flatfield(t, t) for _ in range(ch)]), device=dev)
    ctx = _lib.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    for mode_name, mode in (("overwrite", _lib.FUSE_OVERWRITE), ("feather", _lib.FUSE_FEATHER)):
        for ff_name, ffs, fft in (("none", None, _lib.FF_NONE), ("f32", ffs32, _lib.FF_F32)):
            if args.only and args.only != f"{mode_name}:{ff_name}":
                continue
            def run():
                check(lib.mistitch_fuse(ctx.handle, ptr(tiles), ch * n, t, t, t * t, ptr(d_yx), ptr(d_plane), ptr(ffs),
                                        fft, ptr(d_plane) if ffs is not None else ptr(None), ptr(mosaic), ch, mh, mw,
                                        0, mh, mode))
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            feather_bytes = ch * (2 * n * t * t + 2 * mh * mw)
            print(f"{mode_name:9s} ff={ff_name:4s} {ms:8.3f} ms  {feather_bytes / ms / 1e6:8.1f} GB/s (2B*tile px + 2B*out px)",
                  flush=True)
            if args.chunks:
                ncy, ncx = -(-mh // t), -(-mw // t)
                laid = torch.empty((ch, ncy, ncx, t, t), dtype=torch.int16, device=dev).view(torch.uint16)
                def run_chunked():
                    check(lib.mistitch_fuse_chunked(ctx.handle, ptr(tiles), ch * n, t, t, t * t, ptr(d_yx), ptr(d_plane), ptr(ffs),
                                                    fft, ptr(d_plane) if ffs is not None else ptr(None), ptr(laid), ch, mh, mw,
                                                    t, t, 0, mh, mode))
                def run_layout():
                    check(lib.mistitch_chunk_layout(ctx.handle, ptr(mosaic), 2, ch, mh, mw, t, t, ptr(laid)))
                for label, fn in (("born in chunk order", run_chunked), ("chunk layout of the plain mosaic", run_layout)):
                    fn()
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(args.reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    print(f"   {label}: {e0.elapsed_time(e1) / args.reps:.3f} ms", flush=True)
                del laid
            if args.kernels:
                ctx.profile(True)
                run()
                print("   " + "  ".join(f"{k} {v[0]:.3f}" for k, v in ctx.profile_collect().items()), flush=True)
                ctx.profile(False)


if __name__ == "__main__":
    main()
