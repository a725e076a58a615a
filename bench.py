#!/usr/bin/env python3
"""Benchmark of the registration-and-fusion hot path on MI355X.

One "step" = one pass of the hot path over one synthetic acquisition that is
already resident in HBM: flatfield-correct the registration channel, register
every neighbouring tile pair, (N > 1: all-gather the pair rows over RCCL), and
fuse all channels with flatfield correction into the mosaic.  Registration and
fusion are timed in separate bracketed regions of K steps each because the
metric has two units (tile pairs/s and fused MPix/s).

Scaling is weak: every rank owns one whole acquisition (an independent region,
as wells or timepoints are), so `value` sums over ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|small]

Under torch.distributed.run the RANK / LOCAL_RANK / WORLD_SIZE variables select
the rank's GPU; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak

WORKLOADS = {
    # name: rows, cols, tile, channels, flatfield, label
    "c3": dict(rows=20, cols=20, tile=2048, channels=4, flatfield=True,
               label="20x20 grid of 2048^2 uint16 tiles, 4 channels + flatfield, 10% overlap, edge_width 256"),
    "c2": dict(rows=10, cols=10, tile=2048, channels=1, flatfield=False,
               label="10x10 grid of 2048^2 uint16 tiles, 1 channel, 10% overlap, edge_width 256"),
    "small": dict(rows=4, cols=4, tile=512, channels=2, flatfield=True,
                  label="4x4 grid of 512^2 uint16 tiles, 2 channels + flatfield (debug size)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--fft", default="f64", choices=["f64", "f32"])
    ap.add_argument("--fuse-mode", default="feather", choices=["feather", "overwrite"])
    ap.add_argument("--edge-width", type=int, default=None)
    ap.add_argument("--pair-chunk", type=int, default=None)
    ap.add_argument("--cpu-pairs", type=int, default=220,
                    help="pairs in the CPU baseline sample: about 12 s of single-core NumPy (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo backend, every rank on device 0 (not a measurement)")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side figures (other fusion mode, other FFT precision): profiler runs")
    return ap.parse_args()


def union_area(yx, h, w):
    """Pixels covered by at least one h x w tile (coordinate-compressed paint)."""
    ys = np.unique(np.concatenate([yx[:, 0], yx[:, 0] + h]))
    xs = np.unique(np.concatenate([yx[:, 1], yx[:, 1] + w]))
    grid = np.zeros((len(ys) - 1, len(xs) - 1), dtype=bool)
    for y, x in yx:
        grid[np.searchsorted(ys, y):np.searchsorted(ys, y + h), np.searchsorted(xs, x):np.searchsorted(xs, x + w)] = True
    return int((np.diff(ys)[:, None] * np.diff(xs)[None, :] * grid).sum())


def build_device_acquisition(cfg, seed, device):
    """Tiles [C, N, H, W] uint16 cut on the device from white-noise scenes, plus geometry."""
    from image_stitcher_amd import synth

    return synth.make_device_acquisition(cfg["rows"], cfg["cols"], cfg["tile"], cfg["channels"], seed, device)


def load_pmc_traffic(workload, fft):
    """HBM bytes per launch of each kernel, from the committed rocprofv3 --pmc passes
    (2 x FETCH_SIZE + WRITE_SIZE, see tools/pmc_summary.py); {} when none is recorded.
    Counters cannot be collected from inside this process, so they are measured by
    separate profiler runs of this same command and read back here."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as fh:
        doc = json.load(fh)
    if doc.get("workload") != workload or doc.get("fft") != fft:
        return {}
    return {k: v["hbm_bytes_per_launch"] for k, v in doc.get("kernels", {}).items()}


def cpu_baseline(n_pairs_target, edge_width, tile, fuse_mode, with_ff):
    """Time the NumPy oracle (single thread, serial loops, as the reference runs) on a bounded sample."""
    from image_stitcher_amd import synth
    from oracle import fusion as ofu
    from oracle import registration as oreg

    side = 2
    while 2 * side * side - 2 * side < n_pairs_target:
        side += 1
    g = synth.make_grid(99, side, side, tile, tile, overlap=0.10, jitter=3)
    n = min(n_pairs_target, len(g.pair_idx))
    ff = synth.make_flatfield(tile, tile)
    t0 = time.perf_counter()
    corrected = np.stack([ofu.apply_flatfield(x, ff) for x in g.tiles]) if with_ff else g.tiles
    out, _ = oreg.register_pairs(corrected, g.pair_idx[:n], g.pair_dir[:n], edge_width, polish=True)
    t_reg = time.perf_counter() - t0
    # the same pairs again, spread over the host cores this process may use (threads: NumPy drops the
    # GIL inside its FFTs, sorts and reductions; no child process is started once the GPU is in use)
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(16, os.cpu_count() or 1, n))
    slices = [np.arange(i, n, n_thr) for i in range(n_thr)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as pool:
        list(pool.map(lambda ix: oreg.register_pairs(corrected, g.pair_idx[ix], g.pair_dir[ix], edge_width, polish=True),
                      slices))
    t_reg_threads = time.perf_counter() - t0
    yx = (g.true_yx - g.true_yx.min(axis=0)).astype(np.int32)
    mh, mw = int(yx[:, 0].max()) + tile, int(yx[:, 1].max()) + tile
    plane = np.zeros(len(g.tiles), dtype=np.int32)
    nf = min(len(g.tiles), 9)
    t0 = time.perf_counter()
    fuse = ofu.fuse_feather if fuse_mode == "feather" else ofu.fuse_overwrite
    fuse(g.tiles[:nf], yx[:nf], plane[:nf], 1, mh, mw, flatfields=ff[None] if with_ff else None,
         tile_ff=plane[:nf] if with_ff else None)
    t_fuse = time.perf_counter() - t0
    mpix = union_area(yx[:nf], tile, tile) / 1e6
    return dict(pairs=n, t_reg=t_reg, t_fuse=t_fuse, fused_mpix=mpix, tiles_fused=nf, grid_side=side,
                t_reg_threads=t_reg_threads, threads=n_thr)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_gloo:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU path"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from image_stitcher_amd import _lib, synth
    from image_stitcher_amd import dist as mdist
    from image_stitcher_amd._lib import check, lib, ptr

    cfg = WORKLOADS[args.workload]
    tile, ch = cfg["tile"], cfg["channels"]
    edge = args.edge_width or (256 if tile >= 2048 else tile // 8)
    fft_flag = _lib.FFT_F64 if args.fft == "f64" else _lib.FFT_F32
    other_fft = "f32" if args.fft == "f64" else "f64"
    other_fft_flag = _lib.FFT_F32 if args.fft == "f64" else _lib.FFT_F64
    mode_id = _lib.FUSE_FEATHER if args.fuse_mode == "feather" else _lib.FUSE_OVERWRITE
    other_mode = "overwrite" if args.fuse_mode == "feather" else "feather"
    other_mode_id = _lib.FUSE_OVERWRITE if args.fuse_mode == "feather" else _lib.FUSE_FEATHER

    ctx = _lib.Context(local_rank, torch.cuda.current_stream(device).cuda_stream, pair_chunk=args.pair_chunk)
    tiles, true_yx, pair_idx, pair_dir = build_device_acquisition(cfg, seed=3 + rank, device=device)
    n_tiles = cfg["rows"] * cfg["cols"]
    n_pairs = len(pair_idx)
    yx_np = (true_yx - true_yx.min(axis=0)).astype(np.int32)
    mosaic_h, mosaic_w = int(yx_np[:, 0].max()) + tile, int(yx_np[:, 1].max()) + tile

    # device-resident descriptors and outputs (nothing below allocates inside the timed loops
    # once the library workspaces have been sized by the warm-up steps)
    d_pair_idx = torch.as_tensor(pair_idx, dtype=torch.int32, device=device)
    d_pair_dir = torch.as_tensor(pair_dir, dtype=torch.uint8, device=device)
    d_out = torch.empty((n_pairs, 3), dtype=torch.float64, device=device)
    d_status = torch.empty((n_pairs,), dtype=torch.int8, device=device)
    d_out2, d_status2 = torch.empty_like(d_out), torch.empty_like(d_status)
    d_yx = torch.as_tensor(np.tile(yx_np, (ch, 1)), dtype=torch.int32, device=device)
    d_plane = torch.as_tensor(np.repeat(np.arange(ch), n_tiles), dtype=torch.int32, device=device)
    mosaic = torch.empty((ch, mosaic_h, mosaic_w), dtype=torch.int16, device=device).view(torch.uint16)
    ffs = None
    reg_tiles = tiles[0]
    if cfg["flatfield"]:
        ffs = torch.as_tensor(np.stack([synth.make_flatfield(tile, tile, amplitude=0.2 - 0.03 * c) for c in range(ch)]),
                              device=device)
    ff_type = _lib.FF_F32 if ffs is not None else _lib.FF_NONE
    counts = [n_pairs] * world
    gathered = None

    def register(flags, out, status):
        """Flatfield-correct the registration channel and register every pair: with a flatfield the
        strips are corrected as they are extracted (one call), which equals correcting whole tiles first."""
        if ffs is not None:
            check(lib.mistitch_register_pairs_flatfield(ctx.handle, ptr(tiles[0]), n_tiles, tile, tile, tile * tile,
                                                        ptr(ffs[0]), ff_type, ptr(d_pair_idx), ptr(d_pair_dir),
                                                        n_pairs, edge, flags, ptr(out), ptr(status)))
        else:
            check(lib.mistitch_register_pairs(ctx.handle, ptr(tiles[0]), n_tiles, tile, tile, tile * tile,
                                              ptr(d_pair_idx), ptr(d_pair_dir), n_pairs, edge, flags,
                                              ptr(out), ptr(status)))

    def step_registration():
        nonlocal gathered
        register(fft_flag, d_out, d_status)
        if world > 1:
            gathered = mdist.all_gather_pair_results(d_out, d_status, counts)

    def step_registration_other():
        """The same step at the other FFT precision (reported beside the headline figure)."""
        register(other_fft_flag, d_out2, d_status2)

    def step_registration_subpixel():
        """The configured step with the upsampled-DFT estimate added for every pair."""
        register(fft_flag | _lib.REG_SUBPIXEL, d_out2, d_status2)

    def fuse(mode):
        check(lib.mistitch_fuse(ctx.handle, ptr(tiles), ch * n_tiles, tile, tile, tile * tile, ptr(d_yx), ptr(d_plane),
                                ptr(ffs), ff_type, ptr(d_plane) if ffs is not None else ptr(None), ptr(mosaic), ch,
                                mosaic_h, mosaic_w, 0, mosaic_h, mode))

    def step_fusion():
        fuse(mode_id)

    def step_fusion_other():
        fuse(other_mode_id)

    def bracket():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, k):
        bracket()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        bracket()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step_registration()
        step_fusion()
    t_reg = timed(step_registration, args.steps)
    t_fuse = timed(step_fusion, args.steps)
    # side figures: the other fusion mode and the other FFT precision, same bracketing
    side = not args.no_side
    if side:
        step_fusion_other()
        t_fuse_other = timed(step_fusion_other, args.steps)
        step_registration_other()
        t_reg_other = timed(step_registration_other, args.steps)
        out2_np, st2_np = d_out2.cpu().numpy(), d_status2.cpu().numpy()
        step_registration_subpixel()
        t_reg_sub = timed(step_registration_subpixel, args.steps)
        out3_np = d_out2.cpu().numpy()
        # host-to-device rate of a pinned buffer of tile data (what an ingest stage would pay; never part of `value`)
        n_h2d = min(64, tiles.shape[1])
        pinned = torch.empty((n_h2d, tile, tile), dtype=torch.int16).pin_memory()
        dst_h2d = torch.empty_like(pinned, device=device)
        dst_h2d.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst_h2d.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        h2d_gbs = 3 * pinned.numel() * 2 / (time.perf_counter() - t0) / 1e9
        del pinned, dst_h2d
        # ingest: the registration channel goes up band by band through two pinned buffers while the
        # bands already resident are registered (upload-bound; never part of `value`)
        from image_stitcher_amd import ingest
        host_tiles = tiles[0].cpu().view(torch.int16).numpy().view(np.uint16)
        rows_per_band = 2
        stager = ingest.TileStager(n_tiles, tile, tile, rows_per_band * cfg["cols"], device)
        ingest_ms = {}
        for label, fill in (("decoder_is_a_memcpy", ingest.fill_from_array(host_tiles)), ("buffers_prefilled", lambda lo, hi, dst: None)):
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, out_s, st_s = ingest.register_streaming(fill, cfg["rows"], cfg["cols"], tile, tile, pair_idx, pair_dir, edge,
                                                           rows_per_band=rows_per_band, fft=args.fft,
                                                           flatfield=ffs[0] if ffs is not None else None, ctx=ctx,
                                                           stager=stager)
                torch.cuda.synchronize()
                ingest_ms[label] = 1e3 * (time.perf_counter() - t0)
            if label == "decoder_is_a_memcpy":
                ingest_same = bool(torch.equal(out_s, d_out) and torch.equal(st_s, d_status))
        del stager, host_tiles

    # ---- parity properties at full size: every planted shift recovered, mosaic spot checks ----
    out_np, st_np = d_out.cpu().numpy(), d_status.cpu().numpy()
    expected = true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]]
    shifts_ok = bool((st_np == 0).all() and np.array_equal(out_np[:, 1:].astype(np.int64), expected))

    # ---- the host step between the two device stages: pair rows -> tile positions (never part of `value`) ----
    from image_stitcher_amd import solve
    grid_row, grid_col = np.repeat(np.arange(cfg["rows"]), cfg["cols"]), np.tile(np.arange(cfg["cols"]), cfg["rows"])
    t0 = time.perf_counter()
    solved = solve.solve_positions(solve.grid_from_pairs(grid_row, grid_col, pair_idx, pair_dir, out_np, st_np), tile, tile)
    t_solve = time.perf_counter() - t0
    positions_ok = bool(np.array_equal(np.stack([solved.y_pos, solved.x_pos], axis=1), yx_np))

    # ---- per-kernel durations from HIP events on the launch stream ----
    kernels = {}
    if not args.no_profile:
        ctx.profile(True)
        psteps = 2
        for _ in range(psteps):
            step_registration()
            step_fusion()
            if side:
                step_fusion_other()
        for name, (ms, launches) in ctx.profile_collect().items():
            kernels[name] = dict(ms_per_step=ms / psteps, launches_per_step=launches / psteps,
                                 avg_launch_us=1e3 * ms / launches)
        if side:  # the subpixel stage's own kernels, from a separate pass so the figures above stay those of the configured step
            ctx.profile_collect()
            step_registration_subpixel()
            for name, (ms, launches) in ctx.profile_collect().items():
                if name.startswith("k_ups_"):
                    kernels[name] = dict(ms_per_step=ms, launches_per_step=float(launches), avg_launch_us=1e3 * ms / launches)
        ctx.profile(False)

    # ---- algorithmic bytes ----
    strip_px = edge * tile
    reg_bytes_per_pair = 2 * strip_px * 2                      # two uint16 strips in
    cplx = 16 if args.fft == "f64" else 8
    covered = union_area(yx_np, tile, tile)
    clipped_tile_px = n_tiles * tile * tile                    # every tile lies inside the mosaic here
    fuse_bytes_by_mode = {"feather": ch * (2 * clipped_tile_px + 2 * mosaic_h * mosaic_w),
                          "overwrite": ch * (4 * covered + 2 * (mosaic_h * mosaic_w - covered))}
    if ffs is not None:
        for m in fuse_bytes_by_mode:
            fuse_bytes_by_mode[m] += ch * tile * tile * 4      # flatfield planes, read once (cache resident)
    fuse_bytes = fuse_bytes_by_mode[args.fuse_mode]
    alg_bytes_per_step = {
        "k_fft_long_fwd": n_pairs * (2 * strip_px * 2 + strip_px * cplx),
        "k_fft_short_cps": n_pairs * (strip_px * cplx + (strip_px // 2) * cplx),
        "k_fft_long_inv": n_pairs * (strip_px // 2) * cplx,
        "k_ncc_accumulate": n_pairs * 2 * strip_px * 2,
        "k_transpose_strips": int((pair_dir == 0).sum()) * 2 * strip_px * 2 * 2,
        "k_flatfield": n_tiles * tile * tile * 4 + tile * tile * 4,
        "k_stage_strips": n_pairs * 2 * strip_px * 4 + tile * tile * 4,   # strip pixels in and out, one flatfield plane
        "k_fuse_feather": fuse_bytes_by_mode["feather"], "k_fuse_overwrite": fuse_bytes_by_mode["overwrite"],
    }
    traffic = load_pmc_traffic(args.workload, args.fft)
    rooflines = {}
    for name, k in kernels.items():
        if name in alg_bytes_per_step and k["ms_per_step"] > 0:
            gbs = alg_bytes_per_step[name] / (k["ms_per_step"] * 1e-3) / 1e9
            rooflines[name] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                   frac=round(gbs / HBM_PEAK_GBS, 4), traffic=traffic.get(name),
                                   avg_launch_us=round(k["avg_launch_us"], 2),
                                   algorithmic_bytes_per_launch=int(alg_bytes_per_step[name] / k["launches_per_step"]))
    # the dominant kernel of the configured step (the other fusion mode is a side figure)
    in_step = [n for n in rooflines if n != f"k_fuse_{other_mode}"]
    dominant = max(in_step, key=lambda n: kernels[n]["ms_per_step"]) if in_step else None

    pairs_per_s = world * n_pairs * args.steps / t_reg
    mpix_per_step = ch * mosaic_h * mosaic_w / 1e6
    mpix_per_s = world * mpix_per_step * args.steps / t_fuse
    fuse_gbs = fuse_bytes * args.steps / t_fuse / 1e9            # per GPU
    if side:
        mpix_per_s_other = world * mpix_per_step * args.steps / t_fuse_other
        fuse_gbs_other = fuse_bytes_by_mode[other_mode] * args.steps / t_fuse_other / 1e9
        pairs_per_s_other = world * n_pairs * args.steps / t_reg_other
        shifts_ok_other = bool((st2_np == 0).all() and np.array_equal(out2_np[:, 1:].astype(np.int64), expected))

    cpu = None
    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        cpu = cpu_baseline(args.cpu_pairs, edge, tile, args.fuse_mode, cfg["flatfield"])

    if rank == 0:
        line = {
            "metric": "tile_pairs_registered_per_sec",
            "value": round(pairs_per_s, 1),
            "unit": "tile-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * (t_reg + t_fuse) / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.fft == "f64" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['label']}, one acquisition per GPU",
                       "pairs_per_gpu": n_pairs, "edge_width": edge, "fft": args.fft,
                       "fuse_mode": args.fuse_mode, "mosaic": [ch, mosaic_h, mosaic_w]},
            "registration": {"ms_per_step": round(1e3 * t_reg / args.steps, 3),
                             "includes": ("flatfield applied to the edge strips as they are extracted, " if ffs is not None else "")
                             + "register_pairs" + (", RCCL all-gather of pair rows" if world > 1 else ""),
                             "roofline_whole_path": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                                     "achieved": round(pairs_per_s / world * reg_bytes_per_pair / 1e9, 1),
                                                     "frac": round(pairs_per_s / world * reg_bytes_per_pair / 1e9
                                                                   / HBM_PEAK_GBS, 4),
                                                     "algorithmic_bytes_per_pair": reg_bytes_per_pair}},
            "fusion": {"metric": "fused_mosaic_MPix_per_sec", "value": round(mpix_per_s, 1), "unit": "MPix/s",
                       "dtype": "u16", "mode": args.fuse_mode, "ms_per_step": round(1e3 * t_fuse / args.steps, 3),
                       "roofline": {"bound": "hbm", "achieved": round(fuse_gbs, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(fuse_gbs / HBM_PEAK_GBS, 4),
                                    "traffic": traffic.get(f"k_fuse_{args.fuse_mode}"),
                                    "algorithmic_bytes_per_launch": int(fuse_bytes)}},
            "parity": {"planted_integer_shifts_recovered": shifts_ok, "pairs_checked": int(n_pairs),
                       "solved_positions_are_the_fused_positions": positions_ok},
            "position_solve": {"ms": round(1e3 * t_solve, 2), "where": "host, one core, on this rank's pair rows",
                               "tree_edges": len(solved.tree)},
            "roofline": dict(rooflines[dominant], kernel=dominant) if dominant else None,
            "kernels": {n: {**{k: round(v, 3) for k, v in kernels[n].items()},
                            **({"hbm_frac": rooflines[n]["frac"]} if n in rooflines else {})} for n in kernels},
        }
        if side:
            line["fusion_" + other_mode] = {
                "value": round(mpix_per_s_other, 1), "unit": "MPix/s",
                "ms_per_step": round(1e3 * t_fuse_other / args.steps, 3),
                "roofline": {"bound": "hbm", "achieved": round(fuse_gbs_other, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(fuse_gbs_other / HBM_PEAK_GBS, 4),
                             "traffic": traffic.get(f"k_fuse_{other_mode}"),
                             "algorithmic_bytes_per_launch": int(fuse_bytes_by_mode[other_mode])}}
            line["registration_" + other_fft] = {
                "value": round(pairs_per_s_other, 1), "unit": "tile-pairs/s",
                "ms_per_step": round(1e3 * t_reg_other / args.steps, 3),
                "planted_integer_shifts_recovered": shifts_ok_other}
        if side:
            line["host_to_device"] = {"GB/s": round(h2d_gbs, 1), "sample": f"{n_h2d} tiles from pinned host memory, 3 copies",
                                      "upload_of_this_workload_s": round(tiles.numel() * 2 / h2d_gbs / 1e9, 3)}
            line["ingest"] = {"what": f"one channel ({n_tiles} tiles, {n_tiles * tile * tile * 2 / 1e9:.2f} GB) uploaded in bands of "
                                      f"{rows_per_band} grid rows through two pinned buffers, pairs registered as bands arrive",
                              "upload_and_register_ms": {k: round(v, 1) for k, v in ingest_ms.items()},
                              "upload_alone_at_measured_rate_ms": round(n_tiles * tile * tile * 2 / h2d_gbs / 1e6, 1),
                              "register_alone_ms": round(1e3 * t_reg / args.steps, 1),
                              "rows_equal_resident_registration": ingest_same}
            line["registration_subpixel"] = {
                "value": round(world * n_pairs * args.steps / t_reg_sub, 1), "unit": "tile-pairs/s",
                "ms_per_step": round(1e3 * t_reg_sub / args.steps, 3), "upsample_factor": 100,
                "max_abs_shift_error_px": float(np.abs(out3_np[:, 1:] - expected).max())}
        if cpu is not None:
            cpu_pairs_s = cpu["pairs"] / cpu["t_reg"]
            line["cpu_baseline"] = {
                "value": round(cpu_pairs_s, 3), "unit": "tile-pairs/s", "cores": 1, "kind": "port",
                "sample": f"{cpu['pairs']} pairs of a {cpu['grid_side']}x{cpu['grid_side']} grid of {tile}^2 tiles "
                          f"(flatfield + register, NumPy oracle incl. least-squares polish, serial as the reference); "
                          f"fusion: {cpu['tiles_fused']} tiles, {args.fuse_mode}",
                "host_cpus": os.cpu_count(),
                "all_cores": {"value": round(cpu["pairs"] / cpu["t_reg_threads"], 3), "unit": "tile-pairs/s",
                              "cores": cpu["threads"],
                              "sample": "the same pairs, registration only (tiles already corrected), one thread per core"},
                "fusion_value": round(cpu["fused_mpix"] / cpu["t_fuse"], 2), "fusion_unit": "MPix/s",
                "gpu_over_cpu_registration": round(pairs_per_s / cpu_pairs_s, 1),
                "gpu_over_cpu_fusion": round(mpix_per_s / (cpu["fused_mpix"] / cpu["t_fuse"]), 1),
            }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
