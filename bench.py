#!/usr/bin/env python3
"""Benchmark of the registration-and-fusion hot path on MI355X.

One "step" = one pass of the hot path over one synthetic acquisition that is
already resident in HBM: flatfield-correct the registration channel, register
every neighbouring tile pair, (N > 1: all-gather the pair rows over RCCL), and
fuse all channels with flatfield correction into the mosaic.  Registration and
fusion are timed in separate bracketed regions of K steps each because the
metric has two units (tile pairs/s and fused MPix/s).

N > 1 shards ONE acquisition (strong scaling, the default): rank r registers its
contiguous run of the pair list (the tiles of those grid rows and the halo row
above them are resident), one all-gather hands every rank all pair rows, every rank solves
the positions, and rank r fuses its band of mosaic chunk rows from the tiles
that touch it.  `value` = pairs of the whole grid / the slowest rank's time.
`--scaling weak` runs N independent replicas instead (one acquisition per GPU,
no exchange).

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
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: shard one acquisition over the ranks (strong) or one acquisition per rank (weak)")
    ap.add_argument("--band-align", type=int, default=256,
                    help="mosaic row bands of the ranks are multiples of this many rows (the reference writes chunks of "
                         "min(H, 12288) rows, parameters.py:695-701: a writer gathers whole bands per chunk)")
    ap.add_argument("--as-rank", type=int, default=None,
                    help="one-GPU projection of a sharded run: do the work of rank R of --of ranks (its run of pairs, its band "
                         "of mosaic rows, its resident tiles) without a process group or a collective")
    ap.add_argument("--of", type=int, default=8, help="ranks of the split that --as-rank takes one of")
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


def kernel_source_hash():
    """sha256 over the kernel sources (the stamp tools/pmc_summary.py puts into profiles/pmc_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "image-stitcher_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".hpp", ".h")):
            with open(os.path.join(src, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def load_pmc_traffic(workload, fft):
    """(HBM bytes per launch of each kernel, note) from the committed rocprofv3 --pmc passes
    (2 x FETCH_SIZE + WRITE_SIZE, see tools/pmc_summary.py).  Counters cannot be collected from
    inside this process, so they are measured by separate profiler runs of this same command
    and read back here; they are only reported while the kernel sources are the ones that
    were profiled (same hash), otherwise traffic is null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return {}, "no profiles/pmc_traffic.json"
    with open(path) as fh:
        doc = json.load(fh)
    if doc.get("workload") != workload or doc.get("fft") != fft:
        return {}, "profiles/pmc_traffic.json was collected on another workload or FFT precision"
    if doc.get("kernel_source_sha256") != kernel_source_hash():
        return {}, "profiles/pmc_traffic.json is stale: the kernel sources changed since the counter passes"
    return {k: v["hbm_bytes_per_launch"] for k, v in doc.get("kernels", {}).items()}, \
        f"rocprofv3 --pmc passes of this command ({doc.get('tag')}), kernel sources unchanged since"


def _cpu_worker_init(shm_name, shape):
    """Pool initialiser: attach the corrected tiles the parent placed in shared memory."""
    from multiprocessing import shared_memory
    global _W_SHM, _W_TILES
    _W_SHM = shared_memory.SharedMemory(name=shm_name)
    _W_TILES = np.ndarray(shape, dtype=np.uint16, buffer=_W_SHM.buf)


def _cpu_worker_register(job):
    from oracle import registration as oreg
    pair_idx, pair_dir, edge_width = job
    if len(pair_idx) == 0:
        return 0
    oreg.register_pairs(_W_TILES, pair_idx, pair_dir, edge_width, polish=True)
    return len(pair_idx)


def cpu_baseline(n_pairs_target, edge_width, tile, fuse_mode, with_ff):
    """Time the NumPy oracle on a bounded sample: one core with serial loops (as the reference runs),
    then the same pairs spread over worker PROCESSES, one per core this job may use.  Called before
    anything touches the GPU, so the children are ordinary spawned interpreters."""
    import multiprocessing as mp
    from multiprocessing import shared_memory
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
    t_ff = time.perf_counter() - t0
    t0 = time.perf_counter()
    out, _ = oreg.register_pairs(corrected, g.pair_idx[:n], g.pair_dir[:n], edge_width, polish=True)
    t_reg = time.perf_counter() - t0
    # all cores: the box gives a one-GPU job 16 CPUs (os.cpu_count() shows the whole machine)
    n_proc = max(1, min(16, os.cpu_count() or 1, n))
    t_reg_procs = None
    shm = shared_memory.SharedMemory(create=True, size=corrected.nbytes)
    try:
        np.ndarray(corrected.shape, dtype=np.uint16, buffer=shm.buf)[:] = corrected
        slices = [np.arange(i, n, n_proc) for i in range(n_proc)]
        jobs = [(g.pair_idx[ix], g.pair_dir[ix], edge_width) for ix in slices]
        warm = [(g.pair_idx[:0], g.pair_dir[:0], edge_width)] * n_proc
        with mp.get_context("spawn").Pool(n_proc, initializer=_cpu_worker_init, initargs=(shm.name, corrected.shape)) as pool:
            pool.map(_cpu_worker_register, warm, chunksize=1)      # interpreters up, modules imported
            t0 = time.perf_counter()
            done = sum(pool.map(_cpu_worker_register, jobs, chunksize=1))
            t_reg_procs = time.perf_counter() - t0
            assert done == n
    finally:
        shm.close()
        shm.unlink()
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
    return dict(pairs=n, t_reg=t_reg, t_ff=t_ff, n_tiles=len(g.tiles), t_fuse=t_fuse, fused_mpix=mpix, tiles_fused=nf, grid_side=side,
                t_reg_procs=t_reg_procs, procs=n_proc)


def band_bytes(yx, tile, band_h, mosaic_w, ch, with_ff):
    """Algorithmic bytes of fusing one band of ``band_h`` rows (tile positions relative to the band):
    feather = 2 B x tile pixels inside the band + 2 B x band pixels; overwrite = 4 B x covered + 2 B x
    uncovered band pixels; plus the flatfield planes once."""
    clipped = []
    tile_px = 0
    for y, x in yx:
        y0, y1 = max(int(y), 0), min(int(y) + tile, band_h)
        if y1 > y0:
            clipped.append((y0, y1, int(x), int(x) + tile))
            tile_px += (y1 - y0) * tile
    covered = 0
    if clipped:
        ys = np.unique([v for r in clipped for v in r[:2]])
        xs = np.unique([v for r in clipped for v in r[2:]])
        grid = np.zeros((len(ys) - 1, len(xs) - 1), dtype=bool)
        for y0, y1, x0, x1 in clipped:
            grid[np.searchsorted(ys, y0):np.searchsorted(ys, y1), np.searchsorted(xs, x0):np.searchsorted(xs, x1)] = True
        covered = int((np.diff(ys)[:, None] * np.diff(xs)[None, :] * grid).sum())
    band_px = band_h * mosaic_w
    out = {"feather": ch * (2 * tile_px + 2 * band_px), "overwrite": ch * (4 * covered + 2 * (band_px - covered))}
    if with_ff:
        for m in out:
            out[m] += ch * tile * tile * 4
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = WORKLOADS[args.workload]
    tile, ch = cfg["tile"], cfg["channels"]
    edge = args.edge_width or (256 if tile >= 2048 else tile // 8)

    # The CPU baseline runs FIRST, on rank 0 of a one-GPU run: its worker processes are started
    # before this process has touched the GPU.
    cpu, cpu_error = None, None
    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        try:   # a host-side failure (worker processes, memory) is reported in the line, it does not end the GPU run
            cpu = cpu_baseline(args.cpu_pairs, edge, tile, args.fuse_mode, cfg["flatfield"])
        except Exception as exc:  # noqa: BLE001 - reported, not raised
            cpu_error = f"{type(exc).__name__}: {exc}"[:300]

    import torch
    import torch.distributed as dist

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

    fft_flag = _lib.FFT_F64 if args.fft == "f64" else _lib.FFT_F32
    other_fft = "f32" if args.fft == "f64" else "f64"
    other_fft_flag = _lib.FFT_F32 if args.fft == "f64" else _lib.FFT_F64
    mode_id = _lib.FUSE_FEATHER if args.fuse_mode == "feather" else _lib.FUSE_OVERWRITE
    other_mode = "overwrite" if args.fuse_mode == "feather" else "feather"
    other_mode_id = _lib.FUSE_OVERWRITE if args.fuse_mode == "feather" else _lib.FUSE_FEATHER

    # ---- the partition: strong = one acquisition over all ranks, weak = one acquisition per rank ----
    strong = args.scaling == "strong" or world == 1
    part_world, part_rank = (world, rank) if strong else (1, 0)
    as_rank = args.as_rank is not None
    if as_rank:
        assert world == 1 and 0 <= args.as_rank < args.of, "--as-rank is a one-process projection"
        part_world, part_rank = args.of, args.as_rank
    seed = 3 if strong else 3 + rank
    rows, cols = cfg["rows"], cfg["cols"]
    n_tiles_grid = rows * cols
    true_yx, _, _ = synth.planted_positions(seed, rows, cols, tile)
    pair_idx, pair_dir = synth.grid_pairs(rows, cols)
    n_pairs_grid = len(pair_idx)
    yx_np = (true_yx - true_yx.min(axis=0)).astype(np.int32)
    mosaic_h, mosaic_w = int(yx_np[:, 0].max()) + tile, int(yx_np[:, 1].max()) + tile
    rs = mdist.RegistrationShard(pair_idx, rows, cols, part_world, part_rank)
    fs = mdist.FusionShard(yx_np[:, 0], tile, mosaic_h, part_world, part_rank, align=args.band_align)
    local_ids = np.union1d(rs.tile_ids, fs.tile_ids).astype(np.int64)   # resident tiles: both stages' needs
    n_local = len(local_ids)
    n_pairs = len(rs.pair_ids)                                          # pairs this rank registers

    ctx = _lib.Context(local_rank, torch.cuda.current_stream(device).cuda_stream, pair_chunk=args.pair_chunk)
    tiles, _, _, _ = synth.make_device_acquisition(rows, cols, tile, ch, seed, device, tile_ids=local_ids)

    # device-resident descriptors and outputs (nothing below allocates inside the timed loops
    # once the library workspaces have been sized by the warm-up steps)
    d_pair_idx = torch.as_tensor(rs.local_pairs(local_ids), dtype=torch.int32, device=device)
    d_pair_dir = torch.as_tensor(pair_dir[rs.pair_ids], dtype=torch.uint8, device=device)
    d_out = torch.empty((n_pairs, 3), dtype=torch.float64, device=device)
    d_status = torch.empty((n_pairs,), dtype=torch.int8, device=device)
    d_out2, d_status2 = torch.empty_like(d_out), torch.empty_like(d_status)
    band_yx = fs.local_yx(yx_np, local_ids)                             # relative to the band's first row
    d_yx = torch.as_tensor(np.tile(band_yx, (ch, 1)), dtype=torch.int32, device=device)
    d_plane = torch.as_tensor(np.repeat(np.arange(ch), n_local), dtype=torch.int32, device=device)
    band_h = fs.rows
    mosaic = torch.empty((ch, max(band_h, 1), mosaic_w), dtype=torch.int16, device=device).view(torch.uint16)
    ffs = None
    if cfg["flatfield"]:
        ffs = torch.as_tensor(np.stack([synth.make_flatfield(tile, tile, amplitude=0.2 - 0.03 * c) for c in range(ch)]),
                              device=device)
    ff_type = _lib.FF_F32 if ffs is not None else _lib.FF_NONE
    gathered = [None, None]

    def register(flags, out, status, use_ctx=None):
        """Flatfield-correct the registration channel and register this rank's pairs: with a flatfield the
        strips are corrected as they are extracted (one call), which equals correcting whole tiles first."""
        if n_pairs == 0:
            return
        handle = (use_ctx or ctx).handle
        if ffs is not None:
            check(lib.mistitch_register_pairs_flatfield(handle, ptr(tiles[0]), n_local, tile, tile, tile * tile,
                                                        ptr(ffs[0]), ff_type, ptr(d_pair_idx), ptr(d_pair_dir),
                                                        n_pairs, edge, flags, ptr(out), ptr(status)))
        else:
            check(lib.mistitch_register_pairs(handle, ptr(tiles[0]), n_local, tile, tile, tile * tile,
                                              ptr(d_pair_idx), ptr(d_pair_dir), n_pairs, edge, flags,
                                              ptr(out), ptr(status)))

    def step_registration():
        register(fft_flag, d_out, d_status)
        if part_world > 1 and not as_rank:   # every rank ends up with every pair row
            gathered[0], gathered[1] = rs.gather(d_out, d_status)
        else:
            gathered[0], gathered[1] = d_out, d_status

    def step_registration_other():
        """The same step at the other FFT precision (reported beside the headline figure)."""
        register(other_fft_flag, d_out2, d_status2)

    def step_registration_subpixel():
        """The configured step with the upsampled-DFT estimate added for every pair."""
        register(fft_flag | _lib.REG_SUBPIXEL, d_out2, d_status2)

    def step_registration_full():
        """Every method of the reference's production call (tile_registration.py:568-576 with scikit-image present):
        the consensus stage over the ten-peak candidate set, the upsampled-DFT estimate, the two-peak discrete search."""
        register(fft_flag | _lib.REG_CONSENSUS | _lib.REG_PEAKS10 | _lib.REG_SUBPIXEL, d_out2, d_status2)

    def fuse(mode):
        if band_h == 0:
            return
        check(lib.mistitch_fuse(ctx.handle, ptr(tiles), ch * n_local, tile, tile, tile * tile, ptr(d_yx), ptr(d_plane),
                                ptr(ffs), ff_type, ptr(d_plane) if ffs is not None else ptr(None), ptr(mosaic), ch,
                                band_h, mosaic_w, 0, band_h, mode))

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
    # The registration step replayed from a HIP graph (one-GPU and as-rank runs; never `value`): after a sizing call the
    # library only enqueues kernels, so the whole call is captured once on a stream of its own and replayed - what is
    # left of the gaps between its ten launches, which matter at a rank's share of an 8-way split (a step under 1 ms).
    graph_fig = None
    if world == 1 and not args.no_side and n_pairs > 0:
        try:   # a side figure must never cost the run its headline line
            g_stream = torch.cuda.Stream(device)
            g_ctx = _lib.Context(local_rank, g_stream.cuda_stream, pair_chunk=args.pair_chunk)
            g_out, g_status = torch.empty_like(d_out), torch.empty_like(d_status)
            with torch.cuda.stream(g_stream):
                register(fft_flag, g_out, g_status, g_ctx)   # sizes that context's workspaces
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=g_stream):
                register(fft_flag, g_out, g_status, g_ctx)
            g_out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(g_out, d_out) and torch.equal(g_status, d_status))
            t_graph = timed(graph.replay, args.steps)
            graph_fig = {"ms_per_step": round(1e3 * t_graph / args.steps, 3), "eager_ms_per_step": round(1e3 * t_reg / args.steps, 3),
                         "rows_equal_the_eager_step": same,
                         "what": "mistitch_register_pairs captured once into a HIP graph (torch.cuda.CUDAGraph on a stream of its own) and replayed"}
            del graph, g_ctx
        except Exception as exc:  # noqa: BLE001 - reported, not raised
            graph_fig = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.synchronize()
    # side figures (one-GPU runs): the other fusion mode and the other FFT precision, same bracketing
    side = not args.no_side and world == 1 and not as_rank
    side_error = None
    if side:
        try:   # side figures must never cost the run its headline line: a failure is reported in their place
            step_fusion_other()
            t_fuse_other = timed(step_fusion_other, args.steps)
            step_registration_other()
            t_reg_other = timed(step_registration_other, args.steps)
            out2_np, st2_np = d_out2.cpu().numpy(), d_status2.cpu().numpy()
            step_registration_subpixel()
            t_reg_sub = timed(step_registration_subpixel, args.steps)
            out3_np = d_out2.cpu().numpy()
            step_registration_full()
            t_reg_full = timed(step_registration_full, args.steps)
            out4_np, st4_np = d_out2.cpu().numpy(), d_status2.cpu().numpy()
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
            stager = ingest.TileStager(n_tiles_grid, tile, tile, rows_per_band * cols, device)
            ingest_ms = {}
            for label, fill in (("decoder_is_a_memcpy", ingest.fill_from_array(host_tiles)), ("buffers_prefilled", lambda lo, hi, dst: None)):
                for rep in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, out_s, st_s = ingest.register_streaming(fill, rows, cols, tile, tile, pair_idx, pair_dir, edge,
                                                               rows_per_band=rows_per_band, fft=args.fft,
                                                               flatfield=ffs[0] if ffs is not None else None, ctx=ctx,
                                                               stager=stager)
                    torch.cuda.synchronize()
                    ingest_ms[label] = 1e3 * (time.perf_counter() - t0)
                if label == "decoder_is_a_memcpy":
                    ingest_same = bool(torch.equal(out_s, d_out) and torch.equal(st_s, d_status))
            del stager, host_tiles
        except Exception as exc:  # noqa: BLE001 - reported, not raised
            side, side_error = False, f"{type(exc).__name__}: {exc}"[:300]
            torch.cuda.synchronize()

    # ---- parity properties at full size: every planted shift recovered (on the gathered rows) ----
    step_registration()
    torch.cuda.synchronize()
    out_np, st_np = gathered[0].cpu().numpy(), gathered[1].cpu().numpy()
    expected = true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]]
    if as_rank:   # this rank's rows only: nothing was gathered
        shifts_ok = bool((st_np == 0).all() and np.array_equal(out_np[:, 1:].astype(np.int64), expected[rs.pair_ids]))
        full = np.full((n_pairs_grid, 3), np.nan)
        full[:, 1:] = expected          # the other ranks' rows, as they would arrive: the solve below is timed on a full table
        full[:, 0] = 1.0
        full[rs.pair_ids] = out_np
        out_np, st_np = full, np.zeros(n_pairs_grid, dtype=np.int8)
    else:
        shifts_ok = bool((st_np == 0).all() and np.array_equal(out_np[:, 1:].astype(np.int64), expected))

    # ---- the host step between the two device stages: pair rows -> tile positions (never part of `value`) ----
    from image_stitcher_amd import solve
    grid_row, grid_col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
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
            register(fft_flag, d_out, d_status)
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

    # ---- algorithmic bytes (this rank's share) ----
    strip_px = edge * tile
    reg_bytes_per_pair = 2 * strip_px * 2                      # two uint16 strips in
    cplx = 16 if args.fft == "f64" else 8
    fuse_bytes_by_mode = band_bytes(band_yx, tile, band_h, mosaic_w, ch, ffs is not None)
    fuse_bytes = fuse_bytes_by_mode[args.fuse_mode]
    local_dir = pair_dir[rs.pair_ids]
    alg_bytes_per_step = {
        "k_fft_long_fwd": n_pairs * (2 * strip_px * 2 + strip_px * cplx),
        "k_fft_short_cps": n_pairs * (strip_px * cplx + (strip_px // 2) * cplx),
        "k_fft_long_inv": n_pairs * (strip_px // 2) * cplx,
        "k_ncc_accumulate": n_pairs * 2 * strip_px * 2,
        "k_transpose_strips": int((local_dir == 0).sum()) * 2 * strip_px * 2 * 2,
        "k_flatfield": n_local * tile * tile * 4 + tile * tile * 4,
        "k_stage_strips": n_pairs * 2 * strip_px * 4 + tile * tile * 4,   # strip pixels in and out, one flatfield plane
        "k_fuse_feather": fuse_bytes_by_mode["feather"], "k_fuse_overwrite": fuse_bytes_by_mode["overwrite"],
    }
    traffic, traffic_note = load_pmc_traffic(args.workload, args.fft) if world == 1 else ({}, "counters are collected on one-GPU runs")
    rooflines = {}
    for name, k in kernels.items():
        if name in alg_bytes_per_step and k["ms_per_step"] > 0:
            gbs = alg_bytes_per_step[name] / (k["ms_per_step"] * 1e-3) / 1e9
            rooflines[name] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                   frac=round(gbs / HBM_PEAK_GBS, 4), traffic=traffic.get(name),
                                   avg_launch_us=round(k["avg_launch_us"], 2),
                                   algorithmic_bytes_per_launch=int(alg_bytes_per_step[name] / k["launches_per_step"]))
    # the transform kernels are not bound by HBM alone: their float64 arithmetic against the vector peak, 5 N log2 N
    # flops per N-point complex transform (lines of `tile` points: `edge` forward, edge / 2 inverse; columns of `edge`
    # points: `tile` forward, the stored half inverse)
    FP64_PEAK_TFLOPS = 78.6
    lg = lambda v: float(np.log2(v))
    stored = tile // 2 + 8
    fft_flops_per_pair = {"k_fft_long_fwd": 5.0 * tile * lg(tile) * edge,
                          "k_fft_short_cps": 5.0 * edge * lg(edge) * (tile + stored),
                          "k_fft_long_inv": 5.0 * tile * lg(tile) * ((edge + 1) // 2)}
    for name, fl in fft_flops_per_pair.items():
        if name in kernels and kernels[name]["ms_per_step"] > 0:
            tf = n_pairs * fl / (kernels[name]["ms_per_step"] * 1e-3) / 1e12
            kernels[name]["fp64_tflops"] = tf
            kernels[name]["fp64_frac"] = tf / FP64_PEAK_TFLOPS
    # the first subpixel stage: its matrix products against the float64 matrix-core peak (one column of every
    # conjugate pair x edge frequencies x 32 node displacements, 8 flops per complex multiply-add) and the normalised
    # spectrum it streams against HBM
    if "k_ups_short" in kernels and kernels["k_ups_short"]["ms_per_step"] > 0:
        sec = kernels["k_ups_short"]["ms_per_step"] * 1e-3
        cols_half = tile // 2 + 1
        tf = n_pairs * cols_half * edge * 32 * 8.0 / sec / 1e12
        gbs = n_pairs * cols_half * (edge * 16 + 32 * 16) / sec / 1e9
        kernels["k_ups_short"].update(mfma_fp64_tflops=tf, mfma_fp64_frac=tf / FP64_PEAK_TFLOPS, hbm_gbs=gbs,
                                      hbm_frac=gbs / HBM_PEAK_GBS)
    # the dominant kernel of the configured step (the other fusion mode is a side figure)
    in_step = [n for n in rooflines if n != f"k_fuse_{other_mode}"]
    dominant = max(in_step, key=lambda n: kernels[n]["ms_per_step"]) if in_step else None

    # whole-job figures: strong = the ONE grid / the slowest rank's time; weak = every rank's own grid
    job_pairs = n_pairs_grid if strong else world * n_pairs_grid
    mpix_per_step = ch * mosaic_h * mosaic_w / 1e6 * (1 if strong else world)
    pairs_per_s = job_pairs * args.steps / t_reg
    mpix_per_s = mpix_per_step * args.steps / t_fuse
    fuse_gbs = fuse_bytes * args.steps / t_fuse / 1e9            # this rank's band / the slowest rank's time
    reg_gbs = n_pairs * reg_bytes_per_pair * args.steps / t_reg / 1e9
    # the bytes the registration passes really move per pair: every kernel's inputs read once and outputs written
    # once (the transform of a strip pair does not fit on chip: 2 x 8 MiB complex128 round trips through HBM)
    stored_cols = tile // 2 + 8                                   # columns of the half spectrum the column pass keeps
    pass_bytes_per_pair = dict(
        stage=2 * strip_px * 4 if ffs is not None else strip_px * 4,          # strips in and out (left pairs only without a flatfield)
        fft_long_fwd=2 * strip_px * 2 + strip_px * cplx,
        fft_short_cps=strip_px * cplx + stored_cols * edge * cplx,
        fft_long_inv=stored_cols * edge * cplx,
        ncc=2 * strip_px * 2)
    pass_total = sum(pass_bytes_per_pair.values())
    pass_gbs = n_pairs * pass_total * args.steps / t_reg / 1e9
    if side:
        mpix_per_s_other = mpix_per_step * args.steps / t_fuse_other
        fuse_gbs_other = fuse_bytes_by_mode[other_mode] * args.steps / t_fuse_other / 1e9
        pairs_per_s_other = job_pairs * args.steps / t_reg_other
        shifts_ok_other = bool((st2_np == 0).all() and np.array_equal(out2_np[:, 1:].astype(np.int64), expected))

    # per-rank shares, for the judge of load balance (gathered to rank 0)
    shares = None
    if world > 1:
        mine = torch.tensor([float(n_pairs), float(band_h), float(n_local)], dtype=torch.float64, device=device)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        shares = [dict(rank=r, pairs=int(v[0]), mosaic_rows=int(v[1]), tiles_resident=int(v[2])) for r, v in
                  enumerate(t.cpu().tolist() for t in every)]

    if rank == 0:
        part = (f"one acquisition sharded over {world} GPUs: contiguous runs of pairs, mosaic by row bands (multiples of {args.band_align} rows)"
                if strong and world > 1 else ("one acquisition per GPU" if world > 1 else "one GPU"))
        if as_rank:
            part = (f"PROJECTION on one GPU: the share of rank {part_rank} of {part_world} of one sharded acquisition (its pairs, its "
                    f"mosaic band, its resident tiles), no collective; value = the whole grid's pairs / this rank's time")
        line = {
            "metric": "tile_pairs_registered_per_sec",
            "value": round(pairs_per_s, 1),
            "unit": "tile-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * (t_reg + t_fuse) / args.steps, 3),
            "higher_is_better": True,
            "scaling": args.scaling,   # the series this line belongs to: N = 1 is the first point of either
            "vs_baseline": None,
            "dtype": "f64" if args.fft == "f64" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['label'].replace('edge_width 256', f'edge_width {edge}')}, {part}",
                       "pairs_in_job": job_pairs, "pairs_on_rank0": n_pairs, "edge_width": edge, "fft": args.fft,
                       "fuse_mode": args.fuse_mode, "mosaic": [ch, mosaic_h, mosaic_w],
                       "mosaic_rows_on_rank0": band_h, "tiles_resident_on_rank0": n_local},
            "registration": {"ms_per_step": round(1e3 * t_reg / args.steps, 3),
                             "includes": ("flatfield applied to the edge strips as they are extracted, " if ffs is not None else "")
                             + "register_pairs = featureless test, phase correlation, two-peak 32-candidate NCC search (what the "
                               "reference computes where scikit-image is absent; consensus / subpixel stages: registration_all_methods)"
                             + (", RCCL all-gather of pair rows" if (strong and world > 1 and not as_rank) else ""),
                             "roofline_whole_path": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                                     "achieved": round(reg_gbs, 1), "frac": round(reg_gbs / HBM_PEAK_GBS, 4),
                                                     "algorithmic_bytes_per_pair": reg_bytes_per_pair,
                                                     "per": "GPU (rank 0's pairs / the slowest rank's time)"},
                             "roofline_pass_model": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                                     "achieved": round(pass_gbs, 1), "frac": round(pass_gbs / HBM_PEAK_GBS, 4),
                                                     "bytes_per_pair": int(pass_total),
                                                     "bytes_per_pair_by_pass": {k: int(v) for k, v in pass_bytes_per_pair.items()},
                                                     "what": "inputs read once and outputs written once by every kernel of the "
                                                             "pipeline (the 2-D transform of a strip pair makes two round trips "
                                                             "through HBM); the whole-path figure above counts the two strips only"}},
            "fusion": {"metric": "fused_mosaic_MPix_per_sec", "value": round(mpix_per_s, 1), "unit": "MPix/s",
                       "dtype": "u16", "mode": args.fuse_mode, "ms_per_step": round(1e3 * t_fuse / args.steps, 3),
                       "roofline": {"bound": "hbm", "achieved": round(fuse_gbs, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(fuse_gbs / HBM_PEAK_GBS, 4),
                                    "traffic": traffic.get(f"k_fuse_{args.fuse_mode}"),
                                    "algorithmic_bytes_per_launch": int(fuse_bytes),
                                    "per": "GPU (rank 0's band / the slowest rank's time)"}},
            "parity": {"planted_integer_shifts_recovered": shifts_ok, "pairs_checked": int(len(out_np)),
                       "solved_positions_are_the_fused_positions": positions_ok},
            "position_solve": {"ms": round(1e3 * t_solve, 2), "where": "host, one core, every rank on the gathered pair rows",
                               "tree_edges": len(solved.tree)},
            "roofline": dict(rooflines[dominant], kernel=dominant) if dominant else None,
            "traffic_source": traffic_note,
            "kernels": {n: {**{k: round(v, 4 if k.startswith("fp64") else 3) for k, v in kernels[n].items()},
                            **({"hbm_frac": rooflines[n]["frac"]} if n in rooflines else {}),
                            **({"traffic_over_algorithmic": round(rooflines[n]["traffic"] / rooflines[n]["algorithmic_bytes_per_launch"], 3)}
                               if n in rooflines and rooflines[n].get("traffic") else {})} for n in kernels},
        }
        if shares is not None:
            line["ranks"] = shares
        if graph_fig is not None:
            line["registration_graph_replay"] = graph_fig
        if side_error is not None:
            line["side_figures_error"] = side_error
        if as_rank:
            line["as_rank"] = {"rank": part_rank, "of": part_world, "pairs": n_pairs, "mosaic_rows": band_h, "tiles_resident": n_local,
                               "registration_ms": round(1e3 * t_reg / args.steps, 3), "fusion_ms": round(1e3 * t_fuse / args.steps, 3)}
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
            line["host_to_device"] = {"GB/s": round(h2d_gbs, 1), "sample": f"{n_h2d} tiles from pinned host memory, 3 copies",
                                      "upload_of_this_workload_s": round(tiles.numel() * 2 / h2d_gbs / 1e9, 3)}
            line["ingest"] = {"what": f"one channel ({n_tiles_grid} tiles, {n_tiles_grid * tile * tile * 2 / 1e9:.2f} GB) uploaded in bands of "
                                      f"{rows_per_band} grid rows through two pinned buffers, pairs registered as bands arrive",
                              "upload_and_register_ms": {k: round(v, 1) for k, v in ingest_ms.items()},
                              "upload_alone_at_measured_rate_ms": round(n_tiles_grid * tile * tile * 2 / h2d_gbs / 1e6, 1),
                              "register_alone_ms": round(1e3 * t_reg / args.steps, 1),
                              "rows_equal_resident_registration": ingest_same}
            line["registration_subpixel"] = {
                "value": round(job_pairs * args.steps / t_reg_sub, 1), "unit": "tile-pairs/s",
                "ms_per_step": round(1e3 * t_reg_sub / args.steps, 3), "upsample_factor": 100,
                "max_abs_shift_error_px": float(np.abs(out3_np[:, 1:] - expected).max())}
            line["registration_all_methods"] = {
                "value": round(job_pairs * args.steps / t_reg_full, 1), "unit": "tile-pairs/s",
                "ms_per_step": round(1e3 * t_reg_full / args.steps, 3),
                "what": "consensus stage over the 160 candidates of the ten best peaks + upsampled-DFT estimate (factor 100) + "
                        "two-peak discrete search, first of the largest NCC: every method of the reference's production call "
                        "(tile_registration.py:568-576) when scikit-image is installed; its least-squares polish returns its "
                        "start point and is not run",
                "planted_integer_shifts_recovered": bool((st4_np == 0).all() and
                                                         np.array_equal(np.rint(out4_np[:, 1:]).astype(np.int64), expected))}
        if cpu_error is not None:
            line["cpu_baseline"] = {"error": cpu_error}
        if cpu is not None:
            cpu_pairs_s = cpu["pairs"] / cpu["t_reg"]
            line["cpu_baseline"] = {
                "value": round(cpu_pairs_s, 3), "unit": "tile-pairs/s", "cores": 1, "kind": "port",
                "sample": f"{cpu['pairs']} pairs of a {cpu['grid_side']}x{cpu['grid_side']} grid of {tile}^2 tiles, registration only "
                          f"(tiles already corrected; NumPy oracle: two-peak discrete search + least-squares polish, what the reference "
                          f"computes without scikit-image; serial as the reference); fusion: {cpu['tiles_fused']} tiles, {args.fuse_mode}",
                "flatfield_of_whole_tiles": {"tiles": cpu["n_tiles"], "seconds": round(cpu["t_ff"], 3),
                                             "note": "the reference corrects whole tiles when it loads them; not part of value"},
                "host_cpus": os.cpu_count(),
                "all_cores": {"value": round(cpu["pairs"] / cpu["t_reg_procs"], 3), "unit": "tile-pairs/s",
                              "cores": cpu["procs"],
                              "sample": "the same pairs, registration only (tiles already corrected), one worker PROCESS per "
                                        "core, started before the GPU was touched"},
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
