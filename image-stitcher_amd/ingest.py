"""Tile ingest: host tiles -> device array through two pinned buffers, registration riding along.

The step on the host side of the registration path.  The reference decodes a batch of tiles
into NumPy arrays, optionally corrects them and hands them to the pair loop
(``load_batch_images``, registration/tile_registration.py:2184-2279; batches of neighbourhoods,
:838-940).  Here the decoder writes straight into one of two pinned staging buffers (``fill``
callback), the buffer goes up on a copy stream while the other one is being filled, and as soon
as a band of grid rows is resident its pairs are registered on the compute stream, so that the
device work hides behind the PCIe transfer instead of following it.

Only scheduling lives here (PyTorch streams and events); every pixel operation is libmistitch's.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Band:
    """Grid rows [row0, row1) = tiles [tile0, tile1) of a row-major grid, and the pairs that
    become computable once the band is resident (their second tile lies in it; the first is in
    it or in the row above, which went up with the previous band)."""
    row0: int
    row1: int
    tile0: int
    tile1: int
    pairs: np.ndarray   # indices into the global pair list


def plan_bands(n_rows: int, n_cols: int, pair_idx: np.ndarray, rows_per_band: int) -> List[Band]:
    """Bands of whole grid rows, top to bottom, with the pairs each one releases."""
    if rows_per_band < 1:
        raise ValueError("rows_per_band must be >= 1")
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    if len(pair_idx) and (pair_idx[:, 0] >= pair_idx[:, 1]).any():
        raise ValueError("the first tile of a pair must precede the second in grid order")
    centre_row = pair_idx[:, 1] // n_cols
    bands = []
    for r0 in range(0, n_rows, rows_per_band):
        r1 = min(n_rows, r0 + rows_per_band)
        mine = np.nonzero((centre_row >= r0) & (centre_row < r1))[0]
        bands.append(Band(r0, r1, r0 * n_cols, r1 * n_cols, mine))
    return bands


Fill = Callable[[int, int, np.ndarray], None]
"""``fill(tile0, tile1, dst)`` writes tiles [tile0, tile1) into ``dst`` ([tile1 - tile0, H, W]
uint16, a view of pinned memory): the place for the decoder."""


def fill_from_array(host_tiles: np.ndarray) -> Fill:
    """A ``fill`` callback that copies from tiles already in host memory."""
    def fill(lo: int, hi: int, dst: np.ndarray) -> None:
        np.copyto(dst, host_tiles[lo:hi])
    return fill


class TileStager:
    """Device tile array plus two pinned staging buffers and a copy stream."""

    def __init__(self, n_tiles: int, tile_h: int, tile_w: int, band_tiles: int, device=None):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.shape = (n_tiles, tile_h, tile_w)
        self.tiles = torch.empty(self.shape, dtype=torch.int16, device=self.device)   # uint16 bits
        self.pinned = [torch.empty((band_tiles, tile_h, tile_w), dtype=torch.int16).pin_memory() for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self.free = [torch.cuda.Event(), torch.cuda.Event()]   # buffer k may be refilled
        self.n_uploaded = 0

    def upload(self, bands: Sequence[Band], fill: Fill) -> Iterator[Tuple[Band, "object"]]:
        """Fill and send the bands in turn; yields (band, event) right after a band's copy has
        been enqueued - the event completes when the band is resident."""
        torch = self.torch
        # A stager may be reused: the kernels of the previous call (on the compute stream) may still
        # be reading rows of self.tiles that this call's copies overwrite.
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        for i, band in enumerate(bands):
            k = i & 1
            n = band.tile1 - band.tile0
            if n > self.pinned[k].shape[0]:
                raise ValueError(f"band of {n} tiles exceeds the staging buffers ({self.pinned[k].shape[0]})")
            # the copy that last read this buffer (in this call or the previous one) has finished;
            # an event that was never recorded is complete
            self.free[k].synchronize()
            view = self.pinned[k][:n]
            fill(band.tile0, band.tile1, view.numpy().view(np.uint16))
            with torch.cuda.stream(self.copy_stream):
                self.tiles[band.tile0:band.tile1].copy_(view, non_blocking=True)
                self.free[k].record(self.copy_stream)
                resident = torch.cuda.Event()
                resident.record(self.copy_stream)
            self.n_uploaded = band.tile1
            yield band, resident

    def tiles_u16(self):
        return self.tiles.view(self.torch.uint16)


def register_streaming(fill: Fill, n_rows: int, n_cols: int, tile_h: int, tile_w: int, pair_idx, pair_dir,
                       edge_width: int = 256, rows_per_band: int = 2, fft: str = "f64", flatfield=None,
                       subpixel: bool = False, ctx=None, device=None, stager: Optional[TileStager] = None):
    """Upload a row-major grid band by band and register each band's pairs while the next band
    is in flight.  Returns (tiles [N, H, W] uint16 on the device, out [P, 3] float64, status [P]
    int8), rows in the order of ``pair_idx`` - the same values as one ``register_pairs`` call
    over the resident grid.  ``stager``: a TileStager of matching shape to reuse (pinning its
    buffers is the slow part of building one)."""
    import torch
    from . import registration
    pair_idx = np.asarray(pair_idx, dtype=np.int32).reshape(-1, 2)
    pair_dir = np.asarray(pair_dir, dtype=np.uint8).reshape(-1)
    bands = plan_bands(n_rows, n_cols, pair_idx, rows_per_band)
    if stager is None:
        stager = TileStager(n_rows * n_cols, tile_h, tile_w, rows_per_band * n_cols, device)
    elif stager.shape != (n_rows * n_cols, tile_h, tile_w):
        raise ValueError(f"stager holds {stager.shape}, the grid is {(n_rows * n_cols, tile_h, tile_w)}")
    dev = stager.device
    n_pairs = len(pair_idx)
    out = torch.full((n_pairs, 3), float("nan"), dtype=torch.float64, device=dev)
    status = torch.ones((n_pairs,), dtype=torch.int8, device=dev)
    compute = torch.cuda.current_stream(dev)
    tiles = stager.tiles_u16()
    if flatfield is not None and not isinstance(flatfield, torch.Tensor):
        flatfield = torch.from_numpy(np.ascontiguousarray(flatfield)).to(dev)
    for band, resident in stager.upload(bands, fill):
        if len(band.pairs) == 0:
            continue
        compute.wait_event(resident)
        sel = torch.as_tensor(band.pairs, dtype=torch.long, device=dev)
        o, s = registration.register_pairs(tiles, pair_idx[band.pairs], pair_dir[band.pairs], edge_width, fft=fft,
                                           ctx=ctx, subpixel=subpixel, flatfield=flatfield)
        out[sel] = o
        status[sel] = s
    compute.wait_stream(stager.copy_stream)
    return tiles, out, status


# ------------------------------------------------------------------------------------------------
# fusion in mosaic row bands: upload -> fuse -> download, three streams, two buffers each
# ------------------------------------------------------------------------------------------------
FillTiles = Callable[[np.ndarray, np.ndarray], None]
"""``fill(indices, dst)`` writes the tiles ``indices`` (ascending = placement order) into ``dst``
([len(indices), H, W] uint16, pinned)."""

Drain = Callable[[int, int, np.ndarray], None]
"""``drain(row0, row1, src)`` consumes mosaic rows [row0, row1) of every plane from ``src``
([n_planes, row1 - row0, mosaic_w] uint16, pinned; valid only during the call)."""


def plan_mosaic_bands(tile_y: Sequence[int], tile_h: int, mosaic_h: int, band_rows: int,
                      rows: Optional[Tuple[int, int]] = None) -> List[Tuple[int, int, np.ndarray]]:
    """(row0, row1, indices of the tiles that touch the band, ascending) for consecutive bands of
    ``band_rows`` mosaic rows over ``rows`` (default: the whole mosaic).  A tile that straddles a
    band edge is listed, and uploaded, for both bands."""
    if band_rows < 1:
        raise ValueError("band_rows must be >= 1")
    lo, hi = (0, mosaic_h) if rows is None else (int(rows[0]), int(rows[1]))
    y = np.asarray(tile_y, dtype=np.int64)
    out = []
    for r0 in range(lo, hi, band_rows):
        r1 = min(hi, r0 + band_rows)
        out.append((r0, r1, np.nonzero((y < r1) & (y + tile_h > r0))[0]))
    return out


def fuse_streaming(fill: FillTiles, drain: Drain, tile_yx, tile_plane, n_planes: int, tile_h: int, tile_w: int,
                   mosaic_h: int, mosaic_w: int, band_rows: int = 4096, flatfields=None, tile_ff=None,
                   mode: str = "overwrite", rows: Optional[Tuple[int, int]] = None, ctx=None, device=None,
                   workspace: Optional[dict] = None) -> int:
    """Fuse a mosaic that need not fit on the device, nor its tiles: per band of mosaic rows the
    touching tiles go up through pinned memory, the band is fused as a mosaic of its own (tile
    rows shifted by the band's first row) and comes down through pinned memory, the three steps
    of neighbouring bands overlapping.  A tile that straddles a band edge is decoded and sent
    once: the next band takes it from the previous band's device buffer.
    Tiles are in placement order, as for ``fuse_tiles``; the
    bands handed to ``drain`` equal the rows of the one-shot mosaic.  Returns the number of bands.
    ``workspace``: a dict that keeps the pinned and device buffers between calls of the same
    geometry (pinning gigabytes takes longer than moving them)."""
    import torch
    from . import fusion
    yx = np.asarray(tile_yx, dtype=np.int32).reshape(-1, 2)
    plane = np.asarray(tile_plane, dtype=np.int32).reshape(-1)
    ffi = None if tile_ff is None else np.asarray(tile_ff, dtype=np.int32).reshape(-1)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    bands = plan_mosaic_bands(yx[:, 0], tile_h, mosaic_h, band_rows, rows)
    if not bands:
        return 0
    max_tiles = max(1, max(len(b[2]) for b in bands))
    max_rows = max(b[1] - b[0] for b in bands)
    if flatfields is not None and not isinstance(flatfields, torch.Tensor):
        flatfields = torch.from_numpy(np.ascontiguousarray(flatfields)).to(dev)
    band_px = n_planes * max_rows * mosaic_w    # flat: a band is viewed as [n_planes, its rows, mosaic_w]
    key = (max_tiles, tile_h, tile_w, band_px, str(dev))
    ws = workspace if workspace is not None else {}
    if ws.get("key") != key:
        ws.clear()
        ws["key"] = key
        ws["pin_in"] = [torch.empty((max_tiles, tile_h, tile_w), dtype=torch.int16).pin_memory() for _ in range(2)]
        ws["dev_in"] = [torch.empty((max_tiles, tile_h, tile_w), dtype=torch.int16, device=dev) for _ in range(2)]
        ws["dev_new"] = torch.empty((max_tiles, tile_h, tile_w), dtype=torch.int16, device=dev)
        ws["dev_out"] = [torch.empty((band_px,), dtype=torch.int16, device=dev) for _ in range(2)]
        ws["pin_out"] = [torch.empty((band_px,), dtype=torch.int16).pin_memory() for _ in range(2)]
    pin_in, dev_in, dev_out, pin_out = ws["pin_in"], ws["dev_in"], ws["dev_out"], ws["pin_out"]
    dev_new = ws["dev_new"]
    prev_idx = np.zeros((0,), dtype=np.int64)
    to_dev = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev, non_blocking=True))
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    compute = torch.cuda.current_stream(dev)
    up_done = [None, None]      # H2D of the band using input pair k has finished (pinned side reusable)
    fused = [None, None]        # fuse of that band has finished (device input reusable, device output ready)
    down_done = [None, None]    # D2H of that band has finished (device output reusable, pinned output ready)
    pending = [None, None]      # (row0, row1) waiting in pin_out[k]

    def drain_slot(k):
        if pending[k] is not None:
            down_done[k].synchronize()
            r0, r1 = pending[k]
            h = r1 - r0
            drain(r0, r1, pin_out[k][:n_planes * h * mosaic_w].view(n_planes, h, mosaic_w).numpy().view(np.uint16))
            pending[k] = None

    for i, (r0, r1, idx) in enumerate(bands):
        k = i & 1
        n, h = len(idx), r1 - r0
        drain_slot(k)                           # frees pin_out[k]; its D2H also released dev_out[k]
        if up_done[k] is not None:
            up_done[k].synchronize()
        # tiles the previous band already brought up are copied across on the device
        kept = np.isin(idx, prev_idx)
        new_at, kept_at = np.nonzero(~kept)[0], np.nonzero(kept)[0]
        kept_from = np.searchsorted(prev_idx, idx[kept_at])
        n_new = len(new_at)
        if n_new:
            fill(idx[new_at], pin_in[k][:n_new].numpy().view(np.uint16))
        with torch.cuda.stream(s_in):
            if fused[k] is not None:
                s_in.wait_event(fused[k])       # the band that last read dev_in[k]
            if n_new:
                dev_new[:n_new].copy_(pin_in[k][:n_new], non_blocking=True)
                dev_in[k].index_copy_(0, to_dev(new_at), dev_new[:n_new])
            if len(kept_at):
                dev_in[k].index_copy_(0, to_dev(kept_at), dev_in[k ^ 1].index_select(0, to_dev(kept_from)))
            up_done[k] = torch.cuda.Event()
            up_done[k].record(s_in)
        prev_idx = idx
        # the band's small index arrays go up from pinned memory without stalling this thread
        local_yx = yx[idx].copy()
        local_yx[:, 0] -= r0
        d_yx, d_plane = to_dev(local_yx), to_dev(plane[idx])
        d_ff = None if ffi is None else to_dev(ffi[idx])
        compute.wait_event(up_done[k])
        if down_done[k] is not None:
            compute.wait_event(down_done[k])    # dev_out[k] has been read out
        band_out = dev_out[k][:n_planes * h * mosaic_w].view(n_planes, h, mosaic_w)
        fusion.fuse_tiles(dev_in[k][:n].view(torch.uint16), d_yx, d_plane, n_planes, h, mosaic_w,
                          flatfields=flatfields, tile_ff=d_ff, mode=mode, out=band_out.view(torch.uint16), ctx=ctx)
        fused[k] = torch.cuda.Event()
        fused[k].record(compute)
        with torch.cuda.stream(s_out):
            s_out.wait_event(fused[k])
            pin_out[k][:n_planes * h * mosaic_w].copy_(dev_out[k][:n_planes * h * mosaic_w], non_blocking=True)
            down_done[k] = torch.cuda.Event()
            down_done[k].record(s_out)
        pending[k] = (r0, r1)
    drain_slot(len(bands) & 1)
    drain_slot((len(bands) + 1) & 1)
    compute.wait_stream(s_in)
    compute.wait_stream(s_out)
    return len(bands)


def stitch_streaming(fill_registration: Fill, fill_tiles: FillTiles, drain: Drain, n_rows: int, n_cols: int,
                     tile_h: int, tile_w: int, n_planes: int, tile_plane, tile_grid_index,
                     edge_width: int = 256, pou: float = 3, ncc_threshold: float = 0.5, flatfields=None, tile_ff=None,
                     registration_flatfield=None, mode: str = "overwrite", band_rows: int = 4096,
                     rows_per_band: int = 2, subpixel: bool = False):
    """The whole path for one region, host memory to host memory: the registration channel goes up
    and is registered band by band, the pair rows are solved into tile positions on the host, and
    every plane is fused at those positions band by band.  What ``register_tiles_batched`` followed
    by ``Stitcher.stitch_region`` do in the reference (tile_registration.py:641-1221,
    stitcher.py:313-405), with the stage coordinates replaced by the solved positions.

    fill_registration  Fill for the row-major grid of the registration channel
    fill_tiles         FillTiles for all tiles to fuse, in placement order
    tile_plane         [T] destination plane of every tile to fuse
    tile_grid_index    [T] grid tile (row * n_cols + col) each of them belongs to
    Returns (PositionSolve, mosaic_h, mosaic_w, pair rows [P, 3], status [P])."""
    from . import solve, synth
    pair_idx, pair_dir = synth.grid_pairs(n_rows, n_cols)
    _, out, status = register_streaming(fill_registration, n_rows, n_cols, tile_h, tile_w, pair_idx, pair_dir,
                                        edge_width=edge_width, rows_per_band=rows_per_band,
                                        flatfield=registration_flatfield, subpixel=subpixel)
    out_np, st_np = out.cpu().numpy(), status.cpu().numpy()
    row, col = np.repeat(np.arange(n_rows), n_cols), np.tile(np.arange(n_cols), n_rows)
    pos = solve.solve_positions(solve.grid_from_pairs(row, col, pair_idx, pair_dir, out_np, st_np), tile_h, tile_w,
                                pou, ncc_threshold)
    grid_yx = np.stack([pos.y_pos, pos.x_pos], axis=1).astype(np.int32)
    mosaic_h, mosaic_w = int(grid_yx[:, 0].max()) + tile_h, int(grid_yx[:, 1].max()) + tile_w
    yx = grid_yx[np.asarray(tile_grid_index, dtype=np.int64)]
    fuse_streaming(fill_tiles, drain, yx, tile_plane, n_planes, tile_h, tile_w, mosaic_h, mosaic_w, band_rows=band_rows,
                   flatfields=flatfields, tile_ff=tile_ff, mode=mode)
    return pos, mosaic_h, mosaic_w, out_np, st_np
