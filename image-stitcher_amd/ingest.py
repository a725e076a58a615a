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
        for i, band in enumerate(bands):
            k = i & 1
            n = band.tile1 - band.tile0
            if n > self.pinned[k].shape[0]:
                raise ValueError(f"band of {n} tiles exceeds the staging buffers ({self.pinned[k].shape[0]})")
            if i >= 2:
                self.free[k].synchronize()      # the copy that last read this buffer has finished
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
