"""One-process-per-GPU partitioning of the hot path.

Registration shards naturally by tile pair (each pair needs two tiles and
nothing else); the only exchange is one all-gather of the per-pair rows
``(ncc, y, x)`` + status so that the global position solve sees every pair.
Fusion shards by disjoint mosaic row bands and needs no exchange: every band
resolves its pixels from the same global placement order.

Works with any ``torch.distributed`` backend: ``nccl`` (RCCL over xGMI) for CUDA
tensors, ``gloo`` for the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def band_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``range(n_items)``: first ``n_items % world`` ranks get one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pairs_by_grid_rows(pair_idx: np.ndarray, n_cols: int, n_rows: int, world: int, rank: int
                             ) -> Tuple[np.ndarray, np.ndarray]:
    """Pairs owned by ``rank`` when grid rows are dealt out in contiguous bands.

    A pair belongs to the band of its SECOND tile (the neighbourhood centre), so
    "left" pairs are band-local and a band's first row of "top" pairs needs the
    last tile row of the band above: that halo row is simply loaded by both ranks.
    Returns (indices into the global pair list, sorted unique tile indices needed).
    """
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    r0, r1 = band_bounds(n_rows, world, rank)
    centre_row = pair_idx[:, 1] // n_cols
    mine = np.nonzero((centre_row >= r0) & (centre_row < r1))[0]
    tiles = np.unique(pair_idx[mine].reshape(-1)) if len(mine) else np.zeros((0,), dtype=pair_idx.dtype)
    return mine, tiles


def shard_pairs_evenly(pair_idx: np.ndarray, world: int, rank: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pairs owned by ``rank`` when the pair list (in the reference's visiting order: row-major by
    centre tile, left then top) is cut into ``world`` contiguous runs of equal length (+-1).  A run
    covers about ``n_rows / world`` grid rows plus a partial row at either end; the tiles it needs
    are those of its pairs (the row above the run's first centre tiles is the halo).
    Returns (indices into the global pair list, sorted unique tile indices needed)."""
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    lo, hi = band_bounds(len(pair_idx), world, rank)
    mine = np.arange(lo, hi)
    tiles = np.unique(pair_idx[mine].reshape(-1)) if len(mine) else np.zeros((0,), dtype=pair_idx.dtype)
    return mine, tiles


def shard_mosaic_rows(mosaic_h: int, world: int, rank: int, align: int = 1) -> Tuple[int, int]:
    """Row band ``[row0, row1)`` of the mosaic for ``rank``; band edges are multiples of
    ``align`` (the output chunk height) except the last."""
    n_chunks = (mosaic_h + align - 1) // align
    c0, c1 = band_bounds(n_chunks, world, rank)
    return min(c0 * align, mosaic_h), min(c1 * align, mosaic_h)


def tiles_touching_rows(tile_y: Sequence[int], tile_h: int, row0: int, row1: int) -> np.ndarray:
    """Indices of tiles whose rows intersect the band (a tile straddling a band edge is
    used by both neighbours: duplicated input, no exchange)."""
    y = np.asarray(tile_y)
    return np.nonzero((y < row1) & (y + tile_h > row0))[0]


def all_gather_pair_results(out, status, counts: Sequence[int]):
    """All-gather the per-rank rows into global pair order (rank-major concatenation).

    out [P_local, 3] float64, status [P_local] int8 on any device the process
    group's backend supports; ``counts[r]`` = pairs owned by rank r.  One
    collective each for rows and status, padded to the largest shard.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    pmax = max(int(c) for c in counts) if world else 0
    pad_out = torch.full((pmax, 3), float("nan"), dtype=torch.float64, device=out.device)
    pad_st = torch.ones((pmax,), dtype=torch.int8, device=out.device)
    n = int(out.shape[0])
    pad_out[:n] = out
    pad_st[:n] = status
    all_out = torch.empty((world, pmax, 3), dtype=torch.float64, device=out.device)
    all_st = torch.empty((world, pmax), dtype=torch.int8, device=out.device)
    if dist.get_backend() == "nccl":  # RCCL: one flat collective each, on the device
        dist.all_gather_into_tensor(all_out, pad_out)
        dist.all_gather_into_tensor(all_st, pad_st)
    else:  # gloo (CPU tests, single-GPU rehearsals): list variant on host copies
        parts_out = [torch.empty((pmax, 3), dtype=torch.float64) for _ in range(world)]
        parts_st = [torch.empty((pmax,), dtype=torch.int8) for _ in range(world)]
        dist.all_gather(parts_out, pad_out.cpu())
        dist.all_gather(parts_st, pad_st.cpu())
        all_out = torch.stack(parts_out).to(out.device)
        all_st = torch.stack(parts_st).to(out.device)
    rows = torch.cat([all_out[r, : int(counts[r])] for r in range(world)])
    st = torch.cat([all_st[r, : int(counts[r])] for r in range(world)])
    return rows, st


def scatter_to_global(mine_lists: List[np.ndarray], rows, status, n_pairs: int):
    """Reorder rank-major gathered rows back into the global pair numbering."""
    import torch

    order = np.concatenate([np.asarray(m) for m in mine_lists]) if mine_lists else np.zeros((0,), dtype=np.int64)
    g_rows = torch.full((n_pairs, 3), float("nan"), dtype=torch.float64, device=rows.device)
    g_st = torch.ones((n_pairs,), dtype=torch.int8, device=rows.device)
    idx = torch.as_tensor(order, dtype=torch.long, device=rows.device)
    g_rows[idx] = rows
    g_st[idx] = status
    return g_rows, g_st


# ------------------------------------------------------------------------------------------------
# one acquisition sharded over the ranks of a node: registration by grid-row bands of pairs,
# one all-gather of the pair rows, the (tiny) position solve replicated, fusion by mosaic row bands
# ------------------------------------------------------------------------------------------------
class RegistrationShard:
    """What ``rank`` of ``world`` registers of one grid: its pairs (the reference's serial pair loop,
    tile_registration.py:336-376, cut into contiguous grid-row bands) and the tiles they need."""

    def __init__(self, pair_idx, n_rows: int, n_cols: int, world: int, rank: int, by: str = "pairs"):
        """``by``: "pairs" = contiguous runs of equal pair counts (balanced: 95 pairs each for a 20 x 20
        grid on 8 ranks), "grid_rows" = whole grid rows per rank (3 or 2 rows each there)."""
        self.world, self.rank = int(world), int(rank)
        self.pair_idx = np.asarray(pair_idx, dtype=np.int32).reshape(-1, 2)
        self.n_pairs = len(self.pair_idx)
        if by == "pairs":
            cut = lambda r: shard_pairs_evenly(self.pair_idx, world, r)
        elif by == "grid_rows":
            cut = lambda r: shard_pairs_by_grid_rows(self.pair_idx, n_cols, n_rows, world, r)
        else:
            raise ValueError(f"unknown partition {by!r}")
        self.pair_lists = [cut(r)[0] for r in range(world)]
        self.counts = [len(m) for m in self.pair_lists]
        self.pair_ids, self.tile_ids = cut(rank)
        self._gather_ws = {}

    def local_pairs(self, local_ids) -> np.ndarray:
        """This rank's pairs with tile indices renumbered into ``local_ids`` (sorted global ids of the
        tiles the rank holds; must contain ``tile_ids``)."""
        local_ids = np.asarray(local_ids)
        pos = np.searchsorted(local_ids, self.pair_idx[self.pair_ids].reshape(-1))
        if len(pos) and (pos.max() >= len(local_ids) or (local_ids[pos] != self.pair_idx[self.pair_ids].reshape(-1)).any()):
            raise ValueError("a tile this rank's pairs need is not resident")
        return pos.reshape(-1, 2).astype(np.int32)

    def gather(self, out_local, status_local):
        """Rows of every rank in global pair order: the one exchange of the path (a single all-gather of
        32 bytes per pair: ncc, y, x and the status as a fourth float64); with one rank, or outside a process
        group, a plain reorder.  Over RCCL the result is on the HOST (where the position solve consumes it)."""
        import torch
        import torch.distributed as dist

        if self.world > 1:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("world > 1 needs an initialised process group")
            if dist.get_backend() != "nccl":
                rows, st = all_gather_pair_results(out_local, status_local, self.counts)
                return scatter_to_global(self.pair_lists, rows, st, self.n_pairs)
            # RCCL: ONE collective.  The status rides as a fourth float64 column of the pair rows (32 bytes per
            # pair); the gathered [world * pmax, 4] block comes to the host in one copy, where the position solve
            # runs anyway, and is put into global pair order with NumPy: no indexed device kernels, and after the
            # first call no allocation.  Returned tensors are HOST tensors, reused by the next call.
            dev = out_local.device
            ws = self._gather_ws.get(dev)
            if ws is None:
                pmax = max(self.counts)
                src = np.concatenate([r * pmax + np.arange(c) for r, c in enumerate(self.counts)])   # valid slots of [world, pmax]
                dst = np.concatenate([np.asarray(m) for m in self.pair_lists])                     # their global pair ids
                pin = dev.type == "cuda"
                ws = dict(pmax=pmax, src=src, dst=dst,
                          pad=torch.full((pmax, 4), float("nan"), dtype=torch.float64, device=dev),
                          all=torch.empty((self.world * pmax, 4), dtype=torch.float64, device=dev),
                          host=torch.empty((self.world * pmax, 4), dtype=torch.float64, pin_memory=pin),
                          rows=torch.full((self.n_pairs, 3), float("nan"), dtype=torch.float64),
                          st=torch.ones((self.n_pairs,), dtype=torch.int8))
                self._gather_ws[dev] = ws
            n = int(out_local.shape[0])
            ws["pad"][:n, :3] = out_local
            ws["pad"][:n, 3] = status_local      # int8 -> float64, exact
            dist.all_gather_into_tensor(ws["all"], ws["pad"])
            ws["host"].copy_(ws["all"])           # device -> host: waits for the collective on the current stream
            block = ws["host"].numpy()
            ws["rows"].numpy()[ws["dst"]] = block[ws["src"], :3]
            ws["st"].numpy()[ws["dst"]] = block[ws["src"], 3].astype(np.int8)
            return ws["rows"], ws["st"]
        return scatter_to_global(self.pair_lists, out_local, status_local, self.n_pairs)


class FusionShard:
    """The band of mosaic rows ``rank`` of ``world`` fuses (whole output chunk rows, stitcher.py:388-405
    cut by output row) and the tiles that touch it; a tile straddling a band edge is held by both ranks."""

    def __init__(self, tile_y, tile_h: int, mosaic_h: int, world: int, rank: int, align: int = 2048):
        self.row0, self.row1 = shard_mosaic_rows(int(mosaic_h), world, rank, align)
        self.tile_ids = tiles_touching_rows(np.asarray(tile_y), tile_h, self.row0, self.row1) \
            if self.row1 > self.row0 else np.zeros((0,), dtype=np.int64)

    @property
    def rows(self) -> int:
        return self.row1 - self.row0

    def local_yx(self, tile_yx, ids=None) -> np.ndarray:
        """Positions of ``ids`` (default: the touching tiles) relative to the band's first row: the band is
        fused as a mosaic of its own, rows above it simply fall outside."""
        ids = self.tile_ids if ids is None else np.asarray(ids)
        yx = np.asarray(tile_yx, dtype=np.int32).reshape(-1, 2)[ids].copy()
        yx[:, 0] -= self.row0
        return yx
