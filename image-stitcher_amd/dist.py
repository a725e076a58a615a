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
