"""Global position solve: pairwise translations of a tile grid -> one pixel position per tile.

The step that consumes the gathered pair results.  It is tiny (a few hundred pairs), sequential
and stays on the host; NumPy only, no pandas or networkx.  It follows, in order,

* the overlap estimate ``compute_image_overlap2`` (registration/_stage_model.py:87-163) and the
  way it is called and clipped (registration/tile_registration.py:977-1005),
* the stage repeatability estimate (tile_registration.py:1007-1044),
* ``filter_translations`` (_stage_model.py:178-457): overlap/NCC window, interquartile
  filter, repeatability filter, replacement of rejected translations by medians,
* ``compute_maximum_spanning_tree`` (registration/_global_optimization.py:116-274, weights
  :318-371): raw ("first") translations, NCC plus 10 for pairs that passed the filters,
* ``compute_final_position`` (_global_optimization.py:374-524): accumulation along the tree,
  stage coordinates for tiles outside it, shift to a zero origin, rounding to int32,

with the reference's exception types for the failures a caller can see.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

VALID_TRANSLATION_WEIGHT_BONUS = 10.0      # _global_optimization.py:26
MIN_VALID_WEIGHT = -1.0                    # _global_optimization.py:27
MAX_VALID_WEIGHT = 1.0 + VALID_TRANSLATION_WEIGHT_BONUS
MAX_TRANSLATION_PX = 10000                 # _global_optimization.py:177,187
DEFAULT_PX_PER_MM = 1333.0                 # _global_optimization.py:499
NOMINAL_TILE_WIDTH_PX = 2084.0             # tile_registration.py:1171

DIRECTIONS = ("left", "top")
NCC, Y, X = 0, 1, 2


@dataclass
class TranslationGrid:
    """Tile grid with the measured translation of every tile against its left and top neighbour.

    ``first[d][i]`` = (ncc, y, x) of tile ``i`` in the frame of its neighbour in direction ``d``
    (the ``{d}_{ncc,y,x}_first`` columns of the reference's grid, tile_registration.py:832-836);
    NaN where the tile has no such neighbour or the pair was rejected.
    """
    row: np.ndarray
    col: np.ndarray
    neighbour: Dict[str, np.ndarray]           # index of the left / top neighbour, -1 = none
    first: Dict[str, np.ndarray]               # [n, 3] float64 per direction
    stage_x: Optional[np.ndarray] = None       # mm
    stage_y: Optional[np.ndarray] = None

    @property
    def n_tiles(self) -> int:
        return int(self.row.shape[0])


@dataclass
class FilterConfig:
    """Parameters of the filter chain (_stage_model.py:165-175)."""
    overlap_left: float
    overlap_top: float
    size_x: int
    size_y: int
    pou: float = 3.0
    ncc_threshold: float = 0.1
    iqr_multiplier: float = 1.5
    repeatability: float = 0.0


@dataclass
class FilterResult:
    valid1: Dict[str, np.ndarray]
    valid2: Dict[str, np.ndarray]
    valid3: Dict[str, np.ndarray]
    second: Dict[str, np.ndarray]              # [n, 3] per direction: the "_second" columns
    estimated: Dict[str, np.ndarray]           # True where ``second`` is a median, not a measurement


@dataclass
class TreeEdge:
    """One accepted pair: ``f`` is the tile that owns the measurement, ``t`` its neighbour, and
    (y, x) the position of ``f`` in the frame of ``t`` (_global_optimization.py:202-211)."""
    f: int
    t: int
    direction: str
    weight: float
    y: float
    x: float


@dataclass
class PositionSolve:
    overlap_left: float
    overlap_top: float
    repeatability: float
    filters: FilterResult
    tree: List[TreeEdge]
    y_pos: np.ndarray                          # int32, min 0
    x_pos: np.ndarray
    pixel_size_um: Optional[float] = None
    not_in_tree: List[int] = field(default_factory=list)


# ------------------------------------------------------------------------------------------------
# grid construction
# ------------------------------------------------------------------------------------------------
def neighbour_indices(row: Sequence[int], col: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """Index of the tile at (row, col-1) and at (row-1, col) for every tile, -1 where there is
    none (tile_registration.py:795-815)."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    at = {}
    for i, key in enumerate(zip(col.tolist(), row.tolist())):
        if key in at:
            raise ValueError(f"two tiles at grid position (col={key[0]}, row={key[1]})")
        at[key] = i
    left = np.array([at.get((c - 1, r), -1) for c, r in zip(col.tolist(), row.tolist())], dtype=np.int64)
    top = np.array([at.get((c, r - 1), -1) for c, r in zip(col.tolist(), row.tolist())], dtype=np.int64)
    return left, top


def grid_from_pairs(row, col, pair_idx, pair_dir, results, status=None, stage_x=None, stage_y=None) -> TranslationGrid:
    """Scatter the rows of a batched registration into a grid.

    ``pair_idx[p] = (first, second)``: the result of pair ``p`` belongs to tile ``second`` under
    direction ``pair_dir[p]`` (0 left, 1 top), as the reference files the value of a
    neighbourhood under its centre tile (tile_registration.py:937-940).  Pairs with a non-zero
    ``status`` of "featureless" stay NaN, as do pairs that were never registered.
    """
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    left, top = neighbour_indices(row, col)
    n = row.shape[0]
    first = {d: np.full((n, 3), np.nan, dtype=np.float64) for d in DIRECTIONS}
    pair_idx = np.asarray(pair_idx, dtype=np.int64).reshape(-1, 2)
    pair_dir = np.asarray(pair_dir).reshape(-1)
    results = np.asarray(results, dtype=np.float64).reshape(-1, 3)
    if not (pair_idx.shape[0] == pair_dir.shape[0] == results.shape[0]):
        raise ValueError("pair_idx, pair_dir and results disagree on the number of pairs")
    nb = {"left": left, "top": top}
    for p in range(pair_idx.shape[0]):
        d = DIRECTIONS[int(pair_dir[p])]
        a, b = int(pair_idx[p, 0]), int(pair_idx[p, 1])
        if nb[d][b] != a:
            raise ValueError(f"pair {p}: tile {a} is not the {d} neighbour of tile {b}")
        if status is not None and int(status[p]) == 1:      # featureless: the reference stores nothing
            continue
        first[d][b] = results[p]
    as_f = (lambda v: None if v is None else np.asarray(v, dtype=np.float64))
    return TranslationGrid(row, col, nb, first, as_f(stage_x), as_f(stage_y))


# ------------------------------------------------------------------------------------------------
# statistics that parameterise the filters
# ------------------------------------------------------------------------------------------------
def _between(v: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """Closed interval test that is False for NaN (pandas ``Series.between``)."""
    with np.errstate(invalid="ignore"):
        return (v >= lo) & (v <= hi)


def _above(v: np.ndarray, thr: float) -> np.ndarray:
    with np.errstate(invalid="ignore"):
        return v > thr


def _median(v: np.ndarray) -> float:
    """Median of a short vector without NaN: middle element, or the mean of the two middle ones
    (what ``np.median`` and the pandas median return, minus their dispatch cost)."""
    a = np.sort(v, axis=None)
    n = a.shape[0]
    if n == 0:
        return float("nan")
    return float(a[n // 2]) if n & 1 else float((a[n // 2 - 1] + a[n // 2]) / 2)


def image_overlap(first: np.ndarray, rows: np.ndarray, size_y: int, size_x: int) -> Tuple[float, float]:
    """Median (y / size_y, x / size_x) over the selected pairs (_stage_model.py:87-163)."""
    t = np.stack([first[rows, Y] / size_y, first[rows, X] / size_x])
    t = t[:, np.all(np.isfinite(t), axis=0)]
    if t.shape[1] < 1:
        raise ValueError(f"Insufficient valid translations ({t.shape[1]}). Need at least 1.")
    med = np.median(t, axis=1)
    if np.any(np.abs(med) > 1.0):
        raise ValueError("Invalid overlap values > 100% detected")
    return float(med[0]), float(med[1])


def estimate_overlaps(grid: TranslationGrid, size_y: int, size_x: int, pou: float,
                      ncc_threshold: float) -> Tuple[float, float, bool, bool]:
    """Overlap percentages along x for left pairs and along y for top pairs, clipped to
    [pou, 100 - pou]; 50 for a direction with no pair above the threshold
    (tile_registration.py:977-1005)."""
    good = {d: _above(grid.first[d][:, NCC], ncc_threshold) for d in DIRECTIONS}
    has_left, has_top = bool(good["left"].any()), bool(good["top"].any())
    if not has_left and not has_top:
        raise ValueError("No good initial pairs found - tiles may not have sufficient overlap")
    overlap_left = overlap_top = 50.0
    if has_left:
        overlap_left = float(np.clip(100 - image_overlap(grid.first["left"], good["left"], size_y, size_x)[1] * 100,
                                     pou, 100 - pou))
    if has_top:
        overlap_top = float(np.clip(100 - image_overlap(grid.first["top"], good["top"], size_y, size_x)[0] * 100,
                                    pou, 100 - pou))
    return overlap_left, overlap_top, has_left, has_top


def estimate_repeatability(grid: TranslationGrid, size_y: int, size_x: int, overlap_left: float, overlap_top: float,
                           has_left: bool, has_top: bool, pou: float, ncc_threshold: float) -> float:
    """Half the spread of the accepted translations, the larger of: along the stepping axis over
    the whole grid, and across it within one grid line (tile_registration.py:1007-1044)."""
    rs = []
    for d, along, across, size, overlap, present, line in (
            ("top", Y, X, size_y, overlap_top, has_top, grid.col),
            ("left", X, Y, size_x, overlap_left, has_left, grid.row)):
        f = grid.first[d]
        if not present:
            rs.append(0.0)
            continue
        ok = _between(f[:, along], size * (100 - overlap - pou) / 100, size * (100 - overlap + pou) / 100) \
            & _above(f[:, NCC], ncc_threshold)
        if not ok.any():
            rs.append(0.0)
            continue
        w1 = f[ok, along]
        r1 = float(np.ceil((w1.max() - w1.min()) / 2)) if w1.shape[0] > 1 else 0.0
        spreads = []
        for key in np.unique(line[ok]):
            w2 = f[ok & (line == key), across]
            w2 = w2[~np.isnan(w2)]
            if w2.shape[0] > 1:
                spreads.append(w2.max() - w2.min())
        r2 = float(np.ceil(max(spreads) / 2)) if spreads else 0.0
        rs.append(max(r1, r2))
    return float(max(rs))


# ------------------------------------------------------------------------------------------------
# filter chain
# ------------------------------------------------------------------------------------------------
def filter_by_overlap_and_correlation(t: np.ndarray, ncc: np.ndarray, overlap: float, size: int, pou: float = 3,
                                      ncc_threshold: float = 0.1) -> np.ndarray:
    """_stage_model.py:248-283."""
    if not 0 < overlap < 100:
        raise ValueError("Overlap must be between 0 and 100 percent")
    if pou < 0 or pou >= overlap:
        raise ValueError("Invalid percent overlap uncertainty (pou)")
    if ncc_threshold < 0 or ncc_threshold > 1:
        raise ValueError("NCC threshold must be between 0 and 1")
    return _between(t, size * (100 - overlap - pou) / 100, size * (100 - overlap + pou) / 100) & _above(ncc, ncc_threshold)


def filter_outliers(t: np.ndarray, isvalid: np.ndarray, w: float = 1.5) -> np.ndarray:
    """Interquartile filter with a range of at least one pixel (_stage_model.py:285-318)."""
    if w <= 0:
        raise ValueError("IQR multiplier must be positive")
    sel = t[isvalid]
    if sel.shape[0] < 1:
        return isvalid.copy()
    q1, q3 = np.quantile(sel, (0.25, 0.75))
    iqr = max(1, abs(q3 - q1))
    return isvalid & _between(t, q1 - w * iqr, q3 + w * iqr)


def filter_translations(grid: TranslationGrid, cfg: FilterConfig) -> FilterResult:
    """The whole chain (_stage_model.py:178-245).  Left pairs are judged by x and grouped by grid
    column, top pairs by y and grouped by grid row (_stage_model.py:339-372, :414-415)."""
    spec = {"left": (X, cfg.overlap_left, cfg.size_x, grid.col), "top": (Y, cfg.overlap_top, cfg.size_y, grid.row)}
    valid1, valid2, valid3, second, estimated = {}, {}, {}, {}, {}
    for d in ("top", "left"):
        axis, overlap, size, _ = spec[d]
        f = grid.first[d]
        valid1[d] = filter_by_overlap_and_correlation(f[:, axis], f[:, NCC], overlap, size, cfg.pou, cfg.ncc_threshold)
    for d in ("top", "left"):
        valid2[d] = filter_outliers(grid.first[d][:, spec[d][0]], valid1[d], cfg.iqr_multiplier)
    r = cfg.repeatability
    for d in DIRECTIONS:
        f, line = grid.first[d], spec[d][3]
        v3 = np.zeros(grid.n_tiles, dtype=bool)
        for key in np.unique(line):
            g = line == key
            ok = g & valid2[d]
            if not ok.any():
                continue
            medy, medx = _median(f[ok, Y]), _median(f[ok, X])
            v3[g] = (_between(f[g, Y], medy - r, medy + r) & _between(f[g, X], medx - r, medx + r)
                     & _above(f[g, NCC], cfg.ncc_threshold))
        valid3[d] = v3
    # replacement of what failed (_stage_model.py:375-457)
    for d in DIRECTIONS:
        f, line = grid.first[d], spec[d][3]
        s = np.full((grid.n_tiles, 3), np.nan, dtype=np.float64)
        s[valid3[d]] = f[valid3[d]]
        est = np.zeros(grid.n_tiles, dtype=bool)
        for key in np.unique(line):
            g = line == key
            ok = g & valid3[d]
            if not ok.any():
                continue
            bad = g & ~valid3[d]
            s[bad, Y] = _median(f[ok, Y])
            s[bad, X] = _median(f[ok, X])
            s[bad, NCC] = 0.0
            est |= bad
        for axis in (X, Y):
            hole = np.isnan(s[:, axis])
            if hole.any():
                rest = s[~hole, axis]
                s[hole, axis] = _median(rest)
                s[hole, NCC] = 0.0
                est |= hole
        second[d], estimated[d] = s, est
    for d in DIRECTIONS:
        # the reference ends with a bare assert, which is what a grid with no usable pair in one
        # direction (a single row or column of tiles) runs into (_stage_model.py:453-455)
        assert np.all(np.isfinite(second[d][:, (Y, X)])), f"no finite {d} translation to fall back on"
    return FilterResult(valid1, valid2, valid3, second, estimated)


# ------------------------------------------------------------------------------------------------
# spanning tree and positions
# ------------------------------------------------------------------------------------------------
def candidate_edges(grid: TranslationGrid, valid3: Dict[str, np.ndarray]) -> List[TreeEdge]:
    """Every measured pair that may enter the tree, in the reference's insertion order: tile by
    tile, left before top (_global_optimization.py:139-211, weights :318-371)."""
    ok = {}
    for d in DIRECTIONS:
        f = grid.first[d]
        with np.errstate(invalid="ignore"):
            ok[d] = ((grid.neighbour[d] >= 0) & (f[:, NCC] >= MIN_VALID_WEIGHT) & (f[:, NCC] <= MAX_VALID_WEIGHT)
                     & (np.abs(f[:, Y]) <= MAX_TRANSLATION_PX) & (np.abs(f[:, X]) <= MAX_TRANSLATION_PX))
    edges = []
    for i in np.nonzero(ok["left"] | ok["top"])[0].tolist():
        for d in DIRECTIONS:
            if ok[d][i]:
                ncc, y, x = grid.first[d][i].tolist()
                weight = ncc + (VALID_TRANSLATION_WEIGHT_BONUS if valid3[d][i] else 0.0)
                edges.append(TreeEdge(i, int(grid.neighbour[d][i]), d, weight, y, x))
    return edges


def _graph_order(edges: List[TreeEdge]) -> Tuple[List[int], List[TreeEdge]]:
    """Nodes in insertion order and edges in the order an undirected adjacency structure reports
    them: by first endpoint in node order, then by the order the edges were attached to it."""
    adj: Dict[int, Dict[int, TreeEdge]] = {}
    for e in edges:
        adj.setdefault(e.f, {})
        adj.setdefault(e.t, {})
        adj[e.f][e.t] = e
        adj[e.t][e.f] = e
    seen, ordered = set(), []
    for u, nbrs in adj.items():
        for v, e in nbrs.items():
            if v not in seen:
                ordered.append(e)
        seen.add(u)
    return list(adj.keys()), ordered


class _Forest:
    def __init__(self, nodes):
        self.parent = {u: u for u in nodes}

    def find(self, u):
        while self.parent[u] != u:
            self.parent[u] = self.parent[self.parent[u]]
            u = self.parent[u]
        return u

    def union(self, a, b) -> bool:
        ra, rb = self.find(a), self.find(b)
        if ra == rb:
            return False
        self.parent[rb] = ra
        return True


def maximum_spanning_tree(grid: TranslationGrid, valid3: Dict[str, np.ndarray]) -> List[TreeEdge]:
    """Kruskal over the candidate edges, heaviest first, equal weights in graph order (the
    stable descending sort of networkx's ``maximum_spanning_tree``, which the reference calls at
    _global_optimization.py:264)."""
    edges = candidate_edges(grid, valid3)
    if not edges:
        raise ValueError("No valid connections between tiles")
    nodes, ordered = _graph_order(edges)
    if len(ordered) < len(nodes) - 1:
        raise ValueError(f"Insufficient edges for spanning tree: {len(ordered)} edges for {len(nodes)} nodes. "
                         f"Minimum required: {len(nodes) - 1}. Check NCC threshold or image quality.")
    probe = _Forest(nodes)
    for e in ordered:
        probe.union(e.f, e.t)
    roots: Dict[int, int] = {}
    for u in nodes:
        roots[probe.find(u)] = roots.get(probe.find(u), 0) + 1
    if len(roots) > 1:
        sizes = list(roots.values())
        raise ValueError(f"Tile graph has {len(roots)} disconnected components (sizes: {sizes}). "
                         "Check NCC threshold, image quality, or tile overlap. "
                         "Tiles may be too far apart or have insufficient overlap for registration.")
    forest = _Forest(nodes)
    tree = []
    for e in sorted(ordered, key=lambda e: e.weight, reverse=True):
        if forest.union(e.f, e.t):
            tree.append(e)
    return tree


def estimate_pixel_size_um(grid: TranslationGrid) -> Optional[float]:
    """Pixel size guessed from the stage step of the first tile that has a left neighbour, with
    the nominal tile width the reference hard-codes (tile_registration.py:1161-1174)."""
    if grid.stage_x is None:
        return None
    for i in range(grid.n_tiles - 1):
        nb = int(grid.neighbour["left"][i])
        if nb >= 0:
            dx_mm = abs(float(grid.stage_x[i]) - float(grid.stage_x[nb]))
            if dx_mm > 0:
                return dx_mm * 1000.0 / NOMINAL_TILE_WIDTH_PX
    return None


def final_positions(grid: TranslationGrid, tree: List[TreeEdge], source_index: int = 0,
                    pixel_size_um: Optional[float] = None) -> Tuple[np.ndarray, np.ndarray, List[int]]:
    """Walk the tree from ``source_index`` adding each edge's translation when stepping from the
    neighbour to the measuring tile and subtracting it the other way, place tiles outside the
    tree from their stage coordinates, shift to a zero origin and round half to even
    (_global_optimization.py:374-524)."""
    if not tree:
        raise ValueError("Empty spanning tree")
    n = grid.n_tiles
    if not 0 <= source_index < n:
        raise ValueError(f"source_index {source_index} not found in grid DataFrame")
    try:
        adj: Dict[int, List[Tuple[int, TreeEdge]]] = {}
        for e in tree:
            adj.setdefault(e.f, []).append((e.t, e))
            adj.setdefault(e.t, []).append((e.f, e))
        pos = np.full((n, 2), np.nan, dtype=np.float64)
        todo = set(adj)
        while todo:
            start = source_index if source_index in todo else min(todo)
            pos[start] = 0.0
            queue, done = [start], {start}
            while queue:
                cur = queue.pop(0)
                for nb, e in adj[cur]:
                    if nb in done:
                        continue
                    done.add(nb)
                    queue.append(nb)
                    sign = 1.0 if cur == e.t else -1.0
                    pos[nb, 0] = pos[cur, 0] + sign * e.y
                    pos[nb, 1] = pos[cur, 1] + sign * e.x
            todo -= done
        outside = sorted(set(range(n)) - set(adj))
        if outside:
            if grid.stage_x is None or grid.stage_y is None:
                raise RuntimeError(f"Cannot position {len(outside)} isolated tiles: no stage coordinates available")
            px_per_mm = 1000.0 / pixel_size_um if pixel_size_um is not None else DEFAULT_PX_PER_MM
            ref_y, ref_x = pos[source_index]
            for i in outside:
                pos[i, 1] = ref_x + (grid.stage_x[i] - grid.stage_x[source_index]) * px_per_mm
                pos[i, 0] = ref_y + (grid.stage_y[i] - grid.stage_y[source_index]) * px_per_mm
        if np.isnan(pos).any():
            raise RuntimeError("Failed to compute positions for all tiles")
        pos -= pos.min(axis=0, keepdims=True)
        pos = np.round(pos).astype(np.int32)
        return pos[:, 0].copy(), pos[:, 1].copy(), outside
    except Exception as exc:  # noqa: BLE001 - the reference wraps everything (_global_optimization.py:523-524)
        raise RuntimeError(f"Failed to compute final positions: {exc}")


def solve_positions(grid: TranslationGrid, size_y: int, size_x: int, pou: float = 3, ncc_threshold: float = 0.5
                    ) -> PositionSolve:
    """Pair results -> tile positions, as the tail of the reference's batched registration does
    (tile_registration.py:977-1176; defaults of pou and ncc_threshold from :646-647)."""
    overlap_left, overlap_top, has_left, has_top = estimate_overlaps(grid, size_y, size_x, pou, ncc_threshold)
    r = estimate_repeatability(grid, size_y, size_x, overlap_left, overlap_top, has_left, has_top, pou, ncc_threshold)
    cfg = FilterConfig(overlap_left, overlap_top, size_x, size_y, pou, ncc_threshold, 1.5, r)
    filters = filter_translations(grid, cfg)
    try:
        tree = maximum_spanning_tree(grid, filters.valid3)
    except ValueError:
        raise
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError(f"Failed to compute maximum spanning tree: {exc}")
    px = estimate_pixel_size_um(grid)
    y_pos, x_pos, outside = final_positions(grid, tree, 0, px)
    return PositionSolve(overlap_left, overlap_top, r, filters, tree, y_pos, x_pos, px, outside)
