"""Integer hill-climb refinement of pair translations: CPU restatement.  TEST INFRASTRUCTURE ONLY.

Follows ``image_stitcher/registration/_constrained_refinement.py`` of the reference
(``find_local_max_integer_constrained`` :62-138, ``refine_translations`` :140-307) and is
pinned by ``tests/golden/refine_golden.json``, which ``tests/golden/make_golden_refine.py``
produced by running the reference itself.
"""
from __future__ import annotations

import itertools
from typing import Tuple

import numpy as np

from . import registration as oreg

MAX_ITERATIONS = 100     # cr:27
MIN_IMPROVEMENT = 1e-6   # cr:28


def find_local_max_integer_constrained(func, init_x, limits, max_iter: int = MAX_ITERATIONS):
    """cr:62-138: move to the best of the 3 x 3 integer neighbours (first of equal maxima in
    itertools.product order) while it strictly improves."""
    init_x = np.asarray(init_x, dtype=np.float64)
    limits = np.asarray(limits, dtype=np.float64)
    if not (np.all(limits[:, 0] <= init_x) and np.all(init_x <= limits[:, 1])):
        raise ValueError("Initial point must be within limits")
    value = func(init_x)
    if not np.isfinite(value):
        raise ValueError("Initial function value is not finite")
    x = init_x.copy()
    prev_value = float("-inf")
    for _ in range(max_iter):
        neighbors = [x + np.array(d) for d in itertools.product([-1, 0, 1], repeat=len(x))]
        valid = [n for n in neighbors if np.all(limits[:, 0] <= n) and np.all(n <= limits[:, 1])]
        if not valid:
            break
        vals = np.array([func(n) for n in valid])
        if not np.any(np.isfinite(vals)):
            raise RuntimeError("all neighbours non-finite")
        best = int(np.argmax(vals))
        if vals[best] <= value or abs(vals[best] - prev_value) < MIN_IMPROVEMENT:
            break
        x = np.array(valid[best])
        prev_value = value
        value = vals[best]
    return x, value


def refine_pair(img1: np.ndarray, img2: np.ndarray, direction: str, init_y: int, init_x: int, r: float,
                edge_width: int) -> Tuple[float, int, int]:
    """(ncc, y, x) for one pair as the loop body of refine_translations leaves it in the grid
    (cr:193-295); (nan, init_y, init_x) where the reference falls back."""
    e1, e2, off_y, off_x = oreg.edge_strips(img1, img2, direction, edge_width)
    h, w = img1.shape

    def score(params):
        gy, gx = params.astype(int)
        ty, tx = int(gy) - off_y, int(gx) - off_x
        s1 = oreg.extract_overlap_subregion(e1, ty, tx)
        s2 = oreg.extract_overlap_subregion(e2, -ty, -tx)
        if s1.size == 0 or s2.size == 0:
            return float("-inf")
        return oreg.ncc(s1, s2)

    init = np.array([int(init_y), int(init_x)])
    limits = np.array([[max(-h + 1, init[0] - r), min(h - 1, init[0] + r)],
                       [max(-w + 1, init[1] - r), min(w - 1, init[1] + r)]])
    try:
        x, v = find_local_max_integer_constrained(score, init, limits)
        return float(v), int(np.round(x[0])), int(np.round(x[1]))
    except Exception:
        return float("nan"), int(init[0]), int(init[1])


def refine_pairs(tiles: np.ndarray, pair_idx, directions, start, r: float, edge_width: int) -> np.ndarray:
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    out = np.zeros((len(pair_idx), 3), dtype=np.float64)
    for p, (i1, i2) in enumerate(pair_idx):
        out[p] = refine_pair(tiles[i1], tiles[i2], "top" if directions[p] else "left", start[p][0], start[p][1], r,
                             edge_width)
    return out
