#!/usr/bin/env python3
"""Development-container check (needs /root/reference; no test imports this): random pair tables
through the REFERENCE's position solve and through image_stitcher_amd.solve, compared field by field.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_solve_against_reference.py --cases 500
"""
import argparse
import contextlib
import io
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_solve as mg  # noqa: E402  (sets up the import of the reference)

from image_stitcher_amd import solve  # noqa: E402


def compare(case, exp):
    grid = mg.load_case(case)
    try:
        res = solve.solve_positions(grid, case["size_y"], case["size_x"], case["pou"], case["ncc_threshold"])
    except (ValueError, RuntimeError, AssertionError) as exc:
        return exp.get("raises") == type(exc).__name__, f"raised {type(exc).__name__}: {exc}"
    if "raises" in exp:
        return False, f"reference raised {exp['raises']}: {exp['message']}"
    ok = res.overlap_left == exp["overlap_left"] and res.overlap_top == exp["overlap_top"]
    for d in ("left", "top"):
        for k, flags in ((1, res.filters.valid1), (2, res.filters.valid2), (3, res.filters.valid3)):
            ok = ok and flags[d].tolist() == exp[f"{d}_valid{k}"]
        ok = ok and np.array_equal(res.filters.second[d], np.asarray(exp[f"{d}_second"], dtype=np.float64))
    ok = ok and sorted([e.f, e.t, e.direction, e.weight] for e in res.tree) == exp["tree"]
    ok = ok and res.y_pos.tolist() == exp["y_pos"] and res.x_pos.tolist() == exp["x_pos"]
    return ok, "fields differ"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    logging.disable(logging.CRITICAL)
    rng = np.random.default_rng(args.seed)
    outcomes = {}
    for i in range(args.cases):
        rows, cols = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        if rows * cols < 2:
            cols = 2
        n = rows * cols
        n_pairs = rows * (cols - 1) + cols * (rows - 1)
        sy, sx = int(rng.choice([256, 512, 2048, int(rng.integers(100, 3000))])), int(rng.choice([256, 512, 2048, int(rng.integers(100, 3000))]))
        k_miss = int(rng.integers(0, 1 + n_pairs // 4))
        missing = [(("left", "top")[int(rng.integers(2))], int(rng.integers(n))) for _ in range(k_miss)]
        isolated = [int(rng.integers(n))] if rng.random() < 0.15 else []
        case = mg.planted_case(f"r{i}", int(rng.integers(1 << 30)), rows, cols, sy, sx, float(rng.uniform(0.04, 0.4)),
                               jitter=float(rng.uniform(0, 12)), fractional=bool(rng.random() < 0.5),
                               outliers=int(rng.integers(0, 1 + n_pairs // 5)), weak=int(rng.integers(0, 1 + n_pairs // 4)),
                               missing=missing, isolated=isolated, ties=bool(rng.random() < 0.3),
                               pou=float(rng.choice([3.0, 1.0, 5.0, 10.0])), ncc_threshold=float(rng.choice([0.5, 0.1, 0.8])),
                               stage=bool(rng.random() < 0.8))
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            exp = mg.run_reference(case)
        ok, why = compare(case, exp)
        key = exp.get("raises", "positions")
        outcomes[key] = outcomes.get(key, 0) + 1
        if not ok:
            print("DISAGREEMENT in case", i, {k: case[k] for k in ("rows", "cols", "size_y", "size_x", "pou", "ncc_threshold")}, why)
            print("reference:", {k: v for k, v in exp.items() if k in ("raises", "message", "overlap_left", "overlap_top", "glue")})
            return 1
    print("SOLVE CHECK OK", args.cases, "cases; reference outcomes:", outcomes)
    return 0


if __name__ == "__main__":
    sys.exit(main())
