#!/usr/bin/env python3
"""Development-container check (needs /root/reference; no test imports this): random grids and start
shifts through the REFERENCE's refine_translations and through oracle.refine.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_refine_against_reference.py --cases 200
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
import make_golden_refine as mr  # noqa: E402  (sets up the import of the reference)

from image_stitcher_amd import synth  # noqa: E402
from oracle import refine as oref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    logging.disable(logging.CRITICAL)
    rng = np.random.default_rng(args.seed)
    n_pairs = 0
    for i in range(args.cases):
        spec = dict(name=f"r{i}", seed=int(rng.integers(1 << 30)), rows=int(rng.integers(1, 3)), cols=int(rng.integers(2, 4)),
                    tile_h=int(rng.integers(40, 260)), tile_w=int(rng.integers(40, 260)), overlap=float(rng.uniform(0.1, 0.35)),
                    jitter=int(rng.integers(0, 4)), edge_width=int(rng.integers(8, 80)), r=float(rng.choice([0, 1, 2.5, 6, 20])),
                    perturb_seed=int(rng.integers(1 << 30)), perturb=int(rng.integers(0, 7)))
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref = mr.run(spec)
        g = synth.make_grid(spec["seed"], spec["rows"], spec["cols"], spec["tile_h"], spec["tile_w"], overlap=spec["overlap"],
                            jitter=spec["jitter"])
        got = oref.refine_pairs(g.tiles, g.pair_idx, g.pair_dir, np.asarray(ref["start"]), spec["r"], spec["edge_width"])
        want = np.asarray(ref["results"], dtype=np.float64)
        n_pairs += len(want)
        if not (np.array_equal(got[:, 1:], want[:, 1:]) and np.allclose(got[:, 0], want[:, 0], rtol=0, atol=1e-12, equal_nan=True)):
            print("DISAGREEMENT in case", i, spec)
            print(np.concatenate([got, want], axis=1))
            return 1
    print("REFINE CHECK OK", args.cases, "grids,", n_pairs, "pairs")
    return 0


if __name__ == "__main__":
    sys.exit(main())
