#!/usr/bin/env python3
"""Development-container check (needs /root/reference; no test imports this): random tile pairs through
the REFERENCE's compute_single_translation (NumPy backend, scikit-image absent) and through the
oracle's restatement, which is what the device is compared with everywhere else.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_oracle_against_reference.py --cases 300
"""
import argparse
import contextlib
import io
import logging
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.modules.setdefault("tifffile", types.ModuleType("tifffile"))
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")

import image_stitcher.registration.tile_registration as tr  # noqa: E402

from image_stitcher_amd import synth  # noqa: E402
from oracle import registration as oreg  # noqa: E402

tr.set_tensor_backend("numpy")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    logging.disable(logging.CRITICAL)
    rng = np.random.default_rng(args.seed)
    n_pairs = n_none = 0
    for i in range(args.cases):
        rows, cols = (1, 2) if rng.random() < 0.5 else (2, 1)
        th, tw = ((int(2 ** rng.integers(5, 9)) for _ in range(2)) if rng.random() < 0.5
                  else (int(rng.integers(24, 300)), int(rng.integers(24, 300))))
        th, tw = int(th), int(tw)
        edge = int(rng.choice([1, 8, 16, 32, 64, int(rng.integers(3, 200))]))
        g = synth.make_grid(int(rng.integers(1 << 30)), rows, cols, th, tw, overlap=float(rng.uniform(0.05, 0.6)),
                            jitter=int(rng.integers(0, 12)))
        if rng.random() < 0.2:
            g.tiles[int(rng.integers(2))] = (int(rng.integers(0, 60000)) + float(rng.uniform(100, 200)) * rng.random((th, tw))).astype(np.uint16)
        for (a, b), d in zip(g.pair_idx, g.pair_dir):
            direction = "top" if d else "left"
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                want = tr.compute_single_translation(g.tiles[a], g.tiles[b], direction, th, tw, edge)
            got = oreg.compute_single_translation(g.tiles[a], g.tiles[b], direction, edge)
            n_pairs += 1
            n_none += want is None
            same = (want is None) == (got is None) and (want is None or (
                float(want[1]) == float(got[1]) and float(want[2]) == float(got[2]) and abs(float(want[0]) - float(got[0])) < 1e-12))
            if not same:
                print("DISAGREEMENT in case", i, dict(tile=(th, tw), edge=edge, direction=direction), "reference", want, "oracle", got)
                return 1
    print("ORACLE CHECK OK", n_pairs, "pairs,", n_none, "of them rejected by both")
    return 0


if __name__ == "__main__":
    sys.exit(main())
