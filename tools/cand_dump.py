#!/usr/bin/env python3
"""Candidate tables of random small grids, for comparing two builds of the library (MISTITCH_LIB selects one):
    python tools/cand_dump.py dump out.npz [n_cases]      python tools/cand_dump.py cmp a.npz b.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def case(seed):
    from image_stitcher_amd import synth
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    if rows * cols == 1:
        cols = 2
    th, tw = (int(rng.choice([16, 24, 32, 64, 128, 8 * int(rng.integers(2, 40))])) for _ in range(2))
    edge = int(rng.choice([4, 6, 8, 12, 16, 32, int(rng.integers(4, 40))]))
    g = synth.make_grid(int(rng.integers(1 << 30)), rows, cols, th, tw, overlap=float(rng.uniform(0.1, 0.3)),
                        jitter=int(rng.integers(1, 4)), lowpass=float(rng.choice([0, 200, 2000, 8000])))
    return g, edge, dict(seed=seed, grid=(rows, cols), tile=(th, tw), edge=edge)


def main():
    if sys.argv[1] == "dump":
        from image_stitcher_amd import registration as reg
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
        out = {}
        for seed in range(n):
            g, edge, _ = case(seed)
            for wide in (False, True):
                o, st, table = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, edge, peaks10=wide, return_candidates=True)
                out[f"t{seed}_{int(wide)}"] = table.cpu().numpy()
                out[f"o{seed}_{int(wide)}"] = o.cpu().numpy()
        np.savez_compressed(sys.argv[2], **out)
        print("dumped", n)
    else:
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = 0
        for k in a.files:
            if not np.array_equal(a[k], b[k], equal_nan=True):
                bad += 1
                if bad <= 5:
                    seed = int(k[1:].split("_")[0])
                    print("DIFFERS", k, case(seed)[2])
                    if k[0] == "t":
                        w = np.argwhere(~((a[k] == b[k]) | (np.isnan(a[k]) & np.isnan(b[k]))))
                        for pair, ci, _ in w[:6]:
                            print("  pair", pair, "cand", ci, a[k][pair, ci], b[k][pair, ci])
        print("arrays", len(a.files), "differing", bad)


if __name__ == "__main__":
    main()
