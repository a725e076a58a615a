#!/usr/bin/env python3
"""Raw subpixel shifts (mistitch_phase_cross_correlation) against the NumPy restatement over random strip pairs:
how many agree exactly on the 1 / upsample grid, and the largest difference.  usage: subpixel_exact_probe.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from image_stitcher_amd import registration as reg
from oracle import subpixel as osp
from test_gpu_registration import _subpixel_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(5)
exact = 0
worst = 0.0
for it in range(n):
    r = rng.random()
    if r < 0.4:
        h, w = int(2 ** rng.integers(5, 9)), int(2 ** rng.integers(6, 11))
    elif r < 0.6:
        h, w = int(rng.integers(6, 40)), int(rng.integers(24, 120))  # (very small strips are refused by the stage)
    else:
        h, w = int(rng.integers(30, 260)), int(rng.integers(40, 600))
    if rng.random() < 0.3:
        h, w = w, h
    dy, dx = rng.uniform(-3, 3), rng.uniform(-6, 6)
    uf = int(rng.choice([100, 100, 100, 10, 37, 64]))
    a, b = _subpixel_pair(1000 + it, h, w, dy, dx)
    want = osp.phase_cross_correlation(a, b, uf)
    try:
        got = reg.phase_cross_correlation(a, b, uf)
    except Exception as exc:  # a refused shape is not a disagreement
        print("refused", (h, w), exc)
        continue
    d = float(np.abs(got - want).max())
    exact += d < 1e-9
    if d > worst:
        worst = d
        print("worst so far", (h, w), uf, got, want, flush=True)
print(f"{exact} of {n} identical on the grid, largest difference {worst:.4g} px")
