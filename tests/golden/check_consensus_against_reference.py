#!/usr/bin/env python3
"""Random strips through the reference's interpret_translation_ransac with its consensus stage live (the stand-in for
scikit-image's ransac that raises, as tests/golden/make_golden_consensus.py explains) and through the oracle's
restatement: run in the development container only (it imports the reference), not part of the test suite.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_consensus_against_reference.py [n_cases]
"""
import os
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("tifffile", types.ModuleType("tifffile"))

import numpy as np  # noqa: E402

import image_stitcher.registration.tile_registration as tr  # noqa: E402
import image_stitcher.registration._translation_computation as tc  # noqa: E402
from make_golden_consensus import CFG, strips  # noqa: E402  (patches tc.ransac / RANSAC_AVAILABLE on import)
from oracle import registration as oreg  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    warnings.filterwarnings("error", message="RANSAC")  # the clustering branch must not fall through to :1129
    rng = np.random.default_rng(99)
    stood = wins = 0
    for trial in range(n_cases):
        h = int(rng.choice([4, 6, 8, 12, 16, 24, 32, 48, 64, 100]))
        w = int(rng.choice([4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 200]))
        shift = (int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
        a, b = strips(50000 + trial, (h, w), shift, int(rng.choice([0, 0, 200, 2000, 8000, 30000])))
        yi, xi, _ = tc.multi_peak_max(tc.pcm(a, b))
        want = tc.interpret_translation_ransac(a, b, yi, xi, -h, h, -w, w, 10, CFG, tc.SubpixelConfig(upsample_factor=100),
                                               True, tc.OptimizationConfig(method="trf", max_nfev=50, verbose=0))
        got = oreg.compute_translation_consensus(a, b, yi, xi, -h, h, -w, w, 10)
        assert (got[1], got[2]) == (want[1], want[2]) and (got[0] == want[0] or abs(got[0] - want[0]) < 1e-12), (trial, got, want)
        m1 = oreg.consensus_translation(*oreg.generate_translation_candidates(a, b, yi, xi, -h, h, -w, w, 10))
        if m1 is not None and m1[3]:
            stood += 1
            wins += (m1[1], m1[2]) == (want[1], want[2])
    print(f"{n_cases} strip pairs: reference == oracle throughout; the consensus stood in {stood}, was the final result in {wins}")


if __name__ == "__main__":
    main()
