#!/usr/bin/env python3
"""Golden vectors for the consensus step of the reference's "Method 1", produced by RUNNING THE REFERENCE in
the development container.

``ransac_translation_estimation`` (_translation_computation.py:997-1140) hands ``skimage.measure.ransac`` a model
class whose ``estimate(self, data, *args)`` does ``src, dst = data`` (:1053-1055); ransac calls
``estimate(*samples)`` with a sample of ``min_samples = 3`` rows, so that unpacking raises ValueError and the
reference's own ``except Exception as ransac_error`` branch (:1085-1107) takes over: pairwise-distance clustering
with scipy's ``pdist`` (scipy is installed here).  scikit-image is NOT installed here, so this script sets
``RANSAC_AVAILABLE = True`` and binds the module's ``ransac`` name to a callable that raises that ValueError: the
vectors below are then computed by the reference's own clustering branch, its fallbacks (:1030-1038, :1110-1118) and
its result list (interpret_translation_ransac, :1195-1313; Method 2 stays off: SKIMAGE_AVAILABLE is false).
What is pinned: that branch.  What is not: whether a given scikit-image release reaches it.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_consensus.py

Writes tests/golden/consensus_golden.json: seeds, shapes and crafted tables (inputs are regenerated from them by
``image_stitcher_amd.synth``) and expected outputs only.
"""
import json
import os
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("tifffile", types.ModuleType("tifffile"))  # imported at module top, used for file IO only

import numpy as np  # noqa: E402

from image_stitcher_amd import synth  # noqa: E402
import image_stitcher.registration.tile_registration as tr  # noqa: E402
import image_stitcher.registration._translation_computation as tc  # noqa: E402

tr.set_tensor_backend("numpy")


def _ransac_raises(*args, **kwargs):
    raise ValueError("too many values to unpack (expected 2)")


assert not tc.RANSAC_AVAILABLE, "scikit-image is installed: use its own ransac instead of the stand-in"
tc.RANSAC_AVAILABLE = True
tc.ransac = _ransac_raises
CFG = tc.RANSACConfig(use_ransac=True, residual_threshold=2.0, min_inlier_ratio=0.3)  # tile_registration.py:573


def strips(seed, shape, shift, lowpass):
    h, w = shape
    dy, dx = shift
    big = synth.make_scene(seed, h + 64, w + 64)
    if lowpass:
        f = np.fft.fft2(big.astype(np.float64))
        k = np.fft.fftfreq(h + 64)[:, None] ** 2 + np.fft.fftfreq(w + 64)[None, :] ** 2
        low = np.fft.ifft2(f * np.exp(-k * float(lowpass))).real
        big = (1000 + 50000 * (low - low.min()) / (low.max() - low.min())).astype(np.uint16)
    return (np.ascontiguousarray(big[32:32 + h, 32:32 + w]),
            np.ascontiguousarray(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]))


def table_case(name, translations, scores):
    t, s = np.array(translations, dtype=np.float64).reshape(-1, 2), np.array(scores, dtype=np.float64)
    best, inl, score = tc.ransac_translation_estimation(t, s, CFG)
    return dict(name=name, translations=t.tolist(), scores=s.tolist(),
                best=None if best is None else [float(best[0]), float(best[1])],
                inliers=[bool(v) for v in inl], score=float(score))


def strips_case(seed, shape, shift, lowpass):
    h, w = shape
    a, b = strips(seed, shape, shift, lowpass)
    yi, xi, _ = tc.multi_peak_max(tc.pcm(a, b))
    t, s = tc.generate_translation_candidates(a, b, yi, xi, -h, h, -w, w, 10)
    if len(t) == 0:
        return None
    best, inl, score = tc.ransac_translation_estimation(t, s, CFG)
    final = tc.interpret_translation_ransac(a, b, yi, xi, -h, h, -w, w, 10, CFG, tc.SubpixelConfig(upsample_factor=100),
                                            True, tc.OptimizationConfig(method="trf", max_nfev=50, verbose=0))
    disc = tc.interpret_translation(a, b, yi, xi, -h, h, -w, w, 2)
    bi = int(np.argmax(s))
    # the fallback's mask marks one candidate; a standing consensus with a single inlier needs 1 / n >= 0.3, n = 3
    stood = len(t) >= 3 and not (int(inl.sum()) == 1 and len(t) > 3)
    return dict(seed=seed, shape=[h, w], shift=list(shift), lowpass=lowpass, n=len(t), inliers=int(inl.sum()),
                method1=[float(best[0]), float(best[1])], consensus_score=float(score),
                best_ncc=[float(s[bi]), float(t[bi][0]), float(t[bi][1])],
                discrete=[float(disc[0]), int(disc[1]), int(disc[2])],
                final=[float(final[0]), float(final[1]), float(final[2])], stood=stood)


def grid_case(name, seed, rows, cols, th, tw, edge, overlap, jitter):
    """Rows of the reference's top-level compute_single_translation (tile_registration.py:527-586)."""
    g = synth.make_grid(seed, rows, cols, th, tw, overlap=overlap, jitter=jitter)
    out = []
    for (i1, i2), d in zip(g.pair_idx, g.pair_dir):
        r = tr.compute_single_translation(g.tiles[i1], g.tiles[i2], "top" if d else "left", th, tw, edge)
        out.append(None if r is None else [float(r[0]), float(r[1]), float(r[2])])
    return dict(name=name, seed=seed, rows=rows, cols=cols, tile_h=th, tile_w=tw, edge_width=edge, overlap=overlap,
                jitter=jitter, pairs=out)


def main():
    warnings.filterwarnings("error", message="RANSAC")  # the clustering branch must run clean: a warning would be the outer fallback (:1129)
    tables = [
        table_case("cluster_beats_best_ncc", [[0, 0], [1, 0], [0, 1], [1, 1], [30, 30], [-40, 7]], [.2, .3, .25, .1, .9, .5]),
        table_case("ratio_below_0.3", [[0, 0], [1, 0], [10, 10], [20, 20], [30, 30], [40, 40], [50, 50], [60, 60]],
                   [.2, .3, .1, .1, .95, .5, .1, .2]),
        table_case("ratio_exactly_0.3", [[0, 0], [1, 1], [2, 0], [10, 10], [20, 20], [30, 30], [40, 40], [50, 50], [60, 60], [70, 70]],
                   [.2, .3, .1, .1, .95, .5, .1, .2, .3, .4]),
        table_case("two_candidates", [[3, 4], [3, 5]], [.4, .6]),
        table_case("one_candidate", [[-3, 4]], [.4]),
        table_case("three_far_apart", [[0, 0], [10, 0], [0, 10]], [.1, .7, .3]),     # 1/3 >= 0.3: the first stands
        table_case("first_of_equal_counts", [[5, 5], [6, 5], [50, 5], [51, 5]], [.1, .2, .9, .8]),
        table_case("distance_two_counts_sqrt5_does_not", [[0, 0], [2, 0], [0, 2], [2, 1], [1, 2], [-2, 0]], [.5, .4, .3, .2, .1, .6]),
        table_case("duplicates_count", [[7, -7], [7, -7], [7, -7], [100, 0], [0, 100], [-100, 0], [0, -100], [60, 60], [-60, 60]],
                   [.3, .3, .3, .9, .2, .2, .2, .1, .1]),
        table_case("equal_best_ncc_first_wins", [[1, 0], [40, 0], [80, 0], [120, 0], [160, 0], [200, 0], [240, 0]],
                   [.5, .8, .8, .1, .1, .1, .1]),
    ]
    # seeded strips: a few of every outcome (consensus stands and wins / stands and loses to the discrete search /
    # declined), found by walking seeds in a fixed order
    rng = np.random.default_rng(20240)
    want = {"stands_wins": 6, "stands_loses": 6, "declined": 6}
    picked = []
    for trial in range(4000):
        if not any(want.values()):
            break
        h = int(rng.choice([8, 12, 16, 24, 32, 48, 64]))
        w = int(rng.choice([8, 12, 16, 24, 32, 48, 64, 128]))
        shift = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
        lowpass = int(rng.choice([0, 200, 2000, 8000]))
        if shift == (0, 0):
            continue   # identical strips: the correlation is a delta plus rounding noise, whose ranks 2..10 no two FFTs share
        c = strips_case(7000 + trial, (h, w), shift, lowpass)
        if c is None:
            continue
        stood = c["stood"]
        kind = "declined" if not stood else ("stands_wins" if c["final"][1:] == c["method1"] and c["method1"] != c["best_ncc"][1:]
                                             else "stands_loses" if c["final"][1:] != c["method1"] else None)
        if kind and want[kind] > 0:
            want[kind] -= 1
            c["kind"] = kind
            picked.append(c)
    assert not any(want.values()), want
    doc = dict(
        tables=tables, strips=picked,
        grids=[grid_case("g3x3_64_e16", 71, 3, 3, 64, 64, 16, 0.2, 2),
               grid_case("g2x4_96x80_e12", 72, 2, 4, 96, 80, 12, 0.15, 1),
               grid_case("g3x3_256_e32", 73, 3, 3, 256, 256, 32, 0.125, 3)],
    )
    path = os.path.join(HERE, "consensus_golden.json")
    with open(path, "w") as fh:
        json.dump(doc, fh)
        fh.write("\n")
    kinds = {}
    for c in picked:
        kinds[c["kind"]] = kinds.get(c["kind"], 0) + 1
    print("wrote", path, os.path.getsize(path), "bytes;", kinds, [t["best"] for t in tables])


if __name__ == "__main__":
    main()
