#!/usr/bin/env python3
"""Golden vectors for the deterministic part of the reference's "Method 1" (the candidate set of its
consensus stage), produced by RUNNING THE REFERENCE in the development container:
``generate_translation_candidates(n=10)`` (_translation_computation.py:876-994) on seeded strip
pairs, and the best-NCC fallback of ``ransac_translation_estimation`` (:1019-1038; scikit-image is
absent here, so RANSAC_AVAILABLE is false and that is the path the reference itself takes).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_m1.py

Writes tests/golden/m1_golden.json: seeds and shapes (inputs are regenerated from them by
``image_stitcher_amd.synth``) and expected outputs only.
"""
import json
import os
import sys
import types

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


def strips_case(name, seed, shape, shift, smooth=False):
    h, w = shape
    dy, dx = shift
    big = synth.make_scene(seed, h + 64, w + 64)
    if smooth:   # a low-pass scene ranks many near-equal peaks: candidates beyond the first two matter
        f = np.fft.fft2(big.astype(np.float64))
        k = np.fft.fftfreq(h + 64)[:, None] ** 2 + np.fft.fftfreq(w + 64)[None, :] ** 2
        low = np.fft.ifft2(f * np.exp(-k * 2000.0)).real
        big = (1000 + 50000 * (low - low.min()) / (low.max() - low.min())).astype(np.uint16)
    a = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
    b = np.ascontiguousarray(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w])
    p = tc.pcm(a, b)
    yi, xi, _ = tc.multi_peak_max(p)
    t, s = tc.generate_translation_candidates(a, b, yi, xi, -h, h, -w, w, 10)
    best, _, score = tc.ransac_translation_estimation(t, s, tc.RANSACConfig())
    d = tc.interpret_translation(a, b, yi, xi, -h, h, -w, w, 2)
    return dict(name=name, seed=seed, shape=[h, w], shift=[dy, dx], smooth=smooth,
                peaks=[[int(yi[i]), int(xi[i])] for i in range(10)],
                translations=t.tolist(), scores=[float(v) for v in s],
                fallback=[float(score), float(best[0]), float(best[1])],
                discrete=[float(d[0]), int(d[1]), int(d[2])])


def grid_case(name, seed, rows, cols, th, tw, edge, overlap, jitter):
    g = synth.make_grid(seed, rows, cols, th, tw, overlap=overlap, jitter=jitter)
    rows_out = []
    for (i1, i2), d in zip(g.pair_idx, g.pair_dir):
        a, b = g.tiles[i1], g.tiles[i2]
        if d == 0:
            w = min(edge, tw)
            e1, e2, off = a[:, tw - w:], b[:, :w], (0, tw - w)
        else:
            w = min(edge, th)
            e1, e2, off = a[th - w:, :], b[:w, :], (th - w, 0)
        if tr.is_edge_featureless(e1) or tr.is_edge_featureless(e2):
            rows_out.append(None)
            continue
        eh, ew = e1.shape
        yi, xi, _ = tc.multi_peak_max(tc.pcm(e1, e2))
        t, s = tc.generate_translation_candidates(e1, e2, yi, xi, -eh, eh, -ew, ew, 10)
        best, _, score = tc.ransac_translation_estimation(t, s, tc.RANSACConfig())
        rows_out.append(dict(n=len(s), fallback=[float(score), float(best[0]) + off[0], float(best[1]) + off[1]],
                             score_sum=float(np.sum(s)), first=[t[0].tolist(), float(s[0])], last=[t[-1].tolist(), float(s[-1])]))
    return dict(name=name, seed=seed, rows=rows, cols=cols, tile_h=th, tile_w=tw, edge_width=edge, overlap=overlap,
                jitter=jitter, pairs=rows_out)


def main():
    assert not tc.RANSAC_AVAILABLE, "this generator pins the path the reference takes WITHOUT scikit-image"
    doc = dict(
        strips=[strips_case("tall", 61, (256, 64), (3, -5)), strips_case("wide", 62, (64, 256), (-2, 7)),
                strips_case("smooth_tall", 63, (256, 64), (2, 9), smooth=True),
                strips_case("big_shift", 64, (128, 128), (25, -21))],
        grids=[grid_case("g3x3_256", 65, 3, 3, 256, 256, 32, 0.125, 3),
               grid_case("g2x3_300x420_e45", 66, 2, 3, 300, 420, 45, 0.15, 2)],
    )
    path = os.path.join(HERE, "m1_golden.json")
    with open(path, "w") as fh:
        json.dump(doc, fh)
        fh.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
