#!/usr/bin/env python3
"""Golden vectors for 8-bit tiles (MONO8 cameras), produced by running the REFERENCE in the development
container (see make_golden.py for how it is imported):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_u8.py

The reference takes the dtype of the first image (parameters.py:685) and clips flatfield-corrected pixels to that
type's range (tile_registration.py:2177-2180).  Stored: seeds, shapes and expected outputs only.
"""
import json
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

tr.set_tensor_backend("numpy")


def ramp_u8(h, w):
    ramp = (np.arange(h * w, dtype=np.uint32).reshape(h, w) * 7 % 256).astype(np.uint8)
    ramp[0, :8] = [0, 1, 255, 254, 200, 128, 127, 250]
    return ramp


def flatfield_case():
    h, w = 48, 72
    ramp = ramp_u8(h, w)
    out = {}
    for dt in ("float32", "float64"):
        ff = synth.make_flatfield(h, w, dtype=np.dtype(dt), amplitude=0.6)
        got = tr.apply_flatfield_to_image(ramp, ff)
        assert got.dtype == np.uint8
        out[dt] = dict(sum=int(got.astype(np.int64).sum()), first_row=got[0].tolist(), last_row=got[-1].tolist(),
                       n_clipped=int((got == 255).sum()))
    return dict(shape=[h, w], amplitude=0.6, results=out)


def to_u8(tiles16):
    """The synthetic 16-bit scene (1000 .. 31000) stretched over the 8-bit range."""
    return np.clip((tiles16.astype(np.int64) - 1000) * 255 // 30000, 0, 255).astype(np.uint8)


def grid_case():
    """Pairs of an 8-bit grid (the 16-bit synthetic scene shifted down by 8 bits), raw and flatfield corrected as
    load_batch_images corrects whole tiles before the edges are cut (tile_registration.py:635-637)."""
    spec = dict(seed=41, rows=2, cols=3, tile_h=256, tile_w=384, overlap=0.12, jitter=3, edge_width=32)
    g = synth.make_grid(spec["seed"], spec["rows"], spec["cols"], spec["tile_h"], spec["tile_w"],
                        overlap=spec["overlap"], jitter=spec["jitter"])
    tiles = to_u8(g.tiles)
    dim = (g.tiles >> 8).astype(np.uint8)      # values 3 .. 121: every strip falls below the variance threshold
    ff = synth.make_flatfield(spec["tile_h"], spec["tile_w"], amplitude=0.5)
    res = {}
    for name, src in (("raw", tiles), ("corrected", np.stack([tr.apply_flatfield_to_image(t, ff) for t in tiles])),
                      ("dim", dim)):
        rows = []
        for (i1, i2), d in zip(g.pair_idx, g.pair_dir):
            r = tr.compute_single_translation(src[i1], src[i2], "top" if d else "left", spec["tile_h"], spec["tile_w"],
                                              spec["edge_width"])
            rows.append(None if r is None else [float(r[0]), float(r[1]), float(r[2])])
        res[name] = rows
    return dict(spec, flatfield_amplitude=0.5, results=res)


def main():
    gold = dict(meta=dict(numpy=np.__version__, note="reference NumPy backend, uint8 tiles"),
                flatfield=flatfield_case(), grid=grid_case())
    path = os.path.join(HERE, "u8_golden.json")
    with open(path, "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
