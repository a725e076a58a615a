"""8-bit tiles (MONO8 cameras).  The reference takes the dtype of the first image (parameters.py:685) and clips
flatfield-corrected pixels to that type's range (tile_registration.py:2177-2180, stitcher.py:260-267); here 8-bit tiles
are widened to 16 bits on the device and the kernels clip to 255 (mistitch_set_pixel_max).  Vectors in
tests/golden/u8_golden.json come from the reference itself (tests/golden/make_golden_u8.py)."""
import json
import os

import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import fusion as ofu
from oracle import registration as oreg

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "u8_golden.json")))


def _ramp_u8(h, w):
    ramp = (np.arange(h * w, dtype=np.uint32).reshape(h, w) * 7 % 256).astype(np.uint8)
    ramp[0, :8] = [0, 1, 255, 254, 200, 128, 127, 250]
    return ramp


def _to_u8(tiles16):
    return np.clip((tiles16.astype(np.int64) - 1000) * 255 // 30000, 0, 255).astype(np.uint8)


def test_flatfield_of_8_bit_tiles_matches_the_reference():
    from image_stitcher_amd import fusion as fu
    f = GOLD["flatfield"]
    h, w = f["shape"]
    ramp = _ramp_u8(h, w)
    for dt, exp in f["results"].items():
        ff = synth.make_flatfield(h, w, dtype=np.dtype(dt), amplitude=f["amplitude"])
        got = fu.apply_flatfield_to_image(ramp, ff)
        assert got.dtype == np.uint8 and np.array_equal(got, ofu.apply_flatfield(ramp, ff))
        assert int(got.astype(np.int64).sum()) == exp["sum"] and int((got == 255).sum()) == exp["n_clipped"]
        assert got[0].tolist() == exp["first_row"] and got[-1].tolist() == exp["last_row"]
    # a 16-bit image clipped to the 8-bit range on request (the reference's dtype argument), odd size: scalar tail
    odd = (np.arange(13 * 11, dtype=np.uint16).reshape(13, 11) * 37) % 600
    ffo = synth.make_flatfield(13, 11, amplitude=0.3)
    want = (odd / ffo).clip(0, 255).astype(np.uint8)
    assert np.array_equal(fu.apply_flatfield_to_image(odd, ffo, dtype=np.uint8), want)
    # and the context is back at 16 bits afterwards
    big = np.full((16, 16), 60000, dtype=np.uint16)
    assert fu.apply_flatfield_to_image(big, np.full((16, 16), 0.5, dtype=np.float32)).max() == 65535


def test_registration_of_8_bit_tiles_matches_the_reference():
    from image_stitcher_amd import registration as reg
    spec = GOLD["grid"]
    g = synth.make_grid(spec["seed"], spec["rows"], spec["cols"], spec["tile_h"], spec["tile_w"], overlap=spec["overlap"],
                        jitter=spec["jitter"])
    tiles = _to_u8(g.tiles)
    ff = synth.make_flatfield(spec["tile_h"], spec["tile_w"], amplitude=spec["flatfield_amplitude"])
    for name, src, flat in (("raw", tiles, None), ("corrected", tiles, ff), ("dim", (g.tiles >> 8).astype(np.uint8), None)):
        out, st = reg.register_pairs(src, g.pair_idx, g.pair_dir, spec["edge_width"], flatfield=flat)
        out, st = out.cpu().numpy(), st.cpu().numpy()
        for row, status, exp in zip(out, st, spec["results"][name]):
            if exp is None:
                assert status == 1 and np.isnan(row).all()
            else:
                assert status == 0 and row[1:].tolist() == exp[1:] and abs(row[0] - exp[0]) < 1e-12
    # the oracle agrees on the corrected tiles (clipped to 255, not 65535)
    corrected = np.stack([ofu.apply_flatfield(t, ff) for t in tiles])
    assert corrected.dtype == np.uint8 and corrected.max() == 255
    exp, exp_st = oreg.register_pairs(corrected, g.pair_idx, g.pair_dir, spec["edge_width"], polish=False)
    out, st = reg.register_pairs(tiles, g.pair_idx, g.pair_dir, spec["edge_width"], flatfield=ff)
    assert np.array_equal(st.cpu().numpy(), exp_st) and np.array_equal(out.cpu().numpy()[:, 1:], exp[:, 1:])


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
def test_fusion_of_8_bit_tiles(mode):
    """Mosaic of 8-bit tiles with a flatfield that drives many pixels past 255: uint8 out, equal to the oracle."""
    import torch
    from image_stitcher_amd import fusion as fu
    from tests.fusion_cases import overlapping_scene
    tiles16, yx, plane, mh, mw = overlapping_scene(33, 3, 3, 72, 136, n_planes=2, jitter=3, overlap=0.15)
    tiles = (tiles16 >> 8).astype(np.uint8)
    ffs = np.stack([synth.make_flatfield(72, 136, amplitude=0.6), synth.make_flatfield(72, 136, amplitude=0.2)])
    oracle = ofu.fuse_overwrite if mode == "overwrite" else ofu.fuse_feather
    want = oracle(tiles, yx, plane, 2, mh, mw, flatfields=ffs, tile_ff=plane)
    got = fu.fuse_tiles(tiles, yx, plane, 2, mh, mw, flatfields=ffs, tile_ff=plane, mode=mode)
    assert got.dtype == torch.uint8
    got = got.cpu().numpy()
    assert want.dtype == np.uint8 and (want == 255).sum() > 100 and np.array_equal(got, want)
    born = fu.fuse_tiles(tiles, yx, plane, 2, mh, mw, flatfields=ffs, tile_ff=plane, mode=mode, chunks=(64, 64))
    assert born.dtype == torch.uint8 and np.array_equal(born.cpu().numpy()[:, 0, 0], want[:, :64, :64])
