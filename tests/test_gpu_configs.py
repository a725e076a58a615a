"""The benchmark configurations at FULL size on the device (needs an MI355X).

  c2  10 x 10 grid of 2048^2 tiles, one channel, registration + linear (feather) blend
  c3  20 x 20 grid, 4 channels + flatfield: 760 pairs through the fused flatfield call
  c4  96-well plate, 5 x 5 fields of view per well: 3840 pairs in one call, every well one plane of one
      fuse call, wells sharded over 8 ranks (no halo, no exchange), one well written as OME-Zarr
  c5  50 x 50 grid, 5 z levels of one channel: 4900 pairs on the MIDDLE level, and the
      92 k x 92 k mosaic through ``stitch_region(selected_z=...)``

The NumPy oracle cannot run grids of this size inside a test, so each case checks (a) what
must hold for any correct implementation: every planted shift recovered, the host solve
returning the planted positions, a mosaic of tiles cut from one scene reproducing that
scene in every covered pixel in both modes; and (b) windows of the full-size mosaic
against the oracle run on just the tiles that touch each window, bit for bit (a fused
pixel depends only on the tiles that cover it, so a cropped oracle run is exact).
"""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import fusion as ofu

pytestmark = pytest.mark.gpu

TILE, EDGE = 2048, 256


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda", 0)


def _covered_mask(yx, mh, mw, device):
    import torch
    covered = torch.zeros((mh, mw), dtype=torch.bool, device=device)
    for y, x in yx:
        covered[y:y + TILE, x:x + TILE] = True
    return covered


def _solve(rows, cols, pair_idx, pair_dir, out, status):
    from image_stitcher_amd import solve
    row, col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
    res = solve.solve_positions(solve.grid_from_pairs(row, col, pair_idx, pair_dir, out, status), TILE, TILE)
    return np.stack([res.y_pos, res.x_pos], axis=1).astype(np.int32)


def _oracle_window(tiles_dev, yx, plane_of_tile, plane, window, mode, flatfields=None, tile_ff=None):
    """Oracle mosaic of ``window`` = (y0, y1, x0, x1) of one plane from the tiles that touch it.
    ``tiles_dev`` [T, H, W] in placement order; only the touching tiles are downloaded."""
    y0, y1, x0, x1 = window
    touch = [t for t in range(len(yx)) if plane_of_tile[t] == plane and yx[t][0] < y1 and yx[t][0] + TILE > y0
             and yx[t][1] < x1 and yx[t][1] + TILE > x0]
    if not touch:
        return np.zeros((y1 - y0, x1 - x0), dtype=np.uint16)
    oy, ox = min(yx[t][0] for t in touch), min(yx[t][1] for t in touch)
    ch = max(yx[t][0] for t in touch) + TILE - oy
    cw = max(yx[t][1] for t in touch) + TILE - ox
    sub = np.stack([tiles_dev[t].cpu().numpy() for t in touch])
    sub_yx = np.array([[yx[t][0] - oy, yx[t][1] - ox] for t in touch], dtype=np.int64)
    fuse = ofu.fuse_feather if mode == "feather" else ofu.fuse_overwrite
    ff_idx = None if tile_ff is None else [int(tile_ff[t]) for t in touch]
    canvas = fuse(sub, sub_yx, [0] * len(touch), 1, ch, cw, flatfields=flatfields, tile_ff=ff_idx)[0]
    out = np.zeros((y1 - y0, x1 - x0), dtype=np.uint16)
    # the part of the window that the canvas covers (the rest has no tile: zeros)
    cy0, cy1, cx0, cx1 = max(y0, oy), min(y1, oy + ch), max(x0, ox), min(x1, ox + cw)
    out[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0] = canvas[cy0 - oy:cy1 - oy, cx0 - ox:cx1 - ox]
    return out


def _windows(yx_grid, rows, cols, mh, mw, rng, n_random=3):
    """Windows that matter: a four-tile corner, a horizontal and a vertical overlap band, the
    mosaic's far corner (groups cut by the mosaic edge) and a few random places."""
    def at(r, c):
        return yx_grid[r * cols + c]
    r, c = rows // 2, cols // 2
    y, x = at(r, c)
    wins = [(y - 40, y + 300, x - 40, x + 300),                     # four tiles meet
            (y + 900, y + 1100, x - 30, x + 260),                   # left-right overlap
            (y - 30, y + 260, x + 700, x + 1300),                   # top-bottom overlap
            (mh - 150, mh, mw - 333, mw), (0, 120, 0, 517)]         # mosaic corners
    for _ in range(n_random):
        wy, wx = int(rng.integers(0, mh - 400)), int(rng.integers(0, mw - 700))
        wins.append((wy, wy + 400, wx, wx + 700))
    return [(max(a, 0), min(b, mh), max(c_, 0), min(d, mw)) for a, b, c_, d in wins]


def test_c2_register_solve_and_blend(dev):
    """Config 2: 180 planted shifts, solve == planted, feather and overwrite == scene."""
    import torch
    from image_stitcher_amd import fusion, registration
    rows = cols = 10
    tiles, true_yx, pair_idx, pair_dir, scenes = synth.make_device_acquisition(rows, cols, TILE, 1, 2, dev, keep_scenes=True)
    out, status = registration.register_pairs(tiles[0], pair_idx, pair_dir, EDGE)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert len(pair_idx) == 180 and (status == 0).all()
    assert np.array_equal(out[:, 1:].astype(np.int64), true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]])
    yx = _solve(rows, cols, pair_idx, pair_dir, out, status)
    origin = true_yx.min(axis=0)
    assert np.array_equal(yx, true_yx - origin)
    mh, mw = int(yx[:, 0].max()) + TILE, int(yx[:, 1].max()) + TILE
    want = scenes[0][origin[0]:origin[0] + mh, origin[1]:origin[1] + mw].view(torch.int16)
    covered = _covered_mask(yx, mh, mw, dev)
    plane = np.zeros(rows * cols, dtype=np.int32)
    rng = np.random.default_rng(2)
    for mode in ("feather", "overwrite"):
        mosaic = fusion.fuse_tiles(tiles[0], yx, plane, 1, mh, mw, mode=mode)
        m = mosaic[0].view(torch.int16)
        assert int(((m != want) & covered).sum().item()) == 0, mode
        assert int(((m != 0) & ~covered).sum().item()) == 0, mode
        for win in _windows(yx, rows, cols, mh, mw, rng, n_random=1):
            ref = _oracle_window(tiles[0], yx, plane, 0, win, mode)
            got = mosaic[0, win[0]:win[1], win[2]:win[3]].cpu().numpy()
            assert np.array_equal(got, ref), (mode, win)


def test_c3_four_channels_with_flatfield(dev):
    """Config 3: 760 shifts through the fused flatfield call; all four planes of the mosaic
    against the oracle in windows, both modes; single-coverage pixels against a direct division."""
    import torch
    from image_stitcher_amd import fusion, registration
    rows = cols = 20
    ch = 4
    tiles, true_yx, pair_idx, pair_dir = synth.make_device_acquisition(rows, cols, TILE, ch, 3, dev)
    ffs = np.stack([synth.make_flatfield(TILE, TILE, amplitude=0.2 - 0.03 * c) for c in range(ch)])
    d_ffs = torch.as_tensor(ffs, device=dev)
    out, status = registration.register_pairs(tiles[0], pair_idx, pair_dir, EDGE, flatfield=d_ffs[0])
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert len(pair_idx) == 760 and (status == 0).all()
    assert np.array_equal(out[:, 1:].astype(np.int64), true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]])
    # correcting whole tiles first (the reference's order of operations) gives the same rows
    corrected = fusion.apply_flatfield_to_image(tiles[0], d_ffs[0])
    out_b, status_b = registration.register_pairs(corrected, pair_idx, pair_dir, EDGE)
    assert np.array_equal(out_b.cpu().numpy(), out) and np.array_equal(status_b.cpu().numpy(), status)
    del corrected
    yx = _solve(rows, cols, pair_idx, pair_dir, out, status)
    assert np.array_equal(yx, true_yx - true_yx.min(axis=0))
    mh, mw = int(yx[:, 0].max()) + TILE, int(yx[:, 1].max()) + TILE
    n = rows * cols
    all_yx = np.tile(yx, (ch, 1))
    plane = np.repeat(np.arange(ch), n).astype(np.int32)
    flat_tiles = tiles.view(ch * n, TILE, TILE)
    rng = np.random.default_rng(3)
    for mode in ("feather", "overwrite"):
        mosaic = fusion.fuse_tiles(flat_tiles, all_yx, plane, ch, mh, mw, flatfields=d_ffs, tile_ff=plane, mode=mode)
        for c in range(ch):
            for win in _windows(yx, rows, cols, mh, mw, rng, n_random=1):
                ref = _oracle_window(flat_tiles, all_yx, plane, c, win, mode, flatfields=ffs, tile_ff=plane)
                got = mosaic[c, win[0]:win[1], win[2]:win[3]].cpu().numpy()
                assert np.array_equal(got, ref), (mode, c, win)
        # the interior of one tile per channel that no other tile touches: trunc(clip(v / ff)) by NumPy
        t = 7 * cols + 11
        y, x = yx[t]
        lo, hi = 300, TILE - 300
        for c in range(ch):
            raw = tiles[c, t, lo:hi, lo:hi].cpu().numpy()
            want = ofu.apply_flatfield(raw, ffs[c][lo:hi, lo:hi])
            got = mosaic[c, y + lo:y + hi, x + lo:x + hi].cpu().numpy()
            assert np.array_equal(got, want), (mode, c)
        del mosaic


def test_c4_plate_of_wells_sharded_by_well(dev, tmp_path):
    """Config 4: 96 wells x 5 x 5 fields of view of 2048^2 pixels (one channel).  Wells are independent regions
    (tile_registration.py:2696-2875 loops them; each gets its own mosaic, stitcher.py:791-798): all 3840 pairs
    go through one registration call, every well is solved on its own and fused as one plane of a single fuse
    call, a rank of an 8-way split registers and fuses its 12 wells with the same rows and pixels, and a well
    leaves as an OME-Zarr store that reads back as its mosaic."""
    import torch
    from image_stitcher_amd import dist, fusion, omezarr, registration
    wells, rows, cols = 96, 5, 5
    per_well, pairs_per_well = rows * cols, 2 * rows * cols - rows - cols
    tiles = torch.empty((wells * per_well, TILE, TILE), dtype=torch.int16, device=dev).view(torch.uint16)
    true, scenes_kept = [], {}
    for w in range(wells):
        t, true_yx, pair_idx, pair_dir, scenes = synth.make_device_acquisition(rows, cols, TILE, 1, 400 + w, dev, keep_scenes=True)
        tiles[w * per_well:(w + 1) * per_well] = t[0]
        true.append(true_yx)
        if w in (0, 41, 95):
            scenes_kept[w] = scenes[0]
        del t, scenes
    all_pairs = np.concatenate([pair_idx + w * per_well for w in range(wells)]).astype(np.int32)
    all_dirs = np.tile(pair_dir, wells)
    out, status = registration.register_pairs(tiles, all_pairs, all_dirs, EDGE)
    out_np, st_np = out.cpu().numpy(), status.cpu().numpy()
    assert len(all_pairs) == 3840 and (st_np == 0).all()
    yx_all = np.empty((wells * per_well, 2), dtype=np.int32)
    dims = []
    for w in range(wells):
        sl = slice(w * pairs_per_well, (w + 1) * pairs_per_well)
        assert np.array_equal(out_np[sl, 1:].astype(np.int64), true[w][pair_idx[:, 1]] - true[w][pair_idx[:, 0]]), w
        yx = _solve(rows, cols, pair_idx, pair_dir, out_np[sl], st_np[sl])
        assert np.array_equal(yx, true[w] - true[w].min(axis=0)), w
        yx_all[w * per_well:(w + 1) * per_well] = yx
        dims.append((int(yx[:, 0].max()) + TILE, int(yx[:, 1].max()) + TILE))
    mh, mw = max(d[0] for d in dims), max(d[1] for d in dims)      # planes sized for the largest well
    plane = np.repeat(np.arange(wells), per_well).astype(np.int32)
    mosaic = fusion.fuse_tiles(tiles, yx_all, plane, wells, mh, mw, mode="overwrite")
    for w, scene in scenes_kept.items():
        origin = true[w].min(axis=0)
        h, wd = dims[w]
        want = scene[origin[0]:origin[0] + h, origin[1]:origin[1] + wd].view(torch.int16)
        covered = _covered_mask(yx_all[w * per_well:(w + 1) * per_well], h, wd, dev)
        m = mosaic[w, :h, :wd].view(torch.int16)
        assert int(((m != want) & covered).sum().item()) == 0, w
        assert int(((m != 0) & ~covered).sum().item()) == 0, w
        assert int((mosaic[w, h:] != 0).sum().item()) == 0 and int((mosaic[w, :, wd:] != 0).sum().item()) == 0, w
    # wells over 8 ranks: contiguous runs of whole wells, nothing shared between ranks
    covered_wells = []
    for rank in (0, 5, 7):
        lo, hi = dist.band_bounds(wells, 8, rank)
        covered_wells.append((lo, hi))
        mine = tiles[lo * per_well:hi * per_well]
        pairs = np.concatenate([pair_idx + (w - lo) * per_well for w in range(lo, hi)]).astype(np.int32)
        o, s = registration.register_pairs(mine, pairs, np.tile(pair_dir, hi - lo), EDGE)
        assert torch.equal(o, out[lo * pairs_per_well:hi * pairs_per_well]) and torch.equal(s, status[lo * pairs_per_well:hi * pairs_per_well])
        part = fusion.fuse_tiles(mine, yx_all[lo * per_well:hi * per_well], plane[lo * per_well:hi * per_well] - lo, hi - lo,
                                 mh, mw, mode="overwrite")
        assert torch.equal(part, mosaic[lo:hi])
        del part
    assert covered_wells == [(0, 12), (60, 72), (84, 96)]
    # one well out as OME-Zarr (the reference writes one store per region, stitcher.py:465-650)
    h, wd = dims[41]
    store = str(tmp_path / "D6_stitched.ome.zarr")
    info = omezarr.write_ome_zarr(store, mosaic[41, :h, :wd].contiguous().reshape(1, 1, h, wd), ["Fluorescence_488_nm_Ex"],
                                  [0x1FFF00], px_um=0.752, tile_h=TILE, tile_w=TILE, n_levels=1, name="D6_t0")
    assert info["chunks"] == (1, 1, 1, TILE, TILE) and info["levels"] == [(1, 1, 1, h, wd)]
    back = omezarr.read_level(store, 0)
    assert np.array_equal(back[0, 0, 0], mosaic[41, :h, :wd].cpu().view(torch.int16).numpy().view(np.uint16))


def test_c5_middle_z_whole_slide(dev):
    """Config 5, one channel: 50 x 50 tiles x 5 z levels resident (105 GB); the MIDDLE level is
    registered (4900 pairs), solved, and fused through ``stitch_region(selected_z=[2])`` into a
    92 k x 92 k mosaic that must equal the middle level's scene (other levels hold other scenes)."""
    import torch
    from image_stitcher_amd import fusion, registration
    rows = cols = 50
    n, nz, mid = rows * cols, 5, 2
    tiles, true_yx, pair_idx, pair_dir, scenes = synth.make_device_acquisition(
        rows, cols, TILE, nz, 5, dev, keep_scenes=[mid], scene_seeds=[500 + z for z in range(nz)])
    keep = scenes[0]
    del scenes
    out, status = registration.register_pairs(tiles[mid], pair_idx, pair_dir, EDGE)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert len(pair_idx) == 4900 and (status == 0).all()
    assert np.array_equal(out[:, 1:].astype(np.int64), true_yx[pair_idx[:, 1]] - true_yx[pair_idx[:, 0]])
    yx = _solve(rows, cols, pair_idx, pair_dir, out, status)
    origin = true_yx.min(axis=0)
    assert np.array_equal(yx, true_yx - origin)
    # the stitcher's view of the acquisition: one file per (fov, z); stage positions in mm at 1000 um per
    # pixel, so that the float64 mm -> pixel arithmetic of stitch_region is exact for every position
    infos = [fusion.TileInfo(f"R0_{t}_{z}_Fluorescence_488_nm_Ex.tiff", float(yx[t][1]), float(yx[t][0]),
                             z, "Fluorescence_488_nm_Ex") for z in range(nz) for t in range(n)]
    mh, mw = int(yx[:, 0].max()) + TILE, int(yx[:, 1].max()) + TILE
    want = keep[origin[0]:origin[0] + mh, origin[1]:origin[1] + mw].view(torch.int16)
    covered = _covered_mask(yx, mh, mw, dev)
    for mode in ("overwrite", "feather"):
        region = fusion.stitch_region(tiles.view(nz * n, TILE, TILE), infos, ["Fluorescence_488_nm_Ex"], 1000.0,
                                      selected_z=[mid], mode=mode)
        assert tuple(region.shape) == (1, 1, 1, mh, mw)
        m = region[0, 0, 0].view(torch.int16)
        assert int(((m != want) & covered).sum().item()) == 0, mode
        assert int(((m != 0) & ~covered).sum().item()) == 0, mode
        del region, m


def test_candidate_winner_on_device(dev, golden):
    """The winner of the 2 x 16 candidate search for the golden strip pairs, through
    ``register_pairs``: two tiles whose facing edge strips ARE the golden strips, so the pair's
    row must be the reference's ``interpret_translation`` result plus the strip offset
    (_translation_computation.py:196-205 for the order, :766-792 for the strict '>').
    ``half_width_shift`` is the case with two candidates of equal NCC."""
    from image_stitcher_amd import registration
    from oracle import registration as oreg
    for s in golden["candidates"]:
        h, w = s["shape"]
        dy, dx = s["shift"]
        big = synth.make_scene(s["seed"], h + 64, w + 64)
        a = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
        b = np.ascontiguousarray(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w])
        rng = np.random.default_rng(s["seed"])
        if h >= w:   # tall strips: a "left" pair, the strips are the last / first w columns
            tw = 3 * w
            t1 = rng.integers(1000, 31000, size=(h, tw)).astype(np.uint16)
            t2 = rng.integers(1000, 31000, size=(h, tw)).astype(np.uint16)
            t1[:, tw - w:] = a
            t2[:, :w] = b
            direction, edge, off = 0, w, (0, tw - w)
        else:        # wide strips: a "top" pair, the strips are the last / first h rows
            th = 3 * h
            t1 = rng.integers(1000, 31000, size=(th, w)).astype(np.uint16)
            t2 = rng.integers(1000, 31000, size=(th, w)).astype(np.uint16)
            t1[th - h:, :] = a
            t2[:h, :] = b
            direction, edge, off = 1, h, (th - h, 0)
        out, status = registration.register_pairs(np.stack([t1, t2]), [[0, 1]], [direction], edge)
        row, st = out.cpu().numpy()[0], int(status.cpu()[0])
        ncc_best, y_best, x_best = s["best"]
        assert st == 0, s["name"]
        assert (row[1], row[2]) == (y_best + off[0], x_best + off[1]), s["name"]
        assert row[0] == pytest.approx(ncc_best, abs=1e-12), s["name"]
        # every candidate the reference scored, in its order, scores the same on the device
        for (cy, cx, cn) in s["evaluated"]:
            sub_a, sub_b = oreg.extract_overlap_subregion(a, cy, cx), oreg.extract_overlap_subregion(b, -cy, -cx)
            got = registration.ncc(sub_a, sub_b)
            if np.isfinite(cn):
                assert got == pytest.approx(cn, abs=1e-12), (s["name"], cy, cx)
            else:
                assert got == cn, (s["name"], cy, cx)
