"""Shared fusion test scenes (inputs only; no expectations)."""
import numpy as np

from image_stitcher_amd import synth


def reference_fixture_grid(n_rows, n_cols, size, num_z=1):
    """The scene the reference's own stitcher tests build (testutil.py:58-90):
    abutting constant tiles whose value is ``fov * num_z + z``; 1 um pixels."""
    names, x_mm, y_mm, tiles, zs = [], [], [], [], []
    fov = 0
    for r in range(n_rows):
        for c in range(n_cols):
            for z in range(num_z):
                names.append(f"R0_{fov}_{z}_DAPI.tiff")
                x_mm.append(c * 1.0)
                y_mm.append(r * 1.0)
                zs.append(z)
                tiles.append(np.full((size, size), (fov * num_z + z) % 65536, dtype=np.uint16))
            fov += 1
    return names, x_mm, y_mm, zs, np.stack(tiles)


def overlapping_scene(seed, rows, cols, tile_h, tile_w, n_planes=1, jitter=3, overlap=0.1):
    """Random-content tiles (every tile its own noise, so overlaps disagree and the
    winner is observable) on a jittered grid, in a shuffled placement order."""
    rng = np.random.default_rng(seed)
    pitch_y = tile_h - int(round(tile_h * overlap))
    pitch_x = tile_w - int(round(tile_w * overlap))
    n = rows * cols * n_planes
    tiles = rng.integers(0, 65536, size=(n, tile_h, tile_w), dtype=np.uint16)
    yx, plane = [], []
    for p in range(n_planes):
        for r in range(rows):
            for c in range(cols):
                jy, jx = rng.integers(0, 2 * jitter + 1, size=2)
                yx.append((r * pitch_y + jy, c * pitch_x + jx))
                plane.append(p)
    yx = np.asarray(yx, dtype=np.int32)
    plane = np.asarray(plane, dtype=np.int32)
    perm = rng.permutation(n)
    mosaic_h = int(yx[:, 0].max()) + tile_h - 1  # last rows/cols of the far tiles are clipped
    mosaic_w = int(yx[:, 1].max()) + tile_w - 5
    return tiles[perm], yx[perm], plane[perm], mosaic_h, mosaic_w
