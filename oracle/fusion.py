"""NumPy restatement of the reference's tile fusion arithmetic.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``st`` =
image_stitcher/stitcher.py, ``pa`` = image_stitcher/parameters.py,
``tr`` = image_stitcher/registration/tile_registration.py in the reference.
``st``/``pa`` cannot be imported without dask/zarr/ome_zarr, so this file
follows their text and is pinned by the known-answer pixels of the reference's
``stitcher_test.py`` (tests/test_oracle_fusion.py) and, for the flatfield
arithmetic, by vectors produced by the importable ``tr.apply_flatfield_to_image``.

``feather`` has NO reference counterpart (the reference only overwrites,
st:296-300): it is this project's own specification, parity unpinned.
"""
from __future__ import annotations

import math
from typing import List, Dict, Optional, Sequence, Tuple

import numpy as np


# ---- a-10 host scalars ------------------------------------------------------
def pixel_size_um(sensor_pixel_size_um: float, magnification: float, tube_lens_f_mm: float,
                  tube_lens_mm: float) -> float:
    """pa:374-382."""
    objective_focal_mm = tube_lens_f_mm / magnification
    actual_mag = tube_lens_mm / objective_focal_mm
    return sensor_pixel_size_um / actual_mag


def pixel_offset(pos_mm: float, min_mm: float, px_um: float) -> int:
    """st:392-397: float64 arithmetic, truncation toward zero."""
    return int((pos_mm - min_mm) * 1000 / px_um)


def output_dimensions(x_mm: Sequence[float], y_mm: Sequence[float], tile_h: int, tile_w: int,
                      px_um: float) -> Tuple[int, int]:
    """(width_px, height_px) of the mosaic, pa:1009-1021."""
    width_mm = max(x_mm) - min(x_mm) + (tile_w * px_um / 1000)
    height_mm = max(y_mm) - min(y_mm) + (tile_h * px_um / 1000)
    return int(np.ceil(width_mm * 1000 / px_um)), int(np.ceil(height_mm * 1000 / px_um))


def placement_order(filenames: Sequence[str]) -> np.ndarray:
    """Tiles are placed in sorted-path order (pa:614-621), i.e. lexicographic:
    ``R0_10_0_x.tiff`` before ``R0_2_0_x.tiff``.  Returns the index permutation."""
    return np.asarray(sorted(range(len(filenames)), key=lambda i: filenames[i]), dtype=np.int64)


# ---- a-9 flatfield ----------------------------------------------------------
def apply_flatfield(tile: np.ndarray, flatfield: np.ndarray) -> np.ndarray:
    """tr:2177-2180 / st:260-267: true division, clip to the dtype range, truncating cast."""
    if tile.shape != flatfield.shape:
        raise ValueError(f"Flatfield shape {flatfield.shape} doesn't match image shape {tile.shape}")
    info = np.iinfo(tile.dtype)
    return (tile / flatfield).clip(min=info.min, max=info.max).astype(tile.dtype)


# ---- a-13 projection --------------------------------------------------------
def compute_mip(tiles: Sequence[np.ndarray]) -> np.ndarray:
    """st:102-121."""
    if not tiles:
        raise ValueError("Cannot compute MIP from empty tile list")
    return np.max(np.stack(tiles, axis=0), axis=0)


# ---- a-11 placement ---------------------------------------------------------
def place_single_channel_tile(mosaic5d: np.ndarray, tile: np.ndarray, x_pixel: int, y_pixel: int,
                              z_level: int, channel_idx: int,
                              flatfield: Optional[np.ndarray] = None) -> None:
    """st:272-311: optional flatfield, clip to the mosaic, overwrite."""
    if mosaic5d.ndim != 5:
        raise ValueError(f"Unexpected stitched_region shape: {mosaic5d.shape}. Expected 5D array (t, c, z, y, x).")
    if flatfield is not None:
        tile = apply_flatfield(tile, flatfield)
    y_end = min(y_pixel + tile.shape[0], mosaic5d.shape[3])
    x_end = min(x_pixel + tile.shape[1], mosaic5d.shape[4])
    mosaic5d[0, channel_idx, z_level, y_pixel:y_end, x_pixel:x_end] = tile[: y_end - y_pixel, : x_end - x_pixel]


def fuse_overwrite(tiles: np.ndarray, tile_yx: np.ndarray, tile_plane: Sequence[int], n_planes: int,
                   mosaic_h: int, mosaic_w: int, flatfields: Optional[np.ndarray] = None,
                   tile_ff: Optional[Sequence[int]] = None) -> np.ndarray:
    """Serial placement loop of st:388-405 over tiles given IN PLACEMENT ORDER.

    Returns [n_planes, mosaic_h, mosaic_w]; plane = channel * Z + z.
    """
    out = np.zeros((1, n_planes, 1, mosaic_h, mosaic_w), dtype=tiles.dtype)
    for t in range(len(tiles)):
        ff = None
        if flatfields is not None and tile_ff is not None and tile_ff[t] >= 0:
            ff = flatfields[tile_ff[t]]
        place_single_channel_tile(out, tiles[t], int(tile_yx[t][1]), int(tile_yx[t][0]), 0, int(tile_plane[t]), ff)
    return out[0, :, 0]


def place_tile(mosaic5d: np.ndarray, tile: np.ndarray, x_pixel: int, y_pixel: int, z_level: int, channel: str,
               monochrome_channels: Sequence[str], flatfields: Optional[Dict[int, np.ndarray]] = None) -> None:
    """st:208-253: a 2-D tile goes to its channel's plane; an (H, W, 3) tile to the planes ``{base}_R/_G/_B`` with
    ``base = channel.split("_")[0]``; a (1, H, W) tile is squeezed.  Flatfields are looked up by plane index
    (st:255-270)."""
    def put(plane2d, idx):
        ff = flatfields.get(idx) if flatfields else None
        place_single_channel_tile(mosaic5d, plane2d, x_pixel, y_pixel, z_level, idx, ff)
    if tile.ndim == 2:
        put(tile, list(monochrome_channels).index(channel))
    elif tile.ndim == 3:
        if tile.shape[2] == 3:
            base = channel.split("_")[0]
            for i, color in enumerate(["R", "G", "B"]):
                put(tile[:, :, i], list(monochrome_channels).index(f"{base}_{color}"))
        elif tile.shape[0] == 1:
            put(tile[0], list(monochrome_channels).index(channel))
    else:
        raise ValueError(f"Unexpected tile shape: {tile.shape}")


def stitch_region(tiles: Sequence[np.ndarray], keys: Sequence[Tuple[int, int, str]], filenames: Sequence[str],
                  x_mm: Sequence[float], y_mm: Sequence[float], monochrome_channels: Sequence[str], px_um: float,
                  selected_z: Optional[Sequence[int]] = None, apply_mip: bool = False,
                  flatfields: Optional[Dict[int, np.ndarray]] = None, progress: Optional[list] = None) -> np.ndarray:
    """The tile loop of Stitcher.stitch_region (st:313-410) over tiles already in memory.
    ``keys[i] = (fov, z_level, channel)``; iteration order = sorted file names (pa:614-621).
    Returns the (1, C, Z, H, W) array; Z = 1 with ``apply_mip`` (st:152)."""
    order = [int(i) for i in placement_order(filenames)]
    zs = sorted(set(k[1] for k in keys)) if selected_z is None else sorted(set(selected_z))
    z_index = {z: i for i, z in enumerate(zs)}
    order = [i for i in order if keys[i][1] in z_index]
    th, tw = tiles[0].shape[-3:-1] if (tiles[0].ndim == 3 and tiles[0].shape[2] == 3) else tiles[0].shape[-2:]
    width, height = output_dimensions(list(x_mm), list(y_mm), th, tw, px_um)
    x_min, y_min = min(x_mm), min(y_mm)
    out = np.zeros((1, len(monochrome_channels), 1 if apply_mip else len(zs), height, width), dtype=tiles[0].dtype)
    total, done = len(order), 0
    if apply_mip:
        groups: Dict[Tuple[int, str], list] = {}
        for i in order:
            groups.setdefault((keys[i][0], keys[i][2]), []).append(i)
        for (fov, channel), members in groups.items():
            mip = compute_mip([tiles[i] for i in members])
            first = members[0]
            place_tile(out, mip, pixel_offset(x_mm[first], x_min, px_um), pixel_offset(y_mm[first], y_min, px_um), 0,
                       channel, monochrome_channels, flatfields)
            if progress is not None:
                progress.append((done, total))
            done += len(members)
    else:
        for i in order:
            place_tile(out, tiles[i], pixel_offset(x_mm[i], x_min, px_um), pixel_offset(y_mm[i], y_min, px_um),
                       z_index[keys[i][1]], keys[i][2], monochrome_channels, flatfields)
            if progress is not None:
                progress.append((done, total))
            done += 1
    return out


# ---- a-12 feather (own specification) --------------------------------------
def feather_weights(tile_h: int, tile_w: int) -> np.ndarray:
    """Separable linear ramp ``min(y+1, H-y) * min(x+1, W-x)`` as float32 (exact integers)."""
    wy = np.minimum(np.arange(tile_h) + 1, tile_h - np.arange(tile_h)).astype(np.float32)
    wx = np.minimum(np.arange(tile_w) + 1, tile_w - np.arange(tile_w)).astype(np.float32)
    return wy[:, None] * wx[None, :]


def fuse_feather(tiles: np.ndarray, tile_yx: np.ndarray, tile_plane: Sequence[int], n_planes: int,
                 mosaic_h: int, mosaic_w: int, flatfields: Optional[np.ndarray] = None,
                 tile_ff: Optional[Sequence[int]] = None) -> np.ndarray:
    """out = floor(sum(w v) / sum(w) + 0.5) with every float32 operation rounded on
    its own, accumulated in placement order; uncovered pixels stay 0."""
    h, w = tiles.shape[1:]
    wt = feather_weights(h, w)
    num = np.zeros((n_planes, mosaic_h, mosaic_w), dtype=np.float32)
    den = np.zeros((n_planes, mosaic_h, mosaic_w), dtype=np.float32)
    for t in range(len(tiles)):
        tile = tiles[t]
        if flatfields is not None and tile_ff is not None and tile_ff[t] >= 0:
            tile = apply_flatfield(tile, flatfields[tile_ff[t]])
        y, x = int(tile_yx[t][0]), int(tile_yx[t][1])
        y1, x1 = min(y + h, mosaic_h), min(x + w, mosaic_w)
        p = int(tile_plane[t])
        wv = wt[: y1 - y, : x1 - x]
        num[p, y:y1, x:x1] = num[p, y:y1, x:x1] + wv * tile[: y1 - y, : x1 - x].astype(np.float32)
        den[p, y:y1, x:x1] = den[p, y:y1, x:x1] + wv
    out = np.zeros((n_planes, mosaic_h, mosaic_w), dtype=tiles.dtype)   # the tiles' type, as the overwrite mosaic
    covered = den > 0
    q = np.floor(num[covered] / den[covered] + np.float32(0.5))
    out[covered] = np.minimum(q, np.float32(np.iinfo(tiles.dtype).max)).astype(tiles.dtype)
    return out


# ----------------------------------------------------------------------------
# resolution pyramid                                             st:652-668
# ----------------------------------------------------------------------------
def generate_pyramid(image: np.ndarray, num_levels: int) -> List[np.ndarray]:
    """Level l = np.mean over 2^l x 2^l blocks of the last two axes of the ORIGINAL image,
    excess trimmed: what ``da.coarsen(np.mean, image, {3: s, 4: s}, trim_excess=True)``
    computes (dask reshapes each axis into (n // s, s) and reduces the second part);
    the result is float64.  dask is not installed here: restated from its documented
    behaviour, not executed."""
    out = [image]
    h, w = image.shape[-2:]
    for level in range(1, num_levels):
        s = 2 ** level
        hh, ww = h // s, w // s
        trimmed = image[..., :hh * s, :ww * s]
        blocks = trimmed.reshape(image.shape[:-2] + (hh, s, ww, s))
        out.append(np.mean(blocks, axis=(-3, -1)))
    return out
