"""Deterministic synthetic acquisitions: textured scenes cut into overlapping tiles.

Used by the tests, the golden-vector generator and the benchmark so that every
consumer regenerates identical inputs from a seed instead of shipping pixels.

The scene is white noise ``uint16(1000 + 30000 * U(0, 1))``: the reference's
phase correlation applies no window, and on smooth scenes spectral leakage
puts spurious peaks near the origin, so textured input is required for a
planted shift to be recoverable at all.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


def make_scene(seed: int, height: int, width: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (1000.0 + 30000.0 * rng.random((height, width))).astype(np.uint16)


def lowpass_scene(scene: np.ndarray, strength: float) -> np.ndarray:
    """Gaussian low-pass of a scene in the frequency domain, ``exp(-strength * |f|^2)`` with f in cycles per
    pixel, stretched back to 1000 .. 51000.  Smooth scenes rank many near-equal correlation peaks: the input on
    which the ten-peak candidate set and its consensus stage differ from the two-peak search."""
    h, w = scene.shape
    f = np.fft.fft2(scene.astype(np.float64))
    k = np.fft.fftfreq(h)[:, None] ** 2 + np.fft.fftfreq(w)[None, :] ** 2
    low = np.fft.ifft2(f * np.exp(-k * float(strength))).real
    return (1000 + 50000 * (low - low.min()) / (low.max() - low.min())).astype(np.uint16)


def make_flatfield(height: int, width: int, dtype=np.float32, amplitude: float = 0.2) -> np.ndarray:
    """Smooth illumination profile ``1 + a cos(pi u) cos(pi v)``, mean-normalised."""
    u = np.linspace(-0.5, 0.5, height, dtype=np.float64)[:, None]
    v = np.linspace(-0.5, 0.5, width, dtype=np.float64)[None, :]
    ff = 1.0 + amplitude * np.cos(np.pi * u) * np.cos(np.pi * v)
    ff = ff / ff.mean()
    return ff.astype(dtype)


@dataclass
class SyntheticGrid:
    tiles: np.ndarray          # [N, H, W] uint16, row-major grid order (r * cols + c)
    true_yx: np.ndarray        # [N, 2] int64 position of each tile's origin in the scene
    rows: int
    cols: int
    pitch_y: int
    pitch_x: int
    pair_idx: np.ndarray       # [P, 2] int32 (first tile, second tile); second is right/below
    pair_dir: np.ndarray       # [P] uint8, 0 = left neighbour pair, 1 = top neighbour pair

    @property
    def expected_shift(self) -> np.ndarray:
        """[P, 2] planted (y, x) of the second tile's origin in the first tile's frame."""
        return self.true_yx[self.pair_idx[:, 1]] - self.true_yx[self.pair_idx[:, 0]]


def grid_pairs(rows: int, cols: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pairs in the reference's visiting order: per tile (row-major), left then top."""
    idx, dirs = [], []
    for r in range(rows):
        for c in range(cols):
            t = r * cols + c
            if c > 0:
                idx.append((t - 1, t))
                dirs.append(0)
            if r > 0:
                idx.append((t - cols, t))
                dirs.append(1)
    return np.asarray(idx, dtype=np.int32).reshape(-1, 2), np.asarray(dirs, dtype=np.uint8)


def make_grid(seed: int, rows: int, cols: int, tile_h: int, tile_w: int,
              overlap: float = 0.10, jitter: int = 3,
              scene: Optional[np.ndarray] = None, lowpass: float = 0.0) -> SyntheticGrid:
    """Cut a ``rows x cols`` grid of tiles with integer position jitter from a scene (white noise, or with
    ``lowpass`` > 0 its low-passed version, see ``lowpass_scene``)."""
    pitch_y = tile_h - int(round(tile_h * overlap))
    pitch_x = tile_w - int(round(tile_w * overlap))
    sh = (rows - 1) * pitch_y + tile_h + 2 * jitter
    sw = (cols - 1) * pitch_x + tile_w + 2 * jitter
    if scene is None:
        scene = make_scene(seed, sh, sw)
        if lowpass > 0:
            scene = lowpass_scene(scene, lowpass)
    rng = np.random.default_rng(seed + 7919)
    jit = rng.integers(-jitter, jitter + 1, size=(rows * cols, 2)) if jitter > 0 \
        else np.zeros((rows * cols, 2), dtype=np.int64)
    jit[0] = 0
    tiles = np.empty((rows * cols, tile_h, tile_w), dtype=np.uint16)
    true_yx = np.empty((rows * cols, 2), dtype=np.int64)
    for r in range(rows):
        for c in range(cols):
            t = r * cols + c
            y0 = jitter + r * pitch_y + int(jit[t, 0])
            x0 = jitter + c * pitch_x + int(jit[t, 1])
            true_yx[t] = (y0, x0)
            tiles[t] = scene[y0:y0 + tile_h, x0:x0 + tile_w]
    pair_idx, pair_dir = grid_pairs(rows, cols)
    return SyntheticGrid(tiles, true_yx, rows, cols, pitch_y, pitch_x, pair_idx, pair_dir)


def planted_positions(seed: int, rows: int, cols: int, tile: int, overlap: float = 0.10, jitter: int = 3):
    """(true_yx [N, 2] int64, scene_h, scene_w) of a jittered grid, the arithmetic of ``make_grid``."""
    pitch = tile - int(round(tile * overlap))
    sh = (rows - 1) * pitch + tile + 2 * jitter
    sw = (cols - 1) * pitch + tile + 2 * jitter
    rng = np.random.default_rng(seed + 7919)
    jit = rng.integers(-jitter, jitter + 1, size=(rows * cols, 2))
    jit[0] = 0
    true_yx = np.array([[jitter + r * pitch + jit[r * cols + c, 0], jitter + c * pitch + jit[r * cols + c, 1]]
                        for r in range(rows) for c in range(cols)], dtype=np.int64)
    return true_yx, sh, sw


def make_device_acquisition(rows: int, cols: int, tile: int, channels: int, seed: int, device,
                            keep_scenes: bool = False, scene_seeds=None, tile_ids=None):
    """Tiles ``[C, N, H, W]`` uint16 cut ON THE DEVICE from white-noise scenes (one per channel),
    with the planted geometry: (tiles, true_yx, pair_idx, pair_dir[, scenes]); ``keep_scenes`` is
    True (all) or the channel indices whose scenes to return; ``tile_ids`` (sorted global tile
    indices) cuts only those tiles: ``tiles`` is then ``[C, len(tile_ids), H, W]`` (what one rank of
    a sharded run holds; every rank draws the same scenes from the same seeds).  Full-size
    acquisitions (13 GB for a 20 x 20 x 4 grid) never exist on the host."""
    import torch

    true_yx, sh, sw = planted_positions(seed, rows, cols, tile)
    ids = np.arange(rows * cols) if tile_ids is None else np.asarray(tile_ids, dtype=np.int64)
    tiles = torch.empty((channels, len(ids), tile, tile), dtype=torch.int16, device=device)
    gen = torch.Generator(device=device)
    scenes = []
    band = 4096
    for c in range(channels):
        gen.manual_seed(seed * 1000 + c if scene_seeds is None else int(scene_seeds[c]))
        scene = torch.empty((sh, sw), dtype=torch.int16, device=device)
        for y in range(0, sh, band):  # in bands: the float32 scene never exists whole
            h = min(band, sh - y)
            scene[y:y + h] = (1000.0 + 30000.0 * torch.rand((h, sw), generator=gen, device=device)).to(torch.int16)
        for n, t in enumerate(ids):  # values <= 31000: same bits as uint16
            y0, x0 = true_yx[t]
            tiles[c, n] = scene[y0:y0 + tile, x0:x0 + tile]
        if keep_scenes is True or (keep_scenes and c in keep_scenes):
            scenes.append(scene.view(torch.uint16))
        del scene
    pair_idx, pair_dir = grid_pairs(rows, cols)
    if keep_scenes:
        return tiles.view(torch.uint16), true_yx, pair_idx, pair_dir, scenes
    return tiles.view(torch.uint16), true_yx, pair_idx, pair_dir
