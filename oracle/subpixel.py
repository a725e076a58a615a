"""Upsampled-DFT subpixel registration: CPU restatement.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference delegates this step to a third-party function,
``skimage.registration.phase_cross_correlation(..., upsample_factor=100,
normalization='phase')`` (``_translation_computation.py:1233-1272``, imported at
``:14``); scikit-image is listed without a version in the reference's
``environment.yml`` and is neither vendored in the reference nor installed here.
What follows restates the published algorithm of that function (Guizar-Sicairos,
Thurman, Fienup, "Efficient subpixel image registration algorithms", Opt. Lett.
33, 156 (2008)) as scikit-image 0.19-0.22 implement it:

 1. image_product = fft2(a) * conj(fft2(b)), divided by max(|.|, 100 eps);
 2. coarse shift = argmax |ifft2(image_product)|, wrapped to signed;
 3. the inverse DFT is re-evaluated by two matrix products on a
    ceil(1.5 u) x ceil(1.5 u) grid of spacing 1/u around the coarse shift and the
    largest magnitude taken (first in row-major order on ties).

The reference then scores that shift by NCC at its rounded position and lets
it compete with the discrete search (``:1257-1320``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import registration as oreg


def upsampled_dft(data: np.ndarray, region: int, upsample_factor: float, offsets) -> np.ndarray:
    """Two matrix products: rows of exp(-2 pi i (u - offset) f / upsample) against each axis, last axis first."""
    for n_items, off in list(zip(data.shape, offsets))[::-1]:
        kernel = (np.arange(region) - off)[:, None] * np.fft.fftfreq(n_items, upsample_factor)
        kernel = np.exp(-2j * np.pi * kernel)
        data = np.tensordot(kernel, data, axes=(1, -1))
    return data


def phase_cross_correlation(a: np.ndarray, b: np.ndarray, upsample_factor: int = 100) -> np.ndarray:
    """Shift (dy, dx) that registers ``b`` onto ``a`` (normalization='phase', space='real')."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise ValueError("images must be same shape")
    fa, fb = np.fft.fft2(a), np.fft.fft2(b)
    shape = np.array(fa.shape)
    product = fa * fb.conj()
    eps = np.finfo(np.float64).eps
    product = product / np.maximum(np.abs(product), 100 * eps)
    cc = np.fft.ifft2(product)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    midpoint = np.fix(shape / 2)
    shift = np.array(maxima, dtype=np.float64)
    shift[shift > midpoint] -= shape[shift > midpoint]
    if upsample_factor > 1:
        u = float(upsample_factor)
        shift = np.round(shift * u) / u
        region = int(np.ceil(u * 1.5))
        dftshift = np.fix(region / 2.0)
        offsets = dftshift - shift * u
        cc_up = upsampled_dft(product.conj(), region, u, offsets).conj()
        maxima = np.array(np.unravel_index(np.argmax(np.abs(cc_up)), cc_up.shape), dtype=np.float64)
        maxima -= dftshift
        shift = shift + maxima / u
    for dim in range(a.ndim):
        if shape[dim] == 1:
            shift[dim] = 0.0
    return shift


def subpixel_candidate(strip1: np.ndarray, strip2: np.ndarray, upsample_factor: int = 100
                       ) -> Optional[Tuple[float, float, float]]:
    """The reference's "Method 2" entry (tc:1233-1272): (ncc at the rounded shift, y, x) or None."""
    sy, sx = (float(v) for v in phase_cross_correlation(strip1, strip2, upsample_factor))
    h, w = strip1.shape
    if not (-h <= sy <= h and -w <= sx <= w):
        return None
    yi, xi = int(np.round(sy)), int(np.round(sx))
    s1 = oreg.extract_overlap_subregion(strip1, yi, xi)
    s2 = oreg.extract_overlap_subregion(strip2, -yi, -xi)
    if s1.size == 0 or s2.size == 0:
        return None
    v = oreg.ncc(s1, s2)
    if not np.isfinite(v):
        return None
    return v, sy, sx


def compute_translation_with_subpixel(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, n: int = 10,
                                      upsample_factor: int = 100) -> Tuple[float, float, float]:
    """``compute_translation(strategy=RANSAC)`` with scikit-image present, minus its RANSAC
    consensus step (tc:1197-1230, randomised): results are collected in the order
    (subpixel, discrete) and the first of equal scores wins (tc:1313)."""
    results = []
    c = subpixel_candidate(strip1, strip2, upsample_factor)
    if c is not None:
        results.append(c)
    d = oreg.interpret_translation(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, min(int(n), 2))
    if d[0] != float("-inf"):
        results.append((d[0], float(d[1]), float(d[2])))
    if not results:
        return float("-inf"), 0.0, 0.0
    return max(results, key=lambda r: r[0])


def register_pairs_subpixel(tiles: np.ndarray, pair_idx, directions, edge_width: int, upsample_factor: int = 100):
    """Like ``oracle.registration.register_pairs`` with the subpixel candidate competing."""
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    out = np.full((len(pair_idx), 3), np.nan, dtype=np.float64)
    status = np.zeros(len(pair_idx), dtype=np.int8)
    raw = np.full((len(pair_idx), 2), np.nan, dtype=np.float64)
    for p, (i1, i2) in enumerate(pair_idx):
        e1, e2, off_y, off_x = oreg.edge_strips(tiles[i1], tiles[i2], "top" if directions[p] else "left", edge_width)
        if oreg.is_edge_featureless(e1) or oreg.is_edge_featureless(e2):
            status[p] = 1
            continue
        eh, ew = e1.shape
        yins, xins, _ = oreg.multi_peak_max(oreg.pcm(e1, e2))
        raw[p] = phase_cross_correlation(e1, e2, upsample_factor)
        score, y, x = compute_translation_with_subpixel(e1, e2, yins, xins, -eh, eh, -ew, ew, 10, upsample_factor)
        out[p] = (score, y + off_y, x + off_x)
    return out, status, raw
