"""NumPy restatement of the reference's pairwise registration arithmetic.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Citations are
``path:line`` under the reference checkout, with
``tc`` = image_stitcher/registration/_translation_computation.py and
``tr`` = image_stitcher/registration/tile_registration.py.

The restatement covers what the reference executes when scikit-image is not
installed (its own import guard, tc:13-22, then skips the RANSAC and the
subpixel branches): phase correlation, full peak ranking, the 16-candidate
NCC search over the two best peaks, and the bounded least-squares polish.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PCM_EPS = float(np.finfo(np.float64).eps * 100)  # tc:320
FEATURELESS_VAR = 2e3  # tr:497
MIN_OVERLAP_PIXELS = 25  # tc:535


# ----------------------------------------------------------------------------
# a-1  featureless test                                            tr:497-525
# ----------------------------------------------------------------------------
def is_edge_featureless(edge: np.ndarray, variance_threshold: float = FEATURELESS_VAR) -> bool:
    """Population variance of every 16th sample in float32 below threshold."""
    sample = edge[::16, ::16]
    if sample.size == 0:
        return True
    return bool(np.var(sample.astype(np.float32)) < variance_threshold)


# ----------------------------------------------------------------------------
# a-2  peak correlation matrix                                     tc:282-340
# ----------------------------------------------------------------------------
def hann_window(shape) -> np.ndarray:
    """Separable symmetric Hann window ``outer(hanning(h), hanning(w))`` in float64.  NOT reference
    behaviour (the reference applies no window, tc:308-313): the optional extension's own definition."""
    h, w = shape
    return np.outer(np.hanning(h), np.hanning(w))


def pcm(strip1: np.ndarray, strip2: np.ndarray, window: Optional[str] = None) -> np.ndarray:
    """ifft2 of the cross-power spectrum normalised by (|.| + eps), real part.

    No window (unless the ``window="hann"`` extension is asked for), no mean removal, no
    padding: a circular correlation.  Raises
    ValueError for non 2-D / mismatching / non-finite inputs (tc:241-256).
    """
    if strip1.ndim != 2 or strip2.ndim != 2:
        raise ValueError("Images must be 2-dimensional")
    if strip1.shape != strip2.shape:
        raise ValueError(f"Images must have same shape. Got {strip1.shape} and {strip2.shape}")
    if not (np.isfinite(strip1).all() and np.isfinite(strip2).all()):
        raise ValueError("Images contain non-finite values")
    a, b = np.asarray(strip1, dtype=np.float64), np.asarray(strip2, dtype=np.float64)
    if window == "hann":
        wnd = hann_window(a.shape)
        a, b = a * wnd, b * wnd
    elif window is not None:
        raise ValueError(f"unknown window {window!r}")
    f1 = np.fft.fft2(a)
    f2 = np.fft.fft2(b)
    cross = f1 * np.conjugate(f2)
    cross = cross / (np.abs(cross) + PCM_EPS)
    return np.fft.ifft2(cross).real


# ----------------------------------------------------------------------------
# a-3  peak ranking                                                tc:342-441
# ----------------------------------------------------------------------------
def multi_peak_max(pcm_arr: np.ndarray, max_peaks: Optional[int] = None
                   ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """All PCM positions in descending value order (ascending argsort, flipped)."""
    if pcm_arr is None:
        raise TypeError("PCM cannot be None")
    if pcm_arr.ndim != 2:
        raise ValueError(f"PCM must be 2-dimensional, got {pcm_arr.ndim}D array")
    if pcm_arr.size == 0:
        raise ValueError("PCM cannot be empty")
    if min(pcm_arr.shape) < 2:
        raise ValueError(f"PCM dimensions too small: {pcm_arr.shape}")
    if not np.all(np.isfinite(pcm_arr)):
        raise ValueError("PCM contains non-finite values (NaN or infinity)")
    flat = np.asarray(pcm_arr, dtype=np.float64).flatten()
    order = np.argsort(flat)[::-1]
    if max_peaks is not None and max_peaks > 0:
        order = order[: min(int(max_peaks), flat.size)]
    rows, cols = np.unravel_index(order, pcm_arr.shape)
    rows = rows.astype(np.int64)
    cols = cols.astype(np.int64)
    return rows, cols, flat[order].astype(np.float64)


# ----------------------------------------------------------------------------
# a-5  normalised cross correlation                                tc:535-616
# ----------------------------------------------------------------------------
def ncc(patch1: np.ndarray, patch2: np.ndarray, min_overlap_pixels: int = MIN_OVERLAP_PIXELS) -> float:
    """Pearson correlation in float64, two-pass centred form, clipped to [-1, 1]."""
    if patch1.shape != patch2.shape:
        return float("-inf")
    n = int(np.prod(patch1.shape))
    if n == 0 or n < min_overlap_pixels:
        return float("-inf")
    a = np.asarray(patch1, dtype=np.float64)
    b = np.asarray(patch2, dtype=np.float64)
    a = a - np.mean(a)
    b = b - np.mean(b)
    num = float(np.sum(a * b))
    va = float(np.sum(a * a))
    vb = float(np.sum(b * b))
    den = np.sqrt(va * vb)
    if den == 0.0 or not np.isfinite(den):
        if va == 0.0 and vb == 0.0 and np.isclose(num, 0.0):
            return 1.0
        return float("-inf")
    val = num / den
    if not np.isfinite(val):
        return float("-inf")
    return float(np.clip(val, -1.0, 1.0))


# ----------------------------------------------------------------------------
# a-6  overlap window                                              tc:623-657
# ----------------------------------------------------------------------------
def extract_overlap_subregion(img: np.ndarray, y: int, x: int) -> np.ndarray:
    """The part of ``img`` that stays in frame after shifting by (y, x)."""
    h, w = img.shape[:2]
    if abs(y) >= h or abs(x) >= w:
        return np.zeros((0, 0), dtype=img.dtype)
    y0 = int(max(0, min(y, h)))
    y1 = int(max(0, min(y + h, h)))
    x0 = int(max(0, min(x, w)))
    x1 = int(max(0, min(x + w, w)))
    if x0 >= x1 or y0 >= y1:
        return np.zeros((0, 0), dtype=img.dtype)
    return img[y0:y1, x0:x1]


# ----------------------------------------------------------------------------
# a-4  candidate enumeration + NCC search                          tc:660-794
# ----------------------------------------------------------------------------
def candidate_table(py: int, px: int, size_y: int, size_x: int) -> List[Tuple[int, int]]:
    """The 16 signed interpretations of one PCM peak, in evaluation order.

    Order (tc:196-205): magnitude pair (py|H-py) x (px|W-px) outermost, then
    sign of y in (-1, +1), then sign of x in (-1, +1).
    """
    out = []
    for ymag in (py, size_y - py):
        for xmag in (px, size_x - px):
            for ys in (-1, 1):
                for xs in (-1, 1):
                    out.append((ys * ymag, xs * xmag))
    return out


def _peak_is_valid(py: int, px: int, size_y: int, size_x: int,
                   ymin: int, ymax: int, xmin: int, xmax: int) -> bool:
    """tc:135-167: a peak counts if any of its 16 interpretations is in bounds."""
    return any(ymin <= y <= ymax and xmin <= x <= xmax
               for (y, x) in candidate_table(py, px, size_y, size_x))


def interpret_translation(strip1: np.ndarray, strip2: np.ndarray,
                          yins: Sequence[int], xins: Sequence[int],
                          ymin: int, ymax: int, xmin: int, xmax: int,
                          n: int = 2, return_table: bool = False):
    """Best-NCC integer shift among the candidates of the first ``n`` valid peaks.

    Strict ``>`` keeps the first of equal scores; ``(-inf, 0, 0)`` when nothing
    scores.  With ``return_table`` also returns the evaluated
    ``[(y, x, ncc)]`` list (skipped candidates omitted) for the golden files.
    """
    size_y, size_x = strip1.shape
    yins = np.asarray(yins)
    xins = np.asarray(xins)
    best = (float("-inf"), 0, 0)
    table: List[Tuple[int, int, float]] = []
    if yins.size == 0:
        return (best + (table,)) if return_table else best
    ymin, ymax, xmin, xmax = int(ymin), int(ymax), int(xmin), int(xmax)
    full_bounds = (ymin <= -size_y and ymax >= size_y and xmin <= -size_x and xmax >= size_x)
    picked: List[int] = []
    if full_bounds:
        picked = list(range(min(int(n), yins.size)))
    else:
        for i in range(yins.size):
            if _peak_is_valid(int(yins[i]), int(xins[i]), size_y, size_x, ymin, ymax, xmin, xmax):
                picked.append(i)
                if len(picked) >= int(n):
                    break
    for i in picked:
        for (y, x) in candidate_table(int(yins[i]), int(xins[i]), size_y, size_x):
            if not (ymin <= y <= ymax and xmin <= x <= xmax):
                continue
            sub1 = extract_overlap_subregion(strip1, y, x)
            sub2 = extract_overlap_subregion(strip2, -y, -x)
            if sub1.size == 0 or sub2.size == 0:
                continue
            score = ncc(sub1, sub2)
            if np.isnan(score):
                continue
            table.append((y, x, score))
            if score > best[0]:
                best = (score, y, x)
    return (best + (table,)) if return_table else best


# ----------------------------------------------------------------------------
# a-7  "Method 1", deterministic part                      tc:876-994, 1019-1038, 1112-1118
# ----------------------------------------------------------------------------
def generate_translation_candidates(strip1: np.ndarray, strip2: np.ndarray, yins: Sequence[int], xins: Sequence[int],
                                    ymin: int, ymax: int, xmin: int, xmax: int, n: int = 10):
    """(translations [N, 2] float, ncc_scores [N]) of the candidates of the first ``n`` valid peaks whose
    NCC is finite, in evaluation order (tc:876-994): what the reference hands to its consensus step."""
    size_y, size_x = strip1.shape
    yins, xins = np.asarray(yins), np.asarray(xins)
    if yins.size == 0:
        return np.empty((0, 2)), np.empty((0,))
    ymin, ymax, xmin, xmax = int(ymin), int(ymax), int(xmin), int(xmax)
    picked = [i for i in range(yins.size)
              if _peak_is_valid(int(yins[i]), int(xins[i]), size_y, size_x, ymin, ymax, xmin, xmax)][: int(n)] \
        if not (ymin <= -size_y and ymax >= size_y and xmin <= -size_x and xmax >= size_x) \
        else list(range(min(int(n), yins.size)))
    translations, scores = [], []
    for i in picked:
        for (y, x) in candidate_table(int(yins[i]), int(xins[i]), size_y, size_x):
            if not (ymin <= y <= ymax and xmin <= x <= xmax):
                continue
            sub1 = extract_overlap_subregion(strip1, y, x)
            sub2 = extract_overlap_subregion(strip2, -y, -x)
            if sub1.size == 0 or sub2.size == 0:
                continue
            v = ncc(sub1, sub2)
            if np.isnan(v) or v == float("-inf"):
                continue
            translations.append([float(y), float(x)])
            scores.append(v)
    if not translations:
        return np.empty((0, 2)), np.empty((0,))
    return np.array(translations), np.array(scores)


def best_single_translation(translations: np.ndarray, scores: np.ndarray):
    """What ransac_translation_estimation returns whenever the consensus is unavailable or declined
    (tc:1019-1038, 1112-1118, 1129-1140): the FIRST candidate of the largest NCC."""
    if len(translations) == 0:
        return None
    i = int(np.argmax(scores))
    return float(scores[i]), float(translations[i][0]), float(translations[i][1])


def compute_translation_peaks10(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, n: int = 10):
    """The reference's result list with Method 1 standing on its best-NCC fallback: (method 1, discrete search),
    first of equal scores (tc:1195-1230, 1274-1313)."""
    results = []
    t, sc = generate_translation_candidates(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, n)
    m1 = best_single_translation(t, sc)
    if m1 is not None and ymin <= m1[1] <= ymax and xmin <= m1[2] <= xmax and np.isfinite(m1[0]):
        results.append(m1)
    d = interpret_translation(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, min(int(n), 2))
    if d[0] != float("-inf"):
        results.append((d[0], float(d[1]), float(d[2])))
    if not results:
        return float("-inf"), 0.0, 0.0
    return max(results, key=lambda r: r[0])


def register_pairs_peaks10(tiles: np.ndarray, pair_idx, directions: Sequence[int], edge_width: int):
    """``register_pairs`` with the 160-candidate walk; also returns every pair's (translations, scores)."""
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    out = np.full((len(pair_idx), 3), np.nan, dtype=np.float64)
    status = np.zeros(len(pair_idx), dtype=np.int8)
    tables = []
    for p, (i1, i2) in enumerate(pair_idx):
        e1, e2, off_y, off_x = edge_strips(tiles[i1], tiles[i2], "top" if directions[p] else "left", edge_width)
        if e1.size == 0 or e2.size == 0 or is_edge_featureless(e1) or is_edge_featureless(e2):
            status[p] = 1
            tables.append(None)
            continue
        eh, ew = e1.shape
        yins, xins, _ = multi_peak_max(pcm(e1, e2))
        tables.append(generate_translation_candidates(e1, e2, yins, xins, -eh, eh, -ew, ew, 10))
        score, y, x = compute_translation_peaks10(e1, e2, yins, xins, -eh, eh, -ew, ew, 10)
        out[p] = (score, y + off_y, x + off_x)
    return out, status, tables


# ----------------------------------------------------------------------------
# a-7  "Method 1", the consensus step as the reference executes it     tc:997-1140, 1197-1230
# ----------------------------------------------------------------------------
CONSENSUS_MIN_SAMPLES = 3          # RANSACConfig.min_samples, tc:74
CONSENSUS_THRESHOLD = 2.0          # residual_threshold, tc:75 and tr:573
CONSENSUS_MIN_INLIER_RATIO = 0.3   # min_inlier_ratio, tc:78 and tr:573


def consensus_translation(translations: np.ndarray, scores: np.ndarray):
    """(ncc, y, x, used_consensus) of ``ransac_translation_estimation`` + the re-scoring at tc:1217-1227, or None.

    The reference hands ``skimage.measure.ransac`` a model class whose ``estimate(self, data, *args)`` unpacks
    ``src, dst = data`` (tc:1053-1055).  ``ransac`` calls ``estimate(*samples)``, so ``data`` is the 3-row sample of
    ``src`` alone (min_samples = 3), the unpacking raises, and the reference's own ``except`` branch decides
    (tc:1085-1107): pairwise Euclidean distances, neighbours within the residual threshold (the candidate itself
    counts), the FIRST candidate with the most neighbours.  Fewer neighbours than ``min_inlier_ratio`` of the
    candidates (tc:1110-1118) or fewer than ``min_samples`` candidates (tc:1030-1038): the first candidate of the
    largest NCC.  The chosen shift is integer valued, so the NCC at its rounded position is its own score.
    """
    n = len(translations)
    if n == 0:
        return None
    pick, used = int(np.argmax(scores)), False
    if n >= CONSENSUS_MIN_SAMPLES:
        t = np.asarray(translations, dtype=np.float64)
        diff = t[:, None, :] - t[None, :, :]
        dist = np.sqrt((diff * diff).sum(axis=2))
        neighbours = np.sum(dist <= CONSENSUS_THRESHOLD, axis=1)
        centre = int(np.argmax(neighbours))
        if not (int(neighbours[centre]) / n < CONSENSUS_MIN_INLIER_RATIO):
            pick, used = centre, True
    return float(scores[pick]), float(translations[pick][0]), float(translations[pick][1]), used


def compute_translation_consensus(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, n: int = 10):
    """The reference's result list with its consensus stage live and no subpixel stage: (method 1, discrete
    search); the first of equal scores wins (tc:1195-1230, 1274-1313).  The polish (tc:1285-1307) returns its start
    point and loses the tie, so it is left out here as in ``compute_translation(polish=False)``."""
    results = []
    t, sc = generate_translation_candidates(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, n)
    m1 = consensus_translation(t, sc)
    if m1 is not None and ymin <= m1[1] <= ymax and xmin <= m1[2] <= xmax and np.isfinite(m1[0]):
        results.append(m1[:3])
    d = interpret_translation(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax, min(int(n), 2))
    if d[0] != float("-inf"):
        results.append((d[0], float(d[1]), float(d[2])))
    if not results:
        return float("-inf"), 0.0, 0.0
    return max(results, key=lambda r: r[0])


def register_pairs_consensus(tiles: np.ndarray, pair_idx, directions: Sequence[int], edge_width: int):
    """``register_pairs`` with the consensus stage live; also returns per pair whether the consensus stood."""
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    out = np.full((len(pair_idx), 3), np.nan, dtype=np.float64)
    status = np.zeros(len(pair_idx), dtype=np.int8)
    stood = []
    for p, (i1, i2) in enumerate(pair_idx):
        e1, e2, off_y, off_x = edge_strips(tiles[i1], tiles[i2], "top" if directions[p] else "left", edge_width)
        if e1.size == 0 or e2.size == 0 or min(e1.shape) < 2 or is_edge_featureless(e1) or is_edge_featureless(e2):
            status[p] = 1
            stood.append(None)
            continue
        eh, ew = e1.shape
        yins, xins, _ = multi_peak_max(pcm(e1, e2))
        m1 = consensus_translation(*generate_translation_candidates(e1, e2, yins, xins, -eh, eh, -ew, ew, 10))
        stood.append(None if m1 is None else m1[3])
        score, y, x = compute_translation_consensus(e1, e2, yins, xins, -eh, eh, -ew, ew, 10)
        out[p] = (score, y + off_y, x + off_x)
    return out, status, stood


# ----------------------------------------------------------------------------
# a-8  bounded least-squares polish                               tc:1466-1588
# ----------------------------------------------------------------------------
def optimize_translation_continuous(strip1, strip2, y0: float, x0: float,
                                    ymin: float, ymax: float, xmin: float, xmax: float,
                                    max_nfev: int = 50):
    """scipy 'trf' on sqrt(1 - ncc(round(y), round(x))).

    The residual is piecewise constant, so the finite-difference Jacobian is
    zero and the solver returns its start point; kept because the reference
    pays for it and its result competes (and loses ties) in the final max.
    """
    from scipy.optimize import least_squares

    if not (ymin <= y0 <= ymax):
        raise ValueError("Initial Y out of bounds")
    if not (xmin <= x0 <= xmax):
        raise ValueError("Initial X out of bounds")

    def score_at(y, x):
        yi, xi = int(np.round(y)), int(np.round(x))
        s1 = extract_overlap_subregion(strip1, yi, xi)
        s2 = extract_overlap_subregion(strip2, -yi, -xi)
        if s1.size == 0 or s2.size == 0:
            return None
        return ncc(s1, s2)

    def residual(p):
        v = score_at(p[0], p[1])
        if v is None or np.isnan(v) or np.isinf(v):
            return np.array([1.0])
        return np.array([np.sqrt(np.maximum(0.0, 1.0 - v))])

    res = least_squares(residual, np.array([y0, x0], dtype=np.float64),
                        bounds=([ymin, xmin], [ymax, xmax]), method="trf",
                        max_nfev=max_nfev, ftol=1e-6, xtol=1e-6, gtol=1e-6, verbose=0)
    oy, ox = res.x
    v = score_at(oy, ox)
    if v is None or np.isnan(v) or np.isinf(v):
        v = 0.0
    return v, float(oy), float(ox)


# ----------------------------------------------------------------------------
# a-7  strategy switch, RANSAC entry without scikit-image  tc:797-873, 1143-1320
# ----------------------------------------------------------------------------
def compute_translation(strip1, strip2, yins, xins, ymin, ymax, xmin, xmax,
                        n: int = 10, polish: bool = True) -> Tuple[float, float, float]:
    """What ``compute_translation(strategy=RANSAC)`` evaluates without scikit-image.

    Result list order is (discrete search, polish); ``max`` keeps the first of
    equal scores, so the polish can only win with a strictly larger NCC.
    """
    results = []
    d_ncc, d_y, d_x = interpret_translation(strip1, strip2, yins, xins,
                                            ymin, ymax, xmin, xmax, min(int(n), 2))
    if d_ncc != float("-inf"):
        results.append((d_ncc, float(d_y), float(d_x)))
    if polish and results:
        _, sy, sx = max(results, key=lambda r: r[0])
        try:
            c = optimize_translation_continuous(strip1, strip2, sy, sx,
                                                float(ymin), float(ymax), float(xmin), float(xmax))
            if c[0] != float("-inf"):
                results.append(c)
        except Exception:
            pass
    if not results:
        return float("-inf"), 0.0, 0.0
    return max(results, key=lambda r: r[0])


# ----------------------------------------------------------------------------
# a-0  per-pair driver                                             tr:527-586
# ----------------------------------------------------------------------------
def edge_strips(img1: np.ndarray, img2: np.ndarray, direction: str, edge_width: int):
    """(strip1, strip2, offset_y, offset_x) for 'left' / 'top' (tr:537-548)."""
    size_y, size_x = img1.shape
    if direction == "left":
        w = min(edge_width, size_x)
        return img1[:, size_x - w:], img2[:, :w], 0, size_x - w
    if direction == "top":
        w = min(edge_width, size_y)
        return img1[size_y - w:, :], img2[:w, :], size_y - w, 0
    raise ValueError(direction)


def compute_single_translation(img1: np.ndarray, img2: np.ndarray, direction: str,
                               edge_width: int, polish: bool = True, window: Optional[str] = None
                               ) -> Optional[Tuple[float, float, float]]:
    """(ncc, y, x) of img2's origin in img1's frame, or None (featureless / failed)."""
    try:
        if direction not in ("left", "top"):
            return None
        e1, e2, off_y, off_x = edge_strips(img1, img2, direction, edge_width)
        if e1.size == 0 or e2.size == 0:
            return None
        if is_edge_featureless(e1) or is_edge_featureless(e2):
            return None
        eh, ew = e1.shape
        corr = pcm(e1, e2, window)
        yins, xins, _ = multi_peak_max(corr)
        score, y, x = compute_translation(e1, e2, yins, xins, -eh, eh, -ew, ew, n=10, polish=polish)
        return score, y + off_y, x + off_x
    except Exception:
        return None


# ----------------------------------------------------------------------------
# a-14  batch seam                                                 tr:310-378
# ----------------------------------------------------------------------------
def register_pairs(tiles: np.ndarray, pair_idx: np.ndarray, directions: Sequence[int],
                   edge_width: int, polish: bool = True, window: Optional[str] = None):
    """Serial loop over pairs. ``pair_idx[p] = (i1, i2)``; direction 0=left, 1=top.

    Returns (out[P,3] float64 with NaN rows for failed pairs, status[P] int8:
    0 ok, 1 featureless/failed).
    """
    pair_idx = np.asarray(pair_idx).reshape(-1, 2)
    out = np.full((len(pair_idx), 3), np.nan, dtype=np.float64)
    status = np.zeros(len(pair_idx), dtype=np.int8)
    for p, (i1, i2) in enumerate(pair_idx):
        r = compute_single_translation(tiles[i1], tiles[i2], "top" if directions[p] else "left",
                                       edge_width, polish=polish, window=window)
        if r is None:
            status[p] = 1
        else:
            out[p] = r
    return out, status


def neighborhood_results(pair_idx: np.ndarray, directions: Sequence[int], out: np.ndarray,
                         status: np.ndarray) -> Dict[int, Dict[str, Dict[str, float]]]:
    """Pack per-pair rows into the nested dict the reference's batch seam returns.

    Key = centre tile index (second of the pair); a direction key is absent when
    the pair failed (tr:352-376).
    """
    res: Dict[int, Dict[str, Dict[str, float]]] = {}
    for p, (_, centre) in enumerate(np.asarray(pair_idx).reshape(-1, 2)):
        d = res.setdefault(int(centre), {})
        if status[p] == 0:
            d["top" if directions[p] else "left"] = {
                "ncc": float(out[p, 0]), "y": float(out[p, 1]), "x": float(out[p, 2])}
    return res
