"""Device-side mirror of the reference's pairwise registration interface.

Same names, argument meaning and failure behaviour as
``image_stitcher/registration/tile_registration.py`` and
``image_stitcher/registration/_translation_computation.py`` in the reference,
computed by libmistitch on an MI355X.  NumPy arrays in / Python scalars out for
the per-pair functions; the batched entry point (`register_pairs`) keeps
everything on the device.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from . import _lib
from ._lib import (FFT_F32, FFT_F64, PAIR_FEATURELESS, PAIR_NO_CANDIDATE, PAIR_OK, REG_CONSENSUS, REG_PEAKS10,
                   REG_SUBPIXEL, REG_WINDOW_HANN, check, lib, ptr)

logger = logging.getLogger(__name__)

DEFAULT_EDGE_WIDTH = 256  # tile_registration.py:181


@dataclass
class Neighborhood:
    """A tile and the neighbours it is registered against (tile_registration.py:195-209)."""
    center_idx: int
    left_idx: Optional[int] = None
    top_idx: Optional[int] = None

    def get_all_indices(self) -> Set[int]:
        out = {self.center_idx}
        if self.left_idx is not None:
            out.add(self.left_idx)
        if self.top_idx is not None:
            out.add(self.top_idx)
        return out


def _torch():
    import torch
    return torch


def _fft_flag(fft: str, window: Optional[str] = None) -> int:
    """Library flags for an FFT precision and an optional window ("hann"; the reference applies none)."""
    if window not in (None, "hann"):
        raise ValueError(f"unknown window {window!r}")
    w = REG_WINDOW_HANN if window == "hann" else 0
    if fft in ("f64", "float64", "complex128"):
        return FFT_F64 | w
    if fft in ("f32", "float32", "complex64"):
        return FFT_F32 | w
    raise ValueError(f"unknown fft precision {fft!r}")


def pixel_max_of(arr) -> int:
    """Largest value of the pixel type of ``arr`` (uint8: 255, uint16: 65535): what the reference clips
    flatfield-corrected pixels to (``np.iinfo(dtype).max``, tile_registration.py:2177-2180, stitcher.py:260-267)."""
    torch = _torch()
    if isinstance(arr, torch.Tensor):
        return 255 if arr.dtype == torch.uint8 else 65535
    return 255 if np.asarray(arr).dtype == np.uint8 else 65535


def to_device_u16(arr, device=None):
    """uint16 or uint8 NumPy array (or tensor) -> contiguous uint16 tensor on the GPU.  8-bit tiles (MONO8 cameras;
    the reference takes the dtype of the first image, parameters.py:685) are widened: every kernel works on 16-bit
    pixels, and the arithmetic of registration does not depend on the type (float64 transforms, integer moments)."""
    torch = _torch()
    if isinstance(arr, torch.Tensor):
        t = arr
    else:
        a = np.ascontiguousarray(arr)
        if a.dtype not in (np.uint16, np.uint8):
            raise TypeError(f"tiles must be uint16 or uint8, got {a.dtype}")
        t = torch.from_numpy(a)
    if t.dtype not in (torch.uint16, torch.uint8):
        raise TypeError(f"tiles must be uint16 or uint8, got {t.dtype}")
    if not t.is_cuda:
        t = t.cuda(device)
    if t.dtype == torch.uint8:
        t = t.to(torch.int16).view(torch.uint16)
    return t.contiguous()


# ----------------------------------------------------------------------------
# batched seam
# ----------------------------------------------------------------------------
def register_pairs(tiles, pair_idx, pair_dir, edge_width: int = DEFAULT_EDGE_WIDTH, fft: str = "f64",
                   ctx: Optional[_lib.Context] = None, out=None, status=None, subpixel: bool = False,
                   upsample_factor: int = 100, flatfield=None, peaks10: bool = False, return_candidates: bool = False,
                   window: Optional[str] = None, consensus: bool = False):
    """Register every (first, second) tile pair on the device.

    tiles     [N, H, W] uint16 CUDA tensor (or NumPy array, uploaded)
    pair_idx  [P, 2] int32 (tensor or array-like): second tile is the right
              (dir 0, "left" pair) or lower (dir 1, "top" pair) neighbour
    pair_dir  [P] uint8
    Returns (out [P, 3] float64 CUDA tensor: ncc, y, x; status [P] int8 CUDA tensor).
    Rows of featureless pairs are NaN, as the reference leaves them in its grid.
    ``subpixel`` adds the upsampled-DFT estimate (SubpixelConfig(upsample_factor=100,
    use_subpixel=True) at tile_registration.py:574) as the reference does when
    scikit-image is installed; y and x are then fractional.
    ``flatfield`` ([H, W] float32 or float64): the tiles are raw and the strips are corrected
    on the way, with the results of correcting whole tiles first
    (apply_flatfield_to_image, tile_registration.py:2142-2181).
    ``peaks10``: walk the 16 interpretations of the ten best peaks, the candidate set of the
    reference's consensus stage (generate_translation_candidates(n=10),
    _translation_computation.py:876-994) with its best-NCC fallback (:1112-1118) as the result.
    ``consensus`` (implies ``peaks10``): the consensus stage over those candidates as the reference runs it
    (the deterministic neighbour-count clustering of ransac_translation_estimation,
    _translation_computation.py:1085-1127), its result listed before the subpixel estimate and the
    two-peak discrete search (:1195-1313).
    ``window="hann"``: both strips are multiplied by a separable Hann window before the transforms (own
    extension, off by default: the reference applies none); the candidate NCCs stay on raw pixels.
    ``return_candidates``: also return the table [P, 32 or 160, 3] (y, x, ncc; strip relative) of
    every candidate in evaluation order, NaN ncc where the reference does not score it.
    """
    torch = _torch()
    pixel_max = pixel_max_of(tiles)
    tiles = to_device_u16(tiles)
    if tiles.dim() != 3:
        raise ValueError(f"tiles must be [N, H, W], got {tuple(tiles.shape)}")
    dev = tiles.device
    pi = torch.as_tensor(np.asarray(pair_idx, dtype=np.int32).reshape(-1, 2) if not isinstance(pair_idx, torch.Tensor)
                         else pair_idx).to(device=dev, dtype=torch.int32).contiguous()
    pd_ = torch.as_tensor(np.asarray(pair_dir, dtype=np.uint8).reshape(-1) if not isinstance(pair_dir, torch.Tensor)
                          else pair_dir).to(device=dev, dtype=torch.uint8).contiguous()
    n_pairs = int(pi.shape[0])
    if int(pd_.shape[0]) != n_pairs:
        raise ValueError("pair_idx and pair_dir disagree on the number of pairs")
    if out is None:
        out = torch.empty((n_pairs, 3), dtype=torch.float64, device=dev)
    if status is None:
        status = torch.empty((n_pairs,), dtype=torch.int8, device=dev)
    ctx = ctx or _lib.default_context()
    n, h, w = (int(s) for s in tiles.shape)
    flags = _fft_flag(fft, window)
    if subpixel:
        ctx.set_upsample_factor(upsample_factor)
        flags |= REG_SUBPIXEL
    if peaks10:
        flags |= REG_PEAKS10
    if consensus:
        flags |= REG_CONSENSUS | REG_PEAKS10
        peaks10 = True
    ff, ff_type = None, _lib.FF_NONE
    if flatfield is not None:
        ff = flatfield if isinstance(flatfield, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(flatfield))
        if tuple(ff.shape) != (h, w):
            raise ValueError(f"Flatfield shape {tuple(ff.shape)} doesn't match image shape {(h, w)}")
        if ff.dtype not in (torch.float32, torch.float64):
            ff = ff.to(torch.float64)
        ff = ff.to(dev).contiguous()
        ff_type = _lib.FF_F32 if ff.dtype == torch.float32 else _lib.FF_F64
    if return_candidates:
        table = torch.empty((n_pairs, 160 if peaks10 else 32, 3), dtype=torch.float64, device=dev)
        with ctx.pixel_max(pixel_max):
            check(lib.mistitch_register_pairs_candidates(ctx.handle, ptr(tiles), n, h, w, h * w, ptr(ff), ff_type, ptr(pi),
                                                         ptr(pd_), n_pairs, int(edge_width), flags, ptr(out), ptr(status),
                                                         ptr(table)))
        return out, status, table
    if ff is not None:
        with ctx.pixel_max(pixel_max):   # 8-bit tiles: corrected strips are clipped to 255
            check(lib.mistitch_register_pairs_flatfield(ctx.handle, ptr(tiles), n, h, w, h * w, ptr(ff), ff_type, ptr(pi),
                                                        ptr(pd_), n_pairs, int(edge_width), flags, ptr(out), ptr(status)))
        return out, status
    check(lib.mistitch_register_pairs(ctx.handle, ptr(tiles), n, h, w, h * w, ptr(pi), ptr(pd_), n_pairs,
                                      int(edge_width), flags, ptr(out), ptr(status)))
    return out, status


def refine_pairs(tiles, pair_idx, pair_dir, start_yx, r: float, edge_width: int = DEFAULT_EDGE_WIDTH,
                 ctx: Optional[_lib.Context] = None):
    """Integer hill climb of the strip NCC around ``start_yx`` for every pair: the pair loop of
    ``refine_translations`` (registration/_constrained_refinement.py:140-307) in one device call.

    start_yx [P, 2] integer shifts of the second tile in the first tile's frame (the
    ``{left,top}_{y,x}_second`` columns of the reference's grid); r the search radius.
    Returns (out [P, 3] float64 CUDA tensor: ncc, y, x - NaN ncc and the start shift where the
    reference keeps its start; status [P] int8).
    """
    torch = _torch()
    if not isinstance(r, (int, float)) or r < 0:
        raise ValueError("r must be a non-negative number")
    if not isinstance(edge_width, (int, np.integer)) or edge_width <= 0:
        raise ValueError("edge_width must be a positive integer")
    tiles = to_device_u16(tiles)
    if tiles.dim() != 3:
        raise ValueError(f"tiles must be [N, H, W], got {tuple(tiles.shape)}")
    dev = tiles.device
    pi = torch.as_tensor(np.asarray(pair_idx, dtype=np.int32).reshape(-1, 2)).to(device=dev).contiguous()
    pd_ = torch.as_tensor(np.asarray(pair_dir, dtype=np.uint8).reshape(-1)).to(device=dev).contiguous()
    st = torch.as_tensor(np.asarray(start_yx, dtype=np.int32).reshape(-1, 2)).to(device=dev).contiguous()
    n_pairs = int(pi.shape[0])
    if int(pd_.shape[0]) != n_pairs or int(st.shape[0]) != n_pairs:
        raise ValueError("pair_idx, pair_dir and start_yx disagree on the number of pairs")
    out = torch.empty((n_pairs, 3), dtype=torch.float64, device=dev)
    status = torch.empty((n_pairs,), dtype=torch.int8, device=dev)
    ctx = ctx or _lib.default_context()
    n, h, w = (int(s) for s in tiles.shape)
    check(lib.mistitch_refine_pairs(ctx.handle, ptr(tiles), n, h, w, h * w, ptr(pi), ptr(pd_), n_pairs, int(edge_width),
                                    ptr(st), float(r), ptr(out), ptr(status)))
    return out, status


def compute_single_translation(image1_full: np.ndarray, image2_full: np.ndarray, direction: str,
                               full_sizeY: int, full_sizeX: int, edge_width: int,
                               fft: str = "f64") -> Optional[Tuple[float, float, float]]:
    """(ncc, y, x) for one pair, or None (tile_registration.py:527-586).

    Like the reference, every failure is swallowed into None with a warning.
    """
    try:
        if direction not in ("left", "top"):
            return None
        torch = _torch()
        tiles = to_device_u16(np.stack([np.asarray(image1_full), np.asarray(image2_full)]))
        out, status = register_pairs(tiles, [[0, 1]], [0 if direction == "left" else 1], edge_width, fft)
        st = int(status.cpu()[0])
        if st == PAIR_FEATURELESS:
            return None
        r = out.cpu().numpy()[0]
        return float(r[0]), float(r[1]), float(r[2])
    except Exception as exc:  # noqa: BLE001 - the reference swallows everything here
        logger.warning("Failed to compute translation for direction %s: %s", direction, exc)
        return None


def process_neighborhood_batch(neighborhood_batch: Sequence[Neighborhood], batch_images: Dict[str, np.ndarray],
                               needed_indices_list: List[int], selected_tiff_paths: Sequence,
                               all_filenames: Sequence[str], master_grid, edge_width: int,
                               fft: str = "f64") -> Dict[int, Dict[str, Dict[str, float]]]:
    """Translations for a batch of neighbourhoods in ONE device call (tile_registration.py:310-378).

    Returns ``{center_idx: {"left"|"top": {"ncc", "y", "x"}}}``; a direction key
    is absent when that pair was rejected.
    """
    torch = _torch()
    master_to_batch = {m: b for b, m in enumerate(needed_indices_list)}
    names = [getattr(selected_tiff_paths[i], "name", selected_tiff_paths[i]) for i in needed_indices_list]
    stack = np.ascontiguousarray(np.array([batch_images[nm] for nm in names]))
    tiles = to_device_u16(stack)
    pair_idx, pair_dir, owner = [], [], []
    results: Dict[int, Dict[str, Dict[str, float]]] = {}
    for nb in neighborhood_batch:
        results[nb.center_idx] = {}
        if nb.left_idx is not None:
            pair_idx.append((master_to_batch[nb.left_idx], master_to_batch[nb.center_idx]))
            pair_dir.append(0)
            owner.append((nb.center_idx, "left"))
        if nb.top_idx is not None:
            pair_idx.append((master_to_batch[nb.top_idx], master_to_batch[nb.center_idx]))
            pair_dir.append(1)
            owner.append((nb.center_idx, "top"))
    if not pair_idx:
        return results
    try:
        out, status = register_pairs(tiles, pair_idx, pair_dir, edge_width, fft)
        out = out.cpu().numpy()
        status = status.cpu().numpy()
    except Exception as exc:  # noqa: BLE001
        logger.warning("Failed to compute translations for the batch: %s", exc)
        return results
    for (center, direction), row, st in zip(owner, out, status):
        if st != PAIR_FEATURELESS:
            results[center][direction] = {"ncc": float(row[0]), "y": float(row[1]), "x": float(row[2])}
    return results


# ----------------------------------------------------------------------------
# per-function mirrors
# ----------------------------------------------------------------------------
def _validate_pair(image1, image2) -> None:
    """The checks of validate_image_pair (_translation_computation.py:241-256)."""
    if image1.ndim != 2 or image2.ndim != 2:
        raise ValueError("Images must be 2-dimensional")
    if image1.shape != image2.shape:
        raise ValueError(f"Images must have same shape. Got {image1.shape} and {image2.shape}")


def pcm(image1: np.ndarray, image2: np.ndarray, fft: str = "f64", ctx: Optional[_lib.Context] = None,
        window: Optional[str] = None) -> np.ndarray:
    """Peak correlation matrix of two uint16 strips (_translation_computation.py:282-340)."""
    torch = _torch()
    _validate_pair(image1, image2)
    a, b = to_device_u16(image1), to_device_u16(image2)
    h, w = (int(s) for s in a.shape)
    out = torch.empty((h, w), dtype=torch.float64, device=a.device)
    ctx = ctx or _lib.default_context()
    try:
        check(lib.mistitch_pcm(ctx.handle, ptr(a), w, ptr(b), w, h, w, _fft_flag(fft, window), ptr(out)))
    except _lib.MistitchError as exc:
        raise RuntimeError(f"PCM computation failed: {exc}") from exc
    return out.cpu().numpy()


def pcm_topk(image1: np.ndarray, image2: np.ndarray, k: int = 2, fft: str = "f64",
             ctx: Optional[_lib.Context] = None, window: Optional[str] = None
             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``multi_peak_max(pcm(a, b), max_peaks=k)`` fused on the device
    (_translation_computation.py:342-441): rows, cols (int64), vals (float64) in
    descending order, equal values by descending flat index."""
    torch = _torch()
    _validate_pair(image1, image2)
    if not isinstance(k, (int, np.integer)):
        raise TypeError(f"max_peaks must be an integer, got {type(k)}")
    if k <= 0:
        raise ValueError(f"max_peaks must be positive, got {k}")
    a, b = to_device_u16(image1), to_device_u16(image2)
    h, w = (int(s) for s in a.shape)
    rows = torch.empty((k,), dtype=torch.int32, device=a.device)
    cols = torch.empty((k,), dtype=torch.int32, device=a.device)
    vals = torch.empty((k,), dtype=torch.float64, device=a.device)
    ctx = ctx or _lib.default_context()
    check(lib.mistitch_pcm_topk(ctx.handle, ptr(a), w, ptr(b), w, h, w, _fft_flag(fft, window), int(k),
                                ptr(rows), ptr(cols), ptr(vals)))
    return rows.cpu().numpy().astype(np.int64), cols.cpu().numpy().astype(np.int64), vals.cpu().numpy()


def phase_cross_correlation(reference_image: np.ndarray, moving_image: np.ndarray, upsample_factor: int = 100,
                            fft: str = "f64", ctx: Optional[_lib.Context] = None) -> np.ndarray:
    """Shift (dy, dx) registering ``moving_image`` onto ``reference_image`` by upsampled DFT: what
    ``skimage.registration.phase_cross_correlation(a, b, upsample_factor=u, normalization='phase')``
    returns as its first result, as called at _translation_computation.py:1239-1247."""
    torch = _torch()
    _validate_pair(reference_image, moving_image)
    a, b = to_device_u16(reference_image), to_device_u16(moving_image)
    h, w = (int(s) for s in a.shape)
    out = torch.empty((2,), dtype=torch.float64, device=a.device)
    ctx = ctx or _lib.default_context()
    check(lib.mistitch_subpixel_shift(ctx.handle, ptr(a), w, ptr(b), w, h, w, _fft_flag(fft), int(upsample_factor),
                                      ptr(out)))
    return out.cpu().numpy()


def ncc(image1: np.ndarray, image2: np.ndarray, min_overlap_pixels: int = 25) -> float:
    """Pearson correlation of two uint16 patches (_translation_computation.py:535-616)."""
    torch = _torch()
    if not (hasattr(image1, "shape") and hasattr(image2, "shape")):
        return float("-inf")
    if tuple(image1.shape) != tuple(image2.shape):
        return float("-inf")
    n = int(np.prod(image1.shape))
    if n == 0 or n < min_overlap_pixels:
        return float("-inf")
    a = to_device_u16(np.asarray(image1).reshape(image1.shape[0], -1))
    b = to_device_u16(np.asarray(image2).reshape(image2.shape[0], -1))
    h, w = (int(s) for s in a.shape)
    out = torch.empty((1,), dtype=torch.float64, device=a.device)
    ctx = _lib.default_context()
    check(lib.mistitch_ncc(ctx.handle, ptr(a), w, ptr(b), w, h, w, int(min_overlap_pixels), ptr(out)))
    return float(out.cpu()[0])


def edge_variance(edge: np.ndarray) -> float:
    """Population variance of ``edge[::16, ::16]`` (the statistic of tile_registration.py:497-525)."""
    torch = _torch()
    a = to_device_u16(edge)
    h, w = (int(s) for s in a.shape)
    out = torch.empty((1,), dtype=torch.float64, device=a.device)
    ctx = _lib.default_context()
    check(lib.mistitch_edge_variance(ctx.handle, ptr(a), w, h, w, ptr(out)))
    return float(out.cpu()[0])


def is_edge_featureless(edge: np.ndarray, variance_threshold: float = 2e3) -> bool:
    """tile_registration.py:497-525."""
    if edge[::16, ::16].size == 0:
        return True
    return edge_variance(np.ascontiguousarray(edge)) < variance_threshold


def select_best_registration_channel(acquisition, sample_fovs: Sequence[int],
                                     edge_width: int = DEFAULT_EDGE_WIDTH) -> int:
    """Channel whose right-edge strips have the largest mean 16-strided variance over the
    sampled fields of view: the decision of select_best_registration_channel
    (tile_registration.py:430-485) for tiles already in memory.  File discovery and the random
    choice of fields stay with the caller; ``acquisition`` is [C, N, H, W] uint16 (tensor or
    array), ``sample_fovs`` the tile indices to look at.  Ties go to the lowest channel, as
    Python's max over the reference's insertion-ordered dict does."""
    torch = _torch()
    acq = acquisition
    if not isinstance(acq, torch.Tensor):
        acq = np.asarray(acq)
    if acq.ndim != 4:
        raise ValueError("acquisition must be [C, N, H, W]")
    if len(sample_fovs) == 0:
        return 0
    n_ch, _, h, w = (int(v) for v in acq.shape)
    cw = min(int(edge_width), w)
    ctx = _lib.default_context()
    means = []
    for c in range(n_ch):
        tiles = to_device_u16(acq[c])
        out = torch.empty((len(sample_fovs),), dtype=torch.float64, device=tiles.device)
        for i, t in enumerate(sample_fovs):
            strip = tiles[int(t), :, w - cw:]
            check(lib.mistitch_edge_variance(ctx.handle, _lib.ptr_of(strip), w, h, cw, _lib.ptr_at(out, i)))
        means.append(float(out.mean().cpu()))
    return int(np.argmax(means))
