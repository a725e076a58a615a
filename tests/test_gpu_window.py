"""The optional Hann window on the strips (flag MISTITCH_REG_WINDOW_HANN), needs an MI355X.

Own extension, PARITY UNPINNED by construction: the reference applies no window
(_translation_computation.py:308-313).  Checked: flag off = today's results (every other test);
flag on = the NumPy statement ``pcm(a * w, b * w)`` with ``w = outer(hanning(h), hanning(w))`` within
1e-11; and the case the window is for: a smooth scene whose unwindowed correlation peaks at spectral
leakage near (0, 0) registers with the window, on the device as in the oracle."""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import registration as oreg

pytestmark = pytest.mark.gpu


def _smooth_scene(seed, h, w, sigma2):
    rng = np.random.default_rng(seed)
    big = rng.random((h, w))
    k = np.fft.fftfreq(h)[:, None] ** 2 + np.fft.fftfreq(w)[None, :] ** 2
    low = np.fft.ifft2(np.fft.fft2(big) * np.exp(-k * sigma2)).real
    return (1000 + 60000 * (low - low.min()) / (low.max() - low.min())).astype(np.uint16)


@pytest.mark.parametrize("shape", [(64, 256), (256, 64), (195, 512), (60, 300), (97, 101), (256, 2048)])
def test_windowed_pcm_matches_numpy_statement(shape):
    import torch
    from image_stitcher_amd import registration as reg
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    h, w = shape
    big = synth.make_scene(700 + h, h + 16, w + 16)
    a = np.ascontiguousarray(big[8:8 + h, 8:8 + w])
    b = np.ascontiguousarray(big[8 + 2:8 + 2 + h, 8 - 3:8 - 3 + w])
    want = oreg.pcm(a, b, window="hann")
    got = reg.pcm(a, b, window="hann")
    assert np.abs(got - want).max() < 1e-11
    assert np.abs(got - oreg.pcm(a, b)).max() > 1e-6          # it does change the matrix
    assert np.array_equal(reg.pcm(a, b), reg.pcm(a, b, window=None))
    rows, cols, vals = reg.pcm_topk(a, b, 2, window="hann")
    er, ec, ev = oreg.multi_peak_max(want)
    assert (rows[0], cols[0]) == (er[0], ec[0]) and vals[0] == pytest.approx(ev[0], abs=1e-11)
    if (h & (h - 1)) == 0 and (w & (w - 1)) == 0:             # complex64 path exists for power-of-two strips only
        assert np.abs(reg.pcm(a, b, fft="f32", window="hann") - want).max() < 1e-4


def test_smooth_scene_registers_with_the_window():
    """A scene without fine texture: unwindowed, the reference's phase correlation locks onto leakage and
    returns a wrong shift (device and oracle agree on that too); windowed, the planted shifts come back."""
    from image_stitcher_amd import registration as reg
    tile, edge = 512, 128
    rows, cols, jitter = 2, 3, 3
    pitch = tile - 51
    scene = _smooth_scene(5, (rows - 1) * pitch + tile + 2 * jitter, (cols - 1) * pitch + tile + 2 * jitter, 3000.0)
    g = synth.make_grid(9, rows, cols, tile, tile, overlap=0.10, jitter=jitter, scene=scene)
    plain, st0 = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, edge)
    win, st1 = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, edge, window="hann")
    ref_plain, ref_st0 = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, edge, polish=False)
    ref_win, ref_st1 = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, edge, polish=False, window="hann")
    plain, win = plain.cpu().numpy(), win.cpu().numpy()
    assert np.array_equal(st1.cpu().numpy(), ref_st1) and (ref_st1 == 0).all()
    assert np.array_equal(win[:, 1:], ref_win[:, 1:])
    assert np.allclose(win[:, 0], ref_win[:, 0], rtol=0, atol=1e-12)
    assert np.array_equal(win[:, 1:].astype(np.int64), g.expected_shift)          # the window's purpose
    ok = ref_st0 == 0
    assert np.array_equal(st0.cpu().numpy(), ref_st0)
    wrong = (ref_plain[ok, 1:].astype(np.int64) != g.expected_shift[ok]).any(axis=1)
    assert wrong.any()                                                             # the reference's known weakness
    # (which wrong shift comes out depends on the order of near-equal leakage peaks: not compared)
    assert (plain[ok, 1:].astype(np.int64) != g.expected_shift[ok]).any()


def test_window_with_subpixel_and_any_size():
    from image_stitcher_amd import registration as reg
    g = synth.make_grid(41, 2, 3, 300, 420, overlap=0.15, jitter=2)
    out, st = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 45, window="hann", subpixel=True)
    ref, ref_st = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 45, polish=False, window="hann")
    out = out.cpu().numpy()
    assert np.array_equal(st.cpu().numpy(), ref_st)
    # the upsampled-DFT estimate of a windowed pair sits within a few hundredths of the integer shift
    assert np.array_equal(np.rint(out[:, 1:]), ref[:, 1:]) and np.abs(out[:, 1:] - ref[:, 1:]).max() <= 0.05
