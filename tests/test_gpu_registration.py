"""HIP registration path against the oracle and the reference's golden vectors (needs an MI355X)."""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import registration as oreg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from image_stitcher_amd import registration
    return registration


def _grid(spec):
    return synth.make_grid(spec["seed"], spec["rows"], spec["cols"], spec["tile_h"], spec["tile_w"],
                           overlap=spec["overlap"], jitter=spec["jitter"])


@pytest.mark.parametrize("fft", ["f64", "f32"])
def test_grids_match_reference_golden(reg, golden, fft):
    """Every golden grid, the ones with sides that are not powers of two included
    (edge width 195, 300x420 tiles): those run the chirp-z transforms."""
    for spec in golden["grids"]:
        g = _grid(spec)
        out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, spec["edge_width"], fft=fft)
        out, status = out.cpu().numpy(), status.cpu().numpy()
        assert (status == 0).all(), spec["name"]
        exp = np.array(spec["results"], dtype=np.float64)
        # integer shifts: bit exact; NCC: 1e-12
        assert np.array_equal(out[:, 1:], exp[:, 1:]), spec["name"]
        assert np.allclose(out[:, 0], exp[:, 0], rtol=0, atol=1e-12), spec["name"]
        assert np.array_equal(out[:, 1:].astype(np.int64), g.expected_shift)


@pytest.mark.parametrize("grid", [(2, 1, 320, 4096), (1, 2, 4096, 320)])
@pytest.mark.parametrize("seed", [5, 6, 7, 8])
def test_long_sides_of_4096_match_the_oracle(reg, grid, seed):
    """Strips whose long side is 4096: lines of the largest transform, and candidate sums whose rows take two
    workgroups side by side (a column lag then wraps from one workgroup's part of the row into the other's).  The
    seeds plant odd and even shifts along the long side."""
    rows, cols, th, tw = grid
    g = synth.make_grid(seed, rows, cols, th, tw, overlap=0.2, jitter=5)
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 64)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    exp, exp_status = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 64, polish=False)
    assert np.array_equal(status, exp_status)
    assert np.array_equal(out[:, 1:], exp[:, 1:])
    assert np.allclose(out[:, 0], exp[:, 0], rtol=0, atol=1e-9)
    assert np.array_equal(out[:, 1:].astype(np.int64), g.expected_shift)


def test_unsupported_shape_is_an_error(reg):
    """Sides beyond 4096 are refused with an error, never computed wrongly."""
    tiles = np.zeros((2, 16, 4100), dtype=np.uint16)
    with pytest.raises(Exception):
        reg.register_pairs(tiles, [[0, 1]], [1], 16)


@pytest.mark.parametrize("shape", [(50, 300), (195, 512), (97, 101), (64, 300), (30, 1000), (7, 9), (130, 64),
                                   (100, 256), (45, 1024), (2048, 195), (3, 64), (9, 8),
                                   # short sides that factor into radices up to 13 (mixed-radix passes) ...
                                   (77, 128), (91, 256), (208, 256), (176, 512), (6, 32), (768, 1024), (1000, 1024),
                                   (169, 256), (14, 16),
                                   # ... and ones with a larger prime factor (chirp-z)
                                   (34, 64), (257, 512)])
def test_pcm_of_any_size(reg, shape):
    """PCM and its ranking for strips whose sides are not powers of two (primes, odd, tall)."""
    h, w = shape
    big = synth.make_scene(900 + h + w, h + 16, w + 16)
    a = np.ascontiguousarray(big[8:8 + h, 8:8 + w])
    b = np.ascontiguousarray(big[8 + 2:8 + 2 + h, 8 - 3:8 - 3 + w])
    ref = oreg.pcm(a, b)
    got = reg.pcm(a, b)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-11
    # such strips are transformed in complex128 whatever precision is asked for
    assert np.array_equal(reg.pcm(a, b, fft="f32"), got)
    k = 4
    rows, cols, vals = reg.pcm_topk(a, b, k)
    er, ec, ev = oreg.multi_peak_max(ref)
    for i in range(k):
        assert vals[i] == pytest.approx(ev[i], abs=1e-11)
        if i + 1 < len(ev) and abs(ev[i] - ev[i + 1]) > 1e-9 and (i == 0 or abs(ev[i - 1] - ev[i]) > 1e-9):
            assert (rows[i], cols[i]) == (er[i], ec[i])


@pytest.mark.parametrize("shape", [(8, 64), (256, 2048), (4, 24), (12, 64), (34, 64), (6, 40), (195, 256)])
@pytest.mark.parametrize("which", ["both_nyquist", "long_nyquist", "short_nyquist"])
def test_pcm_with_an_exactly_cancelling_self_conjugate_frequency(reg, shape, which):
    """A strip whose alternating pixel sum is exactly zero has an exactly zero spectrum value at that self-conjugate
    frequency: NumPy's transform delivers the zero and the normalised cross-power value there is 0; a transform that
    leaves a rounding residue instead would normalise it to magnitude 1 (1 / (h w) of every PCM value, enough to
    reorder close peaks).  Shapes where NumPy's own sum is not exact (its radix-3 and larger passes) are skipped."""
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(1000, 60000, size=(h, w)).astype(np.int64)
    b = np.roll(a, (1, -2), axis=(0, 1)) + rng.integers(0, 50, size=(h, w))
    sy = (-1) ** np.arange(h)[:, None] if which != "long_nyquist" else np.ones((h, 1), dtype=np.int64)
    sx = (-1) ** np.arange(w)[None, :] if which != "short_nyquist" else np.ones((1, w), dtype=np.int64)
    sign = sy * sx
    # move the signed sum of `a` to zero, a few pixels at a time, staying inside uint16
    residue = int((sign * a).sum())
    flat, sflat = a.reshape(-1), np.broadcast_to(sign, a.shape).reshape(-1)
    for i in range(flat.size):
        if residue == 0:
            break
        want_down = (residue > 0) == (sflat[i] > 0)  # lowering a +pixel or raising a -pixel lowers the sum
        room = int(flat[i]) if want_down else 65535 - int(flat[i])
        step = min(abs(residue), room, 20000)
        flat[i] += -step if want_down else step
        residue += -step if residue > 0 else step
    assert int((sign * a).sum()) == 0 and a.min() >= 0 and a.max() <= 65535
    ky, kx = (h // 2 if which != "long_nyquist" else 0), (w // 2 if which != "short_nyquist" else 0)
    if (h % 2 and ky) or (w % 2 and kx) or np.fft.fft2(a.astype(np.float64))[ky, kx] != 0:
        pytest.skip("NumPy's own value at this frequency is a rounding residue for this length: nothing to agree with")
    a16, b16 = a.astype(np.uint16), b.astype(np.uint16)
    ref = oreg.pcm(a16, b16)
    got = reg.pcm(a16, b16)
    assert np.abs(got - ref).max() < 1e-10, (shape, which, np.abs(got - ref).max(), 1.0 / (h * w))


def test_matches_oracle_on_fresh_seeds(reg):
    for seed, (rows, cols, th, tw, ew) in enumerate([(3, 4, 256, 256, 32), (2, 5, 128, 512, 64), (4, 2, 512, 128, 64)]):
        g = synth.make_grid(100 + seed, rows, cols, th, tw, overlap=0.15, jitter=4)
        ref, ref_status = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, ew, polish=False)
        out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, ew)
        out, status = out.cpu().numpy(), status.cpu().numpy()
        assert np.array_equal(status != 0, ref_status != 0)
        assert np.array_equal(out[:, 1:], ref[:, 1:])
        assert np.allclose(out[:, 0], ref[:, 0], rtol=0, atol=1e-12)


def test_featureless_pairs(reg, golden):
    f = golden["featureless"]
    rng = np.random.default_rng(f["rng_seed"])
    flat = (5000 + 20 * rng.random((512, 512))).astype(np.uint16)
    tex = synth.make_scene(22, 512, 512)
    tiles = np.stack([flat, tex])
    # pairs: (flat, tex), (tex, flat), (tex, tex)
    out, status = reg.register_pairs(tiles, [[0, 1], [1, 0], [1, 1]], [0, 0, 0], 64)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert status[0] == 1 and status[1] == 1 and status[2] == 0
    assert np.isnan(out[0]).all() and np.isnan(out[1]).all()
    exp = f["pairs"]["both_textured"]
    assert out[2, 1] == exp[1] and out[2, 2] == exp[2]
    assert out[2, 0] == pytest.approx(exp[0], abs=1e-12)
    # mirror of the per-pair function: None for rejected pairs
    assert reg.compute_single_translation(flat, tex, "left", 512, 512, 64) is None
    r = reg.compute_single_translation(tex, tex, "left", 512, 512, 64)
    assert r[1:] == (exp[1], exp[2])
    for probe in f["probes"]:
        t = (5000 + probe["amp"] * rng.random((512, 64))).astype(np.uint16)
        assert reg.is_edge_featureless(t) == probe["featureless"]
        assert reg.edge_variance(t) == pytest.approx(probe["var_f32"], rel=1e-5)


def test_pcm_and_peak_ranking(reg, golden):
    for s in golden["peaks"]["strips"]:
        h, w = s["shape"]
        big = synth.make_scene(s["seed"], h + 64, w + 64)
        a = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
        b = np.ascontiguousarray(big[32 + s["shift"][0]:32 + s["shift"][0] + h, 32 + s["shift"][1]:32 + s["shift"][1] + w])
        ref = oreg.pcm(a, b)
        got = reg.pcm(a, b)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-12
        assert float(got.sum()) == pytest.approx(s["pcm_sum"], rel=1e-9)
        rows, cols, vals = reg.pcm_topk(a, b, 16)
        for i, (er, ec, ev) in enumerate(s["top"]):
            assert (rows[i], cols[i]) == (er, ec), (s["seed"], i)
            assert vals[i] == pytest.approx(ev, rel=1e-9, abs=1e-13)
        got32 = reg.pcm(a, b, fft="f32")
        assert np.abs(got32 - ref).max() < 1e-5


def test_ncc_table(reg, golden):
    n = golden["ncc"]
    rng = np.random.default_rng(n["seed"])
    a = rng.integers(0, 65535, size=tuple(n["shape"])).astype(np.uint16)
    b = rng.integers(0, 65535, size=tuple(n["shape"])).astype(np.uint16)
    const = np.full(tuple(n["shape"]), 1234, dtype=np.uint16)
    got = {
        "random": reg.ncc(a, b), "self": reg.ncc(a, a),
        "negated": reg.ncc(a, (65535 - a).astype(np.uint16)),
        "size_24": reg.ncc(a[:4, :6], b[:4, :6]), "size_25": reg.ncc(a[:5, :5], b[:5, :5]),
        "both_constant": reg.ncc(const, const), "one_constant": reg.ncc(a, const),
        "shape_mismatch": reg.ncc(a, b[:, :29]), "empty": reg.ncc(a[:0], b[:0]),
    }
    for k, exp in n["values"].items():
        if isinstance(exp, str):
            assert got[k] == float(exp), k
        else:
            assert got[k] == pytest.approx(exp, abs=1e-13), k


def test_candidate_winner(reg, golden):
    for s in golden["candidates"]:
        h, w = s["shape"]
        dy, dx = s["shift"]
        big = synth.make_scene(s["seed"], h + 64, w + 64)
        a = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
        b = np.ascontiguousarray(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w])
        rows, cols, _ = reg.pcm_topk(a, b, 2)
        assert [[int(rows[0]), int(cols[0])], [int(rows[1]), int(cols[1])]] == s["peaks"]


def test_process_neighborhood_batch_mirror(reg, golden):
    spec = golden["grids"][0]
    g = _grid(spec)

    class P:  # stands in for pathlib.Path: only .name is used
        def __init__(self, name):
            self.name = name

    paths = [P(f"t{i}.tiff") for i in range(len(g.tiles))]
    images = {p.name: g.tiles[i] for i, p in enumerate(paths)}
    hoods = []
    for r in range(g.rows):
        for c in range(g.cols):
            t = r * g.cols + c
            if c > 0 or r > 0:
                hoods.append(reg.Neighborhood(t, t - 1 if c > 0 else None, t - g.cols if r > 0 else None))
    res = reg.process_neighborhood_batch(hoods, images, list(range(len(paths))), paths, [p.name for p in paths],
                                         None, spec["edge_width"])
    exp = oreg.neighborhood_results(g.pair_idx, g.pair_dir, np.array(spec["results"]),
                                    np.zeros(len(g.pair_idx), dtype=np.int8))
    assert res.keys() == exp.keys()
    for k in exp:
        assert res[k].keys() == exp[k].keys()
        for d in exp[k]:
            assert res[k][d]["y"] == exp[k][d]["y"] and res[k][d]["x"] == exp[k][d]["x"]
            assert res[k][d]["ncc"] == pytest.approx(exp[k][d]["ncc"], abs=1e-12)


def _subpixel_pair(seed, h, w, dy, dx):
    """Two uint16 strips cut from one smooth scene that was resampled with a fractional offset."""
    rng = np.random.default_rng(seed)
    big = rng.random((h + 32, w + 32))
    k = np.fft.fftfreq(h + 32)[:, None] ** 2 + np.fft.fftfreq(w + 32)[None, :] ** 2
    low = np.fft.ifft2(np.fft.fft2(big) * np.exp(-k * 4.0)).real  # mild low-pass, still broadband (phase correlation whitens the spectrum)
    fy, fx = np.fft.fftfreq(h + 32)[:, None], np.fft.fftfreq(w + 32)[None, :]
    moved = np.fft.ifft2(np.fft.fft2(low) * np.exp(-2j * np.pi * (fy * dy + fx * dx))).real
    def to_u16(img):
        img = (img - low.min()) / (low.max() - low.min())
        return np.clip(1000 + 50000 * img, 0, 65535).astype(np.uint16)
    return (np.ascontiguousarray(to_u16(low)[16:16 + h, 16:16 + w]),
            np.ascontiguousarray(to_u16(moved)[16:16 + h, 16:16 + w]))


@pytest.mark.parametrize("fft", ["f64", "f32"])
def test_subpixel_shift_matches_upsampled_dft_oracle(reg, fft):
    """Within 0.01 px of the NumPy restatement of scikit-image's algorithm (parity unpinned:
    scikit-image itself is not installed here), and close to the planted fractional shift."""
    from oracle import subpixel as osp
    for seed, (h, w, dy, dx) in enumerate([(64, 256, 0.37, -1.24), (256, 64, -2.5, 3.75), (128, 512, 0.0, 0.49),
                                           (32, 1024, 3.21, -7.68)]):
        a, b = _subpixel_pair(40 + seed, h, w, dy, dx)
        want = osp.phase_cross_correlation(a, b, 100)
        got = reg.phase_cross_correlation(a, b, 100, fft=fft)
        assert np.abs(got - want).max() <= 0.01 + 1e-9, (h, w, got, want)
        assert np.abs(got - np.array([-dy, -dx])).max() < 0.1, (got, dy, dx)
    a, b = _subpixel_pair(77, 64, 256, 1.3, 0.2)
    assert np.abs(reg.phase_cross_correlation(a, b, 10) - osp.phase_cross_correlation(a, b, 10)).max() <= 0.1 + 1e-9


def test_subpixel_shift_of_any_size(reg):
    """The same for strips whose sides are not powers of two (even, odd, prime, tall): the chirp-z path."""
    from oracle import subpixel as osp
    for seed, (h, w, dy, dx) in enumerate([(60, 300, 0.37, -1.24), (195, 260, -2.5, 3.75), (97, 101, 0.0, 0.49),
                                           (33, 520, 1.21, -4.68), (300, 50, -0.8, 0.15), (64, 300, 0.6, 2.2),
                                           (60, 256, 0.3, 1.7), (45, 512, -1.4, 0.85), (512, 75, 2.6, -0.35)]):
        a, b = _subpixel_pair(60 + seed, h, w, dy, dx)
        want = osp.phase_cross_correlation(a, b, 100)
        got = reg.phase_cross_correlation(a, b, 100)
        assert np.abs(got - want).max() <= 0.01 + 1e-9, (h, w, got, want)
        assert np.abs(got - np.array([-dy, -dx])).max() < 0.1, (got, dy, dx)


def test_subpixel_shift_equals_the_direct_sums_on_the_grid(reg):
    """The upsampled correlation is evaluated at 32 x 32 Chebyshev nodes and interpolated to the grid (exact to
    rounding for a function of its bandwidth), so the argmax on the 1 / upsample grid equals the one of the direct
    matrix products in the NumPy restatement - not merely within 0.01 px.  Short sides up to 256 and up to 512 keep
    the first-stage matrix in registers (two instantiations), longer ones take the rotor kernel.  Small strips with
    an even long side that is not a power of two are the sensitive ones: their Nyquist column carries 1 / L of the
    spectrum and counts as the negative frequency."""
    from oracle import subpixel as osp
    cases = [(64, 256, 0.37, -1.24, 100), (256, 2048, -1.63, 2.18, 100), (128, 512, 0.05, 0.49, 37),
             (300, 600, 1.21, -2.4, 100), (384, 512, -0.71, 0.33, 100), (600, 700, 0.45, 1.9, 100),
             (45, 300, -1.4, 0.85, 10), (2, 64, 0.0, 1.3, 100), (6, 40, 0.4, -0.6, 100), (8, 24, 0.4, -0.6, 100),
             (33, 101, -0.3, 0.7, 100), (40, 8, 0.6, 0.2, 100), (12, 36, 0.4, -0.6, 64)]
    for seed, (h, w, dy, dx, uf) in enumerate(cases):
        a, b = _subpixel_pair(300 + seed, h, w, dy, dx)
        want = osp.phase_cross_correlation(a, b, uf)
        got = reg.phase_cross_correlation(a, b, uf)
        assert np.abs(got - want).max() < 1e-9, (h, w, uf, got, want)


def test_register_pairs_with_subpixel_candidate_any_size(reg):
    """Subpixel candidate on a grid whose strips are not powers of two (300 x 420 tiles, edge 45)."""
    from oracle import subpixel as osp
    g = synth.make_grid(33, 2, 3, 300, 420, overlap=0.15, jitter=2)
    ref, ref_status, _ = osp.register_pairs_subpixel(g.tiles, g.pair_idx, g.pair_dir, 45)
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 45, subpixel=True)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert np.array_equal(status != 0, ref_status != 0)
    ok = status == 0
    assert np.abs(out[ok, 1:] - ref[ok, 1:]).max() <= 0.01 + 1e-9
    assert np.allclose(out[ok, 0], ref[ok, 0], rtol=0, atol=1e-12)


def test_register_pairs_with_subpixel_candidate(reg):
    """The subpixel estimate competes as the reference's Method 2 does (listed first, wins ties)."""
    from oracle import subpixel as osp
    g = synth.make_grid(31, 2, 3, 512, 512, overlap=0.12, jitter=3)
    ref, ref_status, _ = osp.register_pairs_subpixel(g.tiles, g.pair_idx, g.pair_dir, 64)
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 64, subpixel=True)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert np.array_equal(status != 0, ref_status != 0)
    assert np.abs(out[:, 1:] - ref[:, 1:]).max() <= 0.01 + 1e-9
    assert np.allclose(out[:, 0], ref[:, 0], rtol=0, atol=1e-12)
    # without the flag the integer result of the discrete search is unchanged
    plain, _ = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 64)
    assert np.array_equal(np.rint(out[:, 1:]), plain.cpu().numpy()[:, 1:])


def test_refine_pairs_matches_reference_golden(reg):
    """The hill climb against vectors produced by the reference's refine_translations."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refine_golden.json")) as f:
        cases = json.load(f)["cases"]
    for spec in cases:
        g = _grid(spec)
        out, status = reg.refine_pairs(g.tiles, g.pair_idx, g.pair_dir, spec["start"], spec["r"], spec["edge_width"])
        out, status = out.cpu().numpy(), status.cpu().numpy()
        exp = np.array(spec["results"], dtype=np.float64)
        assert (status == 0).all()
        assert np.array_equal(out[:, 1:], exp[:, 1:]), spec["name"]
        assert np.allclose(out[:, 0], exp[:, 0], rtol=0, atol=1e-12, equal_nan=True), spec["name"]


def test_refine_pairs_climbs_and_falls_back(reg):
    """Smooth scenes give the climb a slope to follow; a start outside the tile keeps its value with NaN."""
    from oracle import refine as oref
    rng = np.random.default_rng(9)
    big = rng.random((300, 700))
    k = np.fft.fftfreq(300)[:, None] ** 2 + np.fft.fftfreq(700)[None, :] ** 2
    smooth = np.fft.ifft2(np.fft.fft2(big) * np.exp(-k * 600.0)).real
    smooth = (1000 + 60000 * (smooth - smooth.min()) / (smooth.max() - smooth.min())).astype(np.uint16)
    tiles = np.ascontiguousarray(np.stack([smooth[20:276, 10:266], smooth[23:279, 230:486]]))
    pair_idx, pair_dir = [[0, 1], [0, 1]], [0, 0]
    start = [[0, 214], [900, 5]]
    want = oref.refine_pairs(tiles, pair_idx, pair_dir, start, 12, 64)
    out, _ = reg.refine_pairs(tiles, pair_idx, pair_dir, start, 12, 64)
    out = out.cpu().numpy()
    assert np.array_equal(out[:, 1:], want[:, 1:])
    assert np.allclose(out[:, 0], want[:, 0], rtol=0, atol=1e-12, equal_nan=True)
    assert (out[0, 1], out[0, 2]) == (3.0, 220.0) and np.isnan(out[1, 0])


def test_select_best_registration_channel(reg):
    rng = np.random.default_rng(3)
    acq = np.stack([np.stack([(5000 + amp * rng.random((128, 160))).astype(np.uint16) for _ in range(4)])
                    for amp in (300.0, 4000.0, 900.0)])
    fovs = [0, 2, 3]
    want = int(np.argmax([np.mean([np.var(acq[c, t][:, -64:][::16, ::16].astype(np.float32)) for t in fovs])
                          for c in range(3)]))
    assert reg.select_best_registration_channel(acq, fovs, 64) == want == 1


@pytest.mark.parametrize("shape,edge,dt", [((256, 256), 64, np.float32), ((128, 320), 64, np.float64),
                                           ((200, 300), 50, np.float32), ((256, 320), 45, np.float32),
                                           ((192, 256), 100, np.float64)])
def test_register_pairs_with_fused_flatfield(reg, shape, edge, dt):
    """Strips corrected on the way == whole tiles corrected first (oracle), featureless test included."""
    from oracle import fusion as ofu
    h, w = shape
    g = synth.make_grid(70 + h, 2, 3, h, w, overlap=0.15, jitter=3)
    ff = synth.make_flatfield(h, w, dtype=np.dtype(dt), amplitude=0.5)
    tiles = g.tiles.copy()
    tiles[0] = np.rint(3000.0 * ff.astype(np.float64)).astype(np.uint16)  # flat once corrected: featureless
    corrected = np.stack([ofu.apply_flatfield(t, ff) for t in tiles])
    ref, ref_status = oreg.register_pairs(corrected, g.pair_idx, g.pair_dir, edge, polish=False)
    out, status = reg.register_pairs(tiles, g.pair_idx, g.pair_dir, edge, flatfield=ff)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert np.array_equal(status != 0, ref_status != 0)
    assert (status != 0).any() and (status == 0).any()
    ok = status == 0
    assert np.array_equal(out[ok, 1:], ref[ok, 1:])
    assert np.allclose(out[ok, 0], ref[ok, 0], rtol=0, atol=1e-12)


def test_passes_smaller_than_the_call_give_the_same_rows(reg):
    """More pairs than one pass holds: all strips are staged up front and the candidate overlaps
    summed after the last pass.  Rows must equal those of a single pass, flatfield or not."""
    import torch
    from image_stitcher_amd import _lib
    g = synth.make_grid(123, 3, 4, 256, 256, overlap=0.125, jitter=3)
    ff = synth.make_flatfield(256, 256, amplitude=0.3)
    small = _lib.Context(0, torch.cuda.current_stream().cuda_stream, pair_chunk=5)
    by_pass = _lib.Context(0, torch.cuda.current_stream().cuda_stream, pair_chunk=5)
    by_pass.set_staging_limit(0)    # strips staged and candidates summed pass by pass
    for flat in (None, ff):
        for sub in (False, True):
            want, want_st = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, flatfield=flat, subpixel=sub)
            for ctx in (small, by_pass):
                got, got_st = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, flatfield=flat, subpixel=sub, ctx=ctx)
                assert torch.equal(want_st, got_st)
                assert torch.equal(want, got)
    assert np.array_equal(want.cpu().numpy()[:, 1:].round().astype(np.int64), g.expected_shift)


def test_tall_strips_take_the_wide_accumulators(reg):
    """Strips of 1024 rows: a lane sums more steps than 32-bit wave totals can hold, so the
    64-bit reduction runs; ragged widths exercise the columns beyond the last group of eight."""
    g = synth.make_grid(321, 2, 2, 1024, 200, overlap=0.12, jitter=3)
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 1024)
    exp, exp_st = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 1024)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert np.array_equal(status, exp_st)
    assert np.array_equal(out[:, 1:], exp[:, 1:])
    assert np.allclose(out[:, 0], exp[:, 0], rtol=0, atol=1e-12)
    rng = np.random.default_rng(9)
    a = rng.integers(0, 65536, size=(1500, 701)).astype(np.uint16)
    b = (a // 2 + rng.integers(0, 30000, size=a.shape)).astype(np.uint16)
    assert abs(reg.ncc(a, b) - oreg.ncc(a, b)) < 1e-12
    assert abs(reg.ncc(a[:, :5], b[:, :5]) - oreg.ncc(a[:, :5], b[:, :5])) < 1e-12


def test_single_pair_calls_on_a_fresh_context(reg):
    """pcm / pcm_topk / phase_cross_correlation stage BOTH strips of a tall pair (and of a pair
    with two different row pitches) in the context's staging buffer: on a context that no batched
    call has grown before, that buffer has its minimum size."""
    import torch
    from image_stitcher_amd import _lib
    from image_stitcher_amd._lib import check, lib, ptr_of
    from oracle import subpixel as osp
    h, w = 2048, 195
    big = synth.make_scene(612, h + 16, w + 16)
    a = np.ascontiguousarray(big[8:8 + h, 8:8 + w])
    b = np.ascontiguousarray(big[8 + 2:8 + 2 + h, 8 - 3:8 - 3 + w])
    ref = oreg.pcm(a, b)
    for call in ("pcm", "topk", "subpixel"):
        fresh = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
        # a canary right behind whatever the library allocates next is not possible from here; the
        # results themselves and a clean device afterwards are what is checked
        if call == "pcm":
            assert np.abs(reg.pcm(a, b, ctx=fresh) - ref).max() < 1e-11
        elif call == "topk":
            rows, cols, vals = reg.pcm_topk(a, b, 2, ctx=fresh)
            er, ec, ev = oreg.multi_peak_max(ref)
            assert (rows[0], cols[0]) == (er[0], ec[0]) and vals[0] == pytest.approx(ev[0], abs=1e-11)
        else:
            got = reg.phase_cross_correlation(a, b, 100, ctx=fresh)
            assert np.abs(got - osp.phase_cross_correlation(a, b, 100)).max() <= 0.01 + 1e-9
        fresh.synchronize()
        fresh.close()
    # two different row pitches: both strips are re-pitched into the staging buffer
    hh, ww = 64, 256
    wide1 = torch.from_numpy(synth.make_scene(613, hh + 8, ww + 40).view(np.int16)).cuda()
    wide2 = torch.from_numpy(synth.make_scene(614, hh + 8, ww + 104).view(np.int16)).cuda()
    s1, s2 = wide1[:hh, 8:8 + ww], wide2[:hh, 16:16 + ww]
    fresh = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    out = torch.empty((hh, ww), dtype=torch.float64, device="cuda")
    check(lib.mistitch_pcm(fresh.handle, ptr_of(s1), ww + 40, ptr_of(s2), ww + 104, hh, ww, 0, ptr_of(out)))
    want = oreg.pcm(s1.cpu().numpy().view(np.uint16), s2.cpu().numpy().view(np.uint16))
    assert np.abs(out.cpu().numpy() - want).max() < 1e-11
    fresh.close()


def test_ncc_threshold_and_saturated_patches(reg):
    """min_overlap_pixels below the default is honoured (_translation_computation.py:567-569), and
    patches of almost only 0xFFFF pixels, long enough that a lane's 32-bit byte-plane sums would
    wrap without the periodic flush into 64 bits, still give the exact correlation."""
    rng = np.random.default_rng(77)
    a = rng.integers(0, 65535, size=(4, 4)).astype(np.uint16)
    b = rng.integers(0, 65535, size=(4, 4)).astype(np.uint16)
    assert reg.ncc(a, b) == float("-inf")                          # 16 < 25
    assert reg.ncc(a, b, min_overlap_pixels=4) == pytest.approx(oreg.ncc(a, b, min_overlap_pixels=4), abs=1e-13)
    assert reg.ncc(a, b, min_overlap_pixels=17) == float("-inf")
    for shape in ((1, 12_000_000), (3, 9_000_001)):
        x = np.full(shape, 0xFFFF, dtype=np.uint16)
        y = np.full(shape, 0xFFFF, dtype=np.uint16)
        idx = rng.integers(0, shape[1], size=4000)
        x[0, idx] = rng.integers(0, 65535, size=4000)
        y[0, idx[:3000]] = x[0, idx[:3000]] // 2
        y[-1, rng.integers(0, shape[1], size=500)] = 7
        # exact reference from integer moments (Python integers)
        xi, yi = x.astype(np.int64).ravel(), y.astype(np.int64).ravel()
        n = xi.size
        sa, sb = int(xi.sum()), int(yi.sum())
        saa, sbb, sab = int((xi * xi).sum()), int((yi * yi).sum()), int((xi * yi).sum())
        want = (n * sab - sa * sb) / (float(n * saa - sa * sa) ** 0.5 * float(n * sbb - sb * sb) ** 0.5)
        assert reg.ncc(x, y) == pytest.approx(want, abs=1e-12), shape
