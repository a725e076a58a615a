"""The NumPy oracle reproduces what the reference produced (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import registration as oreg


def _grid(spec):
    return synth.make_grid(spec["seed"], spec["rows"], spec["cols"], spec["tile_h"], spec["tile_w"],
                           overlap=spec["overlap"], jitter=spec["jitter"])


def test_grid_pairs_match_reference(golden):
    for spec in golden["grids"]:
        if spec["tile_h"] >= 2048:
            continue  # covered by test_large_pairs_match_reference
        g = _grid(spec)
        out, status = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, spec["edge_width"])
        for p, exp in enumerate(spec["results"]):
            assert exp is not None and status[p] == 0
            assert out[p, 1] == exp[1] and out[p, 2] == exp[2], (spec["name"], p)
            assert out[p, 0] == pytest.approx(exp[0], abs=1e-12)
        # planted integer shifts are what both recover
        assert np.array_equal(out[:, 1:].astype(np.int64), g.expected_shift)


def test_large_pairs_match_reference(golden):
    for spec in golden["grids"]:
        if spec["tile_h"] < 2048:
            continue
        g = _grid(spec)
        # one pair per direction keeps the CPU suite short
        for p in (0, 1):
            i1, i2 = g.pair_idx[p]
            r = oreg.compute_single_translation(g.tiles[i1], g.tiles[i2],
                                                "top" if g.pair_dir[p] else "left", spec["edge_width"])
            exp = spec["results"][p]
            assert r[1] == exp[1] and r[2] == exp[2]
            assert r[0] == pytest.approx(exp[0], abs=1e-12)


def test_featureless(golden):
    f = golden["featureless"]
    rng = np.random.default_rng(f["rng_seed"])
    flat = (5000 + 20 * rng.random((512, 512))).astype(np.uint16)
    tex = synth.make_scene(22, 512, 512)
    cases = {"flat_first": (flat, tex), "flat_second": (tex, flat), "both_textured": (tex, tex)}
    for name, (a, b) in cases.items():
        r = oreg.compute_single_translation(a, b, "left", 64)
        exp = f["pairs"][name]
        if exp is None:
            assert r is None
        else:
            assert r[1:] == tuple(exp[1:]) and r[0] == pytest.approx(exp[0], abs=1e-12)
    for probe in f["probes"]:
        t = (5000 + probe["amp"] * rng.random((512, 64))).astype(np.uint16)
        assert oreg.is_edge_featureless(t) == probe["featureless"]


def test_peak_ranking(golden):
    for s in golden["peaks"]["strips"]:
        h, w = s["shape"]
        big = synth.make_scene(s["seed"], h + 64, w + 64)
        a = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
        b = np.ascontiguousarray(big[32 + s["shift"][0]:32 + s["shift"][0] + h, 32 + s["shift"][1]:32 + s["shift"][1] + w])
        p = oreg.pcm(a, b)
        assert float(p.sum()) == pytest.approx(s["pcm_sum"], rel=1e-12)
        r, c, v = oreg.multi_peak_max(p, max_peaks=16)
        for i, (er, ec, ev) in enumerate(s["top"]):
            assert (r[i], c[i]) == (er, ec)
            assert v[i] == pytest.approx(ev, rel=1e-12, abs=1e-15)
    t = np.zeros((8, 8))
    for (r, c, v) in golden["peaks"]["tie"]["array_nonzero"]:
        t[r, c] = v
    r, c, v = oreg.multi_peak_max(t)
    got = [[int(r[i]), int(c[i]), float(v[i])] for i in range(5)]
    assert got == golden["peaks"]["tie"]["top"]


def test_ncc_table(golden):
    n = golden["ncc"]
    rng = np.random.default_rng(n["seed"])
    a = rng.integers(0, 65535, size=tuple(n["shape"])).astype(np.uint16)
    b = rng.integers(0, 65535, size=tuple(n["shape"])).astype(np.uint16)
    const = np.full(tuple(n["shape"]), 1234, dtype=np.uint16)
    got = {
        "random": oreg.ncc(a, b), "self": oreg.ncc(a, a),
        "negated": oreg.ncc(a, (65535 - a).astype(np.uint16)),
        "size_24": oreg.ncc(a[:4, :6], b[:4, :6]), "size_25": oreg.ncc(a[:5, :5], b[:5, :5]),
        "both_constant": oreg.ncc(const, const), "one_constant": oreg.ncc(a, const),
        "shape_mismatch": oreg.ncc(a, b[:, :29]), "empty": oreg.ncc(a[:0], b[:0]),
    }
    for k, exp in n["values"].items():
        if isinstance(exp, str):
            assert got[k] == float(exp)
        else:
            assert got[k] == pytest.approx(exp, abs=1e-15)


def test_candidate_order(golden):
    for s in golden["candidates"]:
        h, w = s["shape"]
        dy, dx = s["shift"]
        big = synth.make_scene(s["seed"], h + 64, w + 64)
        a = big[32:32 + h, 32:32 + w]
        b = big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]
        yi, xi, _ = oreg.multi_peak_max(oreg.pcm(a, b))
        assert [[int(yi[0]), int(xi[0])], [int(yi[1]), int(xi[1])]] == s["peaks"]
        ncc_v, y, x, table = oreg.interpret_translation(a, b, yi, xi, -h, h, -w, w, 2, return_table=True)
        assert [y, x] == s["best"][1:] and ncc_v == pytest.approx(s["best"][0], abs=1e-12)
        assert len(table) == len(s["evaluated"])
        for (ty, tx, tv), (ey, ex, ev) in zip(table, s["evaluated"]):
            assert (ty, tx) == (ey, ex)
            assert tv == pytest.approx(ev, abs=1e-12)


def test_polish_never_changes_the_discrete_result(golden):
    spec = golden["grids"][0]
    g = _grid(spec)
    a, _ = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, spec["edge_width"], polish=True)
    b, _ = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, spec["edge_width"], polish=False)
    assert np.array_equal(a, b)


def _m1_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m1_golden.json")) as fh:
        return json.load(fh)


def m1_strips(s):
    """The strip pair of one m1_golden.json case, as tests/golden/make_golden_m1.py builds it."""
    h, w = s["shape"]
    dy, dx = s["shift"]
    big = synth.make_scene(s["seed"], h + 64, w + 64)
    if s["smooth"]:
        f = np.fft.fft2(big.astype(np.float64))
        k = np.fft.fftfreq(h + 64)[:, None] ** 2 + np.fft.fftfreq(w + 64)[None, :] ** 2
        low = np.fft.ifft2(f * np.exp(-k * 2000.0)).real
        big = (1000 + 50000 * (low - low.min()) / (low.max() - low.min())).astype(np.uint16)
    return (np.ascontiguousarray(big[32:32 + h, 32:32 + w]),
            np.ascontiguousarray(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]))


def test_method1_candidates_match_the_reference():
    """The deterministic part of the reference's Method 1 (its candidate set and the best-NCC fallback of its
    consensus step), restated in the oracle, against vectors made by running the reference."""
    doc = _m1_golden()
    for s in doc["strips"]:
        a, b = m1_strips(s)
        h, w = s["shape"]
        yi, xi, _ = oreg.multi_peak_max(oreg.pcm(a, b))
        assert [[int(yi[i]), int(xi[i])] for i in range(10)] == s["peaks"], s["name"]
        t, sc = oreg.generate_translation_candidates(a, b, yi, xi, -h, h, -w, w, 10)
        assert np.array_equal(t, np.array(s["translations"])), s["name"]
        assert np.allclose(sc, np.array(s["scores"]), rtol=0, atol=1e-12), s["name"]
        best = oreg.best_single_translation(t, sc)
        assert best[1:] == tuple(s["fallback"][1:]) and abs(best[0] - s["fallback"][0]) < 1e-12
        got = oreg.compute_translation_peaks10(a, b, yi, xi, -h, h, -w, w, 10)
        assert got[1:] == tuple(s["fallback"][1:])            # never worse than the two-peak search
        assert s["fallback"][0] >= s["discrete"][0]
    for g in doc["grids"]:
        grid = synth.make_grid(g["seed"], g["rows"], g["cols"], g["tile_h"], g["tile_w"], overlap=g["overlap"], jitter=g["jitter"])
        out, status, tables = oreg.register_pairs_peaks10(grid.tiles, grid.pair_idx, grid.pair_dir, g["edge_width"])
        for p, exp in enumerate(g["pairs"]):
            assert (exp is None) == (status[p] != 0)
            if exp is None:
                continue
            assert len(tables[p][1]) == exp["n"]
            assert tuple(out[p, 1:]) == tuple(exp["fallback"][1:]) and abs(out[p, 0] - exp["fallback"][0]) < 1e-12
            assert abs(float(np.sum(tables[p][1])) - exp["score_sum"]) < 1e-9


def _consensus_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consensus_golden.json")) as fh:
        return json.load(fh)


def consensus_strips(s):
    """The strip pair of one consensus_golden.json case, as tests/golden/make_golden_consensus.py builds it."""
    h, w = s["shape"]
    dy, dx = s["shift"]
    big = synth.make_scene(s["seed"], h + 64, w + 64)
    if s["lowpass"]:
        f = np.fft.fft2(big.astype(np.float64))
        k = np.fft.fftfreq(h + 64)[:, None] ** 2 + np.fft.fftfreq(w + 64)[None, :] ** 2
        low = np.fft.ifft2(f * np.exp(-k * float(s["lowpass"]))).real
        big = (1000 + 50000 * (low - low.min()) / (low.max() - low.min())).astype(np.uint16)
    return (np.ascontiguousarray(big[32:32 + h, 32:32 + w]),
            np.ascontiguousarray(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]))


def test_consensus_step_matches_the_reference():
    """The consensus step of the reference's Method 1 (the clustering branch its own code takes when scikit-image's
    ransac raises on its model class), restated in the oracle, against vectors made by running that branch of the
    reference (tests/golden/make_golden_consensus.py)."""
    doc = _consensus_golden()
    for t in doc["tables"]:
        got = oreg.consensus_translation(np.array(t["translations"]).reshape(-1, 2), np.array(t["scores"]))
        assert [got[1], got[2]] == t["best"], t["name"]
    kinds = set()
    for s in doc["strips"]:
        a, b = consensus_strips(s)
        h, w = s["shape"]
        yi, xi, _ = oreg.multi_peak_max(oreg.pcm(a, b))
        tr_, sc = oreg.generate_translation_candidates(a, b, yi, xi, -h, h, -w, w, 10)
        assert len(sc) == s["n"]
        m1 = oreg.consensus_translation(tr_, sc)
        assert [m1[1], m1[2]] == s["method1"] and m1[3] == s["stood"], s
        got = oreg.compute_translation_consensus(a, b, yi, xi, -h, h, -w, w, 10)
        assert [got[1], got[2]] == s["final"][1:] and abs(got[0] - s["final"][0]) < 1e-12, s
        kinds.add(s["kind"])
    assert kinds == {"stands_wins", "stands_loses", "declined"}
    for g in doc["grids"]:
        grid = synth.make_grid(g["seed"], g["rows"], g["cols"], g["tile_h"], g["tile_w"], overlap=g["overlap"], jitter=g["jitter"])
        out, status, _ = oreg.register_pairs_consensus(grid.tiles, grid.pair_idx, grid.pair_dir, g["edge_width"])
        for p, exp in enumerate(g["pairs"]):
            assert (exp is None) == (status[p] != 0), (g["name"], p)
            if exp is not None:
                assert tuple(out[p, 1:]) == tuple(exp[1:]) and abs(out[p, 0] - exp[0]) < 1e-12, (g["name"], p)
