"""The numerical claim behind the subpixel stage's node evaluation, checked on the CPU: the upsampled cross-correlation
around a peak, a trigonometric polynomial with frequencies of at most pi per pixel sampled over 1.5 pixels, equals
the polynomial through its values at 32 x 32 Chebyshev nodes to rounding error, so the argmax on the 1 / upsample
grid is the one of the direct matrix products (oracle/subpixel.py).  The interpolation matrix is built here the way
the library's host code builds it (barycentric form, first-kind nodes)."""
import numpy as np
import pytest


def nodes_and_matrix(n_up: int, uf: int, m: int):
    offsets = (np.arange(n_up) - n_up // 2) / uf
    centre = ((n_up - 1) / 2.0 - n_up // 2) / uf
    half = (n_up - 1) / (2.0 * uf)
    j = np.arange(m)
    t = np.cos(np.pi * (2 * j + 1) / (2 * m))
    w = (-1.0) ** j * np.sin(np.pi * (2 * j + 1) / (2 * m))
    x = (2.0 * np.arange(n_up) - (n_up - 1)) / (n_up - 1)
    p = np.zeros((n_up, m))
    for u in range(n_up):
        d = x[u] - t
        hit = np.flatnonzero(d == 0)
        if hit.size:
            p[u, hit[0]] = 1.0
        else:
            q = w / d
            p[u] = q / q.sum()
    return centre + half * t, offsets, p


@pytest.mark.parametrize("s,l,uf", [(256, 2048, 100), (45, 300, 100), (64, 256, 10), (32, 64, 2), (195, 260, 37)])
def test_node_interpolation_reproduces_the_direct_sums(s, l, uf):
    rng = np.random.default_rng(s * 7 + l)
    n_up = (3 * uf + 1) // 2
    ky, kx = np.fft.fftfreq(s)[:, None], np.fft.fftfreq(l)[None, :]
    r = 0.7 * np.exp(-2j * np.pi * (ky * 3.37 + kx * -5.81)) + 0.3 * np.exp(2j * np.pi * rng.random((s, l)))
    r /= np.abs(r)
    shift = np.array([3.0, -6.0])

    def cc_at(ds, dl):
        es = np.exp(2j * np.pi * np.outer(shift[0] + ds, np.fft.fftfreq(s)))
        el = np.exp(2j * np.pi * np.outer(shift[1] + dl, np.fft.fftfreq(l)))
        return es @ r @ el.T

    nodes, offsets, p = nodes_and_matrix(n_up, uf, 32)
    direct = cc_at(offsets, offsets)
    via_nodes = p @ cc_at(nodes, nodes) @ p.T
    assert np.abs(direct - via_nodes).max() <= 2e-14 * np.abs(direct).max()
    assert np.abs(p).sum(axis=1).max() < 3.5          # Lebesgue constant of 32 first-kind nodes
    assert np.argmax(np.abs(direct)) == np.argmax(np.abs(via_nodes))


def test_sixteen_nodes_would_not_be_enough_for_rounding_level_agreement():
    """Why 32 and not one matrix tile of 16: the truncation term 2 (pi h / 2)^n / n! is ~1e-12 of the largest possible
    |CC| at n = 16 and far below rounding at 24 and beyond."""
    rng = np.random.default_rng(3)
    s, l, uf = 64, 512, 100
    n_up = 150
    r = np.exp(2j * np.pi * rng.random((s, l)))

    def cc_at(ds, dl):
        return np.exp(2j * np.pi * np.outer(ds, np.fft.fftfreq(s))) @ r @ np.exp(2j * np.pi * np.outer(dl, np.fft.fftfreq(l))).T

    errs = {}
    for m in (16, 32):
        nodes, offsets, p = nodes_and_matrix(n_up, uf, m)
        errs[m] = np.abs(cc_at(offsets, offsets) - p @ cc_at(nodes, nodes) @ p.T).max() / (s * l)
    assert errs[32] < 1e-15 < errs[16] < 1e-11
