"""CPU checks of the Winograd transform matrices the HIP kernels carry (csrc/winograd.hip, csrc/winograd7.hip):
the Cook-Toom construction of tools/gen_winograd_f54.py reproduces a direct correlation in float64, has the zero structure the
polyphase forms rely on (point 0 sees tap 0 only, point infinity the last tap only), and the generated functions in the tree
are what the generator emits today."""
import importlib.util
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optical-flow-guided-feature-pytorch_amd", "csrc")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_winograd", os.path.join(ROOT, "tools", "gen_winograd_f54.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("m,r,pts", [(5, 4, [0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0]), (3, 3, [0.0, 1.0, -1.0, 2.0]),
                                     (4, 3, [0.0, 1.0, -1.0, 2.0, -2.0])])
def test_cook_toom_is_a_correlation(m, r, pts):
    g = _gen()
    AT, G, BT = g.cook_toom(m, r, pts)
    n = m + r - 1
    rng = np.random.default_rng(3)
    for _ in range(8):
        d, k = rng.standard_normal(n), rng.standard_normal(r)
        want = np.array([sum(k[u] * d[o + u] for u in range(r)) for o in range(m)])
        assert np.allclose(AT @ ((G @ k) * (BT @ d)), want, atol=1e-11)
        d2, k2 = rng.standard_normal((n, n)), rng.standard_normal((r, r))
        want2 = np.array([[np.sum(d2[i:i + r, j:j + r] * k2) for j in range(m)] for i in range(m)])
        assert np.allclose(AT @ ((G @ k2 @ G.T) * (BT @ d2 @ BT.T)) @ AT.T, want2, atol=1e-9)
    assert np.abs(G[0, 1:]).max() == 0 and np.abs(G[n - 1, :r - 1]).max() == 0      # what the zero-skipping of the polyphase forms uses


@pytest.mark.parametrize("which,src", [(None, "winograd7.hip"), ("f33", "winograd_common.h")])
def test_generated_transforms_in_tree(which, src):
    g = _gen()
    buf = io.StringIO()
    with redirect_stdout(buf):
        if which == "f33":
            g.generate(3, 3, [0.0, 1.0, -1.0, 2.0], ("bt5", "at3", "g5"), "F(3, 3), points 0, 1, -1, 2, infinity")
        else:
            g.generate(g.M_, g.R_, g.PTS, ("bt8", "at5x8", "g8"), "points 0, 1, -1, 1/2, -1/2, 2, -2, infinity")
    text = open(os.path.join(CSRC, src)).read()
    for line in buf.getvalue().splitlines():
        if line.strip():
            assert line in text, line


def test_hand_written_f43_matches_cook_toom():
    """winograd.hip's bt6 / at4 / g6 are hand-factored; evaluate them (transcribed) against the matrices for the points 0, +-1, +-2, inf."""
    g = _gen()
    AT, G, BT = g.cook_toom(4, 3, [0.0, 1.0, -1.0, 2.0, -2.0])

    def bt6(d):
        return np.array([4 * d[0] - 5 * d[2] + d[4], -4 * (d[1] + d[2]) + d[3] + d[4], 4 * (d[1] - d[2]) - d[3] + d[4],
                         -2 * d[1] - d[2] + 2 * d[3] + d[4], 2 * d[1] - d[2] - 2 * d[3] + d[4], 4 * d[1] - 5 * d[3] + d[5]])

    def at4(m):
        a, b, c, e = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
        return np.array([m[0] + a + c, b + 2 * e, a + 4 * c, b + 8 * e + m[5]])

    def g6(k):
        return np.array([k[0] / 4, -(k[0] + k[1] + k[2]) / 6, (-k[0] + k[1] - k[2]) / 6, k[0] / 24 + k[1] / 12 + k[2] / 6,
                         k[0] / 24 - k[1] / 12 + k[2] / 6, k[2]])

    rng = np.random.default_rng(5)
    for _ in range(8):
        d, k = rng.standard_normal(6), rng.standard_normal(3)
        want = np.array([sum(k[u] * d[o + u] for u in range(3)) for o in range(4)])
        assert np.allclose(at4(g6(k) * bt6(d)), want, atol=1e-11)          # the hand-written set is a correlation too
    text = open(os.path.join(CSRC, "winograd_common.h")).read()
    for frag in ("t[0] = 4.f * d[0] - 5.f * d[2] + d[4];", "s[3] = b + 8.f * e + m[5];", "u[5] = g[2];"):
        assert frag in text
