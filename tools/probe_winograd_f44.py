"""Error of Winograd F(m x m, r x r) correlation in fp32 against fp64, Cook-Toom matrices from the interpolation points (numpy, CPU).
What a polyphase Winograd form of the 7x7 / stride 2 conv would have to live with: its phase kernels have four taps -> F(4x4, 4x4), 7 points."""
import numpy as np
from numpy.polynomial import polynomial as P


def cook_toom(m, r, pts):
    n = m + r - 1
    assert len(pts) == n - 1
    a = np.array(pts, dtype=np.float64)
    AT = np.zeros((m, n)); G = np.zeros((n, r)); BT = np.zeros((n, n))
    full = np.array([1.0])
    for aj in a:
        full = P.polymul(full, [-aj, 1.0])
    for i, ai in enumerate(a):
        Mi = np.array([1.0]); Ni = 1.0
        for j, aj in enumerate(a):
            if j != i:
                Mi = P.polymul(Mi, [-aj, 1.0]); Ni *= ai - aj
        AT[:, i] = ai ** np.arange(m)
        G[i, :] = ai ** np.arange(r) / Ni
        BT[i, :len(Mi)] = Mi
    AT[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    BT[n - 1, :len(full)] = full
    return AT, G, BT


def check(m, r, pts, K=320, trials=64, seed=0):
    AT, G, BT = cook_toom(m, r, pts)
    n = m + r - 1
    rng = np.random.default_rng(seed)
    # exactness in fp64
    d = rng.standard_normal((n, n)); g = rng.standard_normal((r, r))
    y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
    ref = np.array([[np.sum(d[i:i + r, j:j + r] * g) for j in range(m)] for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), np.abs(y - ref).max()
    AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
    errs = []
    for _ in range(trials):
        d = np.maximum(rng.standard_normal((K, n, n)), 0).astype(np.float32)
        g = (rng.uniform(-1, 1, (K, r, r)) / np.sqrt(K * r * r)).astype(np.float32)
        U = np.einsum("ia,kab,jb->kij", G32, g, G32).astype(np.float32)
        V = np.einsum("ia,kab,jb->kij", BT32, d, BT32).astype(np.float32)
        Mm = np.zeros((n, n), np.float32)
        for k in range(K):
            Mm = (Mm + U[k] * V[k]).astype(np.float32)
        y = (AT32 @ Mm @ AT32.T).astype(np.float32)
        ref = np.zeros((m, m))
        for i in range(m):
            for j in range(m):
                ref[i, j] = np.sum(d[:, i:i + r, j:j + r].astype(np.float64) * g.astype(np.float64))
        direct = np.zeros((m, m), np.float32)
        for i in range(m):
            for j in range(m):
                direct[i, j] = np.sum((d[:, i:i + r, j:j + r] * g).astype(np.float32), dtype=np.float32)
        errs.append((np.abs(y - ref).max(), np.abs(direct - ref).max(), np.abs(ref).max()))
    e = np.array(errs)
    print("F(%dx%d, %dx%d) points %s: winograd err / max|ref| %.2e (direct fp32 %.2e)" % (m, m, r, r, pts, e[:, 0].max() / e[:, 2].max(), e[:, 1].max() / e[:, 2].max()))


check(4, 3, [0, 1, -1, 2, -2])
check(4, 4, [0, 1, -1, 2, -2, 0.5])
check(4, 4, [0, 1, -1, 0.5, -0.5, 2])
check(2, 4, [0, 1, -1, 2])
check(3, 4, [0, 1, -1, 2, -2])
check(3, 4, [0, 1, -1, 0.5, -0.5])
check(5, 4, [0, 1, -1, 0.5, -0.5, 2, -2])
check(5, 4, [0, 1, -1, 0.5, -0.5, 2, -0.25])
check(5, 4, [0, 1, -1, 0.5, -0.5, 1.5, -1.5])
check(6, 4, [0, 1, -1, 0.5, -0.5, 2, -2, 0.25])
check(7, 4, [0, 1, -1, 0.5, -0.5, 2, -2, 0.25, -0.25])
