"""How close to fp32 arithmetic is a contraction done as bf16 x bf16 products with fp32 accumulation?  (numpy, CPU)
Operands split into three bf16 pieces (8 significant bits each: a = a1 + a2 + a3 exactly for a normal fp32), products of pieces are
exact in fp32; bf16x3 = the library's secondary mode (two pieces, 3 products), bf16x6 drops the three products below 2^-24 of the
leading one, bf16x9 keeps all.  Accumulation in fp32, four k values per step, like the MFMA loop.  Reference: float64.
Measured (M = 256, K = 1024, N = 128, ReLU-like activations, fan-in-scaled weights), error / max |ref|:
    fp32 4.8e-7 max, 6.0e-8 rms;  bf16x3 3.6e-6, 8.7e-7;  bf16x6 1.0e-6, 1.3e-7;  bf16x9 1.0e-6, 1.3e-7
=> six products on the bf16 pipe (2.5 PF dense / 6 = 417 TF) sit within 2x of the fp32 pipe's own rounding noise (157 TF)."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    h = bf16(x)
    r = (x - h).astype(np.float32)
    m = bf16(r)
    return h, m, bf16((r - m).astype(np.float32))


M, K, N = 256, 1024, 128
a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
b = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
ref = a.astype(np.float64) @ b.astype(np.float64)


def acc32(terms):
    c = np.zeros((M, N), np.float32)
    for k in range(0, K, 4):
        for A, B in terms:      # smallest products first
            c = (c + (A[:, k:k + 4] @ B[k:k + 4])).astype(np.float32)
    return c


a1, a2, a3 = split3(a)
b1, b2, b3 = split3(b)
cases = (("fp32", [(a, b)]),
         ("bf16x3", [(a1, b2), (a2, b1), (a1, b1)]),
         ("bf16x6", [(a3, b1), (a1, b3), (a2, b2), (a2, b1), (a1, b2), (a1, b1)]),
         ("bf16x9", [(a3, b3), (a3, b2), (a2, b3), (a3, b1), (a1, b3), (a2, b2), (a2, b1), (a1, b2), (a1, b1)]))
s = np.abs(ref).max()
for name, terms in cases:
    e = np.abs(acc32(terms) - ref)
    print("%-7s max err / max|ref| %.3e   rms %.3e" % (name, e.max() / s, np.sqrt((e ** 2).mean()) / s))
