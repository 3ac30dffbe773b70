"""K2 (sobel_tdiff) bandwidth attribution on MI355X: per site and per algorithm variant.
algo 0 = register rotation over t (shipped), 1 = t across lanes + wavefront shuffle,
2 = temporal half only, 3 = spatial half only (diagnostics).
    python tools/bench_k2.py [--batch 64] [--length 7]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--length", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    B, L = a.batch, a.length
    N, P = B * L, B * (L - 1)
    h = runtime.OffForward(B, L, spec.VARIANT_RGB)
    h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
    feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
    h.off_units(feats)            # fills G_*, D_* in the workspace
    torch.cuda.synchronize()
    full = spec.algorithmic_bytes_sobel_tdiff(B, L)
    hw = sum(H * H for _n, _c, H in spec.SITES)
    tb = B * hw * 4 * (128 * L + 128 * (L - 1))
    sb = B * hw * 4 * (64 * (L - 1))
    lib = h.lib
    print("grouped K2 launch (all nine sites), B=%d L=%d" % (B, L))
    print("algo  what                         us       GB/s (algorithmic bytes of the part)")
    for algo, what, nbytes in ((0, "rotation + spatial (shipped)", full), (1, "shuffle + spatial", full),
                               (2, "temporal only (rotation)", tb), (3, "spatial only", sb),
                               (4, "flat + spatial", full), (5, "temporal only (flat)", tb)):
        for _ in range(3):
            h.sobel_tdiff_all(algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            h.sobel_tdiff_all(algo)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print("%d     %-28s %8.1f  %8.1f" % (algo, what, us, nbytes / us / 1e3), flush=True)
    # reference point: plain device-to-device copy of the same byte count (read+write)
    n = 700 * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda")
    y = torch.empty(n, device="cuda")
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("torch copy_ 700 MiB: %.1f us, %.1f GB/s (read+write)" % (us, 2 * n * 4 / us / 1e3))


if __name__ == "__main__":
    main()
