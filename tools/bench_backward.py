"""Training side of the OFF units on MI355X: train-mode forward (K1 + K2 with dropout) and the units'
backward (K2b + K1b + reductions) at BASELINE config 2 size, with algorithmic bytes / FLOPs.
    python tools/bench_backward.py [--batch 64] [--length 7] [--iters 20]
Under rocprofv3 --kernel-trace --stats the per-kernel split is in the stats CSV."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--length", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--variant", type=int, default=spec.VARIANT_RGB)
    a = ap.parse_args()
    B, L = a.batch, a.length
    N, P = B * L, B * (L - 1)
    h = runtime.OffForward(B, L, a.variant, precision=a.precision, training=True)
    h.load_state_dict(synth.make_weights(a.variant))
    feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
    gen = torch.Generator(device="cuda").manual_seed(5)
    bufs = [torch.randn(P, H, H, C, device="cuda", generator=gen) for H, C in ((28, 320), (14, 1056), (7, 832))]
    views = [(bufs[0], 0), (bufs[0], 160)] + [(bufs[1], 160 * k) for k in range(5)] + [(bufs[2], 0), (bufs[2], 160)]
    grads = h.new_unit_grads()
    t_fwd = timed(lambda: h.off_units_train(feats, 21, 0.8), a.iters)
    t_bwd = timed(lambda: h.off_units_backward(feats, views, 21, 0.8, grads=grads), a.iters)
    hw = sum(H * H for _n, _c, H in spec.SITES)
    # K2b: read dM (160 ch, P rows) + G (128, N) + D (32, P), write dG (128, N) + dD (32, P)
    k2b = B * hw * 4 * ((160 + 32 + 32) * (L - 1) + 256 * L)
    # K1b: read X once + dG + dD (ideal); FLOPs as the forward's K1
    x_bytes = sum(N * C * H * H * 4 for _n, C, H in spec.SITES)
    k1b = x_bytes + B * hw * 4 * (128 * L + 32 * (L - 1))
    flops = sum(2 * N * H * H * C * 128 + 2 * P * H * H * C * 32 for _n, C, H in spec.SITES)
    print(json.dumps({"batch": B, "length": L, "precision_fwd": a.precision,
                      "units_train_forward_ms": round(t_fwd, 4), "units_backward_ms": round(t_bwd, 4),
                      "clips_per_s_fwd_bwd_units": round(B / (t_fwd + t_bwd) * 1e3, 1),
                      "k2b_algorithmic_bytes": k2b, "k1b_algorithmic_bytes": k1b, "k1b_flops": flops,
                      "backward_floor_ms_hbm_8TBs": round((k2b + k1b) / 8e12 * 1e3, 4),
                      "k1b_floor_ms_fp32_mfma_157TF": round(flops / 157e12 * 1e3, 4)}))


if __name__ == "__main__":
    main()
