"""Times every (tile_cfg, splitk) plan of the channels-last conv kernel on the 24 fusion-conv
shapes at a given number of pairs (default P = 384 = BASELINE config 2) and prints / saves the
fastest plan per conv.  GPU only:  python tools/tune_conv.py [--pairs 384] [--out gpurun_out/tune.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec  # noqa: E402

TILE_BN = (128, 64, 64, 64, 128, 256, 128, 64)   # 6 / 7: the LDS-patch kernel (k x k convs)
TILE_BM = (128, 128, 256, 64, 64, 128, 196, 196)


def in_hw(key):
    if key == "motion_conv_trans_28":
        return 28
    if key.endswith(("_28a", "_28b", "_28c")) or key == "motion_conv_trans_14":
        return 14
    return 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=384)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune_conv.json"))
    ap.add_argument("--only", default="")
    ap.add_argument("--precision", type=int, default=0)
    args = ap.parse_args()
    P = args.pairs
    torch.manual_seed(0)
    results = {}
    for key, co, ci, k, s, p in spec.FUSION_CONVS:
        if args.only and args.only not in key:
            continue
        H = in_hw(key)
        x = torch.relu(torch.randn(P, H, H, ci, device="cuda"))
        w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
        b = torch.randn(co, device="cuda")
        wp = torch.empty(co, k, k, ci, device="cuda")
        from offk_amd import _lib
        _lib.check(_lib.load().offk_pack_conv_weight(runtime._stream(), runtime._ptr(w), co, ci, k, k, runtime._ptr(wp)))
        Ho = (H + 2 * p - k) // s + 1
        M = P * Ho * Ho
        flops = 2.0 * M * co * ci * k * k
        nkt = k * k * ci // 32
        ref = None
        rows = []
        for cfg in range(8):
            if co % TILE_BN[cfg]:
                continue
            if cfg >= 6 and k == 1:
                continue
            for sk in (1, 2, 3, 4, 6, 8, 12):
                if sk > 1 and (nkt // sk < 8 or sk * M * co * 4 > 1.2e9):
                    continue
                if cfg >= 6 and sk > ci // 32:
                    continue
                y = None
                try:
                    for _ in range(2):
                        y = runtime.conv2d_nhwc(x, w, b, s, p, tile_cfg=cfg, splitk=sk, w_packed=wp, precision=args.precision)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    part = torch.empty(max(sk, 1) * M * co, device="cuda") if sk > 1 else None
                    yb = torch.empty(P, Ho, Ho, co, device="cuda")
                    lib = _lib.load()
                    e0.record()
                    for _ in range(args.iters):
                        _lib.check(lib.offk_conv2d_ex(runtime._stream(), runtime._ptr(x), ci, 0, P, H, H, ci, runtime._ptr(wp),
                                                      runtime._ptr(b), co, k, k, s, p, None, 0, 0, 0, runtime._ptr(yb), co, 0,
                                                      cfg, sk, runtime._ptr(part), part.numel() if part is not None else 0, args.precision))
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / args.iters
                except Exception as ex:  # noqa: BLE001
                    print(key, cfg, sk, "FAILED", ex)
                    continue
                if ref is None:
                    ref = y
                err = ((y - ref).abs().max() / ref.abs().max()).item()
                rows.append((ms, cfg, sk, err))
        rows.sort()
        best = rows[0]
        blocks = -(-M // TILE_BM[best[1]]) * (co // TILE_BN[best[1]]) * best[2]
        results[key] = {"cfg": best[1], "splitk": best[2], "ms": best[0], "tflops": flops / best[0] / 1e9,
                        "blocks": blocks, "top": [list(r) for r in rows[:6]], "M": M, "gflop": flops / 1e9,
                        "maxerr_vs_first": max(r[3] for r in rows)}
        print("%-30s M=%6d K=%5d N=%4d  best cfg=%d sk=%d  %.3f ms  %.1f TF   | " % (key, M, ci * k * k, co, best[1], best[2],
              best[0], flops / best[0] / 1e9) + "  ".join("c%d/s%d:%.3f" % (r[1], r[2], r[0]) for r in rows[:6]), flush=True)
    tot = sum(v["ms"] for v in results.values())
    print("sum of best: %.3f ms, %.1f TF" % (tot, sum(v["gflop"] for v in results.values()) / tot))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
