"""In-situ plan tuning: coordinate descent over (tile_cfg, splitk) of the heavy fusion convs and the three merged
1x1 convs, timing the WHOLE forward (the isolated per-conv optimum of tools/tune_conv.py ignores L2 state and
neighbouring kernels).  Prints the plans that beat the built-in table.
    python tools/tune_forward.py [--batch 64] [--length 7] [--precision f32split]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

CAND = {
    "motion_conv2_trans_28a": [(1, 1), (7, 1), (3, 1), (4, 1)],
    "motion_conv2_trans_28b": [(1, 1), (7, 1), (3, 1), (4, 1)],
    "motion_conv2_trans_28c": [(1, 1), (7, 1), (3, 1), (4, 1)],
    "motion_conv1_trans_28a": [(3, 1), (1, 1)],
    "motion_conv1_trans_28b": [(3, 1), (1, 1)],
    "motion_conv1_trans_28c": [(3, 1), (1, 1)],
    "motion_conv3_trans_28b": [(3, 1), (1, 1), (4, 1), (0, 1)],
    "motion_conv3_trans_28c": [(3, 1), (1, 1), (4, 1), (0, 1)],
    "merged_28a": [(3, 1), (1, 1), (4, 1), (0, 1)],
    "motion_conv1_trans_14a": [(3, 1), (4, 1), (1, 1)],
    "merged_14a": [(3, 1), (1, 1), (4, 1), (0, 1), (5, 1)],
    "motion_conv1_trans_14b": [(3, 1), (4, 1), (1, 1)],
    "motion_conv1_trans": [(4, 1), (3, 1), (1, 1), (0, 1)],
    "merged_7": [(3, 1), (1, 1), (4, 1), (0, 1), (5, 1)],
    "motion_conv2_trans_14a": [(4, 1), (7, 1), (3, 1)],
    "motion_conv2_trans_14b": [(4, 1), (7, 1), (3, 1)],
    "motion_conv2_trans": [(0, 3), (7, 2), (7, 1), (4, 3)],
    "motion_conv_trans": [(7, 4), (7, 2), (7, 3), (5, 3)],
    "motion_conv_trans_14": [(0, 6), (0, 12), (0, 4), (0, 8)],
    "motion_conv_trans_28": [(10, 4), (10, 5), (10, 3), (1, 6)],
}

# exact-fp32 mode: the LDS-patch kernel (6 = 128 channels per block, 7 = 64) against the built-in generic plans
CAND_FP32 = {
    "motion_conv_trans_28": [(3, 6), (3, 3), (4, 6), (4, 3), (0, 6), (0, 3), (1, 6), (3, 4)],
    "motion_conv1_trans_28a": [(3, 1), (1, 1)],
    "motion_conv2_trans_28a": [(3, 1), (1, 1), (3, 2)],
    "merged_28a": [(3, 1), (1, 1), (4, 1), (0, 1)],
    "motion_conv1_trans_28b": [(3, 1), (1, 1)],
    "motion_conv2_trans_28b": [(3, 1), (1, 1), (3, 2)],
    "motion_conv3_trans_28b": [(3, 1), (1, 1), (4, 1), (0, 1)],
    "motion_conv1_trans_28c": [(3, 1), (1, 1)],
    "motion_conv2_trans_28c": [(3, 1), (1, 1), (3, 2)],
    "motion_conv3_trans_28c": [(3, 1), (1, 1), (4, 1), (0, 1)],
    "motion_conv_trans_14": [(3, 12), (3, 6), (4, 12), (4, 6), (0, 6), (0, 12), (1, 12), (3, 8)],
    "motion_conv1_trans_14a": [(3, 1), (4, 1), (3, 2)],
    "motion_conv2_trans_14a": [(0, 3), (4, 3), (3, 3), (3, 2), (4, 2)],
    "merged_14a": [(3, 1), (4, 1), (0, 1)],
    "motion_conv1_trans_14b": [(3, 2), (3, 1), (4, 1), (3, 4)],
    "motion_conv2_trans_14b": [(0, 3), (4, 3), (3, 3), (3, 2), (4, 2)],
    "motion_conv3_trans_14b": [(4, 1), (0, 1), (3, 1), (4, 2), (6, 2)],
    "motion_conv_trans": [(3, 6), (4, 6), (0, 6), (0, 4), (4, 4), (3, 4), (1, 6)],
    "motion_conv1_trans": [(3, 1), (4, 1), (0, 1)],
    "motion_conv2_trans": [(4, 3), (0, 3), (4, 2), (3, 3), (6, 4)],
    "merged_7": [(3, 1), (4, 1), (0, 1)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--length", type=int, default=7)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--wide", action="store_true", help="every generic tile x a split-K ladder per conv (other batch shapes)")
    a = ap.parse_args()
    B, L = a.batch, a.length
    h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision=a.precision)
    h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
    feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
    arr = h._feat_array(feats)
    out = [torch.empty(h.out_rows(), spec.NUM_CLASSES, device="cuda") for _ in range(3)]

    def t():
        for _ in range(3):
            h.forward_into(arr, *out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            h.forward_into(arr, *out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    base = t()
    print("built-in plans: %.4f ms" % base, flush=True)
    chosen = {}
    best = base
    for rnd in range(a.rounds):
        table = CAND_FP32 if a.precision == "fp32" else CAND
        if a.wide:
            table = {}
            for key in CAND_FP32:
                small = "conv1_" in key or "merged" in key or "conv3_trans_28" in key
                table[key] = [(c, k) for c in (3, 4, 1, 0) for k in ((1, 2) if small else (1, 2, 3, 4, 6, 8, 12))]
                if not small and a.precision != "fp32":
                    table[key] += [(7, 1), (7, 2), (7, 4), (10, 4), (5, 2), (5, 4)]
        for key, cands in table.items():
            if a.precision == "fp32" and key.endswith(("_28a", "_28b", "_28c")):
                continue      # fusion@28's chains run in chain_fused.hip in fp32: no plan
            res = []
            for cfg, sk in cands:
                h.set_conv_plan(key, cfg, sk)
                try:
                    res.append((t(), cfg, sk))
                except runtime._lib.OffkError:      # tile does not divide Co
                    torch.cuda.synchronize()
            res.sort()
            tb, cfg, sk = res[0]
            h.set_conv_plan(key, cfg, sk)
            chosen[key] = (cfg, sk)
            best = tb
            print("round %d %-28s best (%d,%d) %.4f ms   | %s" % (rnd, key, cfg, sk, tb, "  ".join("(%d,%d):%.4f" % (c, s, x) for x, c, s in res[:4])), flush=True)
    print("final %.4f ms (built-in %.4f)" % (t(), base))
    print(chosen)


if __name__ == "__main__":
    main()
