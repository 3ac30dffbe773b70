"""Ablation of the fp32 implicit-GEMM conv (tuning build only, OFFK_CONV_ABLATE bits: 1 no activation loads, 2 no weight
loads, 4 no LDS stores, 8 no MFMAs, 16 no epilogue; they act on the register-staged loader: OFFK_CONV_DMA=0): which part of
the K-loop the matrix pipe waits for.
    OFFK_LIB=tools/_bin/liboffk_tune.so OFFK_CONV_DMA=0 python tools/conv_ablate.py
    CASESET=small BITS=0,8,16,24,31 ... : the short-K convs of the bottleneck chains"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import _lib, runtime  # noqa: E402

P = int(os.environ.get("PAIRS", "384"))
CASES = [("7x7s2 320->64 @28", 320, 64, 7, 2, 3, 28, 3, 3), ("7x7s2 320->64 @28", 320, 64, 7, 2, 3, 28, 3, 1),
         ("7x7s2 320->64 @28", 320, 64, 7, 2, 3, 28, 1, 3), ("7x7s2 320->64 @28", 320, 64, 7, 2, 3, 28, 2, 6),
         ("5x5s2 1056->128 @14", 1056, 128, 5, 2, 2, 14, 4, 12), ("5x5s2 1056->128 @14", 1056, 128, 5, 2, 2, 14, 0, 12),
         ("3x3 832->256 @7", 832, 256, 3, 1, 1, 7, 0, 6), ("1x1 128->512 @7 (merged 14a)", 256, 512, 1, 1, 0, 7, 3, 1)]
if os.environ.get("CASESET") == "small":      # the bottleneck-chain convs: where does a short-K launch spend its time?
    CASES = [("3x3 64->64 @14", 64, 64, 3, 1, 1, 14, 3, 1), ("1x1 64->256 @14", 64, 256, 1, 1, 0, 14, 3, 1),
             ("1x1 256->64 @14", 256, 64, 1, 1, 0, 14, 3, 1), ("1x1 64->64 @14", 64, 64, 1, 1, 0, 14, 3, 1),
             ("3x3 128->128 @7", 128, 128, 3, 1, 1, 7, 3, 2), ("1x1 512->128 @7", 512, 128, 1, 1, 0, 7, 3, 1)]
prec = int(os.environ.get("PREC", "0"))
BITS = [int(b) for b in os.environ.get("BITS", "0,3,7,23,8").split(",")]
for name, ci, co, k, s, p, H, cfg, sk in CASES:
    x = torch.relu(torch.randn(P, H, H, ci, device="cuda"))
    w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
    b = torch.randn(co, device="cuda")
    wp = torch.empty(co, k, k, ci, device="cuda")
    lib = _lib.load()
    _lib.check(lib.offk_pack_conv_weight(runtime._stream(), runtime._ptr(w), co, ci, k, k, runtime._ptr(wp)))
    Ho = (H + 2 * p - k) // s + 1
    M = P * Ho * Ho
    flops = 2.0 * M * co * ci * k * k
    part = torch.empty(max(sk, 1) * M * co, device="cuda")
    yb = torch.empty(P, Ho, Ho, co, device="cuda")
    line = "%-30s cfg %d sk %2d |" % (name, cfg, sk)
    for bits in BITS:
        os.environ["OFFK_CONV_ABLATE"] = str(bits)
        def run():
            _lib.check(lib.offk_conv2d_ex(runtime._stream(), runtime._ptr(x), ci, 0, P, H, H, ci, runtime._ptr(wp), runtime._ptr(b), co, k, k,
                                          s, p, None, 0, 0, 0, runtime._ptr(yb), co, 0, cfg, sk, runtime._ptr(part), part.numel(), prec))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += "  [%d] %.3f ms %5.1f TF" % (bits, ms, flops / ms / 1e9)
    print(line, flush=True)
os.environ["OFFK_CONV_ABLATE"] = "0"
