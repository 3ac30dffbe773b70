"""Times the LDS-patch conv kernel (tile_cfg 6 / 7) against the tuned generic plan on the three big fusion convs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import offk_amd  # noqa: F401
from offk_amd import runtime, _lib

P = 384
CASES = (("7x7s2 320->64 @28", 64, 320, 7, 2, 3, 28, [(1, 6), (10, 1), (10, 2), (10, 4), (10, 5)]),
         ("5x5s2 1056->128 @14", 128, 1056, 5, 2, 2, 14, [(0, 6), (10, 2), (10, 3), (10, 6), (10, 11)]),
         ("3x3 832->256 @7", 256, 832, 3, 1, 1, 7, [(5, 3), (7, 4)]))
lib = _lib.load()
for name, co, ci, k, s, p, H, plans in CASES:
    x = torch.relu(torch.randn(P, H, H, ci, device="cuda"))
    w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
    b = torch.randn(co, device="cuda")
    wp = torch.empty(co, k, k, ci, device="cuda")
    _lib.check(lib.offk_pack_conv_weight(runtime._stream(), runtime._ptr(w), co, ci, k, k, runtime._ptr(wp)))
    ws = torch.empty_like(wp)
    _lib.check(lib.offk_split_bf16x3(runtime._stream(), runtime._ptr(wp), wp.numel(), runtime._ptr(ws)))
    Ho = (H + 2 * p - k) // s + 1
    M = P * Ho * Ho
    yb = torch.empty(P, Ho, Ho, co, device="cuda")
    for cfg, sk in plans:
        part = torch.empty(max(sk, 1) * M * co, device="cuda") if sk > 1 else None
        def run():
            _lib.check(lib.offk_conv2d_ex(runtime._stream(), runtime._ptr(x), ci, 0, P, H, H, ci, runtime._ptr(ws), runtime._ptr(b), co, k, k, s, p,
                                          None, 0, 0, 0, runtime._ptr(yb), co, 0, cfg, sk, runtime._ptr(part), part.numel() if part is not None else 0, 1))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-22s cfg %d splitk %2d  %.3f ms  %.0f TF(fp32-equivalent)" % (name, cfg, sk, ms, 2.0 * M * co * ci * k * k / ms / 1e9), flush=True)
