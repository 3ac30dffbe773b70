"""Producer / consumer cycle split of the bf16x3 conv kernel.  Needs a liboffk whose conv_igemm.hip was compiled
with -DOFFK_CONV_TIMING (hipcc -c csrc/conv_igemm.hip ... -DOFFK_CONV_TIMING, link with the other objects of
csrc/_obj into a second .so) and loaded through OFFK_LIB.  Prints, per conv, the cycle sums the waves spent multiplying, storing tiles, issuing
loads and waiting at the per-K-tile barrier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import offk_amd  # noqa: F401
from offk_amd import runtime, _lib

P = 384
CASES = (("7x7s2 320->64 @28", 64, 320, 7, 2, 3, 28, 1, 6), ("5x5s2 1056->128 @14", 128, 1056, 5, 2, 2, 14, 0, 6),
         ("5x5s2 cfg4", 128, 1056, 5, 2, 2, 14, 4, 6), ("3x3 832->256 @7", 256, 832, 3, 1, 1, 7, 0, 6),
         ("3x3 128->512 @7", 512, 128, 3, 1, 1, 7, 4, 1))
lib = _lib.load()
for name, co, ci, k, s, p, H, cfg, sk in CASES:
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
    part = torch.empty(max(sk, 1) * M * co, device="cuda") if sk > 1 else None

    def run():
        _lib.check(lib.offk_conv2d_ex(runtime._stream(), runtime._ptr(x), ci, 0, P, H, H, ci, runtime._ptr(ws), runtime._ptr(b), co, k, k, s, p,
                                      None, 0, 0, 0, runtime._ptr(yb), co, 0, cfg, sk, runtime._ptr(part), part.numel() if part is not None else 0, 1))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    os.environ["OFFK_CONV_TIMING_DUMP"] = "1"
    run()                       # dumps + resets the sums of the warm-up launches
    os.environ.pop("OFFK_CONV_TIMING_DUMP")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print("== %s cfg %d splitk %d: next line = sums over 5 launches" % (name, cfg, sk), flush=True)
    os.environ["OFFK_CONV_TIMING_DUMP"] = "1"
    run()
    os.environ.pop("OFFK_CONV_TIMING_DUMP")
    torch.cuda.synchronize()
