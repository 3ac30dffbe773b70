"""Ablation of K1 (tuning build only, OFFK_PW_ABLATE bits: 1 no feature-map loads, 2 no weight loads, 4 no LDS stores,
8 no MFMAs).  Times offk_off_units (K1 + K2) minus K2 alone.
    OFFK_LIB=tools/_bin/liboffk_tune.so python tools/pw_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

B, L = 64, 7
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]


def timed(fn, iters=10):
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


for prec in ("fp32",):
    h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision=prec)
    h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
    h.off_units(feats)
    k2 = timed(lambda: h.sobel_tdiff_all(0))
    line = "%-7s K2 %.3f ms | K1:" % (prec, k2)
    for bits in (0, 1, 2, 3, 4, 7, 8, 11, 15):
        os.environ["OFFK_PW_ABLATE"] = str(bits)
        t = timed(lambda: h.off_units(feats)) - k2
        line += "  [%d] %.3f" % (bits, t)
    os.environ["OFFK_PW_ABLATE"] = "0"
    print(line, flush=True)
