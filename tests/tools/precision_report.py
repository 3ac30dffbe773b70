"""Measured error of the two arithmetic modes against the oracle and the fp64-accumulating C
oracle on the golden configurations (GPU only).  python tests/tools/precision_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402
from oracle import c_binding, off_oracle as orc  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def centred(a):
    a = np.asarray(a, np.float64)
    return a - a.mean(0, keepdims=True)


for variant, B, L in ((0, 2, 3), (1, 2, 3), (0, 3, 7)):
    w = synth.make_weights(variant)
    feats = synth.make_features(B, L, 3)
    with torch.no_grad():
        (r7, r14, r28), st = orc.off_forward([torch.from_numpy(f) for f in feats], orc.to_torch_weights(w), B, L, variant,
                                             orc.SLICE_FLAT, consensus=False, return_stages=True)
    c7 = c14 = c28 = None
    if B * L <= 6:
        c7, c14, c28 = c_binding.forward(feats, [w[k] for k in spec.weight_shapes(variant)], B, L, variant, 0, False)
    for prec in ("fp32", "f32split"):
        h = runtime.OffForward(B, L, variant, consensus=False, precision=prec)
        h.load_state_dict(w)
        o7, o14, o28 = (t.cpu().numpy() for t in h.forward([torch.from_numpy(f).cuda() for f in feats]))
        P = B * (L - 1)
        s7 = h.region("sum_7", 1024).view(P, 7, 7, 1024).permute(0, 3, 1, 2).cpu().numpy()
        f28 = h.region("fusion_28", 320).view(P, 28, 28, 320).permute(0, 3, 1, 2).cpu().numpy()
        line = "variant %d B%d L%d %-7s | vs torch oracle: fc7 %.2e fc14 %.2e fc28 %.2e  centred fc7 %.2e  fusion_28 %.2e sum_7 %.2e" % (
            variant, B, L, prec, rel(o7, r7), rel(o14, r14), rel(o28, r28), rel(centred(o7), centred(r7.numpy())),
            rel(f28, st["fusion_28"]), rel(s7, st["sum_7"]))
        if c7 is not None:
            line += " | vs fp64 C oracle: fc7 %.2e fc14 %.2e fc28 %.2e" % (rel(o7, c7), rel(o14, c14), rel(o28, c28))
        print(line, flush=True)
