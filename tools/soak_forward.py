"""Soak: the same forward N times, every result compared bit for bit with the first (logits of the three heads + the sum_7 region) --
a rare hazard (a late store-data read, a wait one instruction short) shows as an occasional mismatch.
    python tools/soak_forward.py [runs] [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = 7
w = synth.make_weights(spec.VARIANT_RGB)
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
for prec in ("fp32", "f32split"):
    h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision=prec)
    h.load_state_dict(w)
    ref = [t.clone() for t in h.forward(feats)] + [h.region("sum_7", 1024).clone(), h.region("fusion_7", 832).clone(), h.region("fusion_14", 1056).clone()]
    bad = 0
    for i in range(runs):
        out = list(h.forward(feats)) + [h.region("sum_7", 1024), h.region("fusion_7", 832), h.region("fusion_14", 1056)]
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
    torch.cuda.synchronize()
    print("%-8s B = %d: %d forwards, %d differ from the first bit for bit" % (prec, B, runs, bad), flush=True)
