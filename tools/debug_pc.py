"""Debug: producer / consumer split units kernel against the two-blocks-per-CU form, element by element."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import offk_amd
from offk_amd import runtime, spec, synth
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 7
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 4)]
outs = {}
for form in ("0", "2"):
    os.environ["OFFK_SPLIT_PC"] = form
    h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision="f32split")
    h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
    h.workspace.zero_()
    h.off_units_fused(feats)
    torch.cuda.synchronize()
    P = B * (L - 1)
    res = []
    for fkey, fd in spec.FUSION.items():
        width = 160 * len(fd["sites"]) + fd["carry"]
        buf = h.region("fusion_" + fkey, width).view(P, fd["H"] * fd["H"], width)
        for i, sname in enumerate(fd["sites"]):
            res.append((sname, buf[..., 160 * i + 32:160 * i + 160].clone().cpu(), h.region("D_" + sname, 32).view(P, fd["H"] * fd["H"], 32).clone().cpu()))
    outs[form] = res
for (n, Ta, Da), (_n, Tb, Db) in zip(outs["0"], outs["2"]):
    for nm, a, b in (("T", Ta, Tb), ("D", Da, Db)):
        bad = ~torch.isclose(a, b, rtol=1e-4, atol=1e-5) | ~torch.isfinite(b)
        print("site %s %s: shape %s, mismatching %d of %d, nonfinite %d" % (n, nm, tuple(a.shape), int(bad.sum()), bad.numel(), int((~torch.isfinite(b)).sum())))
        if bad.any():
            idx = bad.nonzero()
            pairs, pix, ch = idx[:, 0].unique(), idx[:, 1].unique(), idx[:, 2].unique()
            print("   pairs", pairs[:16].tolist(), "pixels", pix[:24].tolist(), "(n=%d)" % len(pix), "channels", ch[:40].tolist(), "(n=%d)" % len(ch))
            i0 = idx[0]
            print("   first:", i0.tolist(), "ref", a[tuple(i0)].item(), "got", b[tuple(i0)].item())
