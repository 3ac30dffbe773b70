"""Per-site comparison of the fused units kernel's two fp32 forms (LDS-DMA vs register-staged): T channels of the fusion
buffers and the D regions.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [(1, 7), (2, 3), (3, 7), (9, 7)]
SITE_FUS = [("fusion_28", 320, 0), ("fusion_28", 320, 160), ("fusion_14", 1056, 0), ("fusion_14", 1056, 160), ("fusion_14", 1056, 320),
            ("fusion_14", 1056, 480), ("fusion_14", 1056, 640), ("fusion_7", 832, 0), ("fusion_7", 832, 160)]
for B, L in shapes:
    res = {}
    MODE = os.environ.get("DMA_MODE", "2")
    for dma in ("0", MODE):
        os.environ["OFFK_PW_DMA"] = dma
        h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision="fp32")
        h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
        feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 1)]
        h.workspace.zero_()
        out = h.forward(feats)
        torch.cuda.synchronize()
        res[dma] = ([h.region(n, cs)[:, off:off + 160].clone() for n, cs, off in SITE_FUS],
                    [h.region("D_" + s, 32).clone() for s in spec.SITE_NAMES], [o.clone() for o in out])
    for si in range(9):
        a, b = res["0"][0][si], res[MODE][0][si]
        da, db = res["0"][1][si], res[MODE][1][si]
        bad = (a != b).any(dim=1).nonzero().flatten()
        badd = (da != db).any(dim=1).nonzero().flatten()
        print("B=%d L=%d site %d: M rows differing %d / %d (first %s), max |diff| %.3g ; D rows differing %d / %d (first %s)"
              % (B, L, si, bad.numel(), a.shape[0], bad[:6].tolist(), (a - b).abs().max().item(), badd.numel(), da.shape[0], badd[:6].tolist()))
        if bad.numel():
            HW = a.shape[0] // (B * (L - 1))
            d = (a != b)
            pairs = sorted(set((bad // HW).tolist()))
            chans = d.any(dim=0).nonzero().flatten().tolist()
            px = sorted(set((bad % HW).tolist()))
            print("   pairs", pairs, "channels", chans[:8], "..", chans[-4:], "n", len(chans), "pixels", px[:8], "..", px[-4:], "n", len(px))
    print("logits identical:", [bool(torch.equal(x, y)) for x, y in zip(res["0"][2], res[MODE][2])],
          "max |diff| / max |ref|:", ["%.2e" % ((x - y).abs().max() / x.abs().max()).item() for x, y in zip(res["0"][2], res[MODE][2])], flush=True)
