"""debug: chain 28a in split form with the branch conv zeroed / the c1 conv's input made non-negative"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import offk_amd
from offk_amd import runtime, spec, synth
from oracle import off_oracle as orc
B, L = 12, 7
feats_np = synth.make_features(B, L, 2)
P = B * (L - 1)
for case in ("plain", "branch_zero", "c1_zero"):
    w = dict(synth.make_weights(0))
    if case == "branch_zero":
        w["motion_conv_branch_28a.weight"] = np.zeros_like(w["motion_conv_branch_28a.weight"]); w["motion_conv_branch_28a.bias"] = np.zeros_like(w["motion_conv_branch_28a.bias"])
    if case == "c1_zero":
        w["motion_conv1_trans_28a.weight"] = np.zeros_like(w["motion_conv1_trans_28a.weight"])
    with torch.no_grad():
        want, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], orc.to_torch_weights(w), B, L, 0, orc.SLICE_FLAT, return_stages=True)
    h = runtime.OffForward(B, L, 0, precision="f32split"); h.load_state_dict(w)
    out = h.forward([torch.from_numpy(f).cuda() for f in feats_np]); torch.cuda.synchronize()
    f14 = h.region("fusion_14", 1056).view(P, 14, 14, 1056)[..., 800:].permute(0, 3, 1, 2).double().cpu()
    ref = st["sum_28c"].double()
    e = (f14 - ref).abs()
    bad = (e > 1e-3 * ref.abs().max())
    print(case, "sum_28c max err / max %.3e" % (e.max() / ref.abs().max()).item(), "bad elems", int(bad.sum()), "of", bad.numel(),
          "rows with bad:", sorted(set(bad.nonzero()[:, 2].tolist()))[:14], "cols:", sorted(set(bad.nonzero()[:, 3].tolist()))[:14], "chans n:", len(set(bad.nonzero()[:, 1].tolist())))
