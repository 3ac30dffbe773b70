import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import offk_amd
from offk_amd import runtime, spec, synth
from oracle import off_oracle as orc
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from test_gpu_backward import cotangents, grad_views, dev
B, L, variant = 2, 3, 0
P = B * (L - 1)
feats = synth.make_features(B, L, 2)
h = runtime.OffForward(B, L, variant, training=True); w = synth.make_weights(variant); h.load_state_dict(w)
w = orc.to_torch_weights(w)
tf = [torch.from_numpy(f) for f in feats]
ref, dm = orc.unit_backward(tf, w, B, L, variant, 0, cotangents(P), None)
df = [dev(f) for f in feats]
h.off_units(df)
flat, got = h.off_units_backward(df, grad_views(dm), 0, 0.0)
torch.cuda.synchronize()
for si in (0, 1, 2):
    site, C, H = spec.SITES[si]
    x = tf[si]
    g = torch.relu(torch.nn.functional.conv2d(x, w["motion_conv_gen_%s.weight" % site], w["motion_conv_gen_%s.bias" % site]))
    dT = dm[si][:, 32:].reshape(B, L - 1, 128, H, H)
    dG = torch.zeros(B, L, 128, H, H)
    dG[:, 1:] += dT
    dG[:, :-1] -= dT
    dG = (dG * (g.reshape(B, L, 128, H, H) > 0)).reshape(B * L, 128, H * H).permute(0, 2, 1)   # [N, HW, 128]
    mine = h.region("dG_" + site, 128).view(B * L, H * H, 128).cpu()
    err = (mine - dG).abs()
    print(site, "max err", float(err.max()), "ref max", float(dG.abs().max()))
    bad = (err > 1e-6).nonzero()
    print("  bad count", len(bad), "frames", sorted(set(bad[:, 0].tolist())), "pix range", (int(bad[:, 1].min()), int(bad[:, 1].max())) if len(bad) else None,
          "ch range", (int(bad[:, 2].min()), int(bad[:, 2].max())) if len(bad) else None)
    Gm = h.region("G_" + site, 128).view(B * L, H * H, 128).cpu()
    print("  G err", float((Gm - g.reshape(B * L, 128, H * H).permute(0, 2, 1)).abs().max()))
