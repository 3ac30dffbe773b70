import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import offk_amd
from offk_amd import runtime, spec, synth
B, L = 64, 7
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
w = synth.make_weights(0)
h = runtime.OffForward(B, L, 0, precision="bf16x3"); h.load_state_dict(w)
for i in range(5):
    o = h.forward(feats); torch.cuda.synchronize()
    if i < 3: continue
    for nm, oi, cc in (("14", 1, 512), ("28", 2, 256)):
        fw = w["fc_action_motion_%s.weight" % nm].astype(np.float64); fb = w["fc_action_motion_%s.bias" % nm].astype(np.float64)
        got = o[oi].cpu().numpy().astype(np.float64)
        p = h.region("pooled_" + nm, cc).cpu().numpy().astype(np.float64)
        full = p @ fw.T + fb
        err = np.abs(got - full); bad = err > 1e-4
        r, c = np.nonzero(bad)
        print(i, "head", nm, "bad", int(bad.sum()), "max %.4f" % err.max(), "class%8 hist:", np.bincount(c % 8, minlength=8), flush=True)
        if i == 4:
            os.makedirs("gpurun_out", exist_ok=True)
            np.save("gpurun_out/dbg_got_%s.npy" % nm, got); np.save("gpurun_out/dbg_p_%s.npy" % nm, p)
