"""Split-K sweep of the two big fusion convs at small batch (per-launch trace times): tools/sweep_splitk_small.py B [B ...].  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

for B in [int(a) for a in sys.argv[1:]] or [1]:
    h = runtime.OffForward(B, 7, spec.VARIANT_RGB, precision="fp32")
    h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
    feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, 7, 1)]
    arr = h._feat_array(feats)
    out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
    for key in ("motion_conv_trans_28", "motion_conv_trans_14", "motion_conv_trans"):
        res = []
        for cfg in (3, 4):
            for sk in (0, 4, 8, 12, 16, 24, 32, 48, 64):
                try:
                    h.set_conv_plan(key, cfg if sk else -1, sk)
                except Exception as e:  # noqa: BLE001
                    res.append("%d/%d: %s" % (cfg, sk, str(e)[:30]))
                    continue
                try:
                    for _ in range(3):
                        h.forward_into(arr, *out)
                except Exception:  # noqa: BLE001   (tile does not divide Co)
                    continue
                h.set_profiling(2)
                for _ in range(20):
                    h.forward_into(arr, *out)
                torch.cuda.synchronize()
                t = [1e3 * ms / c for n, (ms, c) in h.launch_times().items() if n.startswith(key + " ") or n == key]
                h.set_profiling(0)
                res.append("%s %.1f" % ("auto" if not sk else "%d/%d" % (cfg, sk), sum(t)))
                if not sk and cfg == 4:
                    res.pop()
        h.set_conv_plan(key, -1, 0)
        print("B=%d %s: %s" % (B, key, "  ".join(res)), flush=True)
