"""K1 / fused K1T cache-policy sweep in one process (tuning build: OFFK_LIB=tools/_bin/liboffk_tune.so).
OFFK_PW_NT bit 0 = non-temporal feature-map loads, bit 1 = non-temporal G / D / T stores."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

B, L = 64, 7
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
w = synth.make_weights(spec.VARIANT_RGB)
print("fused prec    nt | units stage ms (K1 or K1T) | K2-part ms | whole forward ms")
for fused in ("1", "0"):
    os.environ["OFFK_FUSED_UNITS"] = fused
    for prec in ("fp32",):
        h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision=prec)
        h.load_state_dict(w)
        arr = h._feat_array(feats)
        out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
        for rnd in range(2):
            for nt in (0, 1, 2, 3):
                os.environ["OFFK_PW_NT"] = str(nt)
                for _ in range(3):
                    h.forward_into(arr, *out)
                h.set_profiling(True)
                h.stage_times(reset=True)
                for _ in range(15):
                    h.forward_into(arr, *out)
                st = h.stage_times(reset=True)
                h.set_profiling(False)
                ms = {k: v[0] / v[1] for k, v in st.items()}
                print("%5s %-7s %d  | %.4f | %.4f | %.4f" % (fused, prec, nt, ms["pw_reduce"], ms["sobel_tdiff"], sum(ms.values())), flush=True)
        del h
