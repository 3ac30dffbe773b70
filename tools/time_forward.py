"""One process, one library (OFFK_LIB selects an A/B build): wall-clock ms per forward and the per-launch trace.
    python tools/time_forward.py [batch] [length] [steps] [substring of the launches to list, default: all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
pat = sys.argv[4] if len(sys.argv) > 4 else ""
variant = spec.VARIANT_RGB
h = runtime.OffForward(B, L, variant, precision=os.environ.get("OFFK_PRECISION", "fp32"))
h.load_state_dict(synth.make_weights(variant))
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
arr = h._feat_array(feats)
out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
for _ in range(10):
    h.forward_into(arr, *out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    h.forward_into(arr, *out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print("lib=%s B=%d L=%d: %.4f ms / forward (%.0f clips/s), checksum %.6f" % (
    os.path.basename(os.environ.get("OFFK_LIB", "liboffk.so")), B, L, ms, B / ms * 1e3, float(out[0].double().sum())))
h.set_profiling(2)
h.launch_times(reset=True)
for _ in range(20):
    h.forward_into(arr, *out)
torch.cuda.synchronize()
tot = 0.0
for name, (t, calls) in h.launch_times().items():
    tot += t / max(calls, 1)
    if pat in name:
        print("   %-86s %8.1f us" % (name[:86], t / max(calls, 1) * 1e3))
print("   sum of launch groups %.1f us" % (tot * 1e3))
