"""Per-launch trace of one forward shape (offk_set_profiling(h, 2)): tools/launch_trace.py B L [fp32|f32split] [steps].  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

B, L = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision=prec)
h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 1)]
arr = h._feat_array(feats)
out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
for _ in range(3):
    h.forward_into(arr, *out)
h.set_profiling(2)
for _ in range(steps):
    h.forward_into(arr, *out)
torch.cuda.synchronize()
tot = 0.0
for name, (ms, calls) in h.launch_times().items():
    print("%-78s %8.1f us" % (name[:78], 1e3 * ms / calls))
    tot += ms / calls
print("sum %.3f ms" % tot)
