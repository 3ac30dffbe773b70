"""Cycle split of the fused units kernel (pw_tdiff.hip).  Needs a liboffk whose pw_tdiff.hip and offk_api.hip were compiled
with -DOFFK_PT_TIMING (second .so, loaded through OFFK_LIB) and OFFK_FUSED_UNITS=1."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import offk_amd
from offk_amd import runtime, spec, synth
B, L = 64, 7
h = runtime.OffForward(B, L, 0, precision=os.environ.get("PT_PREC", "fp32")); h.load_state_dict(synth.make_weights(0))
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
arr = h._feat_array(feats)
out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
for _ in range(3): h.forward_into(arr, *out)
torch.cuda.synchronize()
os.environ["OFFK_PT_TIMING_DUMP"] = "1"; h.forward_into(arr, *out); os.environ.pop("OFFK_PT_TIMING_DUMP")
for _ in range(4): h.forward_into(arr, *out)
torch.cuda.synchronize()
print("sums over 5 launches:", flush=True)
os.environ["OFFK_PT_TIMING_DUMP"] = "1"; h.forward_into(arr, *out)
torch.cuda.synchronize()
