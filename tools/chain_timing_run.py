"""Cycle split of the fused bottleneck-chain kernel (chain_fused.hip).  Needs a liboffk compiled with -DOFFK_CHAIN_TIMING
(tools/build_variant.py chain -DOFFK_CHAIN_TIMING) loaded through OFFK_LIB."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import offk_amd
from offk_amd import runtime, spec, synth
B, L = 64, 7
h = runtime.OffForward(B, L, 0, precision="fp32"); h.load_state_dict(synth.make_weights(0))
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
arr = h._feat_array(feats)
out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
for _ in range(3): h.forward_into(arr, *out)
torch.cuda.synchronize()
os.environ["OFFK_CHAIN_TIMING_DUMP"] = "1"
for _ in range(6): h.forward_into(arr, *out); torch.cuda.synchronize()
