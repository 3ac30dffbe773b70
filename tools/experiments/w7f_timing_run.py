"""Cycle split of wino7_fused_kernel: needs a liboffk whose winograd7_fused.hip was compiled with -DOFFK_W7F_TIMING (OFFK_LIB)."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import offk_amd  # noqa: F401
from offk_amd import runtime as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
x = torch.randn(n, 28, 28, 320, device="cuda").clamp_min(0)
w = torch.randn(64, 320, 7, 7, device="cuda") / 125.0
b = torch.randn(64, device="cuda")
y = torch.empty(n, 14, 14, 64, device="cuda")
for _ in range(3):
    rt.winograd_conv7x7s2(x, w, b, y=y, fused=True)
torch.cuda.synchronize()
os.environ["OFFK_W7F_TIMING_DUMP"] = "1"
rt.winograd_conv7x7s2(x, w, b, y=y, fused=True)      # dumps the sums of the launches so far, resets
os.environ.pop("OFFK_W7F_TIMING_DUMP")
for _ in range(4):
    rt.winograd_conv7x7s2(x, w, b, y=y, fused=True)
torch.cuda.synchronize()
print("average over 5 launches:", flush=True)
os.environ["OFFK_W7F_TIMING_DUMP"] = "1"
rt.winograd_conv7x7s2(x, w, b, y=y, fused=True)
torch.cuda.synchronize()
