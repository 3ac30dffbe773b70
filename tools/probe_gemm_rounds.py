"""Is the batched Winograd GEMM launch quantised by rounds of resident blocks?  The 3x3 conv of fusion@7 (832 -> 256 on 7x7 maps:
121 GEMMs of [n x 832] x [832 x 256], 64 x 64 tiles, 1280 resident blocks on 256 CUs) for a sweep of n.  Run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/probe_gemm_rounds.py
and read the conv_igemm_kernel durations in launch order (tools/probe_gemm_rounds.sh prints them)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime  # noqa: E402

CI, CO = int(os.environ.get("PROBE_CI", 832)), int(os.environ.get("PROBE_CO", 256))
NS = [int(v) for v in os.environ.get("PROBE_NS", "256,320,384,448,512,576,640,704,768").split(",")]
torch.manual_seed(0)
w = torch.randn(CO, CI, 3, 3, device="cuda") * 0.01
b = torch.randn(CO, device="cuda")
for n in NS:
    x = torch.randn(n, 7, 7, CI, device="cuda")
    for _ in range(4):
        runtime.winograd_conv3x3(x, w, b)
    torch.cuda.synchronize()
print("NS", NS)
