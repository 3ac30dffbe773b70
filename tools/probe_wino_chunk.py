"""Does the polyphase 7x7 / 5x5 pipeline run faster per image on CHUNKS of the batch whose transformed input fits the 256 MiB Infinity
Cache?  Times offk_winograd_conv7x7s2 / conv5x5s2 (weight transform + input transform + GEMMs + output transform per call) at several
image counts; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime as rt  # noqa: E402

torch.manual_seed(0)
for name, fn, H, ci, co, k in (("7x7/2", rt.winograd_conv7x7s2, 28, 320, 64, 7), ("5x5/2", rt.winograd_conv5x5s2, 14, 1056, 128, 5)):
    w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
    b = torch.randn(co, device="cuda")
    for n in (24, 48, 96, 192, 384):
        x = torch.randn(n, H, H, ci, device="cuda").clamp_min(0)
        y = torch.empty(n, H // 2, H // 2, co, device="cuda")
        for _ in range(3):
            fn(x, w, b, y=y)
        torch.cuda.synchronize()
        reps = max(4, 1536 // n)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(x, w, b, y=y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%s n=%3d: %8.1f us per call, %6.3f us per image" % (name, n, dt * 1e6, dt * 1e6 / n), flush=True)
