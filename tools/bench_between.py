"""Times offk_winograd_between (wino_mid.hip) at P images for the instantiations the forward uses (device time by HIP events).
    python tools/bench_between.py [images]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import _lib  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 384
lib = _lib.load()
torch.manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for cin, phases, gemm in ((128, 4, True), (256, 1, True), (128, 1, False)):
    M = torch.randn(121, P, cin, device="cuda")
    bias = torch.randn(cin, device="cuda")
    w1 = torch.randn(cin, cin, device="cuda") / cin ** 0.5 if gemm else None
    b1 = torch.randn(cin, device="cuda") if gemm else None
    x = torch.empty(P, 7, 7, 2 * cin, device="cuda")
    V = torch.empty(121, P, cin, device="cuda")
    for ns in (0,):      # 0 = the default split of the shape (wino_mid_launch)

        def call():
            _lib.check(lib.offk_winograd_between(st, p(M), p(bias), phases, P, cin, p(x), 2 * cin, cin, p(w1), p(b1), cin, p(V)))

        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        print("Cin %d phases %d gemm %d nsplit %d: %.1f us" % (cin, phases, gemm, ns, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
