"""Times offk_batched_gemm_nt on the shapes of the forward's Winograd GEMMs (B = 64: 384 rows per point), fp32 pipe vs split-fp32:
    python tools/time_gemm.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import offk_amd  # noqa: F401
from offk_amd import _lib, runtime

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [("conv2_14a/b 128->128", 121, 384, 128, 128), ("conv3_14b 128->512", 121, 384, 128, 512),
          ("conv_trans_7 832->256", 121, 384, 832, 256), ("conv2_7 256->256", 121, 384, 256, 256),
          ("conv_trans_14 group 0 (81 x K 4224)", 81, 384, 4224, 128), ("conv_trans_14 group 1 (18 x K 2112)", 18, 384, 2112, 128)]
lib = _lib.load()
for name, batch, M, K, Co in SHAPES:
    x = torch.randn(batch, M, K, device="cuda")
    w = torch.randn(batch, Co, K, device="cuda")
    y = torch.empty(batch, M, Co, device="cuda")
    scratch = torch.empty(batch * Co * K * 6, dtype=torch.uint8, device="cuda")
    row = "%-40s" % name
    for prec in ("fp32", "f32split"):
        pid = _lib.PRECISIONS[prec]
        planes = None
        if prec == "f32split":       # pack once, time the GEMM alone through the handle-less entry's own path: pack + GEMM, minus pack
            pass
        st = torch.cuda.current_stream().cuda_stream

        def call():
            _lib.check(lib.offk_batched_gemm_nt(st, x.data_ptr(), w.data_ptr(), y.data_ptr(), batch, M, K, Co, pid, scratch.data_ptr(),
                                                scratch.numel()))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        row += "  %s %8.1f us (%6.1f TF)" % (prec, us, 2.0 * batch * M * K * Co / us * 1e-6)
    print(row, "(f32split includes packing w)")
