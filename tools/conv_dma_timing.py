"""Per-block cycle split of the generic 1x1 kernel's LDS-DMA form (conv_igemm.hip) on the batched Winograd launches of the forward:
    OFFK_VARIANT_DIR=_ab python tools/build_variant.py ctiming -DOFFK_CONV_TIMING
    OFFK_WINO_GEMM=0 OFFK_LIB=tools/_ab/liboffk_ctiming.so python tools/conv_dma_timing.py
(OFFK_WINO_GEMM=0: since round 4 these launches run in wino_gemm.hip's persistent kernel by default; profiles/r04/conv_block_cycle_split.txt
and wino_gemm_persistent.txt hold the outputs.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime  # noqa: E402

P = 384
torch.manual_seed(0)


def dump(tag, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    os.environ["OFFK_CONV_TIMING_DUMP"] = "1"
    fn()                                   # prints + resets the warm-up sums
    os.environ.pop("OFFK_CONV_TIMING_DUMP")
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    print("== %s: next [LDS-DMA form] line = averages over 5 launches" % tag, file=sys.stderr, flush=True)
    os.environ["OFFK_CONV_TIMING_DUMP"] = "1"
    fn()
    os.environ.pop("OFFK_CONV_TIMING_DUMP")
    torch.cuda.synchronize()


for ci, co in ((832, 256), (256, 256), (128, 512), (128, 128)):
    x = torch.randn(P, 7, 7, ci, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.01
    b = torch.randn(co, device="cuda")
    dump("3x3 %d -> %d @7x7, 121 GEMMs [384 x %d] x [%d x %d]" % (ci, co, ci, ci, co), lambda: runtime.winograd_conv3x3(x, w, b))
x = torch.randn(P, 14, 14, 1056, device="cuda")
w = torch.randn(128, 1056, 5, 5, device="cuda") * 0.01
b = torch.randn(128, device="cuda")
dump("5x5/2 1056 -> 128 @14x14 (polyphase)", lambda: runtime.winograd_conv5x5s2(x, w, b))
x = torch.randn(P, 28, 28, 320, device="cuda")
w = torch.randn(64, 320, 7, 7, device="cuda") * 0.01
b = torch.randn(64, device="cuda")
dump("7x7/2 320 -> 64 @28x28 (polyphase)", lambda: runtime.winograd_conv7x7s2(x, w, b))
