"""How do the forward's own kernels behave on CU-masked streams (hipExtStreamCreateWithCUMask; tools/probe_cu_mask.hip has the
mechanism)?  Run under rocprofv3 --kernel-trace --stats, one process per mode:
    PROBE_MODE=<mode> rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/probe_masked_kernels.py
modes:  t:<n>   the polyphase 7x7 / stride-2 Winograd conv of fusion@28 (input transform, 64 GEMMs, output transform) on a stream
                masked to the first n CUs (n = 0: unmasked)
        u:<n>   the OFF units (fused 1x1 reduce + temporal difference, Sobel blocks) on a stream masked to everything BUT the
                first n CUs
        b:<n>   both at once: units on the big slice, the conv on the small one
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


RAW = []


def masked_stream(lo, hi, ncu=256):
    """torch stream confined to CUs [lo, hi) in mask-bit order (bit i = XCD i % 8)."""
    words = (ctypes.c_uint32 * (ncu // 32))()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ncu // 32, words)
    assert err == 0, err
    RAW.append(s)
    return torch.cuda.ExternalStream(s.value)


mode, n = os.environ.get("PROBE_MODE", "t:0").split(":")
n = int(n)
B, L = 64, 7
P = B * (L - 1)
torch.manual_seed(0)
small = masked_stream(0, n) if n else torch.cuda.current_stream()
big = masked_stream(n, 256) if n else torch.cuda.current_stream()

x = torch.randn(P, 28, 28, 320, device="cuda")
w = torch.randn(64, 320, 7, 7, device="cuda") * 0.01
bias = torch.randn(64, device="cuda")
h = runtime.OffForward(B, L, spec.VARIANT_RGB)
h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
torch.cuda.synchronize()
REPS = 6
for _ in range(REPS):
    if mode in ("t", "b"):
        with torch.cuda.stream(small):
            runtime.winograd_conv7x7s2(x, w, bias)
    if mode in ("u", "b"):
        with torch.cuda.stream(big):
            h.off_units(feats)
    torch.cuda.synchronize()
print("done", mode, n, flush=True)
for raw in RAW:      # (a masked stream alive at interpreter exit segfaults in the profiler's finaliser)
    hip.hipStreamDestroy(raw)
