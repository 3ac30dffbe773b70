"""Latency of the forward at small batch (BASELINE config 1: B = 1) and at the test-time shape
(B = 10 crops x L = 25 segments, test_rgb_off.py:24-25).  GPU only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

# usage: bench_small.py [B,L ...] [--fp32 | --f32split]   (default: the three shapes below, both arithmetic modes)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [(1, 7), (4, 7), (10, 25)]
precs = ("fp32",) if "--fp32" in sys.argv else ("f32split",) if "--f32split" in sys.argv else ("fp32", "f32split")
for B, L in shapes:
    for prec in precs:
        h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision=prec)
        h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
        feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 1)]
        arr = h._feat_array(feats)
        out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
        for _ in range(5):
            h.forward_into(arr, *out)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            h.forward_into(arr, *out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        # the same forward captured once into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed:
        # ~32 launches + the side-stream fork/join become one graph launch
        gdt = float("nan")
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                h.forward_into(arr, *out)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ref = [o.clone() for o in out]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                h.forward_into(arr, *out)
            for o in out:
                o.zero_()
            g.replay()
            torch.cuda.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(ref, out))
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            gdt = (time.perf_counter() - t0) / n
        except Exception as e:   # noqa: BLE001
            same = "capture failed: %s" % str(e)[:80]
        print("B=%2d L=%2d %-7s %8.3f ms/forward  %9.1f clips/s   | hipGraph replay %8.3f ms (bit-identical: %s)"
              % (B, L, prec, dt * 1e3, B / dt, gdt * 1e3, same), flush=True)
