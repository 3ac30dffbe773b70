"""Shape fuzz on the GPU: forward (both variants, both precisions, both slice modes) and units backward against the oracle
for random (B, L).  Test-infrastructure use of the oracle only.   python tests/tools/fuzz_shapes.py [n_cases] [seed] [max_B]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import offk_amd  # noqa: F401
from offk_amd import runtime, spec, synth
from oracle import off_oracle as orc
from test_gpu_backward import cotangents, grad_views, device_relu_masks, unit_drop

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
max_B = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


worst = 0.0
for case in range(n_cases):
    B, L = rng.randint(1, max_B), rng.randint(2, 9)
    variant = rng.choice([spec.VARIANT_RGB, spec.VARIANT_FLOW])
    prec = rng.choice(["fp32", "f32split"])
    sm = rng.choice([spec.SLICE_FLAT, spec.SLICE_PER_CLIP])
    cons = rng.choice([False, True])
    P = B * (L - 1)
    feats = synth.make_features(B, L, 5 + case)
    wnp = synth.make_weights(variant, seed=0xBEEF + case)
    w = orc.to_torch_weights(wnp)
    h = runtime.OffForward(B, L, variant, sm, cons, precision=prec, training=True)
    h.load_state_dict(wnp)
    df = [torch.from_numpy(f).cuda() for f in feats]
    tf = [torch.from_numpy(f) for f in feats]
    out = h.forward(df)
    with torch.no_grad():
        ref = orc.off_forward(tf, w, B, L, variant, sm, consensus=cons)
    e = max(rel(a, b.reshape(a.shape)) for a, b in zip(out, ref))
    # backward with dropout
    seed = case + 3
    h.off_units_train(df, seed, 0.8)
    drops = unit_drop(seed, P)
    slack = 1e-5
    g, dm = orc.unit_backward(tf, w, B, L, variant, sm, cotangents(P), drops, device_relu_masks(h, tf, w, B, L, slack))
    _flat, got = h.off_units_backward(df, grad_views(dm), seed, 0.8)
    eb = max(rel(got[k], g[k]) for k in g)
    worst = max(worst, e, eb)
    print("case %2d B=%d L=%d variant=%d prec=%-6s slice=%d cons=%d  forward %.2e  backward %.2e" % (case, B, L, variant, prec, sm, cons, e, eb), flush=True)
    assert e < 2e-4 and eb < 2e-4
print("worst", worst)
