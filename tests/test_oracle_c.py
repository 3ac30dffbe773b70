"""Pins the C restatement (oracle/off_oracle.c, fp64 accumulation) against the goldens
captured from the reference import, and against the PyTorch oracle.  fp32-vs-fp64
accumulation differences stay far inside the 1e-3 budget; the bound asserted is 5e-5."""
import os

import numpy as np
import pytest
import torch

import offk_amd  # noqa: F401
from offk_amd import spec, synth
from oracle import c_binding, off_oracle as orc


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("tag", ["rgb_b2_l3", "flow_b2_l3"])
def test_c_oracle_matches_reference_golden(tag, golden_dir):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    feats = synth.make_features(B, L, cfg)
    w = synth.make_weights(variant)
    ordered = [w[k] for k in spec.weight_shapes(variant)]
    o7, o14, o28, st = c_binding.forward(feats, ordered, B, L, variant, 0, False, stages=True)
    assert rel(o7, g["fc7"]) < 5e-5 and rel(o14, g["fc14"]) < 5e-5 and rel(o28, g["fc28"]) < 5e-5
    assert rel(st["fusion_7"][:, :160], g["full_motion_5a"]) < 5e-5
    assert rel(st["sum_7"][:, :64], g["full_sum_7"]) < 5e-5
    for k in ("fusion_28", "fusion_14", "fusion_7", "sum_7"):
        cs = g["cs_" + k]
        assert abs(st[k].astype(np.float64).sum() - cs[0]) < 2e-5 * cs[1], k
    if variant == spec.VARIANT_FLOW:
        c7, c14, c28 = c_binding.forward(feats, ordered, B, L, variant, 0, True)
        assert rel(c7, g["cons7"]) < 5e-5 and rel(c14, g["cons14"]) < 5e-5 and rel(c28, g["cons28"]) < 5e-5


def test_c_oracle_per_clip_mode_matches_torch_oracle():
    B, L, variant = 2, 3, spec.VARIANT_RGB
    feats = synth.make_features(B, L, 7)
    w = synth.make_weights(variant)
    ordered = [w[k] for k in spec.weight_shapes(variant)]
    o7, o14, o28 = c_binding.forward(feats, ordered, B, L, variant, 1, False)
    with torch.no_grad():
        r7, r14, r28 = orc.off_forward([torch.from_numpy(f) for f in feats], orc.to_torch_weights(w), B, L, variant,
                                       orc.SLICE_PER_CLIP)
    assert rel(o7, r7.numpy()) < 5e-5 and rel(o14, r14.numpy()) < 5e-5 and rel(o28, r28.numpy()) < 5e-5
