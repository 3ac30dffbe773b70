"""Pins oracle/off_oracle.py against the goldens captured from the reference import.

The goldens (tests/golden/*.npz) were produced by oracle/gen_golden.py running the
reference's own RGB_OFF / Flow_OFF classes on injected synthetic feature maps.  The
oracle must reproduce the reference's ATen op sequence, so on the same machine the
match is bit-for-bit; the assertion allows 2e-6 relative for a host with a different
oneDNN blocking / thread count.
"""
import glob
import os

import numpy as np
import pytest
import torch

import offk_amd  # noqa: F401
from offk_amd import spec, synth
from oracle import off_oracle as orc

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(
    os.path.join(os.path.dirname(__file__), "golden", "*_b?_l?.npz")) if not os.path.basename(p).startswith(("grad_", "ret_")))
RET_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ret_*.npz")))


def sample_idx(n, k=97):
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % n


def close(a, b, rtol=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() <= rtol * scale


def test_golden_present():
    assert len(CASES) == 7         # RGB_OFF x3, Flow_OFF x3, RGB_OFF_v2 x1
    assert len(RET_CASES) == 5


@pytest.mark.parametrize("tag", RET_CASES)
def test_oracle_return_conventions_match_reference(tag, golden_dir):
    """what RGB_OFF_forward / Flow_OFF.forward / RGB_OFF_v2.forward return (tuple order, shapes
    incl. the P == 1 squeeze, consensus, modality_fuse sum), captured from the reference import."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    ref_file = tag.split("_")[1].replace("rgbv2", "rgb_v2")
    feats = [torch.from_numpy(f) for f in synth.make_features(B, L, cfg)]
    w = orc.to_torch_weights(synth.make_weights(variant))
    fgs = torch.from_numpy(g["fgs_raw"])
    with torch.no_grad():
        ret = orc.reference_return(feats, w, B, L, ref_file, fgs, conv2="conv2")
        assert len(ret) == (4 if ref_file == "rgb_v2" else 3)
        for i in range(3):
            assert tuple(ret[i].shape) == g["ret%d" % i].shape, (i, ret[i].shape)
            assert close(ret[i].numpy(), g["ret%d" % i]), i
        if ref_file == "rgb_v2":
            assert ret[3] == "conv2" and tuple(g["ret3_shape"]) == (B * L, 192, 56, 56)
        if ref_file != "rgb":
            fused = orc.reference_return(feats, w, B, L, ref_file, fgs, modality_fuse=True)
            assert close(fused.numpy(), g["ret_fused"]) and fused.shape == (B, 101)
    if tag.endswith("b1_l2") and ref_file == "rgb":
        assert g["ret0"].shape == (101,)       # RGB_OFF.py:786: squeeze dropped the pair axis


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_golden(tag, golden_dir):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    feats = [torch.from_numpy(f) for f in synth.make_features(B, L, cfg)]
    w = orc.to_torch_weights(synth.make_weights(variant))
    with torch.no_grad():
        (fc7, fc14, fc28), st = orc.off_forward(feats, w, B, L, variant, orc.SLICE_FLAT,
                                                consensus=False, return_stages=True)
    exact = all(np.array_equal(x.numpy(), g[k]) for x, k in ((fc7, "fc7"), (fc14, "fc14"), (fc28, "fc28")))
    assert close(fc7.numpy(), g["fc7"]) and close(fc14.numpy(), g["fc14"]) and close(fc28.numpy(), g["fc28"])
    for k, t in st.items():
        cs = g["cs_" + k]
        a = t.double().reshape(-1)
        assert abs(a.sum().item() - cs[0]) <= 1e-6 * cs[1] + 1e-9, k
        assert abs(a.abs().sum().item() - cs[1]) <= 1e-6 * cs[1] + 1e-9, k
        sm = t.reshape(-1)[torch.from_numpy(sample_idx(a.numel()))].numpy()
        assert close(sm, g["sm_" + k]), k
    if variant == spec.VARIANT_FLOW:
        with torch.no_grad():
            c7, c14, c28 = orc.off_forward(feats, w, B, L, variant, orc.SLICE_FLAT)
        assert close(c7.numpy(), g["cons7"]) and close(c14.numpy(), g["cons14"]) and close(c28.numpy(), g["cons28"])
        assert c7.shape == (B, 101)
    if "full_motion_5a" in g.files:
        assert close(st["motion_5a"].numpy(), g["full_motion_5a"])
        assert close(st["sum_7"].numpy()[:, :64], g["full_sum_7"])
    print(tag, "bit-exact" if exact else "within 2e-6")


def test_q1_flat_slice_differs_from_per_clip():
    """Quirk Q1 (RGB_OFF.py:609): for B > 1 the flat slice pairs frames of different
    clips; per_clip mode must differ there and coincide for B == 1."""
    w = orc.to_torch_weights(synth.make_weights(spec.VARIANT_RGB))
    with torch.no_grad():
        f1 = [torch.from_numpy(f) for f in synth.make_features(1, 3, 7)]
        a = orc.off_forward(f1, w, 1, 3, 0, orc.SLICE_FLAT)
        b = orc.off_forward(f1, w, 1, 3, 0, orc.SLICE_PER_CLIP)
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        f2 = [torch.from_numpy(f) for f in synth.make_features(2, 3, 7)]
        a = orc.off_forward(f2, w, 2, 3, 0, orc.SLICE_FLAT)
        b = orc.off_forward(f2, w, 2, 3, 0, orc.SLICE_PER_CLIP)
        assert not torch.equal(a[0], b[0])
        # per_clip is batch-invariant: clip 0 of the pair equals the single-clip result
        c = orc.off_forward([f[:3] for f in f2], w, 1, 3, 0, orc.SLICE_PER_CLIP)
        assert np.allclose(b[0][:2].numpy(), c[0].numpy(), rtol=1e-5, atol=1e-6)
