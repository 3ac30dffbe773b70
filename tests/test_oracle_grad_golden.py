"""Pins the oracle's training side (oracle/off_oracle.py: unit_backward, unit_param_grads_from_dm,
segment_consensus_backward, the ``drop`` multiplier of off_unit) against gradients captured from the
reference import (oracle/gen_golden.py grad: the reference's own graph, torch autograd, the reference's own
SegmentConsensus.backward).  SURVEY.md section 8(f) rank 4.
"""
import glob
import os

import numpy as np
import pytest
import torch

import offk_amd  # noqa: F401
from offk_amd import spec, synth
from oracle import off_oracle as orc

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "grad_*.npz")))
DROP_P = 0.8


def sample_idx(n, k=97):
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % n


def close(a, b, rtol=5e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-30)


def cotangents(P):
    return [torch.from_numpy(synth.uniform_values(0xC07 + i, P * spec.NUM_CLASSES, 1.0).reshape(P, spec.NUM_CLASSES))
            for i in range(3)]


def unit_drop(seed, P):
    if seed < 0:
        return None
    return [torch.from_numpy(synth.dropout_keep(seed, si, P, H, DROP_P)).float() / (1.0 - DROP_P)
            for si, (_n, _c, H) in enumerate(spec.SITES)]


def test_grad_goldens_present():
    assert len(CASES) == 4


@pytest.mark.parametrize("tag", CASES)
def test_oracle_backward_matches_reference(tag, golden_dir):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg, seed = (int(v) for v in g["meta"])
    P = B * (L - 1)
    feats = [torch.from_numpy(f) for f in synth.make_features(B, L, cfg)]
    w = orc.to_torch_weights(synth.make_weights(variant))
    drops = unit_drop(seed, P)
    grads, dm = orc.unit_backward(feats, w, B, L, variant, orc.SLICE_FLAT, cotangents(P), drops)
    assert len(grads) == (54 if variant == spec.VARIANT_RGB else 36)
    for k, t in grads.items():
        cs = g["cs_" + k]
        a = t.double().reshape(-1)
        assert abs(a.sum().item() - cs[0]) <= 2e-6 * cs[1] + 1e-9, k
        assert abs(a.abs().sum().item() - cs[1]) <= 2e-6 * cs[1] + 1e-9, k
        assert close(t.reshape(-1)[torch.from_numpy(sample_idx(a.numel()))].numpy(), g["sm_" + k]), k
        if "full_" + k in g.files:
            assert close(t.numpy(), g["full_" + k]), k
    # gradient w.r.t. the unit outputs = the leading channels of the fusion buffers' gradients
    df = {"fusion_28": torch.cat(dm[0:2], 1), "fusion_14": torch.cat(dm[2:7], 1), "fusion_7": torch.cat(dm[7:9], 1)}
    for name, t in df.items():
        cs = g["cs_d" + name]
        a = t.double().reshape(-1)
        assert abs(a.abs().sum().item() - cs[1]) <= 2e-6 * cs[1] + 1e-9, name
        assert close(t.reshape(-1)[torch.from_numpy(sample_idx(a.numel()))].numpy(), g["sm_d" + name]), name
    # the per-unit restatement (what offk_off_units_backward computes from given dM) agrees with the full graph
    g2 = orc.unit_param_grads_from_dm(feats, w, B, L, variant, orc.SLICE_FLAT, dm, drops)
    for k in grads:
        assert close(g2[k].numpy(), grads[k].numpy(), 1e-6), k


def test_dropout_mask_statistics_and_determinism():
    k1 = synth.dropout_keep(7, 0, 4, 28, DROP_P)
    k2 = synth.dropout_keep(7, 0, 4, 28, DROP_P)
    assert k1.shape == (4, 32, 28, 28) and np.array_equal(k1, k2)
    assert abs(k1.mean() - (1.0 - DROP_P)) < 0.01
    assert not np.array_equal(k1, synth.dropout_keep(7, 1, 4, 28, DROP_P))
    assert synth.dropout_keep(3, 2, 2, 14, 0.0).all()


def test_consensus_backward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "consensus_bwd.npz"))
    B, T, C = (int(v) for v in g["meta"])
    go = torch.from_numpy(synth.uniform_values(0xC10, B * C, 1.0).reshape(B, C))
    gi = orc.segment_consensus_backward(go, T)
    assert np.array_equal(gi.numpy(), g["grad_in"].reshape(B * T, C))
