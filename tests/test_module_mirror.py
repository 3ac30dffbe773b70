"""Host-side mirror of the reference interface (no GPU): state_dict contract, return
conventions plumbing, and loud failure when the HIP path cannot run."""
import json
import os

import pytest
import torch

import offk_amd  # noqa: F401
from offk_amd import _lib, off_module, spec


def ref_off_keys(tag, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))[tag]
    return {k: tuple(v) for k, v in ref.items() if "motion" in k or "sobel" in k}


@pytest.mark.parametrize("tag", ["rgb", "flow", "rgb_v2"])
def test_state_dict_keys_match_reference(tag, golden_dir):
    """Every OFF key of the reference state_dict (RGB_OFF.py:265-334, Flow_OFF.py:51) exists
    in the wrapper with the same shape -- reference-format checkpoints load unchanged."""
    want = ref_off_keys(tag, golden_dir)
    m = off_module.bninception_off(101, 2, 3, variant=tag)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    assert {k: tuple(s) for k, s in spec.weight_shapes(m.off.variant).items()} == want
    assert (m.batch, m.length, m.modality_fuse, m.consensus_type) == (2, 3, False, "avg")


def test_load_reference_format_checkpoint_with_module_prefix(golden_dir):
    want = ref_off_keys("flow", golden_dir)
    sd = {"module." + k: torch.full(s, 0.5) for k, s in want.items()}
    sd["module.conv1_7x7_s2.weight"] = torch.zeros(64, 10, 7, 7)       # backbone keys are ignored
    m = off_module.bninception_off(101, 2, 3, variant="flow")
    m.load_state_dict(sd)
    assert float(m.off.motion_conv_gen_3a.weight.mean()) == 0.5
    with pytest.raises(KeyError):
        m.load_state_dict({k: v for k, v in sd.items() if "fc_action_motion_14" not in k})


def test_sobel_weight_is_the_fixed_diagonal_kernel():
    m = off_module.OFFSubNetwork(variant="flow")
    w = m.sobel_edge_diagonal.conv.weight
    assert not w.requires_grad and w.shape == (32, 1, 3, 3)
    assert w[7, 0].tolist() == [[0.0, 1.0, 0.0], [-1.0, 0.0, 1.0], [0.0, -1.0, 0.0]]   # util.py:61


def test_no_cpu_path():
    m = off_module.bninception_off(101, 1, 3)
    feats = [torch.zeros(s) for s in spec.feature_shapes(1, 3)]
    with pytest.raises(_lib.OffkError):
        m.RGB_OFF_forward(feats)


@pytest.mark.parametrize("tag", ["rgb", "flow"])
def test_off_units_trainable_mirror_keys(tag, golden_dir):
    """OFFUnits (training side): its parameters are exactly the reference's unit tensors that
    train_off.py:39-45 leaves trainable, under the reference's keys; the Sobel weight of the Flow variant is frozen
    (util.py:72); no CPU path."""
    want = {k: s for k, s in ref_off_keys(tag, golden_dir).items()
            if k.startswith(spec.UNIT_PARAM_PREFIXES) or k == spec.SOBEL_KEY}
    m = off_module.OFFUnits(2, 3, tag)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    assert trainable == {k for k in want if k != spec.SOBEL_KEY}
    assert m.param_keys == [k for k in spec.weight_shapes(m.variant) if k.startswith(spec.UNIT_PARAM_PREFIXES)]
    with pytest.raises(_lib.OffkError):
        m([torch.zeros(s) for s in spec.feature_shapes(2, 3)])
    # a whole reference-format checkpoint (with the DataParallel prefix) loads: foreign keys are ignored
    full = {"module." + k: torch.full(s, 0.25) for k, s in ref_off_keys(tag, golden_dir).items()}
    full["module.conv1_7x7_s2.weight"] = torch.zeros(64, 3, 7, 7)
    m.load_state_dict(full)
    assert float(m.motion_conv_gen_5b.weight.mean()) == 0.25
    with pytest.raises(KeyError):
        m.load_state_dict({k: v for k, v in full.items() if "motion_spatial_down_3c" not in k})
