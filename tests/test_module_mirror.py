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


class _FakeRuntime:
    """Stands in for runtime.OffForward on a machine without a GPU: records which weights the module pushes."""

    def __init__(self, *a, **kw):
        self.pushed = []

    def set_weight(self, key, value):
        self.pushed.append(key)


def test_changed_parameters_are_repushed_to_the_library(monkeypatch):
    """liboffk keeps packed copies of the weights: every torch-visible write must reach it before the next forward --
    in-place updates (optimizer steps, copy_), a PARENT module's load_state_dict (which never calls the sub-module's
    override), and storage replacement.  Nothing else is re-sent."""
    monkeypatch.setattr(off_module.runtime, "OffForward", _FakeRuntime)
    net = off_module.OFFSubNetwork(101, 2, 3, "rgb")
    rt = net._handle("cuda:0")
    nkeys = len(net.state_dict())
    assert len(rt.pushed) == nkeys                          # first use: everything
    rt.pushed.clear()
    net._handle("cuda:0")
    assert rt.pushed == []                                  # nothing changed, nothing re-sent
    with torch.no_grad():
        net.motion_conv_gen_4a.weight.mul_(0.5)             # what an optimizer step does
        net.fc_action_motion.bias.copy_(torch.ones(101))
    net._handle("cuda:0")
    assert sorted(rt.pushed) == ["fc_action_motion.bias", "motion_conv_gen_4a.weight"]
    rt.pushed.clear()
    parent = torch.nn.Sequential(net)                       # a wrapper loading a checkpoint: bypasses net.load_state_dict
    sd = {k: v.clone() + 1 for k, v in parent.state_dict().items()}
    parent.load_state_dict(sd)
    net._handle("cuda:0")
    assert len(rt.pushed) == nkeys
    rt.pushed.clear()
    net.motion_conv_trans_28.weight.data = torch.zeros_like(net.motion_conv_trans_28.weight)   # new storage
    net._handle("cuda:0")
    assert rt.pushed == ["motion_conv_trans_28.weight"]
    # an in-place write through .data moves neither the parameter's version counter nor its storage --
    # the documented blind spot; mark_weights_dirty() (or OFFK_ALWAYS_PUSH=1) is the remedy
    rt.pushed.clear()
    net.motion_conv_gen_3a.bias.data.mul_(2.0)
    net._handle("cuda:0")
    assert rt.pushed == []
    net.mark_weights_dirty()
    net._handle("cuda:0")
    assert len(rt.pushed) == nkeys
    rt.pushed.clear()
    monkeypatch.setenv("OFFK_ALWAYS_PUSH", "1")
    net._handle("cuda:0")
    assert len(rt.pushed) == nkeys


class _TinyBackbone(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1_7x7_s2 = torch.nn.Conv2d(3, 4, 7)
        self.last_linear = torch.nn.Linear(4, 101)


def test_bninception_off_routes_reference_checkpoint_keys(golden_dir):
    """Reference checkpoints hold backbone and OFF keys side by side at top level (model_utils.py:188-216), possibly
    under 'module.' (test_flow_off.py:52-58): OFF keys reach the OFF sub-network, the rest the backbone, and the
    result reports what nobody took."""
    want = ref_off_keys("rgb", golden_dir)
    m = off_module.bninception_off(101, 2, 3, variant="rgb", backbone=_TinyBackbone())
    sd = {"module." + k: torch.full(s, 0.5) for k, s in want.items()}
    sd["module.conv1_7x7_s2.weight"] = torch.full((4, 3, 7, 7), 2.0)
    sd["module.conv1_7x7_s2.bias"] = torch.zeros(4)
    sd["module.last_linear.weight"] = torch.zeros(101, 4)
    r = m.load_state_dict(sd, strict=False)
    assert float(m.backbone.conv1_7x7_s2.weight.mean()) == 2.0          # not left at random init
    assert float(m.off.motion_conv_gen_3a.weight.mean()) == 0.5
    assert r.missing_keys == ["backbone.last_linear.bias"] and r.unexpected_keys == []
    sd["module.not_a_layer.weight"] = torch.zeros(1)
    r = m.load_state_dict(sd, strict=False)
    assert r.unexpected_keys == ["not_a_layer.weight"]
    with pytest.raises(RuntimeError, match="unexpected"):
        m.load_state_dict(sd, strict=True)
    # 'backbone.'-prefixed names (this wrapper's own state_dict) load too
    m2 = off_module.bninception_off(101, 2, 3, variant="rgb", backbone=_TinyBackbone())
    r = m2.load_state_dict(m.state_dict())
    assert r.missing_keys == [] and r.unexpected_keys == []
    assert float(m2.backbone.conv1_7x7_s2.weight.mean()) == 2.0


def test_modality_fuse_needs_the_backbone_score(monkeypatch):
    """Flow_OFF.py:881 adds Feature_Generation_Score: without a backbone there is none -- a clear error, not None + tensor."""
    class _Rt(_FakeRuntime):
        def forward(self, feats, want28=True):
            z = torch.zeros(2, 101)
            return z, z, None
    monkeypatch.setattr(off_module.runtime, "OffForward", _Rt)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    m = off_module.bninception_off(101, 2, 3, variant="flow")
    m.modality_fuse = True
    with pytest.raises(ValueError, match="Feature_Generation_Score"):
        m([torch.zeros(s) for s in spec.feature_shapes(2, 3)])
