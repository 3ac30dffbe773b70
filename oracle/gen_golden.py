"""Generate tests/golden/*.npz by running the REFERENCE's own model classes.

Runs only in the dev container (needs /root/reference; nothing on the GPU box reads
it).  The reference forward is monolithic (backbone + OFF sub-network inline in
RGB_OFF.py:360 ``RGB_OFF_forward`` / Flow_OFF.py:370 ``forward``), so the nine
feature maps are injected with forward-pre-hooks that replace the input of every
``motion_conv_gen_<s>`` with ``feat_s`` and of every ``motion_spatial_down_<s>`` with
``feat_s[:B*(L-1)]`` (the reference's own slice, RGB_OFF.py:609); the OFF outputs are
then the reference's own arithmetic on those maps.  Inputs/weights come from the
portable generator (offk_amd.synth), so the fixtures hold only outputs.

Known obstacle: basic_ops.ConsensusModule instantiates a
legacy non-static autograd.Function, which torch >= 1.5 refuses to call.  The shim
below calls the reference's own ``SegmentConsensus.forward`` method directly; the
reference files are untouched.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import spec, synth  # noqa: E402

CASES = (  # (tag, variant, B, L, config_id); tag prefix picks the reference file: rgb_ RGB_OFF, flow_ Flow_OFF, rgbv2_ RGB_OFF_v2
    ("rgb_b1_l7", spec.VARIANT_RGB, 1, 7, 1),
    ("rgb_b2_l3", spec.VARIANT_RGB, 2, 3, 2),
    ("rgb_b3_l7", spec.VARIANT_RGB, 3, 7, 3),
    ("flow_b1_l7", spec.VARIANT_FLOW, 1, 7, 1),
    ("flow_b2_l3", spec.VARIANT_FLOW, 2, 3, 2),
    ("flow_b3_l7", spec.VARIANT_FLOW, 3, 7, 3),
    ("rgbv2_b2_l3", spec.VARIANT_FLOW, 2, 3, 4),     # RGB_OFF_v2.py:613-883: the Flow arithmetic behind a 3-channel backbone
)
SAMPLE = 97  # sampled elements per stage tensor


def sample_idx(n, k=SAMPLE):
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % n


def checksum(t):
    a = t.detach().double().reshape(-1)
    idx = torch.from_numpy(sample_idx(a.numel()))
    return np.array([a.sum().item(), a.abs().sum().item()]), t.detach().reshape(-1)[idx].numpy().copy()


def ref_module(tag):
    """(reference module, input channels) for a golden tag."""
    if tag.startswith("rgbv2_"):
        import RGB_OFF_v2 as ref
        return ref, 3
    if tag.startswith("rgb_"):
        import RGB_OFF as ref
        return ref, 3
    import Flow_OFF as ref
    return ref, 10


def run_case(tag, variant, B, L, config_id):
    import basic_ops
    ref, in_ch = ref_module(tag)
    torch.manual_seed(0)
    model = ref.bninception_off(spec.NUM_CLASSES, B, L).eval()
    weights = synth.make_weights(variant)
    sd = model.state_dict()
    for k, v in weights.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    feats = [torch.from_numpy(f) for f in synth.make_features(B, L, config_id)]
    P = B * (L - 1)

    cap = {"pool_in": [], "cons_out": []}
    for (site, _c, _h), f in zip(spec.SITES, feats):
        getattr(model, "motion_conv_gen_" + site).register_forward_pre_hook(
            lambda m, inp, f=f: (f,))
        getattr(model, "motion_spatial_down_" + site).register_forward_pre_hook(
            lambda m, inp, f=f: (f[:P],))
    for key, name in (("motion_conv_trans_28", "fusion_28"), ("motion_conv_trans_14", "fusion_14"),
                      ("motion_conv_trans", "fusion_7")):
        getattr(model, key).register_forward_pre_hook(
            lambda m, inp, name=name: cap.__setitem__(name, inp[0].detach().clone()))
    for key, name in (("fc_action_motion", "fc7"), ("fc_action_motion_14", "fc14"),
                      ("fc_action_motion_28", "fc28")):
        getattr(model, key).register_forward_hook(
            lambda m, inp, out, name=name: cap.__setitem__(name, out.detach().clone()))
    model.global_pool.register_forward_pre_hook(
        lambda m, inp: cap["pool_in"].append(inp[0].detach().clone()))

    if variant == spec.VARIANT_FLOW:
        class ConsensusShim(torch.nn.Module):
            def forward(self, x):
                out = basic_ops.SegmentConsensus("avg", 1).forward(x)   # the reference's own method
                cap["cons_out"].append(out.detach().clone())
                return out
        model.consensus = ConsensusShim()

    with torch.no_grad():
        if variant == spec.VARIANT_RGB:
            ret = model.RGB_OFF_forward(torch.zeros(B * L, 3, 224, 224))
        else:
            ret = model(torch.zeros(B * L, in_ch, 224, 224))
    if tag.startswith("rgbv2_"):
        assert len(ret) == 4 and tuple(ret[3].shape) == (B * L, 192, 56, 56)     # RGB_OFF_v2.py:891
    out = dict(meta=np.array([variant, B, L, config_id], dtype=np.int64))
    out["fc7"], out["fc14"], out["fc28"] = (cap[k].numpy() for k in ("fc7", "fc14", "fc28"))
    out["ret7"], out["ret14"] = ret[0].numpy(), ret[2].numpy()
    if variant == spec.VARIANT_FLOW:
        # consensus call order in Flow_OFF.py:873-876: FGS, fc7, fc28, fc14
        out["cons7"] = cap["cons_out"][1].squeeze(1).numpy()
        out["cons28"] = cap["cons_out"][2].squeeze(1).numpy()
        out["cons14"] = cap["cons_out"][3].squeeze(1).numpy()
        assert np.array_equal(out["cons7"], out["ret7"]) and np.array_equal(out["cons14"], out["ret14"])
    else:
        assert np.array_equal(out["fc7"], out["ret7"]) and np.array_equal(out["fc14"], out["ret14"])
    # global_pool inputs in call order: inception_5b_out (FGS), maxpooled sum_28c, sum_14b, motion_sum
    stages = dict(fusion_28=cap["fusion_28"], fusion_14=cap["fusion_14"], fusion_7=cap["fusion_7"],
                  pool28=cap["pool_in"][1], sum_14b=cap["pool_in"][2], sum_7=cap["pool_in"][3])
    stages["sum_28c"] = cap["fusion_14"][:, 800:]
    off = {"28": 0, "14": 0, "7": 0}
    for name, fus in spec.FUSION.items():
        for site in fus["sites"]:
            stages["motion_" + site] = cap["fusion_" + name][:, off[name]:off[name] + spec.UNIT_CH]
            off[name] += spec.UNIT_CH
    for k, t in stages.items():
        out["cs_" + k], out["sm_" + k] = checksum(t.contiguous())
    if tag.endswith("b2_l3"):
        out["full_motion_5a"] = stages["motion_5a"].contiguous().numpy()   # [4,160,7,7]
        out["full_sum_7"] = stages["sum_7"].contiguous().numpy()[:, :64]    # [4,64,7,7]
    path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
    np.savez_compressed(path, **out)
    print(tag, "fc7", out["fc7"].shape, "->", os.path.getsize(path), "bytes")


GRAD_CASES = (  # (tag, variant, B, L, config_id, drop_seed or None)
    ("grad_rgb_b2_l3", spec.VARIANT_RGB, 2, 3, 2, None),
    ("grad_rgb_b2_l3_drop", spec.VARIANT_RGB, 2, 3, 2, 7),
    ("grad_rgb_b3_l4_drop", spec.VARIANT_RGB, 3, 4, 3, 11),
    ("grad_flow_b2_l3", spec.VARIANT_FLOW, 2, 3, 2, None),
)
DROP_P = 0.8   # RGB_OFF.py:356


def cotangents(P):
    return [torch.from_numpy(synth.uniform_values(0xC07 + i, P * spec.NUM_CLASSES, 1.0).reshape(P, spec.NUM_CLASSES))
            for i in range(3)]


def run_grad_case(tag, variant, B, L, config_id, drop_seed):
    """Training-side goldens: gradients of sum_k <fc_k, R_k> w.r.t. every
    OFF-unit parameter and w.r.t. the fusion buffers, from the reference's own graph + torch
    autograd (what train_off.py:126-146 runs).  With ``drop_seed`` the reference's single
    ``self.dropout`` module (RGB_OFF.py:356) is replaced by one that applies the repo's reproducible
    keep-mask to the nine spatial-gradient calls (:612 ...) and is the identity for the pooled
    head vectors (:785, :791, :845), so the mask placement is the reference's."""
    import basic_ops
    if variant == spec.VARIANT_RGB:
        import RGB_OFF as ref
    else:
        import Flow_OFF as ref
    torch.manual_seed(0)
    model = ref.bninception_off(spec.NUM_CLASSES, B, L).eval()
    weights = synth.make_weights(variant)
    sd = model.state_dict()
    for k, v in weights.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    feats = [torch.from_numpy(f) for f in synth.make_features(B, L, config_id)]
    P = B * (L - 1)
    cap, gcap = {}, {}
    for (site, _c, _h), f in zip(spec.SITES, feats):
        getattr(model, "motion_conv_gen_" + site).register_forward_pre_hook(lambda m, inp, f=f: (f,))
        getattr(model, "motion_spatial_down_" + site).register_forward_pre_hook(lambda m, inp, f=f: (f[:P],))
    for key, name in (("motion_conv_trans_28", "fusion_28"), ("motion_conv_trans_14", "fusion_14"),
                      ("motion_conv_trans", "fusion_7")):
        def pre(m, inp, name=name):
            inp[0].register_hook(lambda g, name=name: gcap.__setitem__(name, g.detach().clone()))
        getattr(model, key).register_forward_pre_hook(pre)
    for key, name in (("fc_action_motion", "fc7"), ("fc_action_motion_14", "fc14"), ("fc_action_motion_28", "fc28")):
        getattr(model, key).register_forward_hook(lambda m, inp, out, name=name: cap.__setitem__(name, out))
    if drop_seed is not None:
        class MaskDropout(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.calls = 0
            def forward(self, x):
                if x.dim() == 4 and x.shape[1] == spec.DOWN_CH and x.shape[2] > 1:
                    si = self.calls
                    self.calls += 1
                    keep = synth.dropout_keep(drop_seed, si, P, spec.SITES[si][2], DROP_P)
                    return x * (torch.from_numpy(keep).float() / (1.0 - DROP_P))
                return x
        model.dropout = MaskDropout()
    if variant == spec.VARIANT_FLOW:
        class ConsensusShim(torch.nn.Module):
            def forward(self, x):
                return basic_ops.SegmentConsensus("avg", 1).forward(x)
        model.consensus = ConsensusShim()
    if variant == spec.VARIANT_RGB:
        model.RGB_OFF_forward(torch.zeros(B * L, 3, 224, 224))
    else:
        model(torch.zeros(B * L, 10, 224, 224))
    if drop_seed is not None:
        assert model.dropout.calls == spec.NUM_SITES
    cot = cotangents(P)
    loss = (cap["fc7"] * cot[0]).sum() + (cap["fc14"] * cot[1]).sum() + (cap["fc28"] * cot[2]).sum()
    loss.backward()
    out = dict(meta=np.array([variant, B, L, config_id, -1 if drop_seed is None else drop_seed], dtype=np.int64))
    for k, prm in model.named_parameters():
        if k.startswith(("motion_conv_gen_", "motion_spatial_down_", "motion_spatial_grad_")):
            g = prm.grad
            out["cs_" + k], out["sm_" + k] = checksum(g)
            if g.numel() <= 512 or k == "motion_spatial_down_3c.weight":
                out["full_" + k] = g.numpy().copy()
    for name, width in (("fusion_28", 320), ("fusion_14", 800), ("fusion_7", 320)):
        out["cs_d" + name], out["sm_d" + name] = checksum(gcap[name][:, :width].contiguous())
    path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
    np.savez_compressed(path, **out)
    print(tag, "->", os.path.getsize(path), "bytes")


RET_CASES = (  # (tag, variant, B, L, config_id): what the reference's forward RETURNS
    ("ret_rgb_b2_l3", spec.VARIANT_RGB, 2, 3, 2),
    ("ret_rgb_b1_l2", spec.VARIANT_RGB, 1, 2, 5),      # P == 1: torch.squeeze drops the pair axis (RGB_OFF.py:786,792,846)
    ("ret_flow_b2_l3", spec.VARIANT_FLOW, 2, 3, 2),
    ("ret_flow_b1_l2", spec.VARIANT_FLOW, 1, 2, 5),
    ("ret_rgbv2_b2_l3", spec.VARIANT_FLOW, 2, 3, 4),
)


def run_ret_case(tag, variant, B, L, config_id):
    """Return conventions of the reference forward on injected maps: RGB_OFF.py:860 (fc7, FGS, fc14);
    Flow_OFF.py:879-884 (3-tuple, or the sum with modality_fuse); RGB_OFF_v2.py:886-891 (4-tuple).  Stores the
    returned tensors, the backbone's per-frame Feature_Generation_Score (what a stub backbone must hand back, :592-594)
    and, for the consensus variants, the fused return."""
    import basic_ops
    ref, in_ch = ref_module(tag[4:])
    torch.manual_seed(0)
    model = ref.bninception_off(spec.NUM_CLASSES, B, L).eval()
    weights = synth.make_weights(variant)
    sd = model.state_dict()
    for k, v in weights.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    feats = [torch.from_numpy(f) for f in synth.make_features(B, L, config_id)]
    P = B * (L - 1)
    cap = {}
    for (site, _c, _h), f in zip(spec.SITES, feats):
        getattr(model, "motion_conv_gen_" + site).register_forward_pre_hook(lambda m, inp, f=f: (f,))
        getattr(model, "motion_spatial_down_" + site).register_forward_pre_hook(lambda m, inp, f=f: (f[:P],))
    model.last_linear.register_forward_hook(lambda m, inp, o: cap.__setitem__("fgs_raw", o.detach().clone()))
    if variant == spec.VARIANT_FLOW:
        class ConsensusShim(torch.nn.Module):
            def forward(self, x):
                return basic_ops.SegmentConsensus("avg", 1).forward(x)   # the reference's own method
        model.consensus = ConsensusShim()
    x = torch.zeros(B * L, in_ch, 224, 224)
    out = dict(meta=np.array([variant, B, L, config_id], dtype=np.int64))
    with torch.no_grad():
        ret = model.RGB_OFF_forward(x) if variant == spec.VARIANT_RGB else model(x)
        out["fgs_raw"] = cap["fgs_raw"].numpy()
        for i in range(3):
            out["ret%d" % i] = ret[i].numpy()
        if len(ret) == 4:
            out["ret3_shape"] = np.array(ret[3].shape, dtype=np.int64)
        if variant == spec.VARIANT_FLOW:
            model.modality_fuse = True
            out["ret_fused"] = model(x).numpy()
    path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
    np.savez_compressed(path, **out)
    print(tag, [tuple(out["ret%d" % i].shape) for i in range(3)], "->", os.path.getsize(path), "bytes")


def run_consensus_backward():
    """basic_ops.py:29-36 called directly (the legacy Function object cannot be applied through autograd)."""
    import basic_ops
    B, T, C = 3, 6, spec.NUM_CLASSES
    x = torch.from_numpy(synth.uniform_values(0xC0F, B * T * C, 1.0).reshape(B, T, C))
    go = torch.from_numpy(synth.uniform_values(0xC10, B * C, 1.0).reshape(B, 1, C))
    sc = basic_ops.SegmentConsensus("avg", 1)
    y = sc.forward(x)
    gi = sc.backward(go)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "consensus_bwd.npz"), y=y.numpy(), grad_in=gi.contiguous().numpy(),
                        meta=np.array([B, T, C], dtype=np.int64))


def dump_state_dict_keys():
    """The weight-interchange contract: every key/shape of the reference state_dicts
    (data only -- names and shapes), so tests can pin the wrapper's key set offline."""
    import json
    import RGB_OFF
    import Flow_OFF
    import RGB_OFF_v2
    out = {}
    for tag, mod in (("rgb", RGB_OFF), ("flow", Flow_OFF), ("rgb_v2", RGB_OFF_v2)):
        sd = mod.bninception_off(spec.NUM_CLASSES, 1, 3).state_dict()
        out[tag] = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_dict keys:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "forward"):
        for case in CASES:
            run_case(*case)
        dump_state_dict_keys()
    if which in ("all", "ret"):
        for case in RET_CASES:
            run_ret_case(*case)
    if which == "rgbv2":
        run_case(*CASES[-1])
    if which in ("all", "grad"):
        for case in GRAD_CASES:
            run_grad_case(*case)
        run_consensus_backward()
