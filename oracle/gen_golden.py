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

CASES = (  # (tag, variant, B, L, config_id)
    ("rgb_b1_l7", spec.VARIANT_RGB, 1, 7, 1),
    ("rgb_b2_l3", spec.VARIANT_RGB, 2, 3, 2),
    ("rgb_b3_l7", spec.VARIANT_RGB, 3, 7, 3),
    ("flow_b1_l7", spec.VARIANT_FLOW, 1, 7, 1),
    ("flow_b2_l3", spec.VARIANT_FLOW, 2, 3, 2),
    ("flow_b3_l7", spec.VARIANT_FLOW, 3, 7, 3),
)
SAMPLE = 97  # sampled elements per stage tensor


def sample_idx(n, k=SAMPLE):
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % n


def checksum(t):
    a = t.detach().double().reshape(-1)
    idx = torch.from_numpy(sample_idx(a.numel()))
    return np.array([a.sum().item(), a.abs().sum().item()]), t.detach().reshape(-1)[idx].numpy().copy()


def run_case(tag, variant, B, L, config_id):
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
            ret = model(torch.zeros(B * L, 10, 224, 224))
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
    for case in CASES:
        run_case(*case)
    dump_state_dict_keys()
