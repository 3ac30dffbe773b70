"""Static description of the OFF sub-network: tap sites, weight keys, shapes.

Everything here is a restatement of literal shapes in the reference model files
(no code is shared with them):

* tap sites and their (C, H): RGB_OFF.py:395,418,435,458,481,504,527,567,590 (the
  nine ``inception_*_output_out`` concats) and the hard-coded ``view`` sizes at
  RGB_OFF.py:600,622,640,691,709,727,745,799,817.
* OFF-unit layers: RGB_OFF.py:265-334 (``motion_*`` declarations);
  Flow_OFF.py:51 (the shared frozen ``sobel_edge_diagonal``).
* fusion buffer channel order: RGB_OFF.py:656 (28), :760 (14), :832 (7).
"""
from collections import OrderedDict

NUM_SITES = 9
#            name  C     H
SITES = (("3a", 256, 28), ("3b", 320, 28), ("3c", 576, 14),
         ("4a", 576, 14), ("4b", 576, 14), ("4c", 608, 14), ("4d", 608, 14),
         ("5a", 1024, 7), ("5b", 1024, 7))
SITE_NAMES = tuple(s[0] for s in SITES)
# branch widths of each inception block in torch.cat order (RGB_OFF.py:395..590 and the layer
# declarations :43-263): what offk_forward_parts accepts instead of the concatenated map
SITE_PARTS = ((64, 64, 96, 32), (64, 96, 96, 64), (160, 96, 320), (224, 96, 128, 128), (192, 128, 128, 128),
              (160, 160, 160, 128), (96, 192, 192, 128), (352, 320, 224, 128), (352, 320, 224, 128))
GEN_CH = 128      # motion_conv_gen_* output channels   (RGB_OFF.py:266)
DOWN_CH = 32      # motion_spatial_down_* output channels (RGB_OFF.py:267)
UNIT_CH = GEN_CH + DOWN_CH   # one OFF unit = [spatial 32 | temporal 128] (RGB_OFF.py:616)
NUM_CLASSES = 101

VARIANT_RGB = 0    # learned depthwise 3x3 + bias, no consensus   (RGB_OFF.py)
VARIANT_FLOW = 1   # fixed diagonal Sobel, SegmentConsensus avg    (Flow_OFF.py / RGB_OFF_v2.py)
SLICE_FLAT = 0     # reference behaviour: spatial branch uses X[:B*(L-1)] (RGB_OFF.py:609)
SLICE_PER_CLIP = 1  # "what the comment says": drop each clip's last frame

# fusion buffer: (which sites land in it, carried-over channels appended after them)
FUSION = OrderedDict((
    ("28", dict(H=28, sites=("3a", "3b"), carry=0)),                       # 320 ch
    ("14", dict(H=14, sites=("3c", "4a", "4b", "4c", "4d"), carry=256)),   # 800 + 256 = 1056
    ("7", dict(H=7, sites=("5a", "5b"), carry=512)),                       # 320 + 512 = 832
))

# diagonal "Sobel" used by Flow_OFF / RGB_OFF_v2 (util.py:61), cross-correlation taps
DIAG_SOBEL = ((0.0, 1.0, 0.0), (-1.0, 0.0, 1.0), (0.0, -1.0, 0.0))

# (key, out_ch, in_ch, k, stride, pad) for every fusion conv, in forward order
FUSION_CONVS = (
    ("motion_conv_trans_28", 64, 320, 7, 2, 3),
    ("motion_conv1_trans_28a", 64, 64, 1, 1, 0),
    ("motion_conv2_trans_28a", 64, 64, 3, 1, 1),
    ("motion_conv3_trans_28a", 256, 64, 1, 1, 0),
    ("motion_conv_branch_28a", 256, 64, 1, 1, 0),
    ("motion_conv1_trans_28b", 64, 256, 1, 1, 0),
    ("motion_conv2_trans_28b", 64, 64, 3, 1, 1),
    ("motion_conv3_trans_28b", 256, 64, 1, 1, 0),
    ("motion_conv1_trans_28c", 64, 256, 1, 1, 0),
    ("motion_conv2_trans_28c", 64, 64, 3, 1, 1),
    ("motion_conv3_trans_28c", 256, 64, 1, 1, 0),
    ("motion_conv_trans_14", 128, 1056, 5, 2, 2),
    ("motion_conv1_trans_14a", 128, 128, 1, 1, 0),
    ("motion_conv2_trans_14a", 128, 128, 3, 1, 1),
    ("motion_conv3_trans_14a", 512, 128, 1, 1, 0),
    ("motion_conv_expand_trans_14a", 512, 128, 1, 1, 0),
    ("motion_conv1_trans_14b", 128, 512, 1, 1, 0),
    ("motion_conv2_trans_14b", 128, 128, 3, 1, 1),
    ("motion_conv3_trans_14b", 512, 128, 3, 1, 1),
    ("motion_conv_trans", 256, 832, 3, 1, 1),
    ("motion_conv1_trans", 256, 256, 1, 1, 0),
    ("motion_conv2_trans", 256, 256, 3, 1, 1),
    ("motion_conv3_trans", 1024, 256, 1, 1, 0),
    ("motion_conv_branch_trans", 1024, 256, 1, 1, 0),
)
HEADS = (("fc_action_motion", 1024), ("fc_action_motion_28", 256), ("fc_action_motion_14", 512))
SOBEL_KEY = "sobel_edge_diagonal.conv.weight"
# parameters of the OFF units themselves (trainable per train_off.py:39-45; the Sobel weight is frozen, util.py:72)
UNIT_PARAM_PREFIXES = ("motion_conv_gen_", "motion_spatial_down_", "motion_spatial_grad_")


def weight_shapes(variant):
    """OrderedDict state_dict-key -> shape for every OFF-sub-network parameter.

    Key names and shapes follow the reference state_dict.
    """
    d = OrderedDict()
    for name, C, _H in SITES:
        d["motion_conv_gen_%s.weight" % name] = (GEN_CH, C, 1, 1)
        d["motion_conv_gen_%s.bias" % name] = (GEN_CH,)
        d["motion_spatial_down_%s.weight" % name] = (DOWN_CH, C, 1, 1)
        d["motion_spatial_down_%s.bias" % name] = (DOWN_CH,)
        if variant == VARIANT_RGB:
            d["motion_spatial_grad_%s.weight" % name] = (DOWN_CH, 1, 3, 3)
            d["motion_spatial_grad_%s.bias" % name] = (DOWN_CH,)
    if variant == VARIANT_FLOW:
        d[SOBEL_KEY] = (DOWN_CH, 1, 3, 3)
    for key, co, ci, k, _s, _p in FUSION_CONVS:
        d[key + ".weight"] = (co, ci, k, k)
        d[key + ".bias"] = (co,)
    for key, cin in HEADS:
        d[key + ".weight"] = (NUM_CLASSES, cin)
        d[key + ".bias"] = (NUM_CLASSES,)
    return d


def feature_shapes(batch, length):
    n = batch * length
    return [(n, C, H, H) for _name, C, H in SITES]


def algorithmic_bytes_sobel_tdiff(batch, length):
    """read G + read D + write M, fp32, summed over sites."""
    hw = sum(H * H for _n, _C, H in SITES)
    return batch * hw * 4 * (GEN_CH * length + (DOWN_CH + UNIT_CH) * (length - 1))


def flops_per_clip(length):
    """(unit_flops, fusion_head_flops) per clip; 2*MACs, mirrors BASELINE.md section 3."""
    unit = 0
    for _n, C, H in SITES:
        unit += 2 * length * H * H * C * GEN_CH
        unit += 2 * (length - 1) * H * H * C * DOWN_CH
        unit += 2 * (length - 1) * H * H * DOWN_CH * 9
    out_hw = {"28": 14 * 14, "14": 7 * 7, "": 7 * 7}
    fus = 0
    for key, co, ci, k, _s, _p in FUSION_CONVS:
        if key.endswith(("_28", "_28a", "_28b", "_28c")):
            hw = out_hw["28"]
        else:
            hw = 49
        fus += 2 * (length - 1) * hw * co * ci * k * k
    for _key, cin in HEADS:
        fus += 2 * (length - 1) * cin * NUM_CLASSES
    return unit, fus
