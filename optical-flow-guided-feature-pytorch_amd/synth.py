"""Portable, counter-based synthetic inputs for the OFF hot path.

The reference ships no weights or data offline, so
parity tests, goldens and the benchmark all draw feature maps and weights from this
generator.  It is pure integer arithmetic (splitmix64 finaliser on ``seed, index``)
followed by exactly representable float conversions, so the dev container and the
GPU box regenerate bit-identical tensors and the fixtures only need to hold outputs.

* feature maps (stand-ins for ``inception_*_output_out``, RGB_OFF.py:395..590, which
  are concats of post-ReLU branches, i.e. non-negative with many zeros):
  ``max(0, z)``, ``z`` = centred sum of four 16-bit uniforms scaled by 2**-15
  (Irwin-Hall, approx N(0, 1.15**2)); about half the entries are zero.
* weights: uniform(-k, k), k = 1/sqrt(fan_in) -- the nn.Conv2d / nn.Linear default
  scale; the diagonal Sobel weight is the fixed kernel of util.py:61.
"""
import math

import numpy as np

from . import spec

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_CHUNK = 1 << 21


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def raw_u64(seed, start, count):
    """64 random bits for counters start..start+count-1 of stream ``seed``."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([seed], dtype=np.uint64) * _GOLD + _GOLD)
        idx = np.arange(start + 1, start + 1 + count, dtype=np.uint64)
        return _mix(base + idx * _GOLD)


def feature_values(seed, start, count):
    out = np.empty(count, dtype=np.float32)
    for o in range(0, count, _CHUNK):
        n = min(_CHUNK, count - o)
        x = raw_u64(seed, start + o, n)
        s = ((x & np.uint64(0xFFFF)) + ((x >> np.uint64(16)) & np.uint64(0xFFFF)) +
             ((x >> np.uint64(32)) & np.uint64(0xFFFF)) + (x >> np.uint64(48))).astype(np.int64)
        z = (s - 131070).astype(np.float32) * np.float32(2.0 ** -15)
        np.maximum(z, np.float32(0), out=out[o:o + n])
    return out


def uniform_values(seed, count, bound):
    x = raw_u64(seed, 0, count)
    u = ((x >> np.uint64(40)).astype(np.float64) * 2.0 + 1.0) * (2.0 ** -25)   # (0,1), 24 bits
    return ((2.0 * u - 1.0) * float(bound)).astype(np.float32)


def feature_seed(config_id, site_index):
    return 0x0FF0 + 16 * config_id + site_index


def make_features(batch, length, config_id=0, clip_offset=0):
    """Nine fp32 NCHW maps [B*L, C, H, H]; element i of site s is counter i of its stream.

    ``clip_offset`` lets a shard generate exactly its slice of a larger batch.
    """
    feats = []
    for si, (_name, C, H) in enumerate(spec.SITES):
        per_clip = length * C * H * H
        v = feature_values(feature_seed(config_id, si), clip_offset * per_clip, batch * per_clip)
        feats.append(v.reshape(batch * length, C, H, H))
    return feats


# ---- stress distributions for the split-precision contractions (f32split) -----------------------------------------
# feature_values() yields k * 2**-15 with k < 2**17: at most 17 significant bits, so every value splits into bf16
# hi + lo (almost) exactly and the activation split of the split-precision kernels is never stressed.  The maps below have
# full 24-bit mantissas and a realistic dynamic range (real BN-Inception taps are post-ReLU / post-max-pool and
# reach 1e1 .. 1e2); tests/test_gpu_parity.py and bench.py measure the split modes' error on them.
FEATURE_KINDS = ("synth", "full_mantissa", "heavy_tail")


def make_features_kind(batch, length, config_id=0, kind="synth", clip_offset=0):
    """kind 'synth': make_features.  'full_mantissa': the same maps times pi/3 rounded to fp32 (random low mantissa
    bits, same range).  'heavy_tail': expm1(1.151 * z) -- still ~half zeros, median ~1, tail up to 1e2."""
    feats = make_features(batch, length, config_id, clip_offset)
    if kind == "synth":
        return feats
    if kind == "full_mantissa":
        return [(f.astype(np.float64) * (math.pi / 3.0)).astype(np.float32) for f in feats]
    if kind == "heavy_tail":
        return [np.expm1(f.astype(np.float64) * 1.151).astype(np.float32) for f in feats]
    raise ValueError("unknown feature kind %r" % (kind,))


def make_weights(variant, seed=0xBEEF):
    """OrderedDict key -> fp32 ndarray for every OFF parameter of ``variant``."""
    out = {}
    shapes = spec.weight_shapes(variant)
    for li, (key, shape) in enumerate(shapes.items()):
        if key == spec.SOBEL_KEY:
            k = np.asarray(spec.DIAG_SOBEL, dtype=np.float32)
            out[key] = np.ascontiguousarray(np.broadcast_to(k, (spec.DOWN_CH, 1, 3, 3))).copy()
            continue
        base = key.rsplit(".", 1)[0] + ".weight"
        wshape = shapes[base]
        fan_in = int(np.prod(wshape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        n = int(np.prod(shape))
        out[key] = uniform_values(seed + li, n, bound).reshape(shape)
    return out


# ---- reproducible dropout mask of the OFF units' spatial branch (training) ----------
DROP_FIELD_BITS = 16


def dropout_threshold(p):
    """16-bit keep threshold: an element is kept iff its 16-bit field >= threshold."""
    return int(round(float(p) * (1 << DROP_FIELD_BITS)))


def dropout_stream(seed, site_index):
    return ((int(seed) & 0xFFFFFFFFFFFF) << 8) | int(site_index)


def dropout_keep(seed, site_index, pairs, H, p):
    """Keep-mask [P,32,H,H] (bool, NCHW like ``motion_spatial_grad_*`` output, RGB_OFF.py:611-612).

    One 64-bit draw serves the four channels of a (pixel, channel quad): draw index
    ((pair*H*H + pixel)*8 + quad), channel c of the quad uses bits [16c, 16c+16).  The HIP
    kernels (sobel_tdiff.hip / units_bwd.hip) evaluate the same integer function.
    """
    hw = H * H
    n = pairs * hw * (spec.DOWN_CH // 4)
    x = raw_u64(dropout_stream(seed, site_index), 0, n).reshape(pairs, hw, spec.DOWN_CH // 4)
    thr = np.uint64(dropout_threshold(p))
    keep = np.empty((pairs, hw, spec.DOWN_CH // 4, 4), dtype=bool)
    for c in range(4):
        keep[..., c] = ((x >> np.uint64(16 * c)) & np.uint64(0xFFFF)) >= thr
    return np.ascontiguousarray(keep.reshape(pairs, H, H, spec.DOWN_CH).transpose(0, 3, 1, 2))
