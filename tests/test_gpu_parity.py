"""GPU parity tests (run with -m gpu on an MI355X): every check goes through the C ABI of
liboffk.so and is compared with the CPU oracle (oracle/off_oracle.py), plain PyTorch fp32
ops on the CPU, and the goldens captured from the reference import.

Tolerance: BASELINE.json north_star asks for 1e-3 relative fp32.  The HIP path computes
in exact fp32 (v_mfma_f32_32x32x2_f32 is an fmaf chain), so the tests hold it to
RTOL = 2e-4 of the tensor's max magnitude -- five times tighter than the stated bar --
so that a wrong bias / tap / channel offset cannot hide inside the budget.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import offk_amd  # noqa: F401
from offk_amd import spec, synth
from oracle import off_oracle as orc

pytestmark = pytest.mark.gpu
RTOL_NORTH_STAR = 1e-3
RTOL = 2e-4
PRECISIONS = ["fp32"]   # (until round 4 also "bf16x3", the two-plane split mode: retired in ABI v9 in favour of the exact three-plane mode)
# handle-level arithmetic modes: f32split = fp32 operands as three bf16 planes on the bf16 pipe (round 5; its own error tests:
# tests/test_gpu_split.py); kernels without a split form run as in fp32
HANDLE_PRECISIONS = PRECISIONS + ["f32split"]


def rel_err(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def signal_err(a, b):
    """Error of the ROW-TO-ROW signal: the logits are bias-dominated under default init (the part that
    differs between rows is ~2.6 % of their magnitude), so rel_err admits ~1 % of the actual signal.  This removes the
    per-class mean over rows from both sides and normalises by the signal's own maximum."""
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(b).double()
    assert a.shape == b.shape and a.shape[0] >= 2, (a.shape, b.shape)
    da, db = a - a.mean(0, keepdim=True), b - b.mean(0, keepdim=True)
    return ((da - db).abs().max() / (db.abs().max() + 1e-30)).item()


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from offk_amd import runtime
    return runtime


def dev(x):
    return torch.as_tensor(x).to("cuda").contiguous()


def nhwc(x):  # logical NCHW cpu tensor -> channels-last physical, on device
    return x.permute(0, 2, 3, 1).contiguous().to("cuda")


def test_layout_helpers(rt):
    torch.manual_seed(0)
    x = torch.randn(3, 37, 7, 5)
    y = rt.nchw_to_nhwc(dev(x))
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    z = rt.nhwc_to_nchw(y, coff=5, c=20)
    assert torch.equal(z.cpu(), x[:, 5:25].contiguous())


CONV_CASES = [  # (Ci, Co, k, stride, pad, H, n_img)  -- the distinct shapes of spec.FUSION_CONVS at small size
    (320, 64, 7, 2, 3, 28, 2), (64, 64, 1, 1, 0, 14, 3), (64, 64, 3, 1, 1, 14, 2), (64, 256, 1, 1, 0, 14, 2),
    (256, 64, 1, 1, 0, 14, 2), (1056, 128, 5, 2, 2, 14, 2), (128, 128, 3, 1, 1, 7, 5), (128, 512, 1, 1, 0, 7, 3),
    (512, 128, 1, 1, 0, 7, 3), (128, 512, 3, 1, 1, 7, 3), (832, 256, 3, 1, 1, 7, 2), (256, 1024, 1, 1, 0, 7, 3),
]


@pytest.mark.parametrize("prec", [0])
@pytest.mark.parametrize("Ci,Co,k,stride,pad,H,n", CONV_CASES)
def test_conv2d_vs_torch(rt, Ci, Co, k, stride, pad, H, n, prec):
    g = torch.Generator().manual_seed(Ci * 7 + Co + k)
    x = torch.randn(n, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    y = rt.conv2d_nhwc(nhwc(x), dev(w), dev(b), stride, pad, precision=prec)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < RTOL


def test_conv2d_epilogues_and_slices(rt):
    from offk_amd import _lib
    g = torch.Generator().manual_seed(5)
    n, H, Ci, Co = 3, 7, 64, 128
    xs = torch.randn(n, 96, H, H, generator=g)          # conv reads channels 32..95 of a 96-channel buffer
    x = xs[:, 32:96]
    w = torch.randn(Co, Ci, 3, 3, generator=g) / 24.0
    b = torch.randn(Co, generator=g)
    res = torch.randn(n, Co, H, H, generator=g)
    cases = {
        0: F.conv2d(x, w, b, padding=1),
        _lib.CONV_RELU_IN | _lib.CONV_RELU_PRE: torch.relu(F.conv2d(torch.relu(x), w, b, padding=1)),
        _lib.CONV_RELU_POST: torch.relu(F.conv2d(x, w, b, padding=1) + res),
        _lib.CONV_RELU_PRE | _lib.CONV_RELU_POST: torch.relu(torch.relu(F.conv2d(x, w, b, padding=1)) + res),
    }
    for flags, ref in cases.items():
        use_res = bool(flags & _lib.CONV_RELU_POST)
        ybuf = torch.full((n, H, H, Co + 64), 7.0, device="cuda")     # write channels 32..159 of a wider buffer
        rt.conv2d_nhwc(nhwc(xs), dev(w), dev(b), 1, 1, res=nhwc(res) if use_res else None, flags=flags,
                       x_coff=32, y=ybuf, y_coff=32)
        assert rel_err(ybuf[..., 32:32 + Co].permute(0, 3, 1, 2), ref) < RTOL, flags
        assert torch.all(ybuf[..., :32] == 7.0) and torch.all(ybuf[..., 32 + Co:] == 7.0)
    # residual add without any ReLU (fusion@7 tail, RGB_OFF.py:841)
    y = rt.conv2d_nhwc(nhwc(x.contiguous()), dev(w), dev(b), 1, 1, res=nhwc(res), flags=0)
    assert rel_err(y.permute(0, 3, 1, 2), F.conv2d(x, w, b, padding=1) + res) < RTOL


@pytest.mark.parametrize("prec", [0])
def test_conv2d_input_slab_beyond_31_bit_offsets(rt, prec):
    """The buffer-descriptor loader is used only while every byte offset fits 31 bits; a slice of a > 2 GiB slab must
    take the 64-bit pointer loader and give the same answer (the last images sit beyond the 2 GiB mark)."""
    g = torch.Generator().manual_seed(11)
    n, H, cs, Ci, Co, coff = 46, 56, 4096, 32, 64, 2048
    assert n * H * H * cs * 4 > 2 ** 31
    x = torch.randn(n, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    slab = torch.empty(n, H, H, cs, device="cuda")
    slab[..., coff:coff + Ci] = nhwc(x)
    y = rt.conv2d_nhwc(slab, dev(w), dev(b), 1, 1, x_coff=coff, ci=Ci, precision=prec)
    del slab
    ref = F.conv2d(x, w, b, padding=1)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < RTOL
    assert rel_err(y[-1:].permute(0, 3, 1, 2), ref[-1:]) < RTOL


@pytest.mark.parametrize("prec", [0])
@pytest.mark.parametrize("cfg,splitk", [(0, 1), (1, 3), (2, 2), (3, 4), (4, 1), (5, 2), (-1, 0)])
def test_conv2d_tile_plans_and_splitk(rt, cfg, splitk, prec):
    """Every tile configuration and the deterministic split-K reduction give the same conv."""
    from offk_amd import _lib
    g = torch.Generator().manual_seed(77)
    n, H, Ci, Co = 5, 7, 128, 256
    x = torch.randn(n, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / 34.0
    b = torch.randn(Co, generator=g)
    res = torch.randn(n, Co, H, H, generator=g)
    ref = torch.relu(torch.relu(F.conv2d(x, w, b, padding=1)) + res)
    y = rt.conv2d_nhwc(nhwc(x), dev(w), dev(b), 1, 1, res=nhwc(res), flags=_lib.CONV_RELU_PRE | _lib.CONV_RELU_POST,
                       tile_cfg=cfg, splitk=splitk, precision=prec)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < RTOL
    y2 = rt.conv2d_nhwc(nhwc(x), dev(w), dev(b), 1, 1, res=nhwc(res), flags=_lib.CONV_RELU_PRE | _lib.CONV_RELU_POST,
                        tile_cfg=cfg, splitk=splitk, precision=prec)
    assert torch.equal(y, y2)      # split-K sums slabs in a fixed order: bit-reproducible


PATCH_CASES = [  # (k, stride, H, Ci, Co, n_img, cfg, splitk): the four shapes of the LDS-patch kernel, partial last groups
    (7, 2, 28, 320, 64, 3, 7, 1), (7, 2, 28, 64, 128, 2, 6, 2), (5, 2, 14, 1056, 128, 6, 6, 3), (5, 2, 14, 96, 64, 9, 7, 1),
    (3, 1, 14, 64, 64, 3, 7, 2), (3, 1, 14, 128, 256, 2, 6, 1), (3, 1, 7, 832, 256, 7, 6, 4), (3, 1, 7, 128, 128, 5, 7, 1),
    (3, 1, 7, 256, 256, 8, 6, 3),
]


@pytest.mark.parametrize("prec", [0])
@pytest.mark.parametrize("k,stride,H,Ci,Co,n,cfg,splitk", PATCH_CASES)
def test_conv2d_patch_kernel_vs_torch(rt, k, stride, H, Ci, Co, n, cfg, splitk, prec):
    """tile_cfg 6 / 7: the input patch of a 196-pixel output group stays in LDS across the taps (conv_igemm.hip).
    Same conv, same epilogue flags, channel-sliced input / output views, bit-reproducible."""
    from offk_amd import _lib
    g = torch.Generator().manual_seed(k * 1000 + Ci + Co + n)
    pad = k // 2
    xs = torch.randn(n, Ci + 32, H, H, generator=g)          # the conv reads channels 32.. of a wider buffer
    x = xs[:, 32:]
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    Ho = (H + 2 * pad - k) // stride + 1
    res = torch.randn(n, Co, Ho, Ho, generator=g)
    flags = _lib.CONV_RELU_IN | _lib.CONV_RELU_PRE | _lib.CONV_RELU_POST
    ref = torch.relu(torch.relu(F.conv2d(torch.relu(x), w, b, stride=stride, padding=pad)) + res)
    ybuf = torch.full((n, Ho, Ho, Co + 64), 7.0, device="cuda")
    rt.conv2d_nhwc(nhwc(xs), dev(w), dev(b), stride, pad, res=nhwc(res), flags=flags, x_coff=32, ci=Ci, y=ybuf, y_coff=32,
                   tile_cfg=cfg, splitk=splitk, precision=prec)
    assert rel_err(ybuf[..., 32:32 + Co].permute(0, 3, 1, 2), ref) < RTOL
    assert torch.all(ybuf[..., :32] == 7.0) and torch.all(ybuf[..., 32 + Co:] == 7.0)
    y2 = torch.full_like(ybuf, 7.0)
    rt.conv2d_nhwc(nhwc(xs), dev(w), dev(b), stride, pad, res=nhwc(res), flags=flags, x_coff=32, ci=Ci, y=y2, y_coff=32,
                   tile_cfg=cfg, splitk=splitk, precision=prec)
    assert torch.equal(ybuf, y2)
    # plain conv (no flags, no residual) through the same kernel
    y3 = rt.conv2d_nhwc(nhwc(x.contiguous()), dev(w), dev(b), stride, pad, tile_cfg=cfg, splitk=1, precision=prec)
    assert rel_err(y3.permute(0, 3, 1, 2), F.conv2d(x, w, b, stride=stride, padding=pad)) < RTOL


def test_conv2d_patch_kernel_rejects_other_shapes(rt):
    from offk_amd import _lib
    x = torch.randn(2, 10, 10, 64, device="cuda")
    w = torch.randn(64, 64, 3, 3)
    with pytest.raises(_lib.OffkError, match="patch kernel"):
        rt.conv2d_nhwc(x, dev(w), None, 1, 1, tile_cfg=7, splitk=1, precision=0)
    with pytest.raises(_lib.OffkError, match="patch kernel"):
        rt.conv2d_nhwc(torch.randn(2, 7, 7, 64, device="cuda"), dev(w), None, 1, 1, tile_cfg=10, splitk=1, precision=0)


def test_head_and_consensus(rt):
    g = torch.Generator().manual_seed(9)
    for C, H, mp in ((256, 14, True), (512, 7, False), (1024, 7, False)):
        x = torch.randn(5, C, H, H, generator=g)
        w = torch.randn(101, C, generator=g) / C ** 0.5
        b = torch.randn(101, generator=g)
        ref = orc.head(x, {"k.weight": w, "k.bias": b}, "k", mp)
        out = rt.head(nhwc(x), dev(w), dev(b), mp)
        assert rel_err(out, ref) < RTOL
    x = torch.randn(4 * 6, 101, generator=g)
    assert rel_err(rt.segment_consensus(dev(x), 4), orc.segment_consensus(x, 4)) < 1e-6


def make_handle(rt, B, L, variant, slice_mode=spec.SLICE_FLAT, consensus=None, weights=None, precision="fp32"):
    h = rt.OffForward(B, L, variant, slice_mode, consensus, precision=precision)
    w = synth.make_weights(variant) if weights is None else weights
    assert h.load_state_dict(w) == []
    assert h.missing_weights()[0] == 0
    return h, orc.to_torch_weights(w)


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("slice_mode", [spec.SLICE_FLAT, spec.SLICE_PER_CLIP])
@pytest.mark.parametrize("site", range(spec.NUM_SITES))
def test_pw_reduce_vs_oracle(rt, site, slice_mode, prec):
    B, L = 2, 3
    h, w = make_handle(rt, B, L, spec.VARIANT_RGB, slice_mode, precision=prec)
    name, C, H = spec.SITES[site]
    x = torch.from_numpy(synth.make_features(B, L, 4)[site])
    G, D = h.pw_reduce(site, dev(x))
    g_ref = torch.relu(F.conv2d(x, w["motion_conv_gen_%s.weight" % name], w["motion_conv_gen_%s.bias" % name]))
    d_ref = F.conv2d(orc.spatial_frames(x, B, L, slice_mode), w["motion_spatial_down_%s.weight" % name],
                     w["motion_spatial_down_%s.bias" % name])
    assert rel_err(G.view(B * L, H, H, 128).permute(0, 3, 1, 2), g_ref) < RTOL
    assert rel_err(D.view(B * (L - 1), H, H, 32).permute(0, 3, 1, 2), d_ref) < RTOL


@pytest.mark.parametrize("algo", [0, 1, 4])      # temporal difference by register rotation / wavefront shuffle / flat shifted stream
@pytest.mark.parametrize("variant", [spec.VARIANT_RGB, spec.VARIANT_FLOW])
@pytest.mark.parametrize("site", [0, 2, 7])
def test_sobel_tdiff_vs_oracle(rt, site, variant, algo):
    B, L = 2, 4
    h, w = make_handle(rt, B, L, variant)
    name, _C, H = spec.SITES[site]
    g = torch.Generator().manual_seed(site + 10 * variant)
    G = torch.relu(torch.randn(B * L, 128, H, H, generator=g))
    D = torch.randn(B * (L - 1), 32, H, H, generator=g)
    t_ref = orc.temporal_diff(G, B)
    if variant == spec.VARIANT_RGB:
        s_ref = F.conv2d(D, w["motion_spatial_grad_%s.weight" % name], w["motion_spatial_grad_%s.bias" % name],
                         padding=1, groups=32)
    else:
        s_ref = F.conv2d(D, w[spec.SOBEL_KEY], None, padding=1, groups=32)
    M = torch.full((B * (L - 1) * H * H, 352), -3.0, device="cuda")
    h.sobel_tdiff(site, nhwc(G).view(-1, 128), nhwc(D).view(-1, 32), M, 160, algo)
    Mv = M.view(B * (L - 1), H, H, 352).permute(0, 3, 1, 2)
    assert rel_err(Mv[:, 160:192], s_ref) < RTOL
    assert rel_err(Mv[:, 192:320], t_ref) < 1e-6          # a single fp32 subtraction: exact
    assert torch.all(Mv[:, :160] == -3.0) and torch.all(Mv[:, 320:] == -3.0)


GOLDEN = ["rgb_b1_l7", "rgb_b2_l3", "rgb_b3_l7", "flow_b1_l7", "flow_b2_l3", "flow_b3_l7", "rgbv2_b2_l3"]


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("tag", GOLDEN)
def test_forward_vs_golden_and_oracle(rt, tag, golden_dir, prec):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    h, w = make_handle(rt, B, L, variant, consensus=False, precision=prec)
    feats_np = synth.make_features(B, L, cfg)
    out7, out14, out28 = h.forward([dev(f) for f in feats_np])
    torch.cuda.synchronize()
    # (1) the reference's own outputs (captured by oracle/gen_golden.py)
    for out, key in ((out7, "fc7"), (out14, "fc14"), (out28, "fc28")):
        assert rel_err(out, g[key]) < RTOL, key
    # (2) the oracle, including every fusion-stage intermediate
    with torch.no_grad():
        (r7, r14, r28), st = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, variant,
                                             orc.SLICE_FLAT, consensus=False, return_stages=True)
    assert rel_err(out7, r7) < RTOL and rel_err(out14, r14) < RTOL and rel_err(out28, r28) < RTOL
    P = B * (L - 1)
    for name, ch, H in (("fusion_28", 320, 28), ("fusion_14", 1056, 14), ("fusion_7", 832, 7), ("sum_7", 1024, 7)):
        got = h.region(name, ch).view(P, H, H, ch).permute(0, 3, 1, 2)
        assert rel_err(got, st[name]) < RTOL, name
    if "full_motion_5a" in g.files:
        got = h.region("fusion_7", 832).view(P, 7, 7, 832).permute(0, 3, 1, 2)[:, :160]
        assert rel_err(got, g["full_motion_5a"]) < RTOL
    # logits are bias-dominated under default init: also pin the row-to-row signal, at the
    # north_star tolerance (the signal is ~2.6 % of the logit magnitude, so this is ~40x stricter than the check above)
    for out, key in ((out7, "fc7"), (out14, "fc14"), (out28, "fc28")):
        if B * (L - 1) < 2:
            continue
        d = (out - out.mean(0, keepdim=True)).cpu().double()
        dr = torch.from_numpy(g[key]).double()
        dr = dr - dr.mean(0, keepdim=True)
        sig_err = ((d - dr).abs().max() / dr.abs().max()).item()
        print("%s %s %s: row-to-row signal error %.2e" % (tag, prec, key, sig_err))
        assert sig_err < RTOL_NORTH_STAR, (key, sig_err)


@pytest.mark.parametrize("tag", ["flow_b2_l3", "flow_b3_l7"])
def test_forward_consensus_vs_golden(rt, tag, golden_dir):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    h, _w = make_handle(rt, B, L, variant)          # Flow default: consensus avg
    out7, out14, out28 = h.forward([dev(f) for f in synth.make_features(B, L, cfg)])
    assert out7.shape == (B, 101)
    for out, key in ((out7, "cons7"), (out14, "cons14"), (out28, "cons28")):
        assert rel_err(out, g[key]) < RTOL, key


def test_per_clip_mode_is_batch_invariant(rt):
    """per_clip slice mode: clip b of a batch equals the single-clip forward (the property
    that makes clip sharding exact); reference_flat mode is NOT (quirk Q1)."""
    B, L = 3, 4
    feats = synth.make_features(B, L, 6)
    h, w = make_handle(rt, B, L, spec.VARIANT_RGB, spec.SLICE_PER_CLIP)
    out7, out14, _ = h.forward([dev(f) for f in feats])
    h1, _ = make_handle(rt, 1, L, spec.VARIANT_RGB, spec.SLICE_PER_CLIP)
    for b in range(B):
        o7, o14, _ = h1.forward([dev(f[b * L:(b + 1) * L]) for f in feats])
        assert rel_err(out7[b * (L - 1):(b + 1) * (L - 1)], o7) < 1e-5
        assert rel_err(out14[b * (L - 1):(b + 1) * (L - 1)], o14) < 1e-5
    with torch.no_grad():
        r7, r14, _ = orc.off_forward([torch.from_numpy(f) for f in feats], w, B, L, 0, orc.SLICE_PER_CLIP)
    assert rel_err(out7, r7) < RTOL and rel_err(out14, r14) < RTOL


def test_nhwc_feature_layout(rt):
    B, L = 2, 3
    feats = synth.make_features(B, L, 8)
    h, _ = make_handle(rt, B, L, spec.VARIANT_RGB)
    a = h.forward([dev(f) for f in feats])
    h2 = rt.OffForward(B, L, spec.VARIANT_RGB, feat_layout=1)
    h2.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
    b = h2.forward([nhwc(torch.from_numpy(f)) for f in feats])
    for x, y in zip(a, b):
        assert rel_err(x, y) < 1e-5


def test_missing_weight_fails_loudly(rt):
    from offk_amd import _lib
    h = rt.OffForward(1, 3, spec.VARIANT_RGB)
    feats = [dev(f) for f in synth.make_features(1, 3, 0)]
    with pytest.raises(_lib.OffkError, match="weight not set"):
        h.forward(feats)
    with pytest.raises(_lib.OffkError, match="shape mismatch"):
        h.set_weight("fc_action_motion.weight", np.zeros((101, 5), dtype=np.float32))
    with pytest.raises(_lib.OffkError, match="not an OFF"):
        h.set_weight("conv1_7x7_s2.weight", np.zeros((64, 3, 7, 7), dtype=np.float32))


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
def test_full_size_b64_vs_oracle(rt, prec):
    """BASELINE config 2 (RGB_OFF, B=64, L=7) against the oracle at full size, both arithmetic modes."""
    B, L = 64, 7
    feats = synth.make_features(B, L, 2)
    h, w = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    out7, out14, out28 = h.forward([dev(f) for f in feats])
    torch.cuda.synchronize()
    with torch.no_grad():
        r7, r14, r28 = orc.off_forward([torch.from_numpy(f) for f in feats], w, B, L, 0, orc.SLICE_FLAT)
    assert rel_err(out7, r7) < RTOL and rel_err(out14, r14) < RTOL and rel_err(out28, r28) < RTOL
    for name, a, b in (("fc7", out7, r7), ("fc14", out14, r14), ("fc28", out28, r28)):   # north_star tolerance on the signal
        se = signal_err(a, b)
        print("full size RGB B=64 %s %s: row-to-row signal error %.2e" % (prec, name, se))
        assert se < RTOL_NORTH_STAR, (name, se)
    # determinism: same inputs, same bits
    o7b, _, _ = h.forward([dev(f) for f in feats])
    assert torch.equal(out7, o7b)


def test_two_stream_fused_forward(rt):
    """BASELINE config 5 at small size: RGB-OFF + Flow-OFF on two streams, K7 late fusion."""
    from offk_amd import scores, two_stream
    B, L = 2, 3
    wr, wf = synth.make_weights(spec.VARIANT_RGB), synth.make_weights(spec.VARIANT_FLOW, seed=0xF10)
    fr, ff = synth.make_features(B, L, 11), synth.make_features(B, L, 12)
    ts = two_stream.TwoStreamOFF(B, L, precision="fp32")
    ts.load_state_dicts(wr, wf)
    fused, pred = ts.forward([dev(f) for f in fr], [dev(f) for f in ff])
    with torch.no_grad():
        r = orc.off_forward([torch.from_numpy(f) for f in fr], orc.to_torch_weights(wr), B, L, spec.VARIANT_RGB, consensus=True)
        f = orc.off_forward([torch.from_numpy(x) for x in ff], orc.to_torch_weights(wf), B, L, spec.VARIANT_FLOW, consensus=True)
    w = scores.FUSION_BEST
    ref = w[0] * r[0] + w[2] * r[1] + w[3] * f[0] + w[5] * f[1]
    assert rel_err(fused, ref) < RTOL
    assert torch.equal(pred.cpu().long(), ref.argmax(dim=1))


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
def test_forward_from_inception_branch_parts(rt, prec):
    """offk_forward_parts: the maps handed over as the inception branches (before torch.cat) give
    bit-identical results to the concatenated maps."""
    B, L = 2, 3
    feats = synth.make_features(B, L, 13)
    h, _ = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    ref = h.forward([dev(f) for f in feats])
    parts = []
    for f, widths in zip(feats, spec.SITE_PARTS):
        assert sum(widths) == f.shape[1]
        off, grp = 0, []
        for wd in widths:
            grp.append(dev(np.ascontiguousarray(f[:, off:off + wd])))
            off += wd
        parts.append(grp)
    got = h.forward(parts)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # ... and the oracle's on the concatenated maps (RGB_OFF.py:395..590 cat, then :596-847)
    with torch.no_grad():
        want = orc.off_forward([torch.from_numpy(f) for f in feats], orc.to_torch_weights(synth.make_weights(spec.VARIANT_RGB)),
                               B, L, spec.VARIANT_RGB, orc.SLICE_FLAT)
    for a, b in zip(got, want):
        assert rel_err(a, b) < RTOL
    from offk_amd import _lib
    bad = list(parts)
    bad[0] = [dev(np.ascontiguousarray(feats[0][:, :48])), dev(np.ascontiguousarray(feats[0][:, 48:]))]
    with pytest.raises(_lib.OffkError, match="multiples of 32"):
        h.forward(bad)


def test_forward_is_stream_capturable(rt):
    """include/offk.h: offk_forward can be captured into a HIP graph; replaying the graph reproduces the eager bits."""
    B, L = 2, 3
    h, _ = make_handle(rt, B, L, spec.VARIANT_RGB, precision="f32split")
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    arr = h._feat_array(feats)
    out = [torch.empty(h.out_rows(), spec.NUM_CLASSES, device="cuda") for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):           # warm-up off the default stream (weight packing, function attributes)
        h.forward_into(arr, *out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [o.clone() for o in out]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        h.forward_into(arr, *out)
    for _ in range(3):
        for o in out:
            o.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, out))


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("B,L", [(2, 3), (3, 9)])
def test_fused_units_path_matches_unfused(rt, prec, B, L, monkeypatch):
    """The default inference path fuses K1 with the temporal difference (pw_tdiff.hip: G never written to HBM);
    OFFK_FUSED_UNITS=0 at offk_create keeps K1 + K2 apart.  Exact fp32 runs the 16-pixel LDS-DMA form on 16x16x4 MFMA tiles,
    split-fp32 the plane kernel on 16x16x32 bf16 tiles: their k grouping differs from K1's, same values to a few fp32 ulps
    (asserted at 2e-6 of the logits' magnitude), also with two temporal groups (L = 9)."""
    feats = [dev(f) for f in synth.make_features(B, L, 4)]
    monkeypatch.setenv("OFFK_FUSED_UNITS", "0")
    h0, _ = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    monkeypatch.delenv("OFFK_FUSED_UNITS")
    ref = h0.forward(feats)
    parts = []
    for f, widths in zip(synth.make_features(B, L, 4), spec.SITE_PARTS):
        off, grp = 0, []
        for wd in widths:
            grp.append(dev(np.ascontiguousarray(f[:, off:off + wd])))
            off += wd
        parts.append(grp)
    h1, _ = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    # whole maps, and the branches-as-parts entry point (same path)
    for got in (h1.forward(feats), h1.forward(parts)):
        for a, b in zip(ref, got):
            assert rel_err(b, a.cpu()) < 2e-6      # (f32split: the unfused reference side runs the fp32 pipe)


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("variant", [spec.VARIANT_RGB, spec.VARIANT_FLOW])
def test_test_time_shape_vs_oracle(rt, variant, prec):
    """The reference's eval shape: 10 crops x 25 segments per video, model built with batch = 10 (test_rgb_off.py:24-25,
    :184; test_flow_off.py:29-30, :414) -> N = 250 frames, P = 240 pairs.  It runs on its own tuned plan table
    (offk_api.hip kTunedP240*, LDS-patch tiles included), so it is checked against the oracle itself, both variants,
    both arithmetic modes."""
    B, L = 10, 25
    feats_np = synth.make_features(B, L, 6)
    h, w = make_handle(rt, B, L, variant, precision=prec)
    got = h.forward([dev(f) for f in feats_np])
    torch.cuda.synchronize()
    with torch.no_grad():
        want = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, variant, orc.SLICE_FLAT)
    rows = B if variant == spec.VARIANT_FLOW else B * (L - 1)
    for a, b in zip(got, want):
        assert a.shape == (rows, 101)
        assert rel_err(a, b) < RTOL
        se = signal_err(a, b)
        print("test-time shape variant %d %s: row-to-row signal error %.2e" % (variant, prec, se))
        assert se < RTOL_NORTH_STAR, se
    assert torch.equal(got[0], h.forward([dev(f) for f in feats_np])[0])


def test_forward_without_the_28_head(rt):
    """out28 = NULL (the reference never returns the 28x28 head, RGB_OFF.py:860): the other two logits are unchanged."""
    B, L = 2, 3
    h, _ = make_handle(rt, B, L, spec.VARIANT_FLOW, precision="f32split")
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    a7, a14, a28 = h.forward(feats)
    b7, b14, b28 = h.forward(feats, want28=False)
    torch.cuda.synchronize()
    assert b28 is None and a28 is not None
    assert torch.equal(a7, b7) and torch.equal(a14, b14)


# ---- round 2: inputs that stress the split-precision contractions (VERDICT r01 weak #1b) -------------------------

STRESS_KINDS = ["full_mantissa", "heavy_tail"]


@pytest.mark.parametrize("kind", STRESS_KINDS)
@pytest.mark.parametrize("prec", PRECISIONS)
def test_pw_reduce_on_full_range_inputs(rt, prec, kind):
    """K1 on maps with 24-bit mantissas / values up to 1e2 (synth.make_features_kind): the activation split of a split-precision kernel is
    exercised for real.  Checked against an fp64 contraction, all nine sites."""
    B, L = 2, 3
    h, w = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    feats = synth.make_features_kind(B, L, 4, kind)
    worst = 0.0
    for site, (name, C, H) in enumerate(spec.SITES):
        x = torch.from_numpy(feats[site])
        G, D = h.pw_reduce(site, dev(x))
        g_ref = torch.relu(F.conv2d(x.double(), w["motion_conv_gen_%s.weight" % name].double(), w["motion_conv_gen_%s.bias" % name].double()))
        d_ref = F.conv2d(x[:B * (L - 1)].double(), w["motion_spatial_down_%s.weight" % name].double(),
                         w["motion_spatial_down_%s.bias" % name].double())
        eg = rel_err(G.view(B * L, H, H, 128).permute(0, 3, 1, 2), g_ref)
        ed = rel_err(D.view(B * (L - 1), H, H, 32).permute(0, 3, 1, 2), d_ref)
        worst = max(worst, eg, ed)
    print("K1 %s on %s maps: worst error / max|out| over nine sites = %.2e" % (prec, kind, worst))
    assert worst < RTOL


@pytest.mark.parametrize("kind", STRESS_KINDS)
@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
def test_forward_on_full_range_inputs(rt, prec, kind):
    """Whole forward (RGB variant, B = 3, L = 7, quirk Q1 active) on the stress maps against the oracle."""
    B, L = 3, 7
    feats_np = synth.make_features_kind(B, L, 3, kind)
    h, w = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    got = h.forward([dev(f) for f in feats_np])
    torch.cuda.synchronize()
    with torch.no_grad():
        want, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, spec.VARIANT_RGB, orc.SLICE_FLAT,
                                   return_stages=True)
    errs = [rel_err(a, b) for a, b in zip(got, want)]
    P = B * (L - 1)
    for name, ch, H in (("fusion_28", 320, 28), ("fusion_14", 1056, 14), ("fusion_7", 832, 7), ("sum_7", 1024, 7)):
        errs.append(rel_err(h.region(name, ch).view(P, H, H, ch).permute(0, 3, 1, 2), st[name]))
    d = (got[0] - got[0].mean(0, keepdim=True)).cpu().double()
    dr = want[0].double() - want[0].double().mean(0, keepdim=True)
    sig = ((d - dr).abs().max() / dr.abs().max()).item()
    print("forward %s on %s maps: logits %.2e %.2e %.2e, stages %.2e %.2e %.2e %.2e, fc7 row-to-row signal %.2e"
          % ((prec, kind) + tuple(errs) + (sig,)))
    assert max(errs) < RTOL and sig < RTOL_NORTH_STAR


@pytest.mark.parametrize("prec", PRECISIONS)
def test_pw_reduce_cancellation_case(rt, prec):
    """Weights orthogonal to the activations: every channel of a pixel carries the same value a(pixel) and every weight
    row has zero sum, so the exact result is the bias and everything else is rounding.  The error of a contraction is
    bounded by eps * sum_k |w_k x_k| (fp32 MFMA: eps ~ 2**-24 per term), NOT by eps *
    |result|: assert that backward-error bound and report the error relative to max|out| for DESIGN.md section 4."""
    B, L, site = 2, 3, 5
    name, C, H = spec.SITES[site]
    wnp = synth.make_weights(spec.VARIANT_RGB)
    for key in ("motion_conv_gen_%s.weight" % name, "motion_spatial_down_%s.weight" % name):
        wk = wnp[key].astype(np.float64)
        wnp[key] = (wk - wk.mean(axis=1, keepdims=True)).astype(np.float32)
    h, w = make_handle(rt, B, L, spec.VARIANT_RGB, weights=wnp, precision=prec)
    a = synth.make_features_kind(B, L, 4, "heavy_tail")[site][:, :1] + np.float32(0.5)          # [N,1,H,H], > 0
    x = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (B * L, C, H, H))))
    G, D = h.pw_reduce(site, dev(x))
    wd, bd = w["motion_spatial_down_%s.weight" % name].double(), w["motion_spatial_down_%s.bias" % name].double()
    d_ref = F.conv2d(x[:B * (L - 1)].double(), wd, bd)
    mag = F.conv2d(x[:B * (L - 1)].double().abs(), wd.abs())             # sum_k |w_k x_k|
    Dn = D.view(B * (L - 1), H, H, 32).permute(0, 3, 1, 2).double().cpu()
    err = (Dn - d_ref).abs()
    backward = (err / mag).max().item()
    forward = (err.max() / d_ref.abs().max()).item()
    print("K1 %s cancellation case: max error / sum|w x| = %.2e, max error / max|out| = %.2e (out ~ bias, sum|w x| up to %.1f)"
          % (prec, backward, forward, mag.max().item()))
    # the same contraction by the CPU's fp32 convolution (what the reference itself would run): its error is of the
    # same kind -- relative to the bias-sized result even exact-fp32 arithmetic is ~1e-4 here
    cpu32 = F.conv2d(x[:B * (L - 1)], w["motion_spatial_down_%s.weight" % name], w["motion_spatial_down_%s.bias" % name]).double()
    print("   torch CPU fp32 conv on the same case: max error / sum|w x| = %.2e, / max|out| = %.2e"
          % (((cpu32 - d_ref).abs() / mag).max().item(), ((cpu32 - d_ref).abs().max() / d_ref.abs().max()).item()))
    assert backward < 2e-6


# ---- round 2: the drop-in class itself on the GPU (VERDICT r01 missing #1) -------------------

class _StubBackbone(torch.nn.Module):
    """Stands for the TSN backbone (out of scope, RGB_OFF.py:362-594): hands back the nine tap maps, the per-frame
    Feature_Generation_Score the golden captured from the reference's own backbone and, for RGB_OFF_v2, conv2."""

    def __init__(self, feats, fgs, conv2=None):
        super().__init__()
        self.feats, self.fgs, self.conv2 = feats, fgs, conv2

    def forward(self, frames):
        assert frames.shape[0] == self.fgs.shape[0]
        return (self.feats, self.fgs) if self.conv2 is None else (self.feats, self.fgs, self.conv2)


RET_GOLDEN = ["ret_rgb_b2_l3", "ret_rgb_b1_l2", "ret_flow_b2_l3", "ret_flow_b1_l2", "ret_rgbv2_b2_l3"]


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("tag", RET_GOLDEN)
def test_bninception_off_mirror_returns_what_the_reference_returns(rt, tag, prec, golden_dir):
    """off_module.BNInception_OFF -- the class a user of RGB_OFF.py / Flow_OFF.py / RGB_OFF_v2.py would switch to --
    against the tensors the reference's own forward returned (oracle/gen_golden.py run_ret_case): tuple order
    (7x7, backbone, 14x14; RGB_OFF.py:860), the P == 1 squeeze (:786), consensus (Flow_OFF.py:866-876), the
    modality_fuse sum (:881) and the RGB_OFF_v2 4-tuple (RGB_OFF_v2.py:891)."""
    from offk_amd import off_module
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    ref_file = tag.split("_")[1].replace("rgbv2", "rgb_v2")
    feats = [dev(f) for f in synth.make_features(B, L, cfg)]
    fgs = dev(g["fgs_raw"])
    in_ch = 10 if ref_file == "flow" else 3
    conv2 = torch.full((B * L, 192, 2, 2), 3.0, device="cuda") if ref_file == "rgb_v2" else None
    m = off_module.bninception_off(101, B, L, variant=ref_file, backbone=_StubBackbone(feats, fgs, conv2), precision=prec)
    sd = {"module." + k: torch.from_numpy(v) for k, v in synth.make_weights(variant).items()}
    m.load_state_dict(sd, strict=False)
    assert (m.batch, m.length, m.modality_fuse) == (B, L, False)
    frames = torch.zeros(B * L, in_ch, 8, 8, device="cuda")
    ret = m.RGB_OFF_forward(frames) if ref_file == "rgb" else m(frames)
    assert len(ret) == (4 if ref_file == "rgb_v2" else 3)
    for i in range(3):
        assert tuple(ret[i].shape) == g["ret%d" % i].shape, (i, tuple(ret[i].shape))
        assert rel_err(ret[i], g["ret%d" % i]) < RTOL, i
    if ref_file == "rgb":
        assert torch.equal(ret[1], fgs)                  # Feature_Generation_Score passes through untouched (:860)
        for a, b in zip(m(frames), ret):                 # forward() of the rgb mirror is RGB_OFF_forward
            assert torch.equal(a, b)
    else:
        if ref_file == "rgb_v2":
            assert ret[3] is conv2
        m.modality_fuse = True
        fused = m(frames)
        assert tuple(fused.shape) == g["ret_fused"].shape and rel_err(fused, g["ret_fused"]) < RTOL
    # no backbone: the nine maps are the input, FGS is None, fusing without it is an error
    m2 = off_module.bninception_off(101, B, L, variant=ref_file, precision=prec)
    m2.load_state_dict(sd, strict=False)
    r2 = m2(feats) if ref_file != "rgb" else m2.RGB_OFF_forward(feats)
    assert r2[1] is None and torch.equal(r2[0], ret[0]) and torch.equal(r2[2], ret[2])
    if ref_file != "rgb":
        m2.modality_fuse = True
        with pytest.raises(ValueError, match="Feature_Generation_Score"):
            m2(feats)


def test_mirror_picks_up_parameter_updates_on_the_gpu(rt):
    """a weight written after the first forward (optimizer step, copy_, a parent's load_state_dict) must
    reach liboffk's packed copies before the next forward."""
    from offk_amd import off_module
    B, L = 2, 3
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    net = off_module.OFFSubNetwork(101, B, L, "rgb").cuda()
    w0 = synth.make_weights(spec.VARIANT_RGB)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    a = [t.clone() for t in net(feats)]
    with torch.no_grad():
        net.motion_conv_trans_28.weight.mul_(1.5)
        net.fc_action_motion_14.bias.add_(1.0)
    b = net(feats)
    w1 = dict(w0)
    w1["motion_conv_trans_28.weight"] = w0["motion_conv_trans_28.weight"] * np.float32(1.5)
    w1["fc_action_motion_14.bias"] = w0["fc_action_motion_14.bias"] + np.float32(1.0)
    with torch.no_grad():
        want = orc.off_forward([f.cpu() for f in feats], orc.to_torch_weights(w1), B, L, 0, orc.SLICE_FLAT)
    for x, y in zip(b, want):
        assert rel_err(x, y) < RTOL
    assert not torch.equal(a[0], b[0])
    wrapper = torch.nn.Sequential(net)                    # a parent module loading a checkpoint
    wrapper.load_state_dict({"0." + k: torch.from_numpy(v) for k, v in w0.items()})
    for x, y in zip(net(feats), a):
        assert torch.equal(x, y)


# ---- round 2: BASELINE configs 3 and 5 at full per-GPU size (VERDICT r01 missing #4) ---------------------------------

@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
def test_flow_full_size_b64_vs_oracle(rt, prec):
    """BASELINE config 3: Flow_OFF (fixed diagonal Sobel, util.py:52-77; consensus inside, Flow_OFF.py:867-876), B = 64."""
    B, L = 64, 7
    feats = synth.make_features(B, L, 3)
    h, w = make_handle(rt, B, L, spec.VARIANT_FLOW, precision=prec)
    got = h.forward([dev(f) for f in feats])
    torch.cuda.synchronize()
    with torch.no_grad():
        want = orc.off_forward([torch.from_numpy(f) for f in feats], w, B, L, spec.VARIANT_FLOW, orc.SLICE_FLAT)
    for a, b in zip(got, want):
        assert a.shape == (B, 101) and rel_err(a, b) < RTOL
        se = signal_err(a, b)
        print("full size Flow B=64 %s: row-to-row signal error %.2e" % (prec, se))
        assert se < RTOL_NORTH_STAR, se


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
def test_two_stream_b64_vs_oracle(rt, prec):
    """BASELINE config 5 on one GPU at the per-GPU batch: RGB-OFF + Flow-OFF on the same 64 clips, two HIP streams,
    K7 late fusion with the notebook weights (score_fusion.ipynb lines 300-301) incl. both TSN scores."""
    from offk_amd import scores, two_stream
    B, L = 64, 7
    wr, wf = synth.make_weights(spec.VARIANT_RGB), synth.make_weights(spec.VARIANT_FLOW, seed=0xF10)
    fr, ff = synth.make_features(B, L, 2), synth.make_features(B, L, 3)
    tsn_r = torch.from_numpy(synth.uniform_values(0x7501, B * 101, 4.0).reshape(B, 101))
    tsn_f = torch.from_numpy(synth.uniform_values(0x7502, B * 101, 4.0).reshape(B, 101))
    ts = two_stream.TwoStreamOFF(B, L, precision=prec)
    ts.load_state_dicts(wr, wf)
    fused, pred = ts.forward([dev(f) for f in fr], [dev(f) for f in ff], rgb_tsn=dev(tsn_r), flow_tsn=dev(tsn_f))
    torch.cuda.synchronize()
    with torch.no_grad():
        r = orc.off_forward([torch.from_numpy(f) for f in fr], orc.to_torch_weights(wr), B, L, spec.VARIANT_RGB, consensus=True)
        f = orc.off_forward([torch.from_numpy(x) for x in ff], orc.to_torch_weights(wf), B, L, spec.VARIANT_FLOW, consensus=True)
    w = scores.FUSION_BEST
    ref = w[0] * r[0] + w[1] * tsn_r + w[2] * r[1] + w[3] * f[0] + w[4] * tsn_f + w[5] * f[1]
    assert rel_err(fused, ref) < RTOL
    # the TSN scores are per-clip random vectors (signal by construction): also check the OFF part of the fused score
    # alone, where the row-to-row signal is the small quantity
    off_ref = ref - w[1] * tsn_r - w[4] * tsn_f
    off_got = fused.cpu() - w[1] * tsn_r - w[4] * tsn_f
    se = signal_err(off_got, off_ref)
    print("two-stream B=64 %s: row-to-row signal error of the OFF part of the fused score %.2e" % (prec, se))
    assert se < RTOL_NORTH_STAR, se
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref.abs().max()       # ties within the tolerance may flip
    assert torch.equal(pred.cpu().long()[clear], ref.argmax(dim=1)[clear])


# ---- round 3: fused bottleneck chain (chain_fused.hip) and Winograd conv (winograd.hip) through their C-ABI entry points ----
@pytest.mark.parametrize("case", ["28a_merged", "28b_residual", "28c_into_slice"])
@pytest.mark.parametrize("n", [1, 5, 11])
def test_bottleneck_chain14_vs_torch(rt, case, n):
    """offk_bottleneck_chain14 against torch CPU fp32 convolutions (RGB_OFF.py:658-667 / :670-685): the merged form of block 28a
    (pre-ReLU input, branch conv folded into c3's K), the residual form, and output into a channel slice of a wider buffer.
    Odd image counts exercise the grid rounding (blocks of 8 images x 2 halves)."""
    g = torch.Generator().manual_seed(100 + n)
    merged = case == "28a_merged"
    Cin = 64 if merged else 256
    x = torch.randn(n, 14, 14, Cin + (64 if merged else 0), generator=g) * (1.0 if merged else 0.5)
    if not merged:
        x = x.clamp_min(0)                               # sa / sb are post-ReLU
    x_coff = 64 if merged else 0
    k1 = 1.0 / Cin ** 0.5
    w1 = (torch.rand(64, Cin, generator=g) * 2 - 1) * k1
    b1 = (torch.rand(64, generator=g) * 2 - 1) * k1
    w2 = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    b2 = (torch.rand(64, generator=g) * 2 - 1) / 24.0
    K3 = 128 if merged else 64
    w3 = (torch.rand(256, K3, generator=g) * 2 - 1) / K3 ** 0.5
    b3 = (torch.rand(256, generator=g) * 2 - 1) / 8.0
    xin = x[..., x_coff:x_coff + Cin].permute(0, 3, 1, 2).contiguous()
    t1 = F.relu(F.conv2d(F.relu(xin) if merged else xin, w1[:, :, None, None], b1))
    t2 = F.relu(F.conv2d(t1, w2, b2, padding=1))
    cat = torch.cat([t2, xin], 1) if merged else t2
    want = F.conv2d(cat, w3[:, :, None, None], b3)
    if not merged:
        want = want + xin
    want = F.relu(want).permute(0, 2, 3, 1)
    res = None if merged else dev(x)
    if case == "28c_into_slice":
        ybuf = torch.full((n, 14, 14, 352), -3.0, device="cuda")
        rt.bottleneck_chain14(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3), dev(b3), res=res, y=ybuf, y_coff=64)
        got = ybuf[..., 64:320]
        assert torch.all(ybuf[..., :64] == -3.0) and torch.all(ybuf[..., 320:] == -3.0)
    else:
        got = rt.bottleneck_chain14(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3), dev(b3), res=res, relu_in=merged, x_coff=x_coff)
    torch.cuda.synchronize()
    assert rel_err(got, want) < RTOL


@pytest.mark.parametrize("ci,co", [(128, 128), (128, 512), (256, 256), (832, 256)])
@pytest.mark.parametrize("n", [1, 7])
def test_winograd_conv3x3_vs_torch(rt, ci, co, n):
    """offk_winograd_conv3x3 (F(4x4, 3x3), fp32) against torch CPU fp32 F.conv2d for the five shapes it serves (RGB_OFF.py:766-767,
    775-780, 833-834, 837-838), with the epilogue of motion_conv3_trans_14b (ReLU, residual, ReLU) and the per-tile sums the 14-head
    takes its average pool from.  Tolerance as for the direct kernels; the measured error is printed."""
    g = torch.Generator().manual_seed(7 * ci + co + n)
    x = torch.randn(n, 7, 7, ci + 32, generator=g).clamp_min(0)
    w = (torch.rand(co, ci, 3, 3, generator=g) * 2 - 1) / (9 * ci) ** 0.5
    b = (torch.rand(co, generator=g) * 2 - 1) / (9 * ci) ** 0.5
    res = torch.randn(n, 7, 7, co, generator=g).clamp_min(0) * 0.3
    xin = x[..., 32:].permute(0, 3, 1, 2).contiguous()
    want = F.relu(F.relu(F.conv2d(xin, w, b, padding=1)) + res.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    got, pool = rt.winograd_conv3x3(dev(x), dev(w), dev(b), res=dev(res), flags=2 | 4, x_coff=32, want_pool=True)
    torch.cuda.synchronize()
    err = rel_err(got, want)
    print("winograd %d -> %d n=%d: max error / max |ref| = %.2e" % (ci, co, n, err))
    assert err < RTOL
    pooled = pool.view(n, 4, co).sum(1) / 49.0
    assert rel_err(pooled, want.mean(dim=(1, 2))) < RTOL
    # plain conv (no epilogue), output into a channel slice
    ybuf = torch.full((n, 7, 7, co + 64), -3.0, device="cuda")
    rt.winograd_conv3x3(dev(x), dev(w), None, x_coff=32, y=ybuf, y_coff=64)
    torch.cuda.synchronize()
    assert rel_err(ybuf[..., 64:], F.conv2d(xin, w, None, padding=1).permute(0, 2, 3, 1)) < RTOL
    assert torch.all(ybuf[..., :64] == -3.0)


@pytest.mark.parametrize("ci,co,n", [(64, 64, 1), (1056, 128, 3)])
def test_winograd_conv5x5s2_vs_torch(rt, ci, co, n):
    """offk_winograd_conv5x5s2 -- the polyphase form of motion_conv_trans_14 (RGB_OFF.py:762-763: 5x5, stride 2, pad 2, 14x14 ->
    7x7): four 7x7 phase images x 3x3 phase kernels concatenated along K -- against torch CPU fp32 F.conv2d, with the ReLU of
    :763 and output into a channel slice (the [u2 | x1] buffer)."""
    g = torch.Generator().manual_seed(11 * ci + co)
    x = torch.randn(n, 14, 14, ci + 32, generator=g).clamp_min(0)
    w = (torch.rand(co, ci, 5, 5, generator=g) * 2 - 1) / (25 * ci) ** 0.5
    b = (torch.rand(co, generator=g) * 2 - 1) / (25 * ci) ** 0.5
    xin = x[..., 32:].permute(0, 3, 1, 2).contiguous()
    want = F.relu(F.conv2d(xin, w, b, stride=2, padding=2)).permute(0, 2, 3, 1)
    ybuf = torch.full((n, 7, 7, co + 96), -3.0, device="cuda")
    rt.winograd_conv5x5s2(dev(x), dev(w), dev(b), flags=2, x_coff=32, y=ybuf, y_coff=96)
    torch.cuda.synchronize()
    err = rel_err(ybuf[..., 96:], want)
    print("polyphase winograd 5x5/2 %d -> %d n=%d: max error / max |ref| = %.2e" % (ci, co, n, err))
    assert err < RTOL
    assert torch.all(ybuf[..., :96] == -3.0)


@pytest.mark.parametrize("ci,co,n,relu", [(32, 64, 1, False), (320, 64, 3, False), (64, 128, 2, True)])
def test_winograd_conv7x7s2_vs_torch(rt, ci, co, n, relu):
    """offk_winograd_conv7x7s2 -- the polyphase Winograd form F(5x5, 4x4) of motion_conv_trans_28 (RGB_OFF.py:657: 7x7, stride 2,
    pad 3, 28x28 -> 14x14; four 14x14 phase images x 4-tap phase kernels concatenated along K, 64 points in four K groups) --
    against torch CPU fp32 F.conv2d, input from a channel slice, output into a channel slice (the [t2 | x0] buffer of the forward)."""
    g = torch.Generator().manual_seed(7 * ci + co)
    x = torch.randn(n, 28, 28, ci + 32, generator=g).clamp_min(0)
    w = (torch.rand(co, ci, 7, 7, generator=g) * 2 - 1) / (49 * ci) ** 0.5
    b = (torch.rand(co, generator=g) * 2 - 1) / (49 * ci) ** 0.5
    xin = x[..., 32:].permute(0, 3, 1, 2).contiguous()
    want = F.conv2d(xin, w, b, stride=2, padding=3)
    if relu:
        want = F.relu(want)
    want = want.permute(0, 2, 3, 1)
    ybuf = torch.full((n, 14, 14, co + 64), -3.0, device="cuda")
    rt.winograd_conv7x7s2(dev(x), dev(w), dev(b), flags=2 if relu else 0, x_coff=32, y=ybuf, y_coff=64)
    torch.cuda.synchronize()
    err = rel_err(ybuf[..., 64:], want)
    print("polyphase winograd 7x7/2 %d -> %d n=%d: max error / max |ref| = %.2e" % (ci, co, n, err))
    assert err < RTOL
    assert torch.all(ybuf[..., :64] == -3.0)
