"""GPU parity of the training side of the OFF units through the C ABI:
offk_off_units_train, offk_off_units_backward, offk_segment_consensus_backward against the oracle's
autograd (oracle/off_oracle.py, pinned to the reference's gradients by tests/test_oracle_grad_golden.py)
and against the committed gradient goldens.

Tolerance: gradients are sums of up to N*H*W ~ 3.5e5 products; RTOL = 2e-4 of each tensor's max magnitude
(the north_star bar is 1e-3), written here.
"""
import os

import numpy as np
import pytest
import torch

import offk_amd  # noqa: F401
from offk_amd import spec, synth
from oracle import off_oracle as orc

pytestmark = pytest.mark.gpu
RTOL = 2e-4
DROP_P = 0.8


@pytest.fixture(scope="module")
def rt():
    from offk_amd import runtime
    return runtime


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.cuda().contiguous()


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


def cotangents(P):
    return [torch.from_numpy(synth.uniform_values(0xC07 + i, P * spec.NUM_CLASSES, 1.0).reshape(P, spec.NUM_CLASSES))
            for i in range(3)]


def unit_drop(seed, P, p=DROP_P):
    return [torch.from_numpy(synth.dropout_keep(seed, si, P, H, p)).float() / (1.0 - p)
            for si, (_n, _c, H) in enumerate(spec.SITES)]


def grad_views(dm):
    """nine [P,160,H,H] -> the three fusion-buffer gradients, channels-last, + per-site (tensor, coff)."""
    groups = ((0, 1), (2, 3, 4, 5, 6), (7, 8))
    views = [None] * spec.NUM_SITES
    for grp in groups:
        buf = dev(torch.cat([dm[i] for i in grp], dim=1).permute(0, 2, 3, 1).contiguous())
        for k, i in enumerate(grp):
            views[i] = (buf, 160 * k)
    return views


def device_relu_masks(h, feats_cpu, w, B, L, slack=1e-5):
    """The ReLU decisions the device took (saved G > 0) as nine [N,128,H,H] 0/1 tensors, after checking that
    they differ from the oracle's own only for pre-activations within rounding distance of zero."""
    masks = []
    for (site, _c, H), x in zip(spec.SITES, feats_cpu):
        G = h.region("G_" + site, 128).view(B * L, H * H, 128).permute(0, 2, 1).reshape(B * L, 128, H, H).cpu()
        with torch.no_grad():
            pre = torch.nn.functional.conv2d(x, w["motion_conv_gen_%s.weight" % site], w["motion_conv_gen_%s.bias" % site])
        mask = (G > 0)
        flip = mask != (pre > 0)
        assert int(flip.sum()) <= 5 + slack * flip.numel(), site
        if flip.any():
            assert float(pre[flip].abs().max()) < slack * max(1.0, float(pre.abs().max())), site
        masks.append(mask.float())
    return masks


def make(rt, B, L, variant, slice_mode=spec.SLICE_FLAT, precision="fp32"):
    h = rt.OffForward(B, L, variant, slice_mode, precision=precision, training=True)
    w = synth.make_weights(variant)
    assert h.load_state_dict(w) == []
    return h, orc.to_torch_weights(w)


@pytest.mark.parametrize("variant,B,L,slice_mode,seed,prec", [
    (spec.VARIANT_RGB, 2, 3, spec.SLICE_FLAT, None, "fp32"),
    (spec.VARIANT_RGB, 2, 3, spec.SLICE_FLAT, 7, "fp32"),
    (spec.VARIANT_RGB, 3, 4, spec.SLICE_FLAT, 11, "fp32"),
    (spec.VARIANT_RGB, 3, 4, spec.SLICE_PER_CLIP, 5, "fp32"),
    (spec.VARIANT_FLOW, 2, 3, spec.SLICE_FLAT, None, "fp32"),
    (spec.VARIANT_FLOW, 2, 7, spec.SLICE_FLAT, 3, "fp32"),
    (spec.VARIANT_FLOW, 2, 7, spec.SLICE_FLAT, 3, "f32split"),     # (training side of a split-fp32 handle: the fp32 kernels)
])
def test_units_backward_vs_oracle(rt, variant, B, L, slice_mode, seed, prec):
    """Exact-fp32 MFMA weight-gradient GEMM (the two-plane bf16x3 core of rounds 1-4 was retired with its mode in ABI v9)."""
    P = B * (L - 1)
    cfg = 2 if B == 2 else 3
    feats = synth.make_features(B, L, cfg)
    h, w = make(rt, B, L, variant, slice_mode, precision=prec)
    slack = 1e-5 if prec == "fp32" else 1e-4
    drops = None if seed is None else unit_drop(seed, P)
    tf = [torch.from_numpy(f) for f in feats]
    dfeats = [dev(f) for f in feats]
    if seed is None:
        h.off_units(dfeats)
    else:
        h.off_units_train(dfeats, seed, DROP_P)
    ref, dm = orc.unit_backward(tf, w, B, L, variant, slice_mode, cotangents(P), drops,
                                device_relu_masks(h, tf, w, B, L, slack))
    # training-mode forward: the fusion buffers hold [dropout(S) | T]
    with torch.no_grad():
        m_ref = [orc.off_unit(x, w, site, B, L, variant, slice_mode, None if drops is None else drops[si])
                 for si, ((site, _c, _h), x) in enumerate(zip(spec.SITES, tf))]
    f28 = h.region("fusion_28", 320).view(P, 28, 28, 320).permute(0, 3, 1, 2)
    assert rel_err(f28[:, :160], m_ref[0]) < RTOL and rel_err(f28[:, 160:], m_ref[1]) < RTOL
    f7 = h.region("fusion_7", 832).view(P, 7, 7, 832).permute(0, 3, 1, 2)
    assert rel_err(f7[:, 160:320], m_ref[8]) < RTOL
    flat, got = h.off_units_backward(dfeats, grad_views(dm), 0 if seed is None else seed, 0.0 if seed is None else DROP_P)
    torch.cuda.synchronize()
    assert set(got) == set(ref)
    errs = dict((k, rel_err(got[k], ref[k])) for k in ref)
    bad = dict((k, "%.2e" % e) for k, e in errs.items() if not e < RTOL or got[k].shape != ref[k].shape)
    assert not bad, bad
    # accumulate: a second call adds the same gradients; reproducible bits
    flat2, got2 = h.off_units_backward(dfeats, grad_views(dm), 0 if seed is None else seed, 0.0 if seed is None else DROP_P)
    assert torch.equal(flat, flat2)
    h.off_units_backward(dfeats, grad_views(dm), 0 if seed is None else seed, 0.0 if seed is None else DROP_P,
                         grads=flat2, accumulate=True)
    assert rel_err(flat2, 2.0 * flat) < 1e-6


@pytest.mark.parametrize("tag", ["grad_rgb_b2_l3", "grad_rgb_b2_l3_drop", "grad_rgb_b3_l4_drop", "grad_flow_b2_l3"])
def test_units_backward_vs_reference_golden(rt, tag, golden_dir):
    """Against the gradients captured from the reference import (oracle/gen_golden.py grad): the full tensors the
    fixture holds (biases, depthwise weights, one down weight) and checksums / samples of the others."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg, seed = (int(v) for v in g["meta"])
    P = B * (L - 1)
    feats = synth.make_features(B, L, cfg)
    h, w = make(rt, B, L, variant)
    drops = None if seed < 0 else unit_drop(seed, P)
    tf = [torch.from_numpy(f) for f in feats]
    dfeats = [dev(f) for f in feats]
    if seed < 0:
        h.off_units(dfeats)
    else:
        h.off_units_train(dfeats, seed, DROP_P)
    # dM from the oracle's own graph (pinned to the golden's fusion-buffer gradients on the CPU side); the golden
    # was taken with the reference's ReLU decisions, the device took its own for the few pre-activations at the
    # kink: shift the golden by the oracle's difference between the two decisions
    own, dm = orc.unit_backward(tf, w, B, L, variant, orc.SLICE_FLAT, cotangents(P), drops)
    dev_mask = orc.unit_param_grads_from_dm(tf, w, B, L, variant, orc.SLICE_FLAT, dm, drops, device_relu_masks(h, tf, w, B, L))
    _flat, got = h.off_units_backward(dfeats, grad_views(dm), max(seed, 0), 0.0 if seed < 0 else DROP_P)
    idx = lambda n: (np.arange(97, dtype=np.int64) * 2654435761 + 12345) % n   # noqa: E731
    for k, t in got.items():
        shift = (dev_mask[k] - own[k]).double().reshape(-1)
        a = t.detach().cpu().double().reshape(-1) - shift
        cs = g["cs_" + k]
        scale = float(a.abs().max())
        assert abs(a.abs().sum().item() - cs[1]) <= 1e-4 * cs[1], k
        sm = a[torch.from_numpy(idx(a.numel()))].numpy()
        assert np.abs(sm - g["sm_" + k]).max() <= RTOL * scale, k
        if "full_" + k in g.files:
            assert rel_err(a.reshape(t.shape), torch.from_numpy(g["full_" + k])) < RTOL, k


def test_units_backward_full_size_properties(rt):
    """BASELINE config 2 (B = 64, L = 7): size-independent properties -- linearity in dM, zero gradient for
    zero dM, bias gradient = column sums of the kernel's own dGpre / dD, bitwise reproducibility."""
    B, L = 64, 7
    P = B * (L - 1)
    h, _w = make(rt, B, L, spec.VARIANT_RGB, precision="fp32")
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    h.off_units_train(feats, 21, DROP_P)
    gen = torch.Generator(device="cuda").manual_seed(5)
    bufs = [torch.randn(P, H, H, C, device="cuda", generator=gen) for H, C in ((28, 320), (14, 1056), (7, 832))]
    views = [(bufs[0], 0), (bufs[0], 160)] + [(bufs[1], 160 * k) for k in range(5)] + [(bufs[2], 0), (bufs[2], 160)]
    g1, v1 = h.off_units_backward(feats, views, 21, DROP_P)
    g1b, _ = h.off_units_backward(feats, views, 21, DROP_P)
    assert torch.equal(g1, g1b)
    # bias gradients are the column sums of the intermediate gradients the kernel left in the workspace
    dG = h.region("dG_3c", 128).double().sum(dim=0)
    dD = h.region("dD_3c", 32).double().sum(dim=0)
    assert rel_err(v1["motion_conv_gen_3c.bias"], dG) < 1e-4 and rel_err(v1["motion_spatial_down_3c.bias"], dD) < 1e-4
    # the temporal branch telescopes: summed over a clip's frames the un-masked gradient is zero, so with the ReLU
    # mask the gen gradient of frame rows is bounded by the dT magnitudes; check linearity instead
    views2 = [(2.0 * t, c) for t, c in views]
    g2, _ = h.off_units_backward(feats, views2, 21, DROP_P)
    assert rel_err(g2, 2.0 * g1) < 1e-6
    zeros = [(torch.zeros_like(t), c) for t, c in views]
    g0, _ = h.off_units_backward(feats, zeros, 21, DROP_P)
    assert float(g0.abs().max()) == 0.0


def test_segment_consensus_backward(rt, golden_dir):
    g = np.load(os.path.join(golden_dir, "consensus_bwd.npz"))
    B, T, C = (int(v) for v in g["meta"])
    go = dev(synth.uniform_values(0xC10, B * C, 1.0).reshape(B, C))
    gi = rt.segment_consensus_backward(go, T)
    assert np.array_equal(gi.cpu().numpy(), g["grad_in"].reshape(B * T, C))


def test_backward_argument_checks(rt):
    from offk_amd import _lib
    h = rt.OffForward(2, 3, spec.VARIANT_RGB)
    with pytest.raises(_lib.OffkError, match="training=True"):
        h.off_units_backward([None] * 9, [None] * 9)
    ht, _ = make(rt, 2, 3, spec.VARIANT_RGB)
    feats = [dev(f) for f in synth.make_features(2, 3, 2)]
    with pytest.raises(_lib.OffkError, match="dropout probability"):
        ht.off_units_train(feats, 1, 1.0)


@pytest.mark.parametrize("variant", ["rgb", "flow"])
def test_off_units_module_autograd(rt, variant):
    """OFFUnits (the trainable host-side mirror): loss.backward() through torch's own fusion stages fills
    param.grad of every unit parameter with what the oracle's graph gives, and an optimizer step is picked up
    by the next forward."""
    from offk_amd.off_module import OFFUnits
    B, L = 2, 3
    P = B * (L - 1)
    v = spec.VARIANT_RGB if variant == "rgb" else spec.VARIANT_FLOW
    wnp = synth.make_weights(v)
    w = orc.to_torch_weights(wnp)
    units = OFFUnits(B, L, variant).cuda()
    units.load_state_dict({k: torch.from_numpy(a) for k, a in wnp.items() if k in units.state_dict()}, strict=True)
    units.train()
    feats_np = synth.make_features(B, L, 2)
    feats = [dev(f) for f in feats_np]
    wg = {k: t.cuda() for k, t in w.items()}
    m28, m14, m7 = units(feats, drop_seed=7)
    for m in (m28, m14, m7):
        m.retain_grad()
    assert m28.shape == (P, 320, 28, 28) and m14.shape == (P, 800, 14, 14) and m7.shape == (P, 320, 7, 7)
    # the reference's fusion stages and heads as ordinary torch ops on the GPU (the oracle's functions are plain torch)
    s28 = orc.fusion_28(m28, wg)
    s14 = orc.fusion_14(torch.cat((m14, s28), 1), wg)
    s7 = orc.fusion_7(torch.cat((m7, s14), 1), wg)
    cot = [c.cuda() for c in cotangents(P)]
    loss = (orc.head(s7, wg, "fc_action_motion", False) * cot[0]).sum() + (orc.head(s14, wg, "fc_action_motion_14", False) * cot[1]).sum() \
        + (orc.head(s28, wg, "fc_action_motion_28", True) * cot[2]).sum()
    loss.backward()
    tf = [torch.from_numpy(f) for f in feats_np]
    drops = unit_drop(7, P)
    # the gradient that reached the units came through torch's GPU convolutions (MIOpen; its own ReLU kinks):
    # compare the units' part on that same dM, and the whole chain against the oracle's CPU graph more loosely
    dm = [g.detach().cpu() for g in (m28.grad[:, :160], m28.grad[:, 160:])] + \
         [m14.grad[:, 160 * k:160 * k + 160].detach().cpu() for k in range(5)] + \
         [g.detach().cpu() for g in (m7.grad[:, :160], m7.grad[:, 160:])]
    masks = device_relu_masks(units._rt, tf, w, B, L)
    ref = orc.unit_param_grads_from_dm(tf, w, B, L, v, orc.SLICE_FLAT, dm, drops, masks)
    whole, _dm = orc.unit_backward(tf, w, B, L, v, orc.SLICE_FLAT, cotangents(P), drops, masks)
    n = 0
    for k, prm in units.named_parameters():
        if k == spec.SOBEL_KEY:
            assert prm.grad is None
            continue
        assert rel_err(prm.grad, ref[k]) < RTOL, k
        assert rel_err(prm.grad, whole[k]) < 1e-2, k
        n += 1
    assert n == len(ref)
    # an SGD step changes the parameters in place; the next forward must run with the new values
    before = m7.detach().clone()
    with torch.no_grad():
        for prm in units.parameters():
            if prm.grad is not None:
                prm -= 10.0 * prm.grad
    units.eval()
    m28b, m14b, m7b = units(feats)
    w2 = dict(w)
    w2.update((k, prm.detach().cpu()) for k, prm in units.named_parameters())
    with torch.no_grad():
        want = orc.off_unit(tf[8], w2, "5b", B, L, v, orc.SLICE_FLAT)
    assert rel_err(m7b[:, 160:], want) < RTOL and not torch.equal(before, m7b)


@pytest.mark.parametrize("B,L,variant,prec,slice_mode,cons", [
    (3, 2, spec.VARIANT_RGB, "fp32", spec.SLICE_FLAT, True),          # L = 2: one pair per clip
    (5, 9, spec.VARIANT_FLOW, "f32split", spec.SLICE_PER_CLIP, False),  # L > 7: two temporal steps per T-block
    (1, 8, spec.VARIANT_RGB, "f32split", spec.SLICE_FLAT, True),       # one clip
    (5, 3, spec.VARIANT_RGB, "fp32", spec.SLICE_PER_CLIP, False),     # P % 4 != 0: partial 196-pixel groups in the patch conv
])
def test_odd_shapes_forward_and_backward(rt, B, L, variant, prec, slice_mode, cons):
    """Shapes off the benchmark grid (tests/tools/fuzz_shapes.py draws more of them): whole forward and units backward."""
    P = B * (L - 1)
    feats = synth.make_features(B, L, 9)
    wnp = synth.make_weights(variant, seed=0xBEEF + B + L)
    w = orc.to_torch_weights(wnp)
    h = rt.OffForward(B, L, variant, slice_mode, cons, precision=prec, training=True)
    h.load_state_dict(wnp)
    df = [dev(f) for f in feats]
    tf = [torch.from_numpy(f) for f in feats]
    out = h.forward(df)
    with torch.no_grad():
        ref = orc.off_forward(tf, w, B, L, variant, slice_mode, consensus=cons)
    for a, b in zip(out, ref):
        assert rel_err(a, b.reshape(a.shape)) < RTOL
    h.off_units_train(df, 5, DROP_P)
    drops = unit_drop(5, P)
    g, dm = orc.unit_backward(tf, w, B, L, variant, slice_mode, cotangents(P), drops,
                              device_relu_masks(h, tf, w, B, L, 1e-5 if prec == "fp32" else 1e-4))
    _flat, got = h.off_units_backward(df, grad_views(dm), 5, DROP_P)
    for k in g:
        assert rel_err(got[k], g[k]) < RTOL, k


def test_off_units_interleaved_forwards_before_backward(rt):
    """the backward reads the G / D activations of ITS forward, but they live in the handle's one workspace.
    fwd(A), fwd(B), backward(A) (an eval pass, a second micro-batch, a checkpoint recompute in between) must give the
    gradients of A -- the module notices the foreign forward and recomputes K1 + K2 from the saved inputs and seed --
    and a parameter written between forward and backward must raise instead of pairing old activations with new weights."""
    from offk_amd.off_module import OFFUnits
    B, L = 2, 3
    wnp = synth.make_weights(spec.VARIANT_RGB)
    units = OFFUnits(B, L, "rgb").cuda()
    units.load_state_dict({k: torch.from_numpy(a) for k, a in wnp.items() if k in units.state_dict()}, strict=True)
    units.train()
    fa = [dev(f) for f in synth.make_features(B, L, 2)]
    fb = [dev(f) for f in synth.make_features(B, L, 5)]
    g = torch.Generator().manual_seed(3)
    cots = [torch.randn(s, generator=g).cuda() for s in ((4, 320, 28, 28), (4, 800, 14, 14), (4, 320, 7, 7))]

    def grads_of(run):
        for p in units.parameters():
            p.grad = None
        out = run()
        torch.autograd.backward(out, cots)
        return {k: p.grad.clone() for k, p in units.named_parameters() if p.grad is not None}

    alone = grads_of(lambda: units(fa, drop_seed=7))

    def interleaved():
        out = units(fa, drop_seed=7)
        units(fb, drop_seed=9)                      # second micro-batch through the same module
        units.eval()
        units(fb)                                   # and an eval pass
        units.train()
        return out

    mixed = grads_of(interleaved)
    assert alone.keys() == mixed.keys() and len(alone) == 54
    for k in alone:
        assert torch.equal(alone[k], mixed[k]), k
    out = units(fa, drop_seed=7)
    with torch.no_grad():
        units.motion_conv_gen_3a.weight.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified"):
        torch.autograd.backward(out, cots)


@pytest.mark.parametrize("prec", ["fp32", "f32split"])
def test_off_units_bound_weights_follow_updates_on_a_side_stream(rt, prec):
    """the trainable tensors are bound in place (offk_bind_weight), so an optimizer step needs no
    library call at all.  Parameters updated by kernels on a NON-DEFAULT (non-blocking) stream must be the ones the next
    forward on that stream uses -- there is no hidden NULL-stream copy to race with -- and nothing is pushed after the
    first forward."""
    from offk_amd.off_module import OFFUnits
    B, L = 2, 3
    wnp = synth.make_weights(spec.VARIANT_RGB)
    units = OFFUnits(B, L, "rgb", precision=prec).cuda()
    units.load_state_dict({k: torch.from_numpy(a) for k, a in wnp.items() if k in units.state_dict()}, strict=True)
    units.eval()
    feats_np = synth.make_features(B, L, 2)
    feats = [dev(f) for f in feats_np]
    tf = [torch.from_numpy(f) for f in feats_np]
    m7 = units(feats)[2]
    with torch.no_grad():
        want = orc.off_unit(tf[8], orc.to_torch_weights(wnp), "5b", B, L, spec.VARIANT_RGB, orc.SLICE_FLAT)
    assert rel_err(m7[:, 160:], want) < RTOL
    calls = []
    real_set = units._rt.set_weight
    real_bind = units._rt.bind_weight
    units._rt.set_weight = lambda *a, **k: (calls.append("set"), real_set(*a, **k))
    units._rt.bind_weight = lambda *a, **k: (calls.append("bind"), real_bind(*a, **k))
    side = torch.cuda.Stream()                       # torch streams are non-blocking: no implicit NULL-stream ordering
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for step in range(3):
            with torch.no_grad():
                for prm in units.parameters():
                    if prm.requires_grad:
                        prm.mul_(1.25).add_(0.01)         # "optimizer step" on the side stream
            m28, m14, m7 = units(feats)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert calls == []                                # nothing copied, nothing re-bound
    w2 = dict(orc.to_torch_weights(wnp))
    w2.update((k, prm.detach().cpu()) for k, prm in units.named_parameters())
    with torch.no_grad():
        for si, site in ((0, "3a"), (4, "4b"), (8, "5b")):
            want = orc.off_unit(tf[si], w2, site, B, L, spec.VARIANT_RGB, orc.SLICE_FLAT)
            buf, off = ((m28, 0), (m14, 320), (m7, 160))[(0, 4, 8).index(si)]
            assert rel_err(buf[:, off:off + 160], want) < RTOL, site
    # a re-allocated parameter is re-bound (once)
    units.motion_conv_gen_3a.weight.data = units.motion_conv_gen_3a.weight.data.clone()
    units(feats)
    assert calls == ["bind"]


def test_bind_weight_argument_checks(rt):
    from offk_amd import _lib
    h = rt.OffForward(1, 3, spec.VARIANT_RGB, training=True)
    with pytest.raises(_lib.OffkError, match="only the OFF units"):
        h.bind_weight("motion_conv_trans_28.weight", torch.zeros(64, 320, 7, 7, device="cuda"))
    with pytest.raises(_lib.OffkError, match="not an OFF"):
        h.bind_weight("conv1_7x7_s2.weight", torch.zeros(64, 3, 7, 7, device="cuda"))
    with pytest.raises(ValueError):
        h.bind_weight("motion_conv_gen_3a.bias", torch.zeros(128))          # host tensor
    with pytest.raises(_lib.OffkError, match="16-byte aligned"):
        h.bind_weight("motion_conv_gen_3a.bias", torch.zeros(132, device="cuda")[1:129])     # contiguous view, 4 bytes in
    # a tensor of the wrong shape is refused on the host side and by the library itself (raw ABI call)
    with pytest.raises(ValueError, match="reference"):
        h.bind_weight("motion_conv_gen_3a.weight", torch.zeros(128, 192, 1, 1, device="cuda"))
    import ctypes
    t = torch.zeros(64, device="cuda")
    shp = (ctypes.c_int64 * 1)(64)
    rc = h.lib.offk_bind_weight(h._h, b"motion_conv_gen_3a.bias", ctypes.c_void_p(t.data_ptr()), shp, 1)
    assert rc == -1 and b"shape mismatch" in h.lib.offk_last_error(h._h)
    # right shape claimed, allocation behind the pointer too small: [128] floats asked for, 64 floats left in the
    # device allocation (its real extent comes from the HIP runtime: a torch tensor may sit inside a larger cached segment)
    hip = ctypes.CDLL("libamdhip64.so")
    big = torch.zeros(4 << 20, device="cuda")
    base, size = ctypes.c_void_p(), ctypes.c_size_t()
    assert hip.hipMemGetAddressRange(ctypes.byref(base), ctypes.byref(size), ctypes.c_void_p(big.data_ptr())) == 0
    shp = (ctypes.c_int64 * 1)(128)
    rc = h.lib.offk_bind_weight(h._h, b"motion_conv_gen_3a.bias", ctypes.c_void_p(base.value + size.value - 64 * 4), shp, 1)
    assert rc == -1 and b"allocation too small" in h.lib.offk_last_error(h._h)
