"""GPU parity tests for the pair-count-gated kernels of the exact-fp32 forward.

Three default-path kernels are selected by the number of frame pairs P = B (L - 1): the fused bottleneck chains of fusion@28
(`chain14_kernel`, from P = 72), the polyphase Winograd form of the 5x5 / stride 2 conv (from P = 40) and of the 7x7 / stride 2
conv (from P = 12).  The reference-held goldens have P <= 18, so with the default gates they never reach the first two.  Here

* every golden and both stress distributions run with ALL gates forced open (OFFK_CHAIN / OFFK_WINOGRAD_5X5 /
  OFFK_WINOGRAD_7X7 = 2 at offk_create): reference-held outputs and stage tensors pass through those kernels;
* the stress distributions run at P = 72 with the default gates;
* the full-size forwards (BASELINE configs 2 and 3) compare the fusion-stage tensors, not only the logits;
* the Winograd stage kernels run on heavy-tailed inputs and are held to a backward-error bound per element,
  |err| <= c 2^-24 sum_k |w_k x_k| (as test_pw_reduce_cancellation_case does for K1), beside the max-normalised check --
  Winograd's error sits where the outputs are small;
* a fusion-conv weight update at P >= 40 reaches the transformed Winograd weights.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import offk_amd  # noqa: F401
from offk_amd import spec, synth
from oracle import off_oracle as orc

from .test_gpu_parity import GOLDEN, HANDLE_PRECISIONS, RTOL, RTOL_NORTH_STAR, dev, make_handle, rel_err, rt, signal_err  # noqa: F401

pytestmark = pytest.mark.gpu

STAGES = (("fusion_28", 320, 28), ("fusion_14", 1056, 14), ("fusion_7", 832, 7), ("sum_7", 1024, 7))
FORCE = {"OFFK_CHAIN": "2", "OFFK_WINOGRAD_5X5": "2", "OFFK_WINOGRAD_7X7": "2"}


def forced_handle(rt, monkeypatch, *args, **kw):
    for k, v in FORCE.items():
        monkeypatch.setenv(k, v)
    try:
        return make_handle(rt, *args, **kw)
    finally:
        for k in FORCE:
            monkeypatch.delenv(k)


def stage_errs(h, st, P):
    return dict((name, rel_err(h.region(name, ch).view(P, H, H, ch).permute(0, 3, 1, 2), st[name])) for name, ch, H in STAGES)


def launches_of(h, feats):
    """Names of the launch groups one forward enqueues (per-launch trace of the library)."""
    h.set_profiling(2)
    h.forward(feats)
    torch.cuda.synchronize()
    names = list(h.launch_times().keys())
    h.set_profiling(0)
    return names


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("tag", GOLDEN)
def test_goldens_through_the_gated_kernels(rt, tag, golden_dir, monkeypatch, prec):
    """All seven reference goldens with every pair-count gate forced open: the reference's own outputs (fc7 / fc14 / fc28) and the
    oracle's stage tensors after chain14_kernel, the polyphase 5x5 and the 7x7 Winograd kernels -- in both arithmetic modes (split-fp32:
    the GEMMs of both polyphase convs and the chains' contractions on the bf16 pipe see reference-held outputs here)."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    variant, B, L, cfg = (int(v) for v in g["meta"])
    h, w = forced_handle(rt, monkeypatch, B, L, variant, consensus=False, precision=prec)
    feats_np = synth.make_features(B, L, cfg)
    feats = [dev(f) for f in feats_np]
    names = launches_of(h, feats)
    assert any(n.startswith("chain_28a") for n in names), names
    assert any(n.startswith("motion_conv_trans_14 [winograd") for n in names), names
    assert any(n.startswith("motion_conv_trans_28 [winograd") for n in names), names
    out7, out14, out28 = h.forward(feats)
    torch.cuda.synchronize()
    for out, key in ((out7, "fc7"), (out14, "fc14"), (out28, "fc28")):
        assert rel_err(out, g[key]) < RTOL, key
    with torch.no_grad():
        _ref, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, variant, orc.SLICE_FLAT, consensus=False,
                                   return_stages=True)
    P = B * (L - 1)
    errs = stage_errs(h, st, P)
    print("%s %s gates open: stage errors %s" % (tag, prec, " ".join("%s %.1e" % kv for kv in errs.items())))
    assert max(errs.values()) < RTOL, errs
    if P >= 2:
        for out, key in ((out7, "fc7"), (out14, "fc14"), (out28, "fc28")):
            se = signal_err(out, g[key])
            assert se < RTOL_NORTH_STAR, (key, se)


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("kind", ["full_mantissa", "heavy_tail"])
@pytest.mark.parametrize("mode", ["gates_open_p18", "default_gates_p72"])
def test_stress_inputs_through_the_gated_kernels(rt, kind, mode, monkeypatch, prec):
    """Full-mantissa / heavy-tailed maps (synth.make_features_kind) through the chain and polyphase-Winograd kernels: at B = 3
    with the gates forced open, and at B = 12 (P = 72) where the default gates select them."""
    B, L = (3, 7) if mode == "gates_open_p18" else (12, 7)
    feats_np = synth.make_features_kind(B, L, 3, kind)
    if mode == "gates_open_p18":
        h, w = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB, precision=prec)
    else:
        h, w = make_handle(rt, B, L, spec.VARIANT_RGB, precision=prec)
    feats = [dev(f) for f in feats_np]
    names = launches_of(h, feats)
    assert any(n.startswith("chain_28a") for n in names) and any(n.startswith("motion_conv_trans_14 [winograd") for n in names), names
    got = h.forward(feats)
    torch.cuda.synchronize()
    with torch.no_grad():
        want, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, spec.VARIANT_RGB, orc.SLICE_FLAT, return_stages=True)
    P = B * (L - 1)
    errs = stage_errs(h, st, P)
    lerr = [rel_err(a, b) for a, b in zip(got, want)]
    sig = signal_err(got[0], want[0])
    print("forward %s on %s maps, %s: logits %.2e %.2e %.2e, stages %s, fc7 row-to-row signal %.2e"
          % ((prec, kind, mode) + tuple(lerr) + (" ".join("%.1e" % v for v in errs.values()), sig)))
    assert max(lerr) < RTOL and max(errs.values()) < RTOL and sig < RTOL_NORTH_STAR


_ORACLE_B64 = {}


def oracle_b64(variant, feats_np, w):
    """The oracle's B = 64 forward with stage tensors (~25 s of host time): once per variant, shared by both arithmetic modes."""
    if variant not in _ORACLE_B64:
        with torch.no_grad():
            _ORACLE_B64[variant] = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, 64, 7, variant, orc.SLICE_FLAT, consensus=False,
                                                   return_stages=True)
    return _ORACLE_B64[variant]


@pytest.mark.parametrize("prec", HANDLE_PRECISIONS)
@pytest.mark.parametrize("variant", [spec.VARIANT_RGB, spec.VARIANT_FLOW])
def test_full_size_b64_stage_tensors_vs_oracle(rt, variant, prec):
    """BASELINE configs 2 / 3 at full size (B = 64, P = 384 -- where every gated kernel runs by default): the fusion-stage tensors
    against the oracle, beside the logit checks of test_full_size_b64_vs_oracle / test_flow_full_size_b64_vs_oracle; both arithmetic modes
    against the ORACLE (not against each other)."""
    B, L = 64, 7
    feats_np = synth.make_features(B, L, 2 if variant == spec.VARIANT_RGB else 3)
    h, w = make_handle(rt, B, L, variant, consensus=False, precision=prec)
    got = h.forward([dev(f) for f in feats_np])
    torch.cuda.synchronize()
    want, st = oracle_b64(variant, feats_np, w)
    P = B * (L - 1)
    errs = stage_errs(h, st, P)
    print("full size variant %d %s: stage errors %s" % (variant, prec, " ".join("%s %.1e" % kv for kv in errs.items())))
    assert max(errs.values()) < RTOL, errs
    for a, b in zip(got, want):
        assert rel_err(a, b) < RTOL and signal_err(a, b) < RTOL_NORTH_STAR


# ---- Winograd stage kernels on heavy-tailed inputs, backward-error bound per element -------------------------------------------

def heavy_tail(shape, seed):
    """expm1(1.151 z) of the portable generator's ReLU-like z (synth.make_features_kind 'heavy_tail'): ~half zeros, tail to ~1e2."""
    n = int(np.prod(shape))
    z = synth.feature_values(seed, 0, n).astype(np.float64)
    return torch.from_numpy(np.expm1(z * 1.151).astype(np.float32).reshape(shape))


# c of |err| <= c 2^-24 sum |w x|: a direct fp32 contraction sits at c ~ 1-2 (test_pw_reduce_cancellation_case).  A Winograd form
# rounds in the transformed domain, so the error of an output is proportional to the magnitude of its TILE (and to the norms of the
# transform matrices), not to its own sum |w x|: on heavy-tailed maps an output whose own window is small next to a large neighbour
# in the same tile shows the largest ratio.  Measured on the GPU (printed by the tests): F(4, 3) x F(3, 3) c = 38 .. 65, polyphase 5x5 / 2 24 .. 27, F(5x5, 4x4) 49 .. 70.
# The bounds are 2 x the largest measured constant; max-normalised error and the signal check of the logits are asserted beside them.
WINO_BOUND = {"3x3": 130.0, "5x5s2": 54.0, "7x7s2": 140.0}


def backward_error(got, x_nchw, w, b, stride, pad, relu):
    ref = F.conv2d(x_nchw.double(), w.double(), b.double(), stride=stride, padding=pad)
    mag = F.conv2d(x_nchw.double().abs(), w.double().abs(), b.double().abs(), stride=stride, padding=pad)
    if relu:
        ref = F.relu(ref)
    err = (got.permute(0, 3, 1, 2).double().cpu() - ref).abs()
    return (err / mag).max().item() * 2.0 ** 24, (err.max() / ref.abs().max()).item()


@pytest.mark.parametrize("ci,co,n", [(128, 128, 5), (128, 512, 3), (256, 256, 4), (832, 256, 2)])
def test_winograd_conv3x3_heavy_tail_backward_error(rt, ci, co, n):
    g = torch.Generator().manual_seed(3 * ci + co)
    x = heavy_tail((n, 7, 7, ci), 0x3300 + ci)
    w = (torch.rand(co, ci, 3, 3, generator=g) * 2 - 1) / (9 * ci) ** 0.5
    b = (torch.rand(co, generator=g) * 2 - 1) / (9 * ci) ** 0.5
    got = rt.winograd_conv3x3(dev(x), dev(w), dev(b), flags=2)
    torch.cuda.synchronize()
    c, fwd = backward_error(got, x.permute(0, 3, 1, 2), w, b, 1, 1, True)
    print("winograd 3x3 %d -> %d heavy tail: max err / (2^-24 sum|w x|) = %.1f, err / max|ref| = %.2e" % (ci, co, c, fwd))
    assert c < WINO_BOUND["3x3"] and fwd < RTOL


@pytest.mark.parametrize("ci,co,n", [(64, 64, 2), (1056, 128, 3)])
def test_winograd_conv5x5s2_heavy_tail_backward_error(rt, ci, co, n):
    g = torch.Generator().manual_seed(5 * ci + co)
    x = heavy_tail((n, 14, 14, ci), 0x5500 + ci)
    w = (torch.rand(co, ci, 5, 5, generator=g) * 2 - 1) / (25 * ci) ** 0.5
    b = (torch.rand(co, generator=g) * 2 - 1) / (25 * ci) ** 0.5
    got = rt.winograd_conv5x5s2(dev(x), dev(w), dev(b), flags=2)
    torch.cuda.synchronize()
    c, fwd = backward_error(got, x.permute(0, 3, 1, 2), w, b, 2, 2, True)
    print("polyphase winograd 5x5/2 %d -> %d heavy tail: max err / (2^-24 sum|w x|) = %.1f, err / max|ref| = %.2e" % (ci, co, c, fwd))
    assert c < WINO_BOUND["5x5s2"] and fwd < RTOL


@pytest.mark.parametrize("ci,co,n", [(64, 64, 2), (320, 64, 3)])
def test_winograd_conv7x7s2_heavy_tail_backward_error(rt, ci, co, n):
    g = torch.Generator().manual_seed(7 * ci + co)
    x = heavy_tail((n, 28, 28, ci), 0x7700 + ci)
    w = (torch.rand(co, ci, 7, 7, generator=g) * 2 - 1) / (49 * ci) ** 0.5
    b = (torch.rand(co, generator=g) * 2 - 1) / (49 * ci) ** 0.5
    got = rt.winograd_conv7x7s2(dev(x), dev(w), dev(b))
    torch.cuda.synchronize()
    c, fwd = backward_error(got, x.permute(0, 3, 1, 2), w, b, 2, 3, False)
    print("polyphase winograd 7x7/2 %d -> %d heavy tail: max err / (2^-24 sum|w x|) = %.1f, err / max|ref| = %.2e" % (ci, co, c, fwd))
    assert c < WINO_BOUND["7x7s2"] and fwd < RTOL


@pytest.mark.parametrize("n", [3, 11])
def test_bottleneck_chain14_heavy_tail(rt, n):
    """chain14_kernel (residual form) on a heavy-tailed post-ReLU input against an fp64 chain."""
    g = torch.Generator().manual_seed(900 + n)
    x = heavy_tail((n, 14, 14, 256), 0x1400 + n)
    w1 = (torch.rand(64, 256, generator=g) * 2 - 1) / 16.0
    b1 = (torch.rand(64, generator=g) * 2 - 1) / 16.0
    w2 = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    b2 = (torch.rand(64, generator=g) * 2 - 1) / 24.0
    w3 = (torch.rand(256, 64, generator=g) * 2 - 1) / 8.0
    b3 = (torch.rand(256, generator=g) * 2 - 1) / 8.0
    xin = x.permute(0, 3, 1, 2).double()
    t1 = F.relu(F.conv2d(xin, w1.double()[:, :, None, None], b1.double()))
    t2 = F.relu(F.conv2d(t1, w2.double(), b2.double(), padding=1))
    want = F.relu(F.conv2d(t2, w3.double()[:, :, None, None], b3.double()) + xin).permute(0, 2, 3, 1)
    got = rt.bottleneck_chain14(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3), dev(b3), res=dev(x))
    torch.cuda.synchronize()
    err = rel_err(got, want)
    print("chain14 n=%d heavy tail: max error / max |ref| = %.2e" % (n, err))
    assert err < 1e-5


def test_mirror_picks_up_fusion_conv_updates_at_winograd_sizes(rt):
    """A fusion-conv weight written after the first forward at P >= 40 (B = 7, L = 7: P = 42 -- the 7x7 / 5x5 / 3x3 convs all on
    their Winograd paths) must reach the TRANSFORMED weights (wino_u7, wino_u[..]) before the next forward."""
    from offk_amd import off_module
    B, L = 7, 7
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    net = off_module.OFFSubNetwork(101, B, L, "rgb").cuda()
    w0 = synth.make_weights(spec.VARIANT_RGB)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    a = [t.clone() for t in net(feats)]
    keys = ("motion_conv_trans_28", "motion_conv_trans_14", "motion_conv3_trans_14b", "motion_conv_trans", "motion_conv2_trans")
    with torch.no_grad():
        for i, k in enumerate(keys):
            getattr(net, k).weight.mul_(1.25 + 0.125 * i)
    b = net(feats)
    w1 = dict(w0)
    for i, k in enumerate(keys):
        w1[k + ".weight"] = w0[k + ".weight"] * np.float32(1.25 + 0.125 * i)
    with torch.no_grad():
        want = orc.off_forward([f.cpu() for f in feats], orc.to_torch_weights(w1), B, L, 0, orc.SLICE_FLAT)
    for x, y in zip(b, want):
        assert rel_err(x, y) < RTOL and signal_err(x, y) < RTOL_NORTH_STAR
    assert not torch.equal(a[0], b[0])


def test_chain_winograd_weights_follow_a_weight_update(rt, monkeypatch):
    """A 3x3 conv of a bottleneck chain written after the first forward must reach its F(2x2, 3x3) weights (chain_u2, re-made by the
    weight finaliser) before the next forward: gates open (B = 3), against a fresh handle holding the updated weights."""
    B, L = 3, 7
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    h, _w = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB)
    a = [t.clone() for t in h.forward(feats)]
    w1 = dict(synth.make_weights(spec.VARIANT_RGB))
    for i, k in enumerate(("motion_conv2_trans_28a", "motion_conv2_trans_28b", "motion_conv2_trans_28c")):
        w1[k + ".weight"] = w1[k + ".weight"] * np.float32(1.5 + 0.25 * i)
        h.set_weight(k + ".weight", torch.from_numpy(w1[k + ".weight"]))
    b = h.forward(feats)
    h2, _w2 = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB, weights=w1)
    c = h2.forward(feats)
    torch.cuda.synchronize()
    for x, y in zip(b, c):
        assert torch.equal(x, y)
    assert not torch.equal(a[2], b[2])


def test_modality_fuse_keeps_the_backbone_score_differentiable(rt):
    """`fc7 + Feature_Generation_Score + fc14` (Flow_OFF.py:881) runs as K7 on raw pointers at inference; when the
    backbone's score requires grad (fine-tuning the TSN stream through the fused score) the sum must stay on the autograd graph:
    d fused / d fgs = 1 / L per frame (SegmentConsensus avg, basic_ops.py:19-21, 29-33)."""
    from offk_amd import off_module
    from .test_gpu_parity import _StubBackbone
    B, L = 2, 3
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    fgs = torch.randn(B * L, 101, device="cuda", requires_grad=True)
    m = off_module.bninception_off(101, B, L, variant="flow", backbone=_StubBackbone(feats, fgs))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(spec.VARIANT_FLOW).items()}, strict=False)
    m.modality_fuse = True
    frames = torch.zeros(B * L, 10, 8, 8, device="cuda")
    fused = m(frames)
    assert fused.requires_grad
    fused.sum().backward()
    assert torch.allclose(fgs.grad, torch.full_like(fgs, 1.0 / L))
    with torch.no_grad():                                   # inference: the K6 + K7 path, same numbers
        fused2 = m(frames)
    assert not fused2.requires_grad and rel_err(fused2, fused.detach()) < 1e-6


def test_single_rank_rccl_smoke():
    """the RCCL leg of the N > 1 path (librccl init, all_gather_into_tensor and the zero-buffer all_reduce
    form, both through offk_amd.dist) executed on the one GPU of the box with a world_size-1 group, inside bench.py's timed step.
    A child process (spawned, never exec'ed over this one): it owns its process group."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--collective-smoke", "--batch", "8", "--steps", "3",
                        "--warmup", "1", "--no-secondary"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["collective_backend"] == "nccl" and line["n_ranks_seen"] == 1 and line["exchange_ok"] is True
    assert "SINGLE-RANK RCCL SMOKE" in line["config"]["workload"] and line["single_rank_rccl_smoke"]["world_size"] == 1



# ---- round 4: what sits between two Winograd convs (wino_mid.hip) through its C-ABI entry point ----------------------------------

# the 1-D transforms of winograd.hip as matrices: F(4, 3) (points 0, +-1, +-2, inf) and F(3, 3) (points 0, 1, -1, 2, inf)
_AT = {4: np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64),
       3: np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]], dtype=np.float64)}
_BT = {6: np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=np.float64),
       5: np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], dtype=np.float64)}


def _mindex(phases, cy, cx, i, j):
    """Point numbering of winograd.hip (wino_mindex): class (cy, cx) has NY x NX points."""
    ny, nx, cls = (5 if cy else 6), (5 if cx else 6), 2 * cy + cx
    if phases == 1:
        return (0, 36, 66, 96)[cls] + i * nx + j
    offa, offb, offc = (0, 25, 45, 65), (0, 5, 9, 14), (0, 5, 10, 14)
    if i < ny - 1 and j < nx - 1:
        return offa[cls] + i * (nx - 1) + j
    if i == ny - 1 and j < nx - 1:
        return 81 + offb[cls] + j
    if j == nx - 1 and i < ny - 1:
        return 99 + offc[cls] + i
    return 117 + cls


def _between_reference(M, bias, phases, w1, b1):
    """fp64 restatement: x = relu(A^T M A + bias) per class -> [n, 7, 7, Cin]; t = relu(x W1^T + b1); V = B^T t B per class."""
    _pts, n, cin = M.shape
    x = np.zeros((n, 7, 7, cin))
    for cy in (0, 1):
        for cx in (0, 1):
            ny, nx = (5 if cy else 6), (5 if cx else 6)
            m = np.stack([np.stack([M[_mindex(phases, cy, cx, i, j)] for j in range(nx)], 0) for i in range(ny)], 0)   # [ny, nx, n, c]
            y = np.einsum("oi,ijnc,pj->opnc", _AT[3 if cy else 4], m, _AT[3 if cx else 4])
            oy, ox = (4 if cy else 0), (4 if cx else 0)
            x[:, oy:oy + y.shape[0], ox:ox + y.shape[1]] = np.transpose(y, (2, 0, 1, 3))
    x = np.maximum(x + bias, 0.0)
    t = np.maximum(x @ w1.T + b1, 0.0) if w1 is not None else x
    cm = t.shape[-1]
    tp = np.zeros((n, 9, 9, cm))
    tp[:, 1:8, 1:8] = t                                     # zero padding 1
    V = np.zeros((121, n, cm))
    for cy in (0, 1):
        for cx in (0, 1):
            ny, nx = (5 if cy else 6), (5 if cx else 6)
            oy, ox = (4 if cy else 0), (4 if cx else 0)
            d = tp[:, oy:oy + ny, ox:ox + nx]                # window rows oy - 1 .. (padded index = image index + 1)
            v = np.einsum("pi,nijc,qj->pqnc", _BT[ny], d, _BT[nx])
            for i in range(ny):
                for j in range(nx):
                    V[_mindex(1, cy, cx, i, j)] = v[i, j]
    return x, V


# (the forms with a 1x1 conv in both arithmetic modes -- stage B is what runs in split arithmetic; the forms without one have no such mode)
@pytest.mark.parametrize("cin,phases,gemm,prec", [(128, 4, True, "fp32"), (128, 1, True, "fp32"), (256, 1, True, "fp32"), (128, 1, False, "fp32"),
                                                  (256, 1, False, "fp32"), (128, 4, True, "f32split"), (128, 1, True, "f32split"), (256, 1, True, "f32split")])
@pytest.mark.parametrize("n", [1, 5])
def test_winograd_between_vs_fp64(rt, cin, phases, gemm, n, prec):
    """offk_winograd_between (wino_mid.hip: output transform + ReLU [+ 1x1 conv + ReLU] + input transform in one launch) against an
    fp64 restatement with the transform matrices written out, for the five instantiations the forward uses
    (RGB_OFF.py:762-767, :775-780, :833-838), including the store of the first conv's activation into a channel slice."""
    g = np.random.default_rng(1000 * cin + 10 * phases + n)
    M = g.standard_normal((121, n, cin)).astype(np.float32)
    bias = (g.standard_normal(cin) * 0.5).astype(np.float32)
    w1 = (g.standard_normal((cin, cin)) / cin ** 0.5).astype(np.float32) if gemm else None
    b1 = (g.standard_normal(cin) * 0.1).astype(np.float32) if gemm else None
    xbuf = torch.full((n, 7, 7, cin + 64), -3.0, device="cuda")
    V = rt.winograd_between(dev(M), dev(bias), phases, dev(w1) if gemm else None, dev(b1) if gemm else None, x=xbuf, x_coff=32, precision=prec)
    torch.cuda.synchronize()
    x_ref, v_ref = _between_reference(M.astype(np.float64), bias.astype(np.float64), phases,
                                      w1.astype(np.float64) if gemm else None, b1.astype(np.float64) if gemm else None)
    ex = rel_err(xbuf[..., 32:32 + cin], x_ref)
    ev = rel_err(V, v_ref)
    print("winograd between Cin %d phases %d gemm %d n %d %s: x %.2e V %.2e" % (cin, phases, gemm, n, prec, ex, ev))
    assert ex < 1e-5 and ev < 2e-5
    assert torch.all(xbuf[..., :32] == -3.0) and torch.all(xbuf[..., 32 + cin:] == -3.0)
    V2 = rt.winograd_between(dev(M), dev(bias), phases, dev(w1) if gemm else None, dev(b1) if gemm else None, precision=prec)     # no x store
    assert torch.equal(V, V2)


def test_winograd_between_off_matches_on(rt, monkeypatch):
    """OFFK_WINO_MID=0 at offk_create keeps the three-launch form: same transforms, the generic 1x1 kernel in between.  Logits agree
    to fp32 summation-order noise, every stage tensor against the oracle in both forms."""
    B, L = 12, 7
    feats_np = synth.make_features(B, L, 2)
    feats = [dev(f) for f in feats_np]
    h1, w = make_handle(rt, B, L, spec.VARIANT_RGB)
    names = launches_of(h1, feats)
    assert sum("[winograd: between]" in n for n in names) == 3, names
    monkeypatch.setenv("OFFK_WINO_MID", "0")
    h0, _ = make_handle(rt, B, L, spec.VARIANT_RGB)
    monkeypatch.delenv("OFFK_WINO_MID")
    assert not any("[winograd: between]" in n for n in launches_of(h0, feats))
    a, b = h1.forward(feats), h0.forward(feats)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert rel_err(x, y.cpu()) < 5e-6
    with torch.no_grad():
        _want, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, spec.VARIANT_RGB, orc.SLICE_FLAT, return_stages=True)
    for h in (h1, h0):
        errs = stage_errs(h, st, B * (L - 1))
        assert max(errs.values()) < RTOL, errs


@pytest.mark.parametrize("B,kind", [(3, None), (12, None), (12, "heavy_tail"), (64, None)])
def test_chain_winograd_3x3_vs_direct_and_oracle(rt, monkeypatch, B, kind):
    """The 3x3 conv inside chain14_kernel in Winograd F(2x2, 3x3) form (the default; transformed weights re-made by the weight
    finaliser) against the direct form (OFFK_CHAIN_WINO=0) and against the oracle's stage tensors, gates forced open."""
    L = 7
    feats_np = synth.make_features(B, L, 2) if kind is None else synth.make_features_kind(B, L, 3, kind)
    feats = [dev(f) for f in feats_np]
    h1, w = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB)
    monkeypatch.setenv("OFFK_CHAIN_WINO", "0")
    h0, _ = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB)
    monkeypatch.delenv("OFFK_CHAIN_WINO")
    a, b = h1.forward(feats), h0.forward(feats)
    torch.cuda.synchronize()
    P = B * (L - 1)
    f1 = h1.region("fusion_14", 1056).view(P, 14, 14, 1056)[..., 800:]
    f0 = h0.region("fusion_14", 1056).view(P, 14, 14, 1056)[..., 800:]
    e_chain = rel_err(f1, f0.cpu())
    e_logit = max(rel_err(x, y.cpu()) for x, y in zip(a, b))
    print("chain winograd vs direct (B = %d, %s): sum_28c %.2e, logits %.2e" % (B, kind, e_chain, e_logit))
    assert 0 < e_chain < 2e-5 and e_logit < 2e-5
    if B <= 12:
        with torch.no_grad():
            want, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], w, B, L, spec.VARIANT_RGB, orc.SLICE_FLAT, return_stages=True)
        errs = stage_errs(h1, st, P)
        assert max(errs.values()) < RTOL, errs
        assert signal_err(a[0], want[0]) < RTOL_NORTH_STAR


@pytest.mark.parametrize("B", [3, 12, 64])
def test_wino_gemm_persistent_matches_generic(rt, monkeypatch, B):
    """The batched GEMMs of every Winograd conv as one persistent launch (wino_gemm.hip: a block works through its tiles as one stream
    of K-tiles) against OFFK_WINO_GEMM=0 (one block of the generic 1x1 kernel per tile): same tile, same MFMA sequence, same k order --
    logits and every stage tensor must be BIT-identical, with every gate forced open.  B = 3: fewer items than resident blocks; 12:
    a few per block; 64: the benchmark shape, grouped launches with three K lengths."""
    L = 7
    feats = [dev(f) for f in synth.make_features(B, L, 2)]
    h1, _ = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB)
    monkeypatch.setenv("OFFK_WINO_GEMM", "0")
    h0, _ = forced_handle(rt, monkeypatch, B, L, spec.VARIANT_RGB)
    monkeypatch.delenv("OFFK_WINO_GEMM")
    a, b = h1.forward(feats), h0.forward(feats)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    P = B * (L - 1)
    for name, ch, H in STAGES:
        assert torch.equal(h1.region(name, ch), h0.region(name, ch)), name
    for name in ("wino_m",):       # the GEMM output of the last Winograd conv of the forward itself
        assert torch.equal(h1.region(name, 1), h0.region(name, 1)), name


