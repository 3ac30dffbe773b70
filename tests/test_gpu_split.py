"""Split-fp32 arithmetic (OFFK_PRECISION_F32SPLIT, round 5): fp32 operands as three bf16 planes on the bf16 matrix pipe.

The claim these tests pin: the split kernels are fp32 arithmetic in the sense that matters for parity -- against an fp64
contraction of the same fp32 inputs their error is NO LARGER than the error of the library's own fp32-pipe kernels
(v_mfma_f32_16x16x4_f32 / 32x32x2_f32: a chain of fp32 FMAs) -- on every input distribution the parity suite uses: the
synthetic maps, 24-bit mantissas, a heavy tail, and the cancellation case.  Reference arithmetic: RGB_OFF.py:597-610 (fp32
nn.Conv2d); the tolerance north_star states is 1e-3 relative, both modes sit four orders inside it.

Reported per mode (print, -s shows it; bench.py carries the same numbers in `f32split_mode.error_vs_fp64`):
    max |err| / max |ref|, rms err / max |ref|, and c of |err| <= c 2^-24 sum_k |w_k x_k| (max and rms over elements).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import offk_amd  # noqa: F401
from offk_amd import spec, synth

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -24
KINDS = ["synth", "full_mantissa", "heavy_tail"]


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from offk_amd import runtime
    return runtime


def dev(x):
    return torch.as_tensor(x).to("cuda").contiguous()


def make_handle(rt, B, L, precision, weights=None, variant=spec.VARIANT_RGB):
    h = rt.OffForward(B, L, variant, spec.SLICE_FLAT, None, precision=precision)
    w = synth.make_weights(variant) if weights is None else weights
    assert h.load_state_dict(w) == []
    return h, w


def units_reference_fp64(feats, w, B, L):
    """Per site: (T, magT, D, magD) in fp64.  T[b (L-1) + t] = relu(G)[b L + t + 1] - relu(G)[b L + t]  (RGB_OFF.py:597-604),
    D = down(X[:P]) (quirk Q1, :609-610); mag = sum_k |w_k x_k| (+ |bias|), for T the two frames' magnitudes added."""
    out = []
    P = B * (L - 1)
    for (name, C, H), x in zip(spec.SITES, feats):
        xd = torch.from_numpy(x).double()
        wg = torch.from_numpy(w["motion_conv_gen_%s.weight" % name]).double()
        bg = torch.from_numpy(w["motion_conv_gen_%s.bias" % name]).double()
        wd = torch.from_numpy(w["motion_spatial_down_%s.weight" % name]).double()
        bd = torch.from_numpy(w["motion_spatial_down_%s.bias" % name]).double()
        G = torch.relu(F.conv2d(xd, wg, bg)).view(B, L, 128, H, H)
        mG = (F.conv2d(xd.abs(), wg.abs()) + bg.abs().view(1, -1, 1, 1)).view(B, L, 128, H, H)
        T = (G[:, 1:] - G[:, :-1]).reshape(P, 128, H, H)
        mT = (mG[:, 1:] + mG[:, :-1]).reshape(P, 128, H, H)
        D = F.conv2d(xd[:P], wd, bd)
        mD = F.conv2d(xd[:P].abs(), wd.abs()) + bd.abs().view(1, -1, 1, 1)
        out.append((T, mT, D, mD))
    return out


def units_outputs(h, B, L):
    """(T, D) per site as the fused units left them: T in the fusion buffers (channels coff + 32 .. coff + 160), D in D_<site>."""
    P = B * (L - 1)
    res = []
    for fkey, fd in spec.FUSION.items():
        width = 160 * len(fd["sites"]) + fd["carry"]
        buf = h.region("fusion_" + fkey, width).view(P, fd["H"], fd["H"], width)
        for i, sname in enumerate(fd["sites"]):
            T = buf[..., 160 * i + 32:160 * i + 160].permute(0, 3, 1, 2).double().cpu()
            D = h.region("D_" + sname, 32).view(P, fd["H"], fd["H"], 32).permute(0, 3, 1, 2).double().cpu()
            res.append((T, D))
    return res


def error_stats(pairs):
    """pairs: [(got, ref, mag)] -> dict of the four numbers, over all elements of all tensors (each tensor normalised by its own max)."""
    mx, cmx, se, sc, n = 0.0, 0.0, 0.0, 0.0, 0
    for got, ref, mag in pairs:
        e = (got - ref).abs()
        s = ref.abs().max().clamp_min(1e-30)
        c = e / (mag.clamp_min(1e-30) * EPS)
        mx = max(mx, (e.max() / s).item())
        cmx = max(cmx, c.max().item())
        se += ((e / s) ** 2).sum().item()
        sc += (c ** 2).sum().item()
        n += e.numel()
    return {"max_over_max": mx, "rms_over_max": (se / n) ** 0.5, "c_max": cmx, "c_rms": (sc / n) ** 0.5}


def run_units(rt, B, L, precision, feats_np, weights=None):
    h, w = make_handle(rt, B, L, precision, weights)
    h.off_units_fused([dev(f) for f in feats_np])
    torch.cuda.synchronize()
    return units_outputs(h, B, L), w


@pytest.mark.parametrize("kind", KINDS)
def test_units_split_error_is_no_larger_than_the_fp32_pipes(rt, kind):
    B, L = 2, 7
    feats_np = synth.make_features_kind(B, L, 4, kind)
    stats = {}
    ref = None
    for prec in ("fp32", "f32split"):
        outs, w = run_units(rt, B, L, prec, feats_np)
        if ref is None:
            ref = units_reference_fp64(feats_np, w, B, L)
        pairs = []
        for (T, D), (Tr, mT, Dr, mD) in zip(outs, ref):
            pairs += [(T, Tr, mT), (D, Dr, mD)]
        stats[prec] = error_stats(pairs)
        print("units %-8s on %-13s maps vs fp64: max %.3e  rms %.3e  c_max %.3f  c_rms %.4f"
              % (prec, kind, stats[prec]["max_over_max"], stats[prec]["rms_over_max"], stats[prec]["c_max"], stats[prec]["c_rms"]))
    for key in ("max_over_max", "rms_over_max", "c_max", "c_rms"):
        assert stats["f32split"][key] <= stats["fp32"][key], (kind, key, stats)
    assert stats["f32split"]["max_over_max"] < 2e-6 and stats["f32split"]["c_max"] < 16.0


def test_units_split_cancellation_case(rt):
    """tests/test_gpu_parity.py::test_pw_reduce_cancellation_case on the fused kernels: every channel of a pixel carries the
    same value and every weight row sums to zero -- the exact G is relu(bias), the exact T is 0, the exact D is the bias."""
    B, L = 2, 7
    wnp = synth.make_weights(spec.VARIANT_RGB)
    for name, _C, _H in spec.SITES:
        for key in ("motion_conv_gen_%s.weight" % name, "motion_spatial_down_%s.weight" % name):
            wk = wnp[key].astype(np.float64)
            wnp[key] = (wk - wk.mean(axis=1, keepdims=True)).astype(np.float32)
    base = synth.make_features_kind(B, L, 4, "heavy_tail")
    feats_np = [np.ascontiguousarray(np.broadcast_to(f[:, :1] + np.float32(0.5), f.shape)) for f in base]
    stats, ref = {}, None
    for prec in ("fp32", "f32split"):
        outs, w = run_units(rt, B, L, prec, feats_np, wnp)
        if ref is None:
            ref = units_reference_fp64(feats_np, w, B, L)
        pairs = [(D, Dr, mD) for (_T, D), (_Tr, _mT, Dr, mD) in zip(outs, ref)]
        stats[prec] = error_stats(pairs)
        print("units %-8s cancellation case (D rows) vs fp64: max err / max|out| %.3e  c_max %.3f  c_rms %.4f"
              % (prec, stats[prec]["max_over_max"], stats[prec]["c_max"], stats[prec]["c_rms"]))
    for key in ("max_over_max", "rms_over_max", "c_max", "c_rms"):
        assert stats["f32split"][key] <= stats["fp32"][key], (key, stats)
    assert stats["f32split"]["c_max"] < 4.0


@pytest.mark.parametrize("B,L", [(1, 2), (3, 3), (2, 9), (5, 7)])
def test_units_split_shapes(rt, B, L):
    """Short clips (frames past the group read zeros), two temporal groups (L = 9), odd batches (packed 14x14 leftovers, the 7x7
    quad stream crossing clip boundaries): every T and D element against fp64."""
    feats_np = synth.make_features(B, L, 5)
    outs, w = run_units(rt, B, L, "f32split", feats_np)
    ref = units_reference_fp64(feats_np, w, B, L)
    for (name, _C, _H), (T, D), (Tr, _mT, Dr, _mD) in zip(spec.SITES, outs, ref):
        assert ((T - Tr).abs().max() / Tr.abs().max()).item() < 2e-6, name
        assert ((D - Dr).abs().max() / Dr.abs().max()).item() < 2e-6, name


# ---- the batched GEMMs of the Winograd convs in split-fp32 arithmetic (wino_gemm_split.hip) ----

def gemm_inputs(batch, M, K, Co, kind, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, M, K, generator=g)
    w = torch.randn(batch, Co, K, generator=g) * (1.0 / K ** 0.5)
    if kind == "relu":                # what V looks like behind a ReLU and B^T . B: mostly small, some zeros
        x = torch.relu(x)
    elif kind == "heavy_tail":
        x = x * torch.exp(2.0 * torch.randn(batch, M, 1, generator=g))
    elif kind == "cancellation":      # every row of w sums to zero, x constant along k (+ noise of 2^-12): the result is the noise's
        w = w - w.mean(dim=2, keepdim=True)
        x = 3.0 + torch.randn(batch, M, 1, generator=g) + torch.randn(batch, M, K, generator=g) * 2.0 ** -12
    return x.float().contiguous(), w.float().contiguous()


@pytest.mark.parametrize("batch,M,K,Co", [(5, 384, 832, 256), (121, 384, 256, 256), (3, 100, 128, 128), (7, 64, 64, 512),
                                          (2, 777, 1056, 128), (9, 200, 320, 64), (3, 130, 1280, 192), (70, 3456, 64, 64)])
def test_batched_gemm_split_shapes(rt, batch, M, K, Co):
    """Item streams longer than the persistent grid (121 x 6 x 2 items; 70 x 54 of the 64-channel form), row tails (M % 64 != 0), the
    shortest K the kernel takes (two K-tiles: the three cursors cross item boundaries on consecutive steps), Co = 64 / 192 (the
    64-channel form), every element against fp64."""
    x, w = gemm_inputs(batch, M, K, Co, "relu", 11)
    ref = torch.einsum("bmk,bnk->bmn", x.double(), w.double())
    y = rt.batched_gemm_nt(dev(x), dev(w), "f32split").double().cpu()
    assert ((y - ref).abs().max() / ref.abs().max()).item() < 1e-6
    y32 = rt.batched_gemm_nt(dev(x), dev(w), "fp32").double().cpu()
    assert ((y32 - ref).abs().max() / ref.abs().max()).item() < 3e-6


@pytest.mark.parametrize("kind", ["relu", "normal", "heavy_tail", "cancellation"])
def test_batched_gemm_split_error_is_no_larger_than_the_fp32_pipes(rt, kind):
    batch, M, K, Co = 6, 384, 832, 256
    x, w = gemm_inputs(batch, M, K, Co, kind, 5)
    ref = torch.einsum("bmk,bnk->bmn", x.double(), w.double())
    mag = torch.einsum("bmk,bnk->bmn", x.double().abs(), w.double().abs())
    stats = {}
    for prec in ("fp32", "f32split"):
        y = rt.batched_gemm_nt(dev(x), dev(w), prec).double().cpu()
        stats[prec] = error_stats([(y, ref, mag)])
        print("batched GEMM %-8s on %-12s inputs vs fp64: max %.3e  rms %.3e  c_max %.3f  c_rms %.4f"
              % (prec, kind, stats[prec]["max_over_max"], stats[prec]["rms_over_max"], stats[prec]["c_max"], stats[prec]["c_rms"]))
    for key in ("max_over_max", "rms_over_max", "c_max", "c_rms"):
        assert stats["f32split"][key] <= stats["fp32"][key], (kind, key, stats)


@pytest.mark.parametrize("B", [2, 12])
def test_forward_split_gemms_against_fp32_pipe_gemms(rt, monkeypatch, B):
    """A split-fp32 handle with its GEMMs / 1x1 convs on the bf16 pipe (default) against the same handle with OFFK_SPLIT_GEMM=0 (only the
    units kernel split): same logits and stage tensors to summation-order noise (the convs' own Winograd error is 1e-5 of max |ref|), and
    the 7-head's folded pool of the split kernel's epilogue gives the same logits."""
    L = 7
    feats = [dev(f) for f in synth.make_features(B, L, 3)]
    h1, _ = make_handle(rt, B, L, "f32split")
    monkeypatch.setenv("OFFK_SPLIT_GEMM", "0")
    h0, _ = make_handle(rt, B, L, "f32split")
    monkeypatch.delenv("OFFK_SPLIT_GEMM")
    a, b = h1.forward(feats), h0.forward(feats)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert ((x - y).abs().max() / y.abs().max()).item() < 2e-5
        sig = (y - y.mean(dim=0, keepdim=True)).abs().max().item()        # the logits are bias-dominated: compare on the row-to-row signal
        assert (x - y).abs().max().item() < 1e-3 * sig
    for name, ch in (("sum_7", 1024), ("fusion_7", 832)):
        x, y = h1.region(name, ch), h0.region(name, ch)
        assert ((x - y).abs().max() / y.abs().max()).item() < 5e-5, name


# ---- the whole forward of both modes against an fp64 evaluation of the oracle ----

@pytest.mark.parametrize("case", ["rgb_golden_inputs", "flow_golden_inputs", "rgb_heavy_tail"])
def test_whole_forward_error_vs_fp64_oracle(rt, case):
    """oracle/off_oracle.py evaluated in fp64 (torch CPU, the same op sequence on double tensors: RGB_OFF.py:596-847) on the inputs of the
    B = 3 x 7 goldens (synth.make_features(3, 7, cfg) -- what gen_golden.py fed the reference) and on heavy-tailed maps: the error of
    `sum_7` and of the three logit tensors of BOTH arithmetic modes.  The split mode may not be further from fp64 than the fp32 pipe is
    (10 % slack: what both share -- the fp32 Winograd transforms -- dominates the whole-forward error and is the same code)."""
    from oracle import off_oracle as orc
    B, L = 3, 7
    variant = spec.VARIANT_FLOW if case.startswith("flow") else spec.VARIANT_RGB
    cfg = 2 if variant == spec.VARIANT_RGB else 3
    feats_np = synth.make_features_kind(B, L, cfg, "heavy_tail" if case.endswith("heavy_tail") else "synth")
    wnp = synth.make_weights(variant)
    w64 = dict((k, v.double()) for k, v in orc.to_torch_weights(wnp).items())
    with torch.no_grad():
        ref, st = orc.off_forward([torch.from_numpy(f).double() for f in feats_np], w64, B, L, variant, orc.SLICE_FLAT, consensus=False,
                                  return_stages=True)
    P = B * (L - 1)
    errs = {}
    for prec in ("fp32", "f32split"):
        h = rt.OffForward(B, L, variant, spec.SLICE_FLAT, False, precision=prec)
        assert h.load_state_dict(wnp) == []
        out = h.forward([dev(f) for f in feats_np])
        torch.cuda.synchronize()
        s7 = h.region("sum_7", 1024).view(P, 7, 7, 1024).permute(0, 3, 1, 2).double().cpu()
        e7 = (s7 - st["sum_7"]).abs()
        el = [(o.double().cpu() - r).abs() for o, r in zip(out, ref)]
        sig = [(r - r.mean(0, keepdim=True)).abs().max() for r in ref]      # the logits are bias-dominated: normalise by the row-to-row signal
        errs[prec] = {"sum_7_max": (e7.max() / st["sum_7"].abs().max()).item(), "sum_7_rms": ((e7 ** 2).mean().sqrt() / st["sum_7"].abs().max()).item(),
                      "logit_signal_max": max((e.max() / s).item() for e, s in zip(el, sig))}
        print("whole forward %-8s %-18s vs fp64 oracle: sum_7 max %.2e rms %.2e, logits / row-to-row signal %.2e"
              % (prec, case, errs[prec]["sum_7_max"], errs[prec]["sum_7_rms"], errs[prec]["logit_signal_max"]))
        assert errs[prec]["sum_7_max"] < 2e-4 and errs[prec]["logit_signal_max"] < 1e-3
    for key in ("sum_7_max", "sum_7_rms", "logit_signal_max"):
        assert errs["f32split"][key] <= 1.1 * errs["fp32"][key], (key, errs)


def test_split_nonfinite_and_tiny_inputs(rt):
    """what include/offk.h states next to OFFK_PRECISION_F32SPLIT.  +-Inf in an operand: NaN in every output it touches (the
    fp32 pipe: +-Inf or NaN) and nowhere else; operands down to 2^-100 (all three planes still bf16 normals or subnormals): the same result
    as the fp32 pipe to the usual error."""
    batch, M, K, Co = 2, 64, 128, 64
    x, w = gemm_inputs(batch, M, K, Co, "normal", 3)
    x[0, 5, 17] = float("inf")
    x[1, 9, 3] = float("-inf")
    y = rt.batched_gemm_nt(dev(x), dev(w), "f32split").cpu()
    y32 = rt.batched_gemm_nt(dev(x), dev(w), "fp32").cpu()
    bad = torch.zeros(batch, M, dtype=torch.bool)
    bad[0, 5] = bad[1, 9] = True
    assert torch.isnan(y[bad]).all() and not torch.isfinite(y32[bad]).any()
    assert torch.isfinite(y[~bad]).all() and torch.isfinite(y32[~bad]).all()
    ref = torch.einsum("bmk,bnk->bmn", x.double(), w.double())
    assert ((y[~bad].double() - ref[~bad]).abs().max() / ref[~bad].abs().max()).item() < 1e-6
    # tiny operands: x scaled by 2^-100 (planes down to 2^-116: bf16 normals), w by 2^-20 -- products ~2^-120, sums fp32 normals
    xs, ws = gemm_inputs(batch, M, K, Co, "normal", 4)
    xs, ws = xs * 2.0 ** -100, ws * 2.0 ** -20
    yt = rt.batched_gemm_nt(dev(xs), dev(ws), "f32split").double().cpu()
    reft = torch.einsum("bmk,bnk->bmn", xs.double(), ws.double())
    assert ((yt - reft).abs().max() / reft.abs().max()).item() < 1e-6


# ---- the bottleneck chains of fusion@28 in split-fp32 arithmetic (chain_split.hip) through their C-ABI entry point (ABI v10) ----

def _chain_inputs(case, n, kind, seed):
    g = torch.Generator().manual_seed(seed)
    branch = case == "28a_branch"
    Cin = 64 if branch else 256
    x = torch.randn(n, 14, 14, Cin + (64 if branch else 0), generator=g) * (1.0 if branch else 0.5)
    if kind == "heavy_tail":
        x = x * torch.exp(1.5 * torch.randn(n, 14, 14, 1, generator=g))
    if not branch:
        x = x.clamp_min(0)                               # sa / sb are post-ReLU
    k1 = 1.0 / Cin ** 0.5
    w1 = (torch.rand(64, Cin, generator=g) * 2 - 1) * k1
    b1 = (torch.rand(64, generator=g) * 2 - 1) * k1
    w2 = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    b2 = (torch.rand(64, generator=g) * 2 - 1) / 24.0
    w3 = (torch.rand(256, 64, generator=g) * 2 - 1) / 8.0
    b3 = (torch.rand(256, generator=g) * 2 - 1) / 8.0
    wb = (torch.rand(256, 64, generator=g) * 2 - 1) / 8.0
    bb = (torch.rand(256, generator=g) * 2 - 1) / 8.0
    return x.float().contiguous(), w1, b1, w2, b2, w3, b3, wb, bb


@pytest.mark.parametrize("kind", ["normal", "heavy_tail"])
@pytest.mark.parametrize("case", ["28a_branch", "28b_residual", "28c_into_slice"])
@pytest.mark.parametrize("n", [1, 5, 11])
def test_bottleneck_chain14_split_vs_fp64(rt, case, n, kind):
    """offk_bottleneck_chain14_split (RGB_OFF.py:658-667 / :670-685) against an fp64 chain: the branch form of block 28a (c1 on relu(x0), the
    branch conv on the pre-ReLU x0, inside the kernel), the residual form, output into a channel slice; odd image counts exercise the grid
    rounding.  Beside it the fp32-pipe kernel (offk_bottleneck_chain14, direct 3x3): the split chain may not be further from fp64."""
    branch = case == "28a_branch"
    x, w1, b1, w2, b2, w3, b3, wb, bb = _chain_inputs(case, n, kind, 300 + n)
    Cin, x_coff = (64, 64) if branch else (256, 0)
    xin = x[..., x_coff:x_coff + Cin].permute(0, 3, 1, 2).double()
    t1 = F.relu(F.conv2d(F.relu(xin) if branch else xin, w1.double()[:, :, None, None], b1.double()))
    t2 = F.relu(F.conv2d(t1, w2.double(), b2.double(), padding=1))
    want = F.conv2d(t2, w3.double()[:, :, None, None], b3.double())
    want = want + (F.conv2d(xin, wb.double()[:, :, None, None], bb.double()) if branch else xin)
    want = F.relu(want).permute(0, 2, 3, 1)
    res = None if branch else dev(x)
    kw = dict(res=res, branch=(dev(wb), dev(bb)) if branch else None, relu_in=branch, x_coff=x_coff)
    if case == "28c_into_slice":
        ybuf = torch.full((n, 14, 14, 352), -3.0, device="cuda")
        rt.bottleneck_chain14_split(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3), dev(b3), y=ybuf, y_coff=64, **kw)
        got = ybuf[..., 64:320]
        assert torch.all(ybuf[..., :64] == -3.0) and torch.all(ybuf[..., 320:] == -3.0)
    else:
        got = rt.bottleneck_chain14_split(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3), dev(b3), **kw)
    # the fp32-pipe kernel on the same chain (28a: its merged form, c3 over [t2 | x0] with K3 = 128)
    if branch:
        w3m = torch.cat([w3, wb], 1).contiguous()
        got32 = rt.bottleneck_chain14(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3m), dev(b3 + bb), relu_in=True, x_coff=x_coff)
    else:
        got32 = rt.bottleneck_chain14(dev(x), dev(w1), dev(b1), dev(w2), dev(b2), dev(w3), dev(b3), res=res)
    torch.cuda.synchronize()
    s = want.abs().max()
    e = ((got.double().cpu() - want).abs().max() / s).item()
    e32 = ((got32.double().cpu() - want).abs().max() / s).item()
    rms = (((got.double().cpu() - want) ** 2).mean().sqrt() / s).item()
    rms32 = (((got32.double().cpu() - want) ** 2).mean().sqrt() / s).item()
    print("chain %-15s n = %2d %-10s vs fp64: split max %.2e rms %.2e | fp32 pipe max %.2e rms %.2e" % (case, n, kind, e, rms, e32, rms32))
    assert e < 2e-6 and e <= 1.05 * e32 and rms <= 1.05 * rms32
