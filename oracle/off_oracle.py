"""ORACLE -- test infrastructure only (never imported by the product path).

CPU restatement, in plain PyTorch functional ops, of the reference's OFF sub-network
forward.  It is the checker the HIP path is compared against and the ``cpu_baseline``
leg of bench.py; only tests/, __graft_entry__.smoke() and bench.py may import it.

Parity status: PINNED.  tests/test_oracle_golden.py checks this file bit-for-bit (max
abs diff 0.0) against tests/golden/*.npz, which oracle/gen_golden.py produced by
importing the reference's own model classes from /root/reference and injecting the
same synthetic feature maps (see that script).

Every function cites the reference lines it follows (paths relative to the reference
repository root).  Shapes: B clips, L segments, N = B*L frames (clip-major),
P = B*(L-1) adjacent pairs; all tensors fp32 NCHW.
"""
import torch
import torch.nn.functional as F

SITES = (("3a", 256, 28), ("3b", 320, 28), ("3c", 576, 14), ("4a", 576, 14), ("4b", 576, 14),
         ("4c", 608, 14), ("4d", 608, 14), ("5a", 1024, 7), ("5b", 1024, 7))
VARIANT_RGB, VARIANT_FLOW = 0, 1
SLICE_FLAT, SLICE_PER_CLIP = 0, 1


def _conv(x, w, key, stride=1, pad=0, groups=1):
    return F.conv2d(x, w[key + ".weight"], w.get(key + ".bias"), stride=stride, padding=pad,
                    groups=groups)


def temporal_diff(g, batch):
    """RGB_OFF.py:599-604: view(B, L*128, H, H); [:,128:] - [:,:-128]; view(-1,128,H,H)."""
    ch, h, wd = g.shape[1], g.shape[2], g.shape[3]
    r = g.reshape(batch, -1, h, wd)
    t = r[:, ch:, :, :] - r[:, :-ch, :, :]
    return t.reshape(-1, ch, h, wd)


def spatial_frames(x, batch, length, slice_mode):
    """RGB_OFF.py:609 -- the reference slices the FLAT frame axis: X[:B*(L-1)] (quirk Q1).

    ``SLICE_PER_CLIP`` is the build's documented alternative (drop each clip's last
    frame, which is what the comment at RGB_OFF.py:608 says); it is not reference
    behaviour and coincides with it only for B == 1.
    """
    p = batch * (length - 1)
    if slice_mode == SLICE_FLAT:
        return x[:p]
    c, h, wd = x.shape[1:]
    return x.reshape(batch, length, c, h, wd)[:, :length - 1].reshape(p, c, h, wd)


def off_unit(x, w, site, batch, length, variant, slice_mode, drop=None, gen_mask=None):
    """One OFF unit -> motion_<site> [P,160,H,H] = cat(spatial 32, temporal 128).

    RGB_OFF.py:596-616 (site 3a; the other eight sites are identical up to names);
    Flow_OFF.py:606-627 for the diagonal-Sobel variant (util.py:52-77).
    ``drop``: training mode -- the multiplier nn.Dropout(p) (:356, applied at :612) would use,
    keep_mask / (1 - p), as a [P,32,H,H] tensor; None = eval (identity).
    ``gen_mask``: gradient tests only -- a [N,128,H,H] 0/1 tensor that replaces the ReLU's own decision
    (g = pre * mask).  Which side of the kink a pre-activation within rounding distance of zero falls on is
    implementation-defined, and one flipped element moves a whole row of the weight gradient; the tests
    pass the device's decision (after checking it differs from this file's only where |pre| ~ 1e-6) so both
    sides differentiate the same piecewise-linear function.
    """
    g = _conv(x, w, "motion_conv_gen_" + site)                             # :597
    g = torch.relu(g) if gen_mask is None else g * gen_mask                # :598
    t = temporal_diff(g, batch)                                            # :599-604
    d = _conv(spatial_frames(x, batch, length, slice_mode), w, "motion_spatial_down_" + site)  # :609-610
    if variant == VARIANT_RGB:
        s = _conv(d, w, "motion_spatial_grad_" + site, pad=1, groups=32)   # :611
    else:
        s = F.conv2d(d, w["sobel_edge_diagonal.conv.weight"], None, padding=1, groups=32)  # Flow_OFF.py:622
    if drop is not None:
        s = s * drop                                                       # :612 (train); identity in eval
    return torch.cat((s, t), dim=1)                                        # :616


def fusion_28(f28, w):
    """RGB_OFF.py:655-685.  Note the 28a branch takes the PRE-ReLU 7x7 output (:665)."""
    x0 = _conv(f28, w, "motion_conv_trans_28", stride=2, pad=3)            # :657
    a = torch.relu(x0)                                                     # :658
    a = torch.relu(_conv(a, w, "motion_conv1_trans_28a"))                  # :659-660
    a = torch.relu(_conv(a, w, "motion_conv2_trans_28a", pad=1))           # :661-662
    a = _conv(a, w, "motion_conv3_trans_28a")                              # :663
    s = torch.relu(a + _conv(x0, w, "motion_conv_branch_28a"))             # :665-667
    for blk in ("28b", "28c"):                                             # :670-685
        a = torch.relu(_conv(s, w, "motion_conv1_trans_" + blk))
        a = torch.relu(_conv(a, w, "motion_conv2_trans_" + blk, pad=1))
        a = _conv(a, w, "motion_conv3_trans_" + blk)
        s = torch.relu(a + s)
    return s                                                               # motion_sum_28c


def fusion_14(f14, w):
    """RGB_OFF.py:759-780."""
    x1 = torch.relu(_conv(f14, w, "motion_conv_trans_14", stride=2, pad=2))   # :762-763
    a = torch.relu(_conv(x1, w, "motion_conv1_trans_14a"))                    # :764-765
    a = torch.relu(_conv(a, w, "motion_conv2_trans_14a", pad=1))              # :766-767
    a = _conv(a, w, "motion_conv3_trans_14a")                                 # :768
    s = torch.relu(a + _conv(x1, w, "motion_conv_expand_trans_14a"))          # :769-771
    a = torch.relu(_conv(s, w, "motion_conv1_trans_14b"))                     # :773-774
    a = torch.relu(_conv(a, w, "motion_conv2_trans_14b", pad=1))              # :775-776
    a = torch.relu(_conv(a, w, "motion_conv3_trans_14b", pad=1))              # :777-778 (3x3, own ReLU)
    return torch.relu(s + a)                                                  # :779-780  motion_sum_14b


def fusion_7(f7, w):
    """RGB_OFF.py:831-841 -- no ReLU after the final add."""
    x2 = torch.relu(_conv(f7, w, "motion_conv_trans", pad=1))                 # :833-834
    a = torch.relu(_conv(x2, w, "motion_conv1_trans"))                        # :835-836
    a = torch.relu(_conv(a, w, "motion_conv2_trans", pad=1))                  # :837-838
    a = _conv(a, w, "motion_conv3_trans")                                     # :839
    return a + _conv(x2, w, "motion_conv_branch_trans")                       # :840-841


def head(x, w, key, maxpool):
    """RGB_OFF.py:782-787 (28), :789-793 (14), :843-847 (7).

    MaxPool2d(3, 2, ceil_mode=True) (:353) then AvgPool2d(7) (:262); dropout is the
    identity in eval; torch.squeeze also drops the pair axis when P == 1 (:786).
    """
    if maxpool:
        x = F.max_pool2d(x, 3, stride=2, ceil_mode=True)
    x = F.avg_pool2d(x, 7, stride=1, padding=0, ceil_mode=True, count_include_pad=True)
    x = torch.squeeze(x)
    return F.linear(x, w[key + ".weight"], w[key + ".bias"])


def segment_consensus(x, batch):
    """basic_ops.py:19-21 ('avg': mean(dim=1, keepdim=True)) applied as in
    Flow_OFF.py:867-876: view(B, L-1, C) -> consensus -> squeeze(1)."""
    return x.reshape(batch, -1, x.shape[-1]).mean(dim=1, keepdim=True).squeeze(1)


def off_forward(feats, w, batch, length, variant=VARIANT_RGB, slice_mode=SLICE_FLAT,
                consensus=None, return_stages=False, unit_drop=None, unit_gen_mask=None):
    """The whole OFF sub-network on nine injected feature maps.

    feats: list of nine [N,C,H,H] fp32 tensors (inception_{3a..5b}_output_out).
    w: dict state_dict-key -> tensor.
    Returns (fc7, fc14, fc28), each [P,101] (or [B,101] after consensus; the
    reference applies it in Flow_OFF.py / RGB_OFF_v2.py only, ``consensus=None``
    picks that default).  With return_stages also a dict of intermediates.
    """
    if consensus is None:
        consensus = (variant == VARIANT_FLOW)
    m = {}
    for si, ((site, _c, _h), x) in enumerate(zip(SITES, feats)):
        m[site] = off_unit(x, w, site, batch, length, variant, slice_mode,
                           None if unit_drop is None else unit_drop[si],
                           None if unit_gen_mask is None else unit_gen_mask[si])
    f28 = torch.cat((m["3a"], m["3b"]), dim=1)                                      # :656
    sum_28c = fusion_28(f28, w)
    f14 = torch.cat((m["3c"], m["4a"], m["4b"], m["4c"], m["4d"], sum_28c), dim=1)  # :760
    sum_14b = fusion_14(f14, w)
    fc28 = head(sum_28c, w, "fc_action_motion_28", True)                            # :782-787
    fc14 = head(sum_14b, w, "fc_action_motion_14", False)                           # :789-793
    f7 = torch.cat((m["5a"], m["5b"], sum_14b), dim=1)                              # :832
    s7 = fusion_7(f7, w)
    fc7 = head(s7, w, "fc_action_motion", False)                                    # :843-847
    if consensus:
        fc7, fc14, fc28 = (segment_consensus(v, batch) for v in (fc7, fc14, fc28))
    if return_stages:
        st = dict(("motion_" + k, v) for k, v in m.items())
        st.update(fusion_28=f28, sum_28c=sum_28c, fusion_14=f14, sum_14b=sum_14b,
                  fusion_7=f7, sum_7=s7)
        return (fc7, fc14, fc28), st
    return fc7, fc14, fc28


def reference_return(feats, w, batch, length, ref_file, fgs_raw, modality_fuse=False, conv2=None,
                     slice_mode=SLICE_FLAT):
    """What the reference's forward hands back, given the backbone's per-frame
    Feature_Generation_Score [N,101] (RGB_OFF.py:592-594) and, for RGB_OFF_v2, its conv2_relu_3x3_out.

    ref_file "rgb":    RGB_OFF.py:860      (fc7, FGS, fc14) -- per-pair logits, no consensus (:849-858 commented out);
                                           torch.squeeze at :786/:792/:846 also drops the pair axis when P == 1.
    ref_file "flow":   Flow_OFF.py:866-884 view(B, L-1, -1) / view(B, L, -1), consensus avg on all four scores, then
                                           fc7 + FGS + fc14 if modality_fuse else (fc7, FGS, fc14).
    ref_file "rgb_v2": RGB_OFF_v2.py:873-891 as flow, the tuple carries conv2_relu_3x3_out as a fourth item.
    """
    variant = VARIANT_RGB if ref_file == "rgb" else VARIANT_FLOW
    fc7, fc14, _fc28 = off_forward(feats, w, batch, length, variant, slice_mode, consensus=False)
    if ref_file == "rgb":
        return fc7, fgs_raw, fc14
    fgs = segment_consensus(fgs_raw.reshape(batch, length, -1), batch)                  # Flow_OFF.py:866, :873
    fc7 = segment_consensus(fc7.reshape(batch, length - 1, -1), batch)                  # :867, :874
    fc14 = segment_consensus(fc14.reshape(batch, length - 1, -1), batch)                # :869, :876
    if modality_fuse:
        return fc7 + fgs + fc14                                                         # :881
    if ref_file == "rgb_v2":
        return fc7, fgs, fc14, conv2                                                    # RGB_OFF_v2.py:891
    return fc7, fgs, fc14                                                               # Flow_OFF.py:884


UNIT_PARAM_PREFIXES = ("motion_conv_gen_", "motion_spatial_down_", "motion_spatial_grad_")


def unit_backward(feats, w, batch, length, variant, slice_mode, cotangents, unit_drop=None, unit_gen_mask=None):
    """Gradients of loss = sum_k <out_k, cotangent_k> (k = fc7, fc14, fc28, per-pair logits) w.r.t.
    every OFF-unit parameter and w.r.t. the nine unit outputs, by autograd through this file's own
    forward -- the same ATen backward kernels the reference's train loop runs
    (train_off.py:126-146; only ``*motion*`` parameters train, :39-45, so the feature maps get no
    gradient).  Returns (param_grads: key -> tensor, dM: list of nine [P,160,H,H]).
    """
    wl = {}
    for k, v in w.items():
        v = v.detach().clone()
        if k.startswith(UNIT_PARAM_PREFIXES):
            v.requires_grad_(True)
        wl[k] = v
    (fc7, fc14, fc28), st = off_forward(feats, wl, batch, length, variant, slice_mode,
                                        consensus=False, return_stages=True, unit_drop=unit_drop,
                                        unit_gen_mask=unit_gen_mask)
    ms = [st["motion_" + site] for site, _c, _h in SITES]
    for m in ms:
        m.retain_grad()
    loss = (fc7 * cotangents[0]).sum() + (fc14 * cotangents[1]).sum() + (fc28 * cotangents[2]).sum()
    loss.backward()
    grads = dict((k, v.grad) for k, v in wl.items() if v.requires_grad)
    return grads, [m.grad for m in ms]


def unit_param_grads_from_dm(feats, w, batch, length, variant, slice_mode, dm, unit_drop=None, unit_gen_mask=None):
    """Unit-parameter gradients for GIVEN output gradients dM (list of nine [P,160,H,H]): the
    restatement of what offk_off_units_backward computes."""
    grads = {}
    for si, ((site, _c, _h), x) in enumerate(zip(SITES, feats)):
        wl = {}
        for k, v in w.items():
            if k.startswith(UNIT_PARAM_PREFIXES) and k.split(".")[0].endswith("_" + site):
                wl[k] = v.detach().clone().requires_grad_(True)
        if variant != VARIANT_RGB:
            wl["sobel_edge_diagonal.conv.weight"] = w["sobel_edge_diagonal.conv.weight"]
        m = off_unit(x, wl, site, batch, length, variant, slice_mode, None if unit_drop is None else unit_drop[si],
                     None if unit_gen_mask is None else unit_gen_mask[si])
        m.backward(dm[si])
        grads.update((k, v.grad) for k, v in wl.items() if v.requires_grad)
    return grads


def segment_consensus_backward(grad_out, length_m1):
    """basic_ops.py:29-33: grad_in = grad_output.expand(shape) / shape[dim]; grad_out [B,C] -> [B*(L-1),C]."""
    b, c = grad_out.shape
    return (grad_out.reshape(b, 1, c).expand(b, length_m1, c) / float(length_m1)).reshape(b * length_m1, c)


def to_torch_weights(weights):
    return dict((k, torch.from_numpy(v) if not torch.is_tensor(v) else v) for k, v in weights.items())
