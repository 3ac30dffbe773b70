"""Host-side mirror of the reference's model interface for the OFF path.

The reference has no module boundary around the OFF sub-network (it is inline in
BNInception_OFF.RGB_OFF_forward, RGB_OFF.py:360-860, and BNInception_OFF.forward,
Flow_OFF.py:370-887).  This file provides

* ``OFFSubNetwork`` -- an nn.Module that owns the OFF parameters under the reference's
  exact state_dict key names (so reference-format checkpoints load) and whose forward
  takes the nine tapped feature maps and runs liboffk;
* ``BNInception_OFF`` / ``bninception_off`` -- same constructor arguments, attributes
  (``batch``, ``length``, ``modality_fuse``, ``consensus_type``) and return conventions
  as the reference classes (RGB_OFF.py:32,860,1346-1360; Flow_OFF.py:45,879-884,
  1371-1385; RGB_OFF_v2.py:891), with the TSN backbone supplied by the caller as an
  ordinary PyTorch module (it is out of scope and stays on stock PyTorch-ROCm).

The nn.Conv2d / nn.Linear children are parameter containers only: they are never
called.  All arithmetic happens in the HIP library; without it every forward raises.
"""
import os

import torch
import torch.nn as nn

from . import runtime, spec

_VARIANTS = {"rgb": spec.VARIANT_RGB, "flow": spec.VARIANT_FLOW, "rgb_v2": spec.VARIANT_FLOW}


class _SobelHolder(nn.Module):
    """state_dict key 'sobel_edge_diagonal.conv.weight' (util.py:52-77): frozen diagonal kernel."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(spec.DOWN_CH, spec.DOWN_CH, 3, 1, 1, bias=False, groups=spec.DOWN_CH)
        k = torch.tensor(spec.DIAG_SOBEL, dtype=torch.float32)
        self.conv.weight = nn.Parameter(k.expand(spec.DOWN_CH, 1, 3, 3).clone(), requires_grad=False)


class OFFSubNetwork(nn.Module):
    def __init__(self, num_classes=spec.NUM_CLASSES, batch=16, length=7, variant="rgb",
                 slice_mode=spec.SLICE_FLAT, consensus=None, feat_layout=0, precision="fp32"):
        super().__init__()
        self.precision = precision
        if variant not in _VARIANTS:
            raise ValueError("variant must be one of %s" % sorted(_VARIANTS))
        self.variant_name = variant
        self.variant = _VARIANTS[variant]
        self.batch, self.length, self.num_classes = batch, length, num_classes
        self.slice_mode, self.feat_layout = slice_mode, feat_layout
        self.consensus_avg = (self.variant == spec.VARIANT_FLOW) if consensus is None else bool(consensus)
        # parameter containers, names and shapes as in the reference (RGB_OFF.py:265-334, Flow_OFF.py:51)
        for name, C, _H in spec.SITES:
            setattr(self, "motion_conv_gen_" + name, nn.Conv2d(C, spec.GEN_CH, 1, 1))
            setattr(self, "motion_spatial_down_" + name, nn.Conv2d(C, spec.DOWN_CH, 1, 1))
            if self.variant == spec.VARIANT_RGB:
                setattr(self, "motion_spatial_grad_" + name,
                        nn.Conv2d(spec.DOWN_CH, spec.DOWN_CH, 3, 1, 1, groups=spec.DOWN_CH, bias=True))
        if self.variant == spec.VARIANT_FLOW:
            self.sobel_edge_diagonal = _SobelHolder()
        for key, co, ci, k, s, p in spec.FUSION_CONVS:
            setattr(self, key, nn.Conv2d(ci, co, k, s, p))
        for key, cin in spec.HEADS:
            setattr(self, key, nn.Linear(cin, num_classes))
        self._rt = None
        self._rt_key = None
        self._pushed = {}      # key -> (data_ptr, _version) of the tensor the library last copied

    # -- weights -> library ---------------------------------------------------------
    def mark_weights_dirty(self):
        """Force a re-push of every weight on the next forward.  Needed after writing a parameter through a path that
        bypasses the parameter's version counter: a raw-pointer write by foreign code, and IN-PLACE WRITES THROUGH
        ``.data`` (``p.data.copy_(w)``, ``p.data.mul_(0.9)`` -- common in TSN-era code; ``.data`` is a tensor with its own
        version counter, so ``p._version`` does not move and the storage pointer stays the same).  ``p.copy_`` / ``p.mul_``
        under ``torch.no_grad()``, optimizer steps, ``load_state_dict`` and ``p.data = new_tensor`` are all seen without this
        call.  OFFK_ALWAYS_PUSH=1 in the environment re-pushes everything on every forward (debugging stale weights; slow).
        The same blind spot exists for the parameter-written-between-forward-and-backward check of OFFUnits."""
        self._pushed = {}

    def load_state_dict(self, state_dict, strict=True, **kw):
        own = set(self.state_dict().keys())
        sd = {}
        for k, v in state_dict.items():
            kk = k[7:] if k.startswith("module.") else k     # DataParallel checkpoints (test_flow_off.py:52-58)
            if kk in own:
                sd[kk] = v
        missing = sorted(own - set(sd))
        if strict and missing:
            raise KeyError("missing OFF weights: %s" % ", ".join(missing[:5]))
        return super().load_state_dict(sd, strict=False, **kw)

    def _handle(self, device):
        key = (self.batch, self.length, self.variant, self.slice_mode, self.consensus_avg, self.feat_layout, str(device), self.precision)
        if self._rt is None or self._rt_key != key:
            self._rt = runtime.OffForward(self.batch, self.length, self.variant, self.slice_mode,
                                          self.consensus_avg, self.num_classes, self.feat_layout, device, self.precision)
            self._rt_key = key
            self._pushed = {}
        # liboffk keeps packed copies: push every tensor that is new or was written since the last push.  Any writer
        # that goes through torch -- load_state_dict of this module OR of a parent (which calls _load_from_state_dict
        # and never reaches the override above), optimizer steps, param.copy_(), param.data = ... -- either bumps the
        # tensor's version counter or replaces its storage, and both are part of the tag.
        if os.environ.get("OFFK_ALWAYS_PUSH") == "1":
            self._pushed = {}
        for k, v in self.state_dict(keep_vars=True).items():
            tag = (v.data_ptr(), v._version)
            if self._pushed.get(k) != tag:
                self._rt.set_weight(k, v)
                self._pushed[k] = tag
        return self._rt

    # -- forward ----------------------------------------------------------------------
    def forward(self, feats, want28=True):
        """feats: the nine ``inception_{3a..5b}_output_out`` maps [B*L, C, H, H] fp32 on a HIP
        device -- each either the concatenated tensor or the list of its inception branches in
        torch.cat order (then the concat never has to exist, offk_forward_parts).  Returns (fc_action_motion_7, fc_action_motion_14, fc_action_motion_28):
        [B*(L-1), classes] each, or [B, classes] with the consensus average."""
        feats = [f.contiguous() if torch.is_tensor(f) else [g.contiguous() for g in f] for f in feats]
        first = feats[0] if torch.is_tensor(feats[0]) else feats[0][0]
        if not first.is_cuda:
            raise runtime._lib.OffkError("OFFSubNetwork has no CPU path: feature maps must live on an MI355X")
        rt = self._handle(first.device)
        with torch.no_grad():
            return rt.forward(feats, want28=want28)


class _OFFUnitsFn(torch.autograd.Function):
    """autograd node around offk_off_units(_train) / offk_off_units_backward.  Inputs after the three
    bookkeeping arguments are the unit parameters in ``OFFUnits.param_keys`` order; the feature maps come from
    the frozen backbone (train_off.py:39-56) and get no gradient."""

    @staticmethod
    def forward(ctx, mod, feats, drop, *params):
        rt = mod._handle(feats[0].device, params)
        mod._run_units(rt, feats, drop)
        ctx.mod, ctx.feats, ctx.drop = mod, feats, drop
        # the backward reads the G / D activations this call leaves in the handle's ONE workspace: remember which
        # call that was (and which parameter values it used) so that a later forward through the same module
        # -- an eval pass, a second micro-batch, a checkpoint recompute -- is detected instead of silently used
        ctx.gen = mod._generation
        ctx.versions = dict((k, (prm.data_ptr(), prm._version)) for k, prm in zip(mod.param_keys, params))
        P = rt.P
        # copies: the workspace is rewritten by the next forward, autograd consumers may outlive it
        outs = []
        for name, H, C, width in (("fusion_28", 28, 320, 320), ("fusion_14", 14, 1056, 800), ("fusion_7", 7, 832, 320)):
            buf = rt.region(name, C).view(P, H, H, C)[..., :width].contiguous()
            outs.append(buf.permute(0, 3, 1, 2))       # NCHW view of channels-last memory
        return tuple(outs)

    @staticmethod
    def backward(ctx, g28, g14, g7):
        mod, feats, (seed, p) = ctx.mod, ctx.feats, ctx.drop
        rt = mod._rt
        if rt is None or any((mod._param(k).data_ptr(), mod._param(k)._version) != v for k, v in ctx.versions.items()):
            raise RuntimeError("OFFUnits: a parameter was modified (or the module moved) between this forward and its "
                               "backward; the saved activations no longer belong to the current weights")
        if mod._generation != ctx.gen:
            # another forward has overwritten G / D since: recompute them (K1 + K2 only, same seed => same mask)
            mod._run_units(rt, feats, ctx.drop)
        bufs = [g.permute(0, 2, 3, 1).contiguous() for g in (g28, g14, g7)]      # channels-last rows
        views = [(bufs[0], 0), (bufs[0], 160)] + [(bufs[1], 160 * k) for k in range(5)] + [(bufs[2], 0), (bufs[2], 160)]
        _flat, grads = rt.off_units_backward(feats, views, seed, p)
        return (None, None, None) + tuple(grads[k] for k in mod.param_keys)


class OFFUnits(nn.Module):
    """The nine OFF units as a trainable module on liboffk: parameters under the
    reference's state_dict keys (``motion_conv_gen_*``, ``motion_spatial_down_*``, ``motion_spatial_grad_*``;
    the Sobel weight of the Flow variant is frozen, util.py:72), forward = K1 + K2 (in ``train()`` mode with
    nn.Dropout(p=0.8) on the spatial gradients, RGB_OFF.py:356/:612, reproducible from ``drop_seed``), backward
    = offk_off_units_backward.  Returns the three motion maps the fusion stages start from
    (cat(motion_3a, motion_3b) [P,320,28,28], cat(motion_3c..4d) [P,800,14,14], cat(motion_5a, motion_5b)
    [P,320,7,7]; RGB_OFF.py:656, :760, :832), channels-last in memory.  The reference's fusion convolutions
    and heads stay ordinary PyTorch modules in training."""

    def __init__(self, batch=16, length=7, variant="rgb", slice_mode=spec.SLICE_FLAT, precision="fp32", drop_p=0.8):
        super().__init__()
        if variant not in _VARIANTS:
            raise ValueError("variant must be one of %s" % sorted(_VARIANTS))
        self.variant = _VARIANTS[variant]
        self.batch, self.length, self.slice_mode, self.precision, self.drop_p = batch, length, slice_mode, precision, float(drop_p)
        for name, C, _H in spec.SITES:
            setattr(self, "motion_conv_gen_" + name, nn.Conv2d(C, spec.GEN_CH, 1, 1))
            setattr(self, "motion_spatial_down_" + name, nn.Conv2d(C, spec.DOWN_CH, 1, 1))
            if self.variant == spec.VARIANT_RGB:
                setattr(self, "motion_spatial_grad_" + name,
                        nn.Conv2d(spec.DOWN_CH, spec.DOWN_CH, 3, 1, 1, groups=spec.DOWN_CH, bias=True))
        if self.variant == spec.VARIANT_FLOW:
            self.sobel_edge_diagonal = _SobelHolder()
        self.param_keys = [k for k in spec.weight_shapes(self.variant) if k.startswith(spec.UNIT_PARAM_PREFIXES)]
        self._rt, self._bound, self.drop_seed = None, {}, 0
        self._generation = 0    # bumped by every units forward: tells a backward whether G / D in the workspace are its own

    def _run_units(self, rt, feats, drop):
        seed, p = drop
        if p > 0.0:
            rt.off_units_train(feats, seed, p)
        else:
            rt.off_units(feats)
        self._generation += 1

    def _param(self, key):
        mod, attr = key.rsplit(".", 1)
        return getattr(getattr(self, mod), attr)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Takes a reference-format checkpoint: the unit keys are used, every other key (backbone, fusion stages,
        heads) is ignored; a DataParallel 'module.' prefix is accepted (test_flow_off.py:52-58)."""
        own = set(self.state_dict().keys())
        sd = {}
        for k, v in state_dict.items():
            kk = k[7:] if k.startswith("module.") else k
            if kk in own:
                sd[kk] = v
        missing = sorted(own - set(sd))
        if strict and missing:
            raise KeyError("missing OFF unit weights: %s" % ", ".join(missing[:5]))
        return super().load_state_dict(sd, strict=False, **kw)

    def _handle(self, device, params):
        if self._rt is None or self._rt.device != torch.device(device):
            self._rt = runtime.OffForward(self.batch, self.length, self.variant, self.slice_mode, False,
                                          device=device, precision=self.precision, training=True)
            self._bound = {}
            if self.variant == spec.VARIANT_FLOW:
                self._rt.set_weight(spec.SOBEL_KEY, self.sobel_edge_diagonal.conv.weight)
        # The trainable tensors are BOUND, not copied (offk_bind_weight): liboffk reads the parameters' own storage at
        # launch time, so an optimizer step costs the library nothing -- no 54 blocking copies per step, no staging, no
        # re-packing -- and is ordered like any other kernel on the stream.  Only a re-allocated parameter is re-bound.
        for key, prm in zip(self.param_keys, params):
            if not prm.is_contiguous():
                raise ValueError("OFFUnits parameter %s must be contiguous" % key)
            if self._bound.get(key) != prm.data_ptr():
                self._rt.bind_weight(key, prm)
                self._bound[key] = prm.data_ptr()
        return self._rt

    def forward(self, feats, drop_seed=None):
        feats = tuple(f.contiguous() for f in feats)
        if not feats[0].is_cuda:
            raise runtime._lib.OffkError("OFFUnits has no CPU path: feature maps must live on an MI355X")
        if self.training and self.drop_p > 0.0:
            if drop_seed is None:
                self.drop_seed += 1
                drop_seed = self.drop_seed
            drop = (int(drop_seed), self.drop_p)
        else:
            drop = (0, 0.0)
        return _OFFUnitsFn.apply(self, feats, drop, *[self._param(k) for k in self.param_keys])


class BNInception_OFF(nn.Module):
    """Drop-in for the reference class of the same name, OFF part on liboffk.

    ``backbone`` is any module mapping the frame batch [B*L, 3|10, 224, 224] to
    ``(feats, Feature_Generation_Score)`` or ``(feats, Feature_Generation_Score, conv2_relu_3x3_out)``
    -- i.e. the reference's own layers up to RGB_OFF.py:594.  Without a backbone the
    ``input`` of forward must already be the list of nine feature maps (then
    Feature_Generation_Score is returned as None).
    """

    def __init__(self, num_classes=1000, batch=16, length=7, variant="rgb", backbone=None,
                 slice_mode=spec.SLICE_FLAT, precision="fp32"):
        super().__init__()
        self.batch, self.length = batch, length
        self.modality_fuse = False                       # Flow_OFF.py:45
        self.consensus_type = "avg"                      # RGB_OFF.py:39
        self.variant_name = variant
        self.backbone = backbone
        self.off = OFFSubNetwork(num_classes, batch, length, variant, slice_mode, precision=precision)

    def state_dict(self, *a, **kw):
        sd = super().state_dict(*a, **kw)
        # flatten 'off.' so the keys are the reference's (motion_*, fc_action_motion*, sobel_edge_diagonal.*)
        return type(sd)((k[4:] if k.startswith("off.") else k, v) for k, v in sd.items())

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Reference-format checkpoints (model_utils.py:188-216, Flow_OFF.py:1398-1413): OFF keys and backbone keys side
        by side at top level, optionally under a DataParallel 'module.' prefix (test_flow_off.py:52-58).  OFF keys go to
        the OFF sub-network, everything else to the backbone (plain or 'backbone.'-prefixed names).  Returns the
        (missing_keys, unexpected_keys) named tuple of nn.Module.load_state_dict; with strict=True missing OFF keys
        raise KeyError and keys nobody takes raise RuntimeError."""
        from torch.nn.modules.module import _IncompatibleKeys
        plain = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        own = set(self.off.state_dict().keys())
        self.off.load_state_dict({k: v for k, v in plain.items() if k in own}, strict=strict)
        missing = sorted(own - set(plain))
        rest = {(k[9:] if k.startswith("backbone.") else k): v for k, v in plain.items() if k not in own}
        unexpected = []
        if self.backbone is not None:
            r = self.backbone.load_state_dict(rest, strict=False)
            missing += ["backbone." + k for k in r.missing_keys]
            unexpected = list(r.unexpected_keys)
        else:
            unexpected = sorted(rest)      # no backbone attached: nobody takes the TSN keys
        if strict and self.backbone is not None and unexpected:
            raise RuntimeError("unexpected keys in state_dict: %s" % ", ".join(unexpected[:5]))
        return _IncompatibleKeys(missing, unexpected)

    def _features(self, input):
        if self.backbone is None:
            return list(input), None, None
        r = self.backbone(input)
        return list(r[0]), r[1], (r[2] if len(r) > 2 else None)

    @staticmethod
    def _squeeze(x):
        # torch.squeeze at RGB_OFF.py:786,792,846 also drops the pair axis when P == 1
        return x[0] if x.shape[0] == 1 else x

    def RGB_OFF_forward(self, input):
        """RGB_OFF.py:360-860: returns (fc_action_motion_7, Feature_Generation_Score, fc_action_motion_14)."""
        feats, fgs, _ = self._features(input)
        fc7, fc14, _fc28 = self.off(feats, want28=False)
        return self._squeeze(fc7), fgs, self._squeeze(fc14)

    def forward(self, input):
        """Flow_OFF.py:370-887 / RGB_OFF_v2.py:377-894 for the flow / rgb_v2 variants
        (consensus average inside, optional fused score); for the rgb variant this is
        RGB_OFF_forward (the reference's plain ``forward`` there is the backbone only)."""
        if self.off.variant == spec.VARIANT_RGB:
            return self.RGB_OFF_forward(input)
        feats, fgs, conv2 = self._features(input)
        fc7, fc14, _fc28 = self.off(feats, want28=False)
        # K6 / K7 are raw library calls on data_ptr: they do not record an autograd graph.  The reference uses this path at test time
        # only (test_flow_off.py:343 under volatile=True); when a caller DOES want gradients through the backbone's score (grad mode
        # on and the score requires grad) the same arithmetic runs as torch ops so the graph stays intact.
        wants_grad = torch.is_grad_enabled() and fgs is not None and fgs.requires_grad
        if fgs is not None and fgs.shape[0] == self.batch * self.length:
            if wants_grad:
                fgs = fgs.float().reshape(self.batch, self.length, -1).mean(dim=1)          # basic_ops.py:19-21
            else:
                fgs = runtime.segment_consensus(fgs.contiguous().float(), self.batch)   # Flow_OFF.py:867,873 (K6)
        if self.modality_fuse:
            if fgs is None:
                raise ValueError("modality_fuse adds the backbone's Feature_Generation_Score (Flow_OFF.py:881): "
                                 "attach a backbone that returns it")
            if wants_grad or fc7.requires_grad or fc14.requires_grad:
                return fc7 + fgs.to(fc7.dtype) + fc14                                    # Flow_OFF.py:881, differentiable
            # Flow_OFF.py:881 `fc7 + fgs + fc14`: K7 with unit weights and one "crop" (offk_score_fusion), not a torch op
            fused, _ = runtime.score_fusion([fc7, fgs.to(fc7.dtype).contiguous(), fc14], (1.0, 1.0, 1.0), want_pred=False)
            return fused
        if self.variant_name == "rgb_v2":
            return fc7, fgs, fc14, conv2                                       # RGB_OFF_v2.py:891
        return fc7, fgs, fc14                                                  # Flow_OFF.py:884


def bninception_off(num_classes=101, batch=16, num_seg=7, variant="rgb", backbone=None, **kw):
    """RGB_OFF.py:1346-1360 / Flow_OFF.py:1371-1385 factory."""
    return BNInception_OFF(num_classes=num_classes, batch=batch, length=num_seg, variant=variant,
                           backbone=backbone, **kw)
