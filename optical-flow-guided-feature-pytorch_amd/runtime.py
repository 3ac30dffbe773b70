"""Thin PyTorch-ROCm host wrapper around the liboffk handle.

PyTorch is plumbing here (device memory, the current HIP stream); all arithmetic runs
in liboffk's HIP kernels.  Nothing in this file computes on the CPU and nothing falls
back to torch ops.
"""
import ctypes

import numpy as np
import torch

from . import _lib, spec


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device=None):
    """HIP stream the call is enqueued on: torch's current stream OF THE DEVICE THE DATA LIVES ON (a handle on cuda:1
    must not launch on cuda:0's stream just because cuda:0 is torch's current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check_dev(t, name, device):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous fp32 CUDA/HIP tensor" % name)
    if t.device != device:
        raise ValueError("%s lives on %s, handle on %s" % (name, t.device, device))


class OffForward:
    """One liboffk handle for a fixed (batch, length, variant).

    Stands for the OFF part of BNInception_OFF (reference RGB_OFF.py:265-358 declarations,
    :596-860 forward; Flow_OFF.py:606-887).
    """

    def __init__(self, batch, length, variant=spec.VARIANT_RGB, slice_mode=spec.SLICE_FLAT,
                 consensus=None, num_classes=spec.NUM_CLASSES, feat_layout=0, device=None, precision=0,
                 training=False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.OffkError("no HIP device visible: the OFF forward has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if consensus is None:  # reference default: only the Flow / v2 files apply it
            consensus = variant == spec.VARIANT_FLOW
        self.batch, self.length, self.variant = int(batch), int(length), int(variant)
        self.slice_mode, self.consensus, self.num_classes = int(slice_mode), bool(consensus), int(num_classes)
        self.feat_layout = int(feat_layout)
        self.precision = _lib.PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        self.N = self.batch * self.length
        self.P = self.batch * (self.length - 1)
        cfg = _lib.OffkConfig(self.batch, self.length, self.variant, self.slice_mode, int(self.consensus),
                              self.num_classes, self.feat_layout, self.device.index or 0, self.precision)
        h = ctypes.c_void_p()
        _lib.check(self.lib.offk_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self._ws = None
        self.training = bool(training)   # allocate the workspace superset the units' backward needs

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.offk_destroy(h)
            self._h = None

    # ---- weights ----------------------------------------------------------------
    def set_weight(self, key, value):
        if torch.is_tensor(value):
            t = value.detach().to(dtype=torch.float32).contiguous()
            shape = tuple(t.shape)
            ptr, keep = ctypes.c_void_p(t.data_ptr()), t
        else:
            a = np.ascontiguousarray(value, dtype=np.float32)
            shape = a.shape
            ptr, keep = ctypes.c_void_p(a.ctypes.data), a
        arr = (ctypes.c_int64 * len(shape))(*shape)
        _lib.check(self.lib.offk_set_weight(self._h, key.encode(), ptr, arr, len(shape)), self._h)
        del keep

    def bind_weight(self, key, tensor):
        """offk_bind_weight: the library reads ``tensor`` (a contiguous fp32 parameter on this device, reference layout)
        in place from now on -- no copy now, none after an optimizer step.  The caller keeps it alive."""
        _check_dev(tensor, key, self.device)
        kk = key[7:] if key.startswith("module.") else key
        want = spec.weight_shapes(self.variant).get(kk)
        if want is not None and tuple(tensor.shape) != tuple(want):
            raise ValueError("%s has shape %s, the reference's is %s" % (key, tuple(tensor.shape), tuple(want)))
        shape = (ctypes.c_int64 * tensor.dim())(*tensor.shape)
        _lib.check(self.lib.offk_bind_weight(self._h, key.encode(), ctypes.c_void_p(tensor.data_ptr()), shape, tensor.dim()), self._h)

    def load_state_dict(self, state_dict, strict=True):
        """Accepts a reference-format state_dict (extra backbone keys are ignored;
        a 'module.' prefix is accepted, test_flow_off.py:52-58)."""
        want = spec.weight_shapes(self.variant)
        seen = set()
        for k, v in state_dict.items():
            kk = k[7:] if k.startswith("module.") else k
            if kk in want:
                self.set_weight(kk, v)
                seen.add(kk)
        missing = [k for k in want if k not in seen]
        if strict and missing:
            raise KeyError("missing OFF weights: %s" % ", ".join(missing[:5]))
        return missing

    def missing_weights(self):
        buf = ctypes.create_string_buffer(256)
        n = self.lib.offk_missing_weights(self._h, buf, 256)
        return n, buf.value.decode()

    # ---- workspace ----------------------------------------------------------------
    @property
    def workspace_bytes(self):
        if self.training:
            return int(self.lib.offk_train_workspace_bytes(self._h))
        return int(self.lib.offk_workspace_bytes(self._h))

    @property
    def workspace(self):
        if self._ws is None:
            self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def region(self, name, channels):
        """View of a named workspace region as [rows, channels] fp32 (channels-last)."""
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self.lib.offk_workspace_region(self._h, name.encode(), ctypes.byref(off), ctypes.byref(nb)), self._h)
        flat = self.workspace[off.value:off.value + nb.value].view(torch.float32)
        return flat.view(-1, channels)

    # ---- forward --------------------------------------------------------------------
    def out_rows(self):
        return self.batch if self.consensus else self.P

    def _feat_array(self, feats):
        if len(feats) != spec.NUM_SITES:
            raise ValueError("need nine feature maps")
        shapes = spec.feature_shapes(self.batch, self.length)
        for i, (f, s) in enumerate(zip(feats, shapes)):
            _check_dev(f, "feats[%d]" % i, self.device)
            want = s if self.feat_layout == 0 else (s[0], s[2], s[3], s[1])
            if tuple(f.shape) != tuple(want):
                raise ValueError("feats[%d] has shape %s, expected %s" % (i, tuple(f.shape), tuple(want)))
        return (ctypes.c_void_p * spec.NUM_SITES)(*[f.data_ptr() for f in feats])

    def _parts_array(self, feats):
        """feats[i] is a tensor or a sequence of 1..4 tensors (channel groups in concat order)."""
        shapes = spec.feature_shapes(self.batch, self.length)
        arr = (_lib.OffkFeatParts * spec.NUM_SITES)()
        for i, (f, s) in enumerate(zip(feats, shapes)):
            parts = [f] if torch.is_tensor(f) else list(f)
            if not 1 <= len(parts) <= 4:
                raise ValueError("feats[%d]: 1..4 channel groups" % i)
            arr[i].n_parts = len(parts)
            for q, t in enumerate(parts):
                _check_dev(t, "feats[%d][%d]" % (i, q), self.device)
                c = t.shape[1] if self.feat_layout == 0 else t.shape[3]
                want = (s[0], c, s[2], s[3]) if self.feat_layout == 0 else (s[0], s[2], s[3], c)
                if tuple(t.shape) != want:
                    raise ValueError("feats[%d][%d] has shape %s, expected %s" % (i, q, tuple(t.shape), want))
                arr[i].channels[q] = c
                arr[i].data[q] = t.data_ptr()
        return arr

    def forward(self, feats, want28=True):
        if any(not torch.is_tensor(f) for f in feats):
            arr = self._parts_array(feats)
            rows = self.out_rows()
            out7 = torch.empty(rows, self.num_classes, dtype=torch.float32, device=self.device)
            out14 = torch.empty_like(out7)
            out28 = torch.empty_like(out7) if want28 else None
            _lib.check(self.lib.offk_forward_parts(self._h, _stream(self.device), arr, _ptr(out7), _ptr(out14), _ptr(out28),
                                                   _ptr(self.workspace)), self._h)
            return out7, out14, out28
        arr = self._feat_array(feats)
        rows = self.out_rows()
        out7 = torch.empty(rows, self.num_classes, dtype=torch.float32, device=self.device)
        out14 = torch.empty_like(out7)
        out28 = torch.empty_like(out7) if want28 else None
        _lib.check(self.lib.offk_forward(self._h, _stream(self.device), arr, _ptr(out7), _ptr(out14), _ptr(out28),
                                         _ptr(self.workspace)), self._h)
        return out7, out14, out28

    def forward_into(self, feat_array, out7, out14, out28):
        """Launch-only variant for benchmarking: pre-validated ctypes array + outputs."""
        _lib.check(self.lib.offk_forward(self._h, _stream(self.device), feat_array, _ptr(out7), _ptr(out14), _ptr(out28),
                                         _ptr(self.workspace)), self._h)

    def off_units(self, feats):
        arr = self._feat_array(feats)
        _lib.check(self.lib.offk_off_units(self._h, _stream(self.device), arr, _ptr(self.workspace)), self._h)

    def off_units_fused(self, feats):
        """The units as forward() runs them (fused K1T + S-blocks, the handle's arithmetic); results in the fusion_* / D_* regions."""
        arr = self._feat_array(feats)
        _lib.check(self.lib.offk_off_units_fused(self._h, _stream(self.device), arr, _ptr(self.workspace)), self._h)

    # ---- training side of the units ----------------------------
    def off_units_train(self, feats, drop_seed=0, drop_p=0.8):
        """K1+K2 in training mode: nn.Dropout(p) (RGB_OFF.py:356, :612) on the spatial gradients with the
        reproducible mask of synth.dropout_keep; leaves G/D in the workspace for off_units_backward."""
        arr = self._feat_array(feats)
        _lib.check(self.lib.offk_off_units_train(self._h, _stream(self.device), arr, _ptr(self.workspace),
                                                 ctypes.c_uint64(int(drop_seed)), float(drop_p)), self._h)

    def unit_grad_slots(self):
        """OrderedDict key -> (offset, shape) of every unit parameter in the flat gradient buffer."""
        from collections import OrderedDict
        out = OrderedDict()
        for key, shape in spec.weight_shapes(self.variant).items():
            if not key.startswith(spec.UNIT_PARAM_PREFIXES):
                continue
            off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self.lib.offk_unit_grad_slot(self._h, key.encode(), ctypes.byref(off), ctypes.byref(cnt)), self._h)
            assert cnt.value == int(np.prod(shape))
            out[key] = (off.value, shape)
        return out

    def new_unit_grads(self):
        return torch.zeros(int(self.lib.offk_unit_grad_floats(self._h)), dtype=torch.float32, device=self.device)

    def off_units_backward(self, feats, grad_views, drop_seed=0, drop_p=0.0, grads=None, accumulate=False):
        """Gradients of the units' parameters.  grad_views: nine (tensor, coff) pairs -- a channels-last
        gradient buffer [P, H, W, Cs] (or [P*H*W, Cs]) and the first of the unit's 160 channels in it.
        Needs training=True (workspace superset) and the G/D state of the matching forward call.
        Returns (flat grads tensor, dict key -> view in the reference's parameter shape)."""
        if not self.training:
            raise _lib.OffkError("create the handle with training=True for the units' backward")
        arr = self._feat_array(feats)
        gv = (_lib.OffkGradView * spec.NUM_SITES)()
        for i, ((t, coff), (_n, _c, H)) in enumerate(zip(grad_views, spec.SITES)):
            _check_dev(t, "grad_views[%d]" % i, self.device)
            if t.numel() != self.P * H * H * t.shape[-1]:
                raise ValueError("grad_views[%d] has %d elements, expected P*H*W*%d" % (i, t.numel(), t.shape[-1]))
            gv[i].data, gv[i].cstride, gv[i].coff = t.data_ptr(), t.shape[-1], int(coff)
        if grads is None:
            grads = self.new_unit_grads()
        _check_dev(grads, "grads", self.device)
        _lib.check(self.lib.offk_off_units_backward(self._h, _stream(self.device), arr, gv, _ptr(self.workspace),
                                                    ctypes.c_uint64(int(drop_seed)), float(drop_p), _ptr(grads),
                                                    int(bool(accumulate))), self._h)
        views = dict((k, grads[off:off + int(np.prod(shape))].view(shape)) for k, (off, shape) in self.unit_grad_slots().items())
        return grads, views

    # ---- stage entry points -----------------------------------------------------------
    def pw_reduce(self, site, feat):
        _name, _C, H = spec.SITES[site]
        G = torch.empty(self.N * H * H, spec.GEN_CH, dtype=torch.float32, device=self.device)
        D = torch.zeros(self.P * H * H, spec.DOWN_CH, dtype=torch.float32, device=self.device)
        _check_dev(feat, "feat", self.device)
        _lib.check(self.lib.offk_pw_reduce(self._h, _stream(self.device), site, _ptr(feat), _ptr(G), _ptr(D)), self._h)
        return G, D

    def sobel_tdiff(self, site, G, D, M, m_coff, algo=0):
        _check_dev(G, "G", self.device)
        _check_dev(D, "D", self.device)
        _check_dev(M, "M", self.device)
        _lib.check(self.lib.offk_sobel_tdiff(self._h, _stream(self.device), site, _ptr(G), _ptr(D), _ptr(M),
                                             M.shape[-1], m_coff, algo), self._h)

    def sobel_tdiff_all(self, algo=0):
        """Grouped K2 launch over the workspace G/D regions (after off_units / forward)."""
        _lib.check(self.lib.offk_sobel_tdiff_all(self._h, _stream(self.device), _ptr(self.workspace), algo), self._h)

    def set_conv_plan(self, conv_key, tile_cfg, splitk):
        _lib.check(self.lib.offk_set_conv_plan(self._h, conv_key.encode(), tile_cfg, splitk), self._h)

    # ---- profiling ---------------------------------------------------------------------
    def set_profiling(self, on):
        """False / 0 off, True / 1 per-stage events (stage_times), 2 per-launch trace (launch_times)."""
        _lib.check(self.lib.offk_set_profiling(self._h, int(on)), self._h)

    def launch_times(self, reset=True, max_entries=128):
        """Per-launch trace: OrderedDict name -> (accumulated ms, calls), in first-launch order."""
        from collections import OrderedDict
        ms = (ctypes.c_double * max_entries)()
        calls = (ctypes.c_int64 * max_entries)()
        buf = ctypes.create_string_buffer(64 * max_entries)
        n = self.lib.offk_launch_times(self._h, buf, len(buf), ms, calls, max_entries, int(reset))
        if n < 0:
            _lib.check(n, self._h)
        names = buf.value.decode().split("\n")
        return OrderedDict((names[i], (ms[i], int(calls[i]))) for i in range(min(n, max_entries)))

    def stage_times(self, reset=True):
        ms = (ctypes.c_double * _lib.NUM_STAGES)()
        calls = (ctypes.c_int64 * _lib.NUM_STAGES)()
        _lib.check(self.lib.offk_stage_times(self._h, ms, calls, int(reset)), self._h)
        return dict((n, (ms[i], int(calls[i]))) for i, n in enumerate(_lib.STAGE_NAMES))


def segment_consensus_backward(grad_out, length_m1):
    """basic_ops.py:29-33: grad_out [B, C] -> grad_in [B*(L-1), C] = grad_out / (L-1), repeated."""
    lib = _lib.load()
    B, C = grad_out.shape
    gi = torch.empty(B * length_m1, C, dtype=torch.float32, device=grad_out.device)
    _lib.check(lib.offk_segment_consensus_backward(_stream(grad_out.device), _ptr(grad_out.contiguous()), B, int(length_m1), C, _ptr(gi)))
    return gi


# ---- handle-less stage kernels (channels-last tensors) ----------------------------------
def conv2d_nhwc(x, w_oihw, bias, stride, pad, res=None, flags=0, x_coff=0, ci=None, y=None, y_coff=0,
                tile_cfg=-1, splitk=0, w_packed=None, precision=0):
    """x: [n, H, W, Cs] fp32 CUDA; uses channels [x_coff, x_coff+Ci).  Returns y [n, Ho, Wo, Co]
    (or writes channels [y_coff, y_coff+Co) of the given y)."""
    lib = _lib.load()
    n, H, W, cs = x.shape
    Co, Ci, KH, KW = w_oihw.shape
    wp = w_packed
    if wp is None:
        wp = torch.empty(Co, KH, KW, Ci, dtype=torch.float32, device=x.device)   # library K order [Co][Ci/32][KH*KW][32]
        _lib.check(lib.offk_pack_conv_weight(_stream(x.device), _ptr(w_oihw.contiguous()), Co, Ci, KH, KW, _ptr(wp)))
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    if y is None:
        y = torch.empty(n, Ho, Wo, Co, dtype=torch.float32, device=x.device)
    part, nfl = None, 0
    if splitk > 1:
        nfl = splitk * n * Ho * Wo * Co
        part = torch.empty(nfl, dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_conv2d_ex(_stream(x.device), _ptr(x), cs, x_coff, n, H, W, Ci, _ptr(wp), _ptr(bias), Co, KH, KW,
                                  stride, pad, _ptr(res), res.shape[-1] if res is not None else 0, 0, flags,
                                  _ptr(y), y.shape[-1], y_coff, tile_cfg, splitk, _ptr(part), nfl, precision))
    return y


def pack_conv_weight(w_oihw):
    """[Co][Ci][KH][KW] -> the library's K order [Co][Ci/32][KH*KW][32] (device tensor)."""
    lib = _lib.load()
    Co, Ci, KH, KW = w_oihw.shape
    wp = torch.empty(Co, KH, KW, Ci, dtype=torch.float32, device=w_oihw.device)
    _lib.check(lib.offk_pack_conv_weight(_stream(w_oihw.device), _ptr(w_oihw.contiguous()), Co, Ci, KH, KW, _ptr(wp)))
    return wp


def bottleneck_chain14(x, w1, b1, w2_oihw, b2, w3, b3, res=None, relu_in=False, x_coff=0, y=None, y_coff=0):
    """One 1x1 -> 3x3 -> 1x1 (+ residual) chain of fusion@28 in one launch (offk_bottleneck_chain14).  x: [n, 14, 14, Cs] fp32 CUDA,
    the chain reads channels [x_coff, x_coff + Cin); w1 [64, Cin], w2_oihw [64, 64, 3, 3], w3 [256, K3] (K3 = 128: contracts
    [t2 | x]); res: [n, 14, 14, 256] or None.  Returns y [n, 14, 14, 256] (or writes channels [y_coff, y_coff + 256) of y)."""
    lib = _lib.load()
    n, H, W, cs = x.shape
    assert H == 14 and W == 14
    Cin, K3 = w1.shape[1], w3.shape[1]
    if y is None:
        y = torch.empty(n, 14, 14, 256, dtype=torch.float32, device=x.device)
    w2p = pack_conv_weight(w2_oihw)
    _lib.check(lib.offk_bottleneck_chain14(_stream(x.device), _ptr(x), cs, x_coff, n, Cin, int(relu_in), _ptr(w1.contiguous()),
                                           _ptr(b1), _ptr(w2p), _ptr(b2), _ptr(w3.contiguous()), _ptr(b3), K3, _ptr(res),
                                           res.shape[-1] if res is not None else 0, 0, _ptr(y), y.shape[-1], y_coff))
    return y


def bottleneck_chain14_split(x, w1, b1, w2_oihw, b2, w3, b3, res=None, branch=None, relu_in=False, x_coff=0, y=None, y_coff=0):
    """offk_bottleneck_chain14_split: the chain in split-fp32 arithmetic (chain_split.hip).  w3 [256, 64]; branch = (w [256, 64], b [256]):
    chain 28a's branch 1x1 on the chain input before relu_in's ReLU (then Cin = 64 and no res)."""
    lib = _lib.load()
    n, H, W, cs = x.shape
    assert H == 14 and W == 14 and w3.shape[1] == 64
    Cin = w1.shape[1]
    if y is None:
        y = torch.empty(n, 14, 14, 256, dtype=torch.float32, device=x.device)
    w2p = pack_conv_weight(w2_oihw)
    scratch = torch.empty(6 * (64 * Cin + 64 * 576 + 2 * 256 * 64), dtype=torch.uint8, device=x.device)
    bw, bb = (branch[0].contiguous(), branch[1].contiguous()) if branch is not None else (None, None)
    _lib.check(lib.offk_bottleneck_chain14_split(_stream(x.device), _ptr(x), cs, x_coff, n, Cin, int(relu_in), _ptr(w1.contiguous()), _ptr(b1),
                                                 _ptr(w2p), _ptr(b2), _ptr(w3.contiguous()), _ptr(b3), _ptr(bw), _ptr(bb), _ptr(res),
                                                 res.shape[-1] if res is not None else 0, 0, _ptr(y), y.shape[-1], y_coff,
                                                 _ptr(scratch), scratch.numel()))
    return y


def winograd_conv3x3(x, w_oihw, bias, res=None, flags=0, x_coff=0, y=None, y_coff=0, want_pool=False):
    """3x3 / stride 1 / pad 1 conv on 7x7 maps as Winograd F(4x4, 3x3) (offk_winograd_conv3x3).  x: [n, 7, 7, Cs]; returns y
    [n, 7, 7, Co] (and, want_pool, the per-tile sums [4 n, Co])."""
    lib = _lib.load()
    n, H, W, cs = x.shape
    assert H == 7 and W == 7
    Co, Ci = w_oihw.shape[:2]
    if y is None:
        y = torch.empty(n, 7, 7, Co, dtype=torch.float32, device=x.device)
    nfl = 121 * (Co * Ci + n * (Ci + Co))
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    pool = torch.empty(4 * n, Co, dtype=torch.float32, device=x.device) if want_pool else None
    _lib.check(lib.offk_winograd_conv3x3(_stream(x.device), _ptr(x), cs, x_coff, n, Ci, _ptr(pack_conv_weight(w_oihw)), _ptr(bias), Co,
                                         _ptr(res), res.shape[-1] if res is not None else 0, 0, flags, _ptr(y), y.shape[-1], y_coff,
                                         _ptr(scratch), nfl, _ptr(pool)))
    return (y, pool) if want_pool else y


def winograd_conv5x5s2(x, w_oihw, bias, flags=0, x_coff=0, y=None, y_coff=0):
    """5x5 / stride 2 / pad 2 conv on 14x14 maps in polyphase Winograd form (offk_winograd_conv5x5s2).  x: [n, 14, 14, Cs]; returns
    y [n, 7, 7, Co]."""
    lib = _lib.load()
    n, H, W, cs = x.shape
    assert H == 14 and W == 14
    Co, Ci = w_oihw.shape[:2]
    if y is None:
        y = torch.empty(n, 7, 7, Co, dtype=torch.float32, device=x.device)
    nfl = 400 * Ci * (Co + n) + 121 * n * Co
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_winograd_conv5x5s2(_stream(x.device), _ptr(x), cs, x_coff, n, Ci, _ptr(pack_conv_weight(w_oihw)), _ptr(bias), Co,
                                           None, 0, 0, flags, _ptr(y), y.shape[-1], y_coff, _ptr(scratch), nfl))
    return y


def winograd_conv7x7s2(x, w_oihw, bias, flags=0, x_coff=0, y=None, y_coff=0):
    """7x7 / stride 2 / pad 3 conv on 28x28 maps in polyphase Winograd form F(5x5, 4x4) (offk_winograd_conv7x7s2).
    x: [n, 28, 28, Cs]; returns y [n, 14, 14, Co]."""
    lib = _lib.load()
    n, H, W, cs = x.shape
    assert H == 28 and W == 28
    Co, Ci = w_oihw.shape[:2]
    if y is None:
        y = torch.empty(n, 14, 14, Co, dtype=torch.float32, device=x.device)
    nfl = 225 * Ci * (Co + 9 * n) + 64 * 9 * n * Co
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_winograd_conv7x7s2(_stream(x.device), _ptr(x), cs, x_coff, n, Ci, _ptr(pack_conv_weight(w_oihw)), _ptr(bias), Co,
                                           flags, _ptr(y), y.shape[-1], y_coff, _ptr(scratch), nfl))
    return y


def winograd_between(M, bias_in, phases_in, w1=None, b1=None, x=None, x_coff=0, precision="fp32"):
    """offk_winograd_between: M [121, n, Cin] (Winograd-domain GEMM output of the conv in front) -> V [121, n, Cmid] (GEMM input of
    the conv behind), through relu(A^T M A + bias_in), optionally a 1x1 conv w1 [Cmid, Cin] + b1 + ReLU, and B^T . B.  x: optional
    [n, 7, 7, Cs] buffer that also receives relu(A^T M A + bias_in) at channels [x_coff, x_coff + Cin)."""
    lib = _lib.load()
    pts, n, cin = M.shape
    assert pts == 121
    cmid = w1.shape[0] if w1 is not None else cin
    V = torch.empty(121, n, cmid, dtype=torch.float32, device=M.device)
    if precision == "f32split":      # offk_winograd_between_ex: stage B (the 1x1 conv) in split-fp32 arithmetic
        scratch = torch.empty(cmid * cin * 6, dtype=torch.uint8, device=M.device)
        _lib.check(lib.offk_winograd_between_ex(_stream(M.device), _ptr(M.contiguous()), _ptr(bias_in), int(phases_in), n, cin, _ptr(x),
                                                x.shape[-1] if x is not None else 0, x_coff, _ptr(w1.contiguous()), _ptr(b1), cmid, _ptr(V),
                                                _lib.PRECISIONS[precision], _ptr(scratch), scratch.numel()))
        return V
    _lib.check(lib.offk_winograd_between(_stream(M.device), _ptr(M.contiguous()), _ptr(bias_in), int(phases_in), n, cin, _ptr(x),
                                         x.shape[-1] if x is not None else 0, x_coff, _ptr(w1.contiguous() if w1 is not None else None),
                                         _ptr(b1), cmid, _ptr(V)))
    return V


def batched_gemm_nt(x, w, precision="fp32"):
    """offk_batched_gemm_nt: y[b] = x[b] @ w[b].T for x [batch, M, K], w [batch, Co, K] (the GEMMs of a conv on a Winograd path);
    precision "f32split": split-fp32 arithmetic on the bf16 matrix pipe (wino_gemm_split.hip)."""
    lib = _lib.load()
    batch, m, k = x.shape
    co = w.shape[1]
    assert w.shape == (batch, co, k)
    y = torch.empty(batch, m, co, dtype=torch.float32, device=x.device)
    scratch = torch.empty(batch * co * k * 6 if precision == "f32split" else 16, dtype=torch.uint8, device=x.device)
    _lib.check(lib.offk_batched_gemm_nt(_stream(x.device), _ptr(x.contiguous()), _ptr(w.contiguous()), _ptr(y), batch, m, k, co,
                                        _lib.PRECISIONS[precision], _ptr(scratch), scratch.numel()))
    return y


def head(x, fc_w, fc_b, maxpool, x_coff=0, c=None):
    lib = _lib.load()
    n, H, W, cs = x.shape
    C = cs if c is None else c
    out = torch.empty(n, fc_w.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_head(_stream(x.device), _ptr(x), cs, x_coff, n, H, W, C, int(maxpool), _ptr(fc_w.contiguous()),
                             _ptr(fc_b.contiguous()), fc_w.shape[0], _ptr(out)))
    return out


def segment_consensus(x, batch):
    lib = _lib.load()
    T = x.shape[0] // batch
    out = torch.empty(batch, x.shape[1], dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_segment_consensus(_stream(x.device), _ptr(x.contiguous()), batch, T, x.shape[1], _ptr(out)))
    return out


def nchw_to_nhwc(x):
    lib = _lib.load()
    n, C, H, W = x.shape
    out = torch.empty(n, H, W, C, dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_nchw_to_nhwc(_stream(x.device), _ptr(x.contiguous()), n, C, H * W, _ptr(out)))
    return out


def nhwc_to_nchw(x, coff=0, c=None):
    lib = _lib.load()
    n, H, W, cs = x.shape
    C = cs - coff if c is None else c
    out = torch.empty(n, C, H, W, dtype=torch.float32, device=x.device)
    _lib.check(lib.offk_nhwc_to_nchw(_stream(x.device), _ptr(x), cs, coff, n, C, H * W, _ptr(out)))
    return out


def score_fusion(score_sets, weights, want_pred=True):
    """score_sets: list of [videos, crops, classes] (or [videos, classes]) fp32 CUDA tensors.
    Returns (fused [videos, classes], pred [videos] int32 or None) -- K7, include/offk.h."""
    lib = _lib.load()
    sets = [s.contiguous() if s.dim() == 3 else s.contiguous().unsqueeze(1) for s in score_sets]
    v, k, c = sets[0].shape
    for s in sets:
        if tuple(s.shape) != (v, k, c) or not s.is_cuda or s.dtype != torch.float32:
            raise ValueError("score sets must be same-shape fp32 CUDA tensors")
    fused = torch.empty(v, c, dtype=torch.float32, device=sets[0].device)
    pred = torch.empty(v, dtype=torch.int32, device=sets[0].device) if want_pred else None
    ptrs = (ctypes.c_void_p * len(sets))(*[s.data_ptr() for s in sets])
    w = (ctypes.c_float * len(sets))(*[float(x) for x in weights])
    _lib.check(lib.offk_score_fusion(_stream(sets[0].device), ptrs, w, len(sets), v, k, c, _ptr(fused), _ptr(pred)))
    return fused, pred
