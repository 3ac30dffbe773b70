"""ctypes binding of liboffk.so (include/offk.h).  No fallback: if the library is missing
or a call fails, this raises -- the product path never runs without the HIP extension."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFFK_LIB") or os.path.join(HERE, "liboffk.so")   # OFFK_LIB: A/B builds in tools
NUM_SITES = 9
NUM_STAGES = 6
STAGE_NAMES = ("pw_reduce", "sobel_tdiff", "fusion_28", "fusion_14", "fusion_7", "heads")

CONV_RELU_IN, CONV_RELU_PRE, CONV_RELU_POST = 1, 2, 4
PRECISION_FP32, PRECISION_F32SPLIT = 0, 2
PRECISIONS = {"fp32": 0, "f32split": 2}      # (1 was "bf16x3", retired in ABI v9)


class OffkError(RuntimeError):
    pass


class OffkConfig(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("length", ctypes.c_int32), ("variant", ctypes.c_int32),
                ("slice_mode", ctypes.c_int32), ("consensus", ctypes.c_int32),
                ("num_classes", ctypes.c_int32), ("feat_layout", ctypes.c_int32),
                ("device", ctypes.c_int32), ("precision", ctypes.c_int32)]


class OffkFeatParts(ctypes.Structure):
    _fields_ = [("n_parts", ctypes.c_int32), ("channels", ctypes.c_int32 * 4), ("data", ctypes.c_void_p * 4)]


class OffkGradView(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("cstride", ctypes.c_int32), ("coff", ctypes.c_int32)]


_c = ctypes
_P = ctypes.c_void_p
_F = ctypes.c_void_p       # float* passed as integer address (tensor.data_ptr())
_I = ctypes.c_int

# name -> (restype, argtypes); every symbol include/offk.h declares
SIGNATURES = {
    "offk_abi_version": (_I, []),
    "offk_last_error": (_c.c_char_p, [_P]),
    "offk_create": (_I, [_c.POINTER(OffkConfig), _c.POINTER(_P)]),
    "offk_destroy": (_I, [_P]),
    "offk_set_weight": (_I, [_P, _c.c_char_p, _F, _c.POINTER(_c.c_int64), _I]),
    "offk_bind_weight": (_I, [_P, _c.c_char_p, _F, _c.POINTER(_c.c_int64), _I]),
    "offk_missing_weights": (_I, [_P, _c.c_char_p, _c.c_size_t]),
    "offk_workspace_bytes": (_c.c_size_t, [_P]),
    "offk_forward": (_I, [_P, _P, _c.POINTER(_F), _F, _F, _F, _P]),
    "offk_forward_parts": (_I, [_P, _P, _c.POINTER(OffkFeatParts), _F, _F, _F, _P]),
    "offk_workspace_region": (_I, [_P, _c.c_char_p, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t)]),
    "offk_set_profiling": (_I, [_P, _I]),
    "offk_stage_times": (_I, [_P, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64), _I]),
    "offk_launch_times": (_I, [_P, _c.c_char_p, _c.c_size_t, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64), _I, _I]),
    "offk_pw_reduce": (_I, [_P, _P, _I, _F, _F, _F]),
    "offk_sobel_tdiff": (_I, [_P, _P, _I, _F, _F, _F, _I, _I, _I]),
    "offk_sobel_tdiff_all": (_I, [_P, _P, _P, _I]),
    "offk_off_units": (_I, [_P, _P, _c.POINTER(_F), _P]),
    "offk_off_units_fused": (_I, [_P, _P, _c.POINTER(_F), _P]),
    "offk_conv2d": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _F, _F, _I, _I, _I, _I, _I, _F, _I, _I, _I, _F, _I, _I]),
    "offk_conv2d_ex": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _F, _F, _I, _I, _I, _I, _I, _F, _I, _I, _I, _F, _I, _I,
                            _I, _I, _F, _c.c_size_t, _I]),
    "offk_bottleneck_chain14": (_I, [_P, _F, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _I, _F, _I, _I, _F, _I, _I]),
    "offk_bottleneck_chain14_split": (_I, [_P, _F, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _F, _I, _I, _P, ctypes.c_size_t]),
    "offk_winograd_conv3x3": (_I, [_P, _F, _I, _I, _I, _I, _F, _F, _I, _F, _I, _I, _I, _F, _I, _I, _F, _c.c_size_t, _F]),
    "offk_winograd_conv5x5s2": (_I, [_P, _F, _I, _I, _I, _I, _F, _F, _I, _F, _I, _I, _I, _F, _I, _I, _F, _c.c_size_t]),
    "offk_winograd_conv7x7s2": (_I, [_P, _F, _I, _I, _I, _I, _F, _F, _I, _I, _F, _I, _I, _F, _c.c_size_t]),
    "offk_winograd_between": (_I, [_P, _F, _F, _I, _I, _I, _F, _I, _I, _F, _F, _I, _F]),
    "offk_winograd_between_ex": (_I, [_P, _F, _F, _I, _I, _I, _F, _I, _I, _F, _F, _I, _F, _I, _P, ctypes.c_size_t]),
    "offk_batched_gemm_nt": (_I, [_P, _F, _F, _F, _I, _I, _I, _I, _I, _P, _c.c_size_t]),
    "offk_pack_conv_weight": (_I, [_P, _F, _I, _I, _I, _I, _F]),
    "offk_set_conv_plan": (_I, [_P, _c.c_char_p, _I, _I]),
    "offk_head": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _F]),
    "offk_segment_consensus": (_I, [_P, _F, _I, _I, _I, _F]),
    "offk_score_fusion": (_I, [_P, _c.POINTER(_F), _c.POINTER(_c.c_float), _I, _I, _I, _I, _F, _F]),
    "offk_train_workspace_bytes": (_c.c_size_t, [_P]),
    "offk_off_units_train": (_I, [_P, _P, _c.POINTER(_F), _P, _c.c_uint64, _c.c_double]),
    "offk_unit_grad_floats": (_c.c_size_t, [_P]),
    "offk_unit_grad_slot": (_I, [_P, _c.c_char_p, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t)]),
    "offk_off_units_backward": (_I, [_P, _P, _c.POINTER(_F), _c.POINTER(OffkGradView), _P, _c.c_uint64, _c.c_double, _F, _I]),
    "offk_segment_consensus_backward": (_I, [_P, _F, _I, _I, _I, _F]),
    "offk_nchw_to_nhwc": (_I, [_P, _F, _I, _I, _I, _F]),
    "offk_nhwc_to_nchw": (_I, [_P, _F, _I, _I, _I, _I, _I, _F]),
}

_lib = None


def load():
    """Load liboffk.so (once).  Raises OffkError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # Load PyTorch's HIP runtime first: liboffk.so then binds to the libamdhip64 already in the
    # process.  Loaded the other way round the process ends up with two HIP runtimes (the system
    # one and the one bundled with torch) and the second to initialise sees no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise OffkError("liboffk.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or optical-flow-guided-feature-pytorch_amd/build.py); there is no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.offk_abi_version() != 10:
        raise OffkError("liboffk.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != 0:
        msg = load().offk_last_error(handle)
        raise OffkError("liboffk error %d: %s" % (rc, msg.decode() if msg else "?"))
