"""ctypes binding of the C restatement (oracle/off_oracle.c) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboff_oracle_c.so")


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "off_oracle.c")):
        subprocess.run(["make", "-s", "-C", HERE], check=True)
    lib = ctypes.CDLL(LIB)
    lib.off_oracle_forward.restype = ctypes.c_int
    return lib


def forward(feats, weights_ordered, B, L, variant, slice_mode, consensus, ncls=101, stages=False):
    """feats: nine fp32 NCHW ndarrays; weights_ordered: list of ndarrays in
    offk_amd.spec.weight_shapes(variant) order.  Returns (out7, out14, out28[, stage dict])."""
    lib = load()
    P = B * (L - 1)
    rows = B if consensus else P
    fp = ctypes.POINTER(ctypes.c_float)
    keep = [np.ascontiguousarray(f, dtype=np.float32) for f in feats]
    wkeep = [np.ascontiguousarray(w, dtype=np.float32) for w in weights_ordered]
    fa = (fp * 9)(*[k.ctypes.data_as(fp) for k in keep])
    wa = (fp * len(wkeep))(*[k.ctypes.data_as(fp) for k in wkeep])
    outs = [np.empty((rows, ncls), dtype=np.float32) for _ in range(3)]
    st = {}
    if stages:
        st = dict(fusion_28=np.empty((P, 320, 28, 28), np.float32), fusion_14=np.empty((P, 1056, 14, 14), np.float32),
                  fusion_7=np.empty((P, 832, 7, 7), np.float32), sum_7=np.empty((P, 1024, 7, 7), np.float32))
    sp = [st[k].ctypes.data_as(fp) if stages else None for k in ("fusion_28", "fusion_14", "fusion_7", "sum_7")]
    rc = lib.off_oracle_forward(fa, wa, B, L, variant, slice_mode, int(consensus), ncls,
                                *[o.ctypes.data_as(fp) for o in outs], *sp)
    assert rc == 0
    return (outs[0], outs[1], outs[2], st) if stages else tuple(outs)
