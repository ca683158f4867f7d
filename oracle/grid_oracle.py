"""ctypes wrapper of oracle/_build/libgce_oracle.so (oracle/gce_oracle.c): hash-grid encoder.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libgce_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.org_level_scales.argtypes = [i32, C.c_float, u32, vp]
        L.org_forward.argtypes = [vp, vp, vp, vp, u32, i32, u32, u32, vp, i32, vp, u32, i32]
        L.org_backward_embeddings.argtypes = [vp, vp, vp, vp, u32, i32, u32, u32, vp, u32, i32]
        L.org_backward_inputs.argtypes = [vp, vp, vp, u32, i32, u32, u32]
        for f in (L.org_level_scales, L.org_forward, L.org_backward_embeddings, L.org_backward_inputs):
            f.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def level_scales(L, S, H):
    out = np.empty(L, np.float32)
    lib().org_level_scales(int(L), float(S), int(H), _p(out))
    return out


def forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False):
    """-> outputs [L,B,C], dy_dx [B,L,D,C] or None  (grid_encoder_ext.forward, grid_encoder_ext.cu:459-494)."""
    inputs = np.ascontiguousarray(inputs, np.float32)
    emb = np.ascontiguousarray(embeddings, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = inputs.shape
    Cc, L = emb.shape[1], len(offsets) - 1
    sc = level_scales(L, S, H)
    out = np.empty((L, B, Cc), np.float32)
    dd = np.empty((B, L, D, Cc), np.float32) if calc_grad_inputs else None
    lib().org_forward(_p(inputs), _p(emb), _p(offsets), _p(out), B, D, Cc, L, _p(sc), int(calc_grad_inputs),
                      _p(dd) if dd is not None else None, int(gridtype), int(align_corners))
    return out, dd


def backward(grad, inputs, embeddings_shape, offsets, S, H, dy_dx=None, gridtype=0, align_corners=False):
    """grad [L,B,C] -> grad_embeddings, grad_inputs or None  (grid_encoder_ext.backward, :496-539)."""
    grad = np.ascontiguousarray(grad, np.float32)
    inputs = np.ascontiguousarray(inputs, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = inputs.shape
    L, _, Cc = grad.shape
    sc = level_scales(L, S, H)
    ge = np.zeros(embeddings_shape, np.float32)
    lib().org_backward_embeddings(_p(grad), _p(inputs), _p(offsets), _p(ge), B, D, Cc, L, _p(sc), int(gridtype),
                                  int(align_corners))
    gi = None
    if dy_dx is not None:
        gi = np.empty((B, D), np.float32)
        lib().org_backward_inputs(_p(grad), _p(np.ascontiguousarray(dy_dx, np.float32)), _p(gi), B, D, Cc, L)
    return ge, gi
