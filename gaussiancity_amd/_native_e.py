"""ctypes binding of libgce_hip.so (C ABI in include/gce.h): multi-resolution hash-grid encoder.
The library is the product: no Python/CPU fallback -- a missing library or failing call raises RuntimeError."""
import ctypes as C
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libgce_hip.so")
EXPORTED_SYMBOLS = ("gce_abi_version", "gce_last_error", "gce_level_scales", "gce_forward", "gce_backward",
                    "gce_set_option", "gce_get_stage_ms", "gce_forward_t", "gce_backward_t")
DTYPE_F32, DTYPE_F16, DTYPE_F64 = 0, 1, 2  # enum gce_dtype
STAGE_NAMES = ("forward", "backward_embeddings", "backward_inputs")
ABI_VERSION = 2
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libgce_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C gaussiancity_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
    L.gce_abi_version.restype = i32
    L.gce_last_error.restype = C.c_char_p
    L.gce_level_scales.restype = i32
    L.gce_level_scales.argtypes = [u32, f32, u32, C.POINTER(f32)]
    L.gce_forward.restype = i32
    L.gce_forward.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, i32, vp, u32, i32, vp]
    L.gce_backward.restype = i32
    L.gce_backward.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, i32, vp, vp, u32, i32, vp]
    L.gce_forward_t.restype = i32
    L.gce_forward_t.argtypes = [i32, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, i32, vp, u32, i32, vp]
    L.gce_backward_t.restype = i32
    L.gce_backward_t.argtypes = [i32, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, i32, vp, vp, u32, i32, vp]
    L.gce_set_option.restype = i32
    L.gce_set_option.argtypes = [C.c_char_p, i32]
    L.gce_get_stage_ms.restype = i32
    L.gce_get_stage_ms.argtypes = [C.POINTER(f32), i32]
    if L.gce_abi_version() != ABI_VERSION:
        raise RuntimeError("libgce_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (gce_status %d): %s" % (what, rc, lib().gce_last_error().decode("utf-8", "replace")))
    return rc


def set_option(name, value):
    return lib().gce_set_option(name.encode(), int(value))


def stage_ms():
    buf = (C.c_float * len(STAGE_NAMES))()
    n = lib().gce_get_stage_ms(buf, len(STAGE_NAMES))
    return {STAGE_NAMES[i]: float(buf[i]) for i in range(n)}
