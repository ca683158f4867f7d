"""ctypes binding of libgcr_hip.so (C ABI in include/gcr.h).

The library is the product: there is NO Python/CPU fallback.  If it is missing or a call
fails, a RuntimeError is raised (the reference surfaces native failures the same way:
C++ exceptions -> RuntimeError, dgr/rasterize_points.cu:46-48, cr/auxiliary.h:158-167).
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# GCR_LIB_PATH: tools/ point it at the experiment build (make -C csrc experiments); nothing else sets it
LIB_PATH = os.environ.get("GCR_LIB_PATH") or os.path.join(_CSRC, "libgcr_hip.so")

# Every symbol include/gcr.h declares; tests check that the built library exports all of them.
EXPORTED_SYMBOLS = (
    "gcr_abi_version", "gcr_last_error", "gcr_geometry_bytes", "gcr_image_bytes",
    "gcr_binning_bytes", "gcr_get_layout", "gcr_forward", "gcr_forward_preprocess", "gcr_forward_render",
    "gcr_backward", "gcr_build_cull_cache", "gcr_cull_cache_bytes", "gcr_mark_visible", "gcr_rasterize_forward", "gcr_set_option",
    "gcr_get_stage_ms", "gcr_grad_record_floats", "gcr_grad_record_floats_opt", "gcr_binning_bytes_lean",
    "gcr_forward_async", "gcr_ticket_poll", "gcr_ticket_wait", "gcr_host_words_alloc", "gcr_host_words_free", "gcr_rescue_count",
    "gcr_get_option", "gcr_rescue_dropped_count",
)
GRAD_REC_FLOATS = 16  # gcr_grad_record_floats() by default (32 under option "deterministic_backward": ext asks per call)

STAGE_NAMES = ("preprocess", "scan", "emit", "sort", "ranges", "blend_fwd", "blend_bwd",
               "preprocess_bwd")


class Options(C.Structure):
    """gcr_options: per-call overrides of the gcr_set_option() defaults (-1 = the default)."""
    _fields_ = [(n, C.c_int32) for n in (
        "lazy_sort", "sort_in_blend", "bwd_piece", "deterministic_backward", "split_preprocess",
        "force_radix", "force_global_cursor", "bwd_wave_units")]

    def __init__(self, **kw):
        super().__init__(*([-1] * 8))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("unknown rasterizer option %r" % k)
            setattr(self, k, int(v))


class Camera(C.Structure):
    """gcr_camera == GaussianRasterizationSettings (dgr/__init__.py:203-215)."""
    _fields_ = [
        ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("bg", C.c_void_p), ("view_matrix", C.c_void_p),
        ("proj_matrix", C.c_void_p), ("campos", C.c_void_p),
        ("host_camera", C.c_int32),  # bg/view/proj/campos are host pointers (copied into the kernel arguments)
        ("flip_x", C.c_int32), ("flip_y", C.c_int32),  # mirrored image store / gradient load
        ("win_x", C.c_int32), ("win_y", C.c_int32), ("win_w", C.c_int32), ("win_h", C.c_int32),  # output window
        ("backward", C.c_int32),  # 1: a backward call will follow (the forward blend leaves its per-piece state)
        ("options", C.POINTER(Options)),  # per-call options or NULL
        ("out_u8", C.c_int32),  # out_color is a uint8 [H,W,3] video frame (inference frames)
    ]


class Gaussians(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("M", C.c_int32),
        ("means3D", C.c_void_p), ("opacities", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p),
        ("stride_means3D", C.c_int32), ("stride_opacities", C.c_int32), ("stride_colors", C.c_int32),
        ("stride_scales", C.c_int32), ("stride_rotations", C.c_int32),
        ("cull_cache", C.c_void_p),  # ABI v8: gcr_cull_cache_bytes(P) bytes of gcr_build_cull_cache, or NULL
    ]


class Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D",
        "dL_dsh", "dL_dscales", "dL_drotations")] + [
        ("stride_means3D", C.c_int32), ("stride_opacity", C.c_int32), ("stride_colors", C.c_int32),
        ("stride_scales", C.c_int32), ("stride_rotations", C.c_int32),
        ("packed", C.c_void_p), ("packed_floats", C.c_int64)]


class Layout(C.Structure):
    _fields_ = [
        ("geom_rec", C.c_size_t), ("geom_cov3D", C.c_size_t), ("geom_clamped", C.c_size_t),
        ("geom_tiles_touched", C.c_size_t), ("geom_block_sums", C.c_size_t),
        ("geom_vis_list", C.c_size_t), ("geom_vis_count", C.c_size_t),
        ("geom_num_rendered", C.c_size_t), ("geom_block_tiles", C.c_size_t), ("geom_total", C.c_size_t),
        ("img_final_T", C.c_size_t), ("img_n_contrib", C.c_size_t), ("img_ranges", C.c_size_t),
        ("img_tile_cursor", C.c_size_t), ("img_tile_table", C.c_size_t), ("img_tile_lazy", C.c_size_t),
        ("img_total", C.c_size_t),
        ("bin_keys", C.c_size_t * 2), ("bin_vals", C.c_size_t * 2), ("bin_hist", C.c_size_t),
        ("bin_sorted", C.c_size_t), ("bin_work", C.c_size_t), ("bin_mask", C.c_size_t), ("bin_ckpt", C.c_size_t),
        ("bin_total", C.c_size_t), ("bin_lean_total", C.c_size_t), ("bin_staged", C.c_size_t),
        ("geom_vis_rec", C.c_size_t),  # ABI v9
    ]


class FrameInfo(C.Structure):
    _fields_ = [("num_rendered", C.c_int64), ("max_tile_instances", C.c_int64)]


ABI_VERSION = 9
TICKET_WORDS = 8  # 64-bit pinned host words per asynchronous frame (include/gcr.h, gcr_forward_async)
RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_lib = None


def build(force=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "-s", "-j4"]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load libgcr_hip.so; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgcr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C gaussiancity_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.gcr_abi_version.restype = C.c_int
    L.gcr_last_error.restype = C.c_char_p
    L.gcr_geometry_bytes.restype = C.c_size_t
    L.gcr_geometry_bytes.argtypes = [C.c_int32]
    L.gcr_image_bytes.restype = C.c_size_t
    L.gcr_image_bytes.argtypes = [C.c_int32, C.c_int32]
    L.gcr_binning_bytes.restype = C.c_size_t
    L.gcr_binning_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    L.gcr_get_layout.restype = C.c_int
    L.gcr_get_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(Layout)]
    L.gcr_forward_preprocess.restype = C.c_int
    L.gcr_forward_preprocess.argtypes = [C.POINTER(Camera), C.POINTER(Gaussians), C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.POINTER(FrameInfo), C.c_void_p]
    L.gcr_forward.restype = C.c_int
    L.gcr_forward.argtypes = [C.POINTER(Camera), C.POINTER(Gaussians), C.c_void_p, C.c_size_t,
                              C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                              C.c_void_p, C.c_void_p, C.POINTER(FrameInfo), C.c_void_p]
    L.gcr_binning_bytes_lean.restype = C.c_size_t
    L.gcr_binning_bytes_lean.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    L.gcr_host_words_alloc.restype = C.c_void_p
    L.gcr_host_words_alloc.argtypes = [C.c_size_t]
    L.gcr_host_words_free.restype = None
    L.gcr_host_words_free.argtypes = [C.c_void_p]
    L.gcr_forward_async.restype = C.c_int
    L.gcr_forward_async.argtypes = [C.POINTER(Camera), C.POINTER(Gaussians), C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.gcr_ticket_poll.restype = C.c_int
    L.gcr_ticket_poll.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.POINTER(FrameInfo)]
    L.gcr_ticket_wait.restype = C.c_int
    L.gcr_ticket_wait.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_void_p, C.POINTER(FrameInfo)]
    L.gcr_rescue_count.restype = C.c_long
    L.gcr_rescue_dropped_count.restype = C.c_long
    L.gcr_grad_record_floats_opt.restype = C.c_int
    L.gcr_grad_record_floats_opt.argtypes = [C.POINTER(Options)]
    L.gcr_forward_render.restype = C.c_int
    L.gcr_forward_render.argtypes = [C.POINTER(Camera), C.POINTER(Gaussians), C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.POINTER(FrameInfo), C.c_void_p, C.c_void_p]
    L.gcr_backward.restype = C.c_int
    L.gcr_backward.argtypes = [C.POINTER(Camera), C.POINTER(Gaussians), C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                               C.c_int64, C.c_void_p, C.POINTER(Grads), C.c_void_p]
    L.gcr_cull_cache_bytes.restype = C.c_size_t
    L.gcr_cull_cache_bytes.argtypes = [C.c_int32]
    L.gcr_build_cull_cache.restype = C.c_int
    L.gcr_build_cull_cache.argtypes = [C.POINTER(Gaussians), C.c_float, C.c_void_p, C.c_void_p]
    L.gcr_mark_visible.restype = C.c_int
    L.gcr_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    L.gcr_rasterize_forward.restype = C.c_int64
    L.gcr_rasterize_forward.argtypes = [RESIZE_FN, C.c_void_p, RESIZE_FN, C.c_void_p, RESIZE_FN,
                                        C.c_void_p, C.POINTER(Camera), C.POINTER(Gaussians),
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    L.gcr_set_option.restype = C.c_int
    L.gcr_set_option.argtypes = [C.c_char_p, C.c_int]
    L.gcr_get_option.restype = C.c_int
    L.gcr_get_option.argtypes = [C.c_char_p]
    L.gcr_get_stage_ms.restype = C.c_int
    L.gcr_get_stage_ms.argtypes = [C.POINTER(C.c_float), C.c_int]
    L.gcr_grad_record_floats.restype = C.c_int
    if L.gcr_grad_record_floats() not in (GRAD_REC_FLOATS, 2 * GRAD_REC_FLOATS):
        raise RuntimeError("libgcr_hip.so gradient record size mismatch")
    if L.gcr_abi_version() != ABI_VERSION:
        raise RuntimeError("libgcr_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc < 0:
        msg = lib().gcr_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (gcr_status %d): %s" % (what, rc, msg))
    return rc


def get_layout(P, W, H, R):
    out = Layout()
    check(lib().gcr_get_layout(int(P), int(W), int(H), int(R), C.byref(out)), "gcr_get_layout")
    return out


def set_option(name, value):
    return lib().gcr_set_option(name.encode(), int(value))


def get_option(name):
    """Current process-wide default of a gcr_set_option() option (gcr_get_option, ABI v7: a plain read, safe beside other
    threads' frames).  Per-call values travel in gcr_options (ext.options), not here."""
    v = lib().gcr_get_option(name.encode())
    if v == -2 ** 31:
        raise KeyError("unknown rasterizer option %r" % name)
    return v


def stage_ms():
    buf = (C.c_float * len(STAGE_NAMES))()
    n = lib().gcr_get_stage_ms(buf, len(STAGE_NAMES))
    return {STAGE_NAMES[i]: float(buf[i]) for i in range(n)}
