"""ctypes/numpy binding of oracle/gcr_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gaussiancity_amd/) never does.

The oracle restates /root/reference/extensions/diff_gaussian_rasterization/cuda_rasterizer/
(forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h); see gcr_oracle.c for the
per-function citations and for the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgcr_oracle.so")
_SO64 = os.path.join(_HERE, "_build", "libgcr_oracle_f64.so")  # the same statements in binary64 (gcr_oracle.c, ORC_F64)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gcr_oracle.c")
    if (force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src)
            or not os.path.exists(_SO64) or os.path.getmtime(_SO64) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class _Camera(C.Structure):
    _fields_ = [
        ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
        ("bg", C.c_void_p), ("view_matrix", C.c_void_p),
        ("proj_matrix", C.c_void_p), ("campos", C.c_void_p),
    ]


class _Camera64(C.Structure):
    _fields_ = [(n, C.c_double if t is C.c_float else t) for n, t in _Camera._fields_]


class _Gaussians(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("M", C.c_int32),
        ("means3D", C.c_void_p), ("opacities", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("scales", C.c_void_p),
        ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
    ]


class _Geom(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "depths", "clamped", "radii", "means2D", "cov3D", "conic_opacity", "rgb",
        "tiles_touched", "point_offsets")]


class _Binning(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("keys_unsorted", "keys", "list_unsorted", "list")]


class _Image(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ranges", "n_contrib", "accum_alpha")]


class _Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dcov3D",
        "dL_dsh", "dL_dscale", "dL_drot")]


_lib = None
_lib64 = None


def lib64():
    """The binary64 build (Frame64): a rounding-noise yardstick for tools/, never a parity target."""
    global _lib64
    if _lib64 is None:
        build()
        L = C.CDLL(_SO64)
        L.orc_preprocess.restype = C.c_int64
        L.orc_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bin.restype = None
        L.orc_bin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_render.restype = None
        L.orc_render.argtypes = [C.c_void_p] * 6
        L.orc_render_backward.restype = None
        L.orc_render_backward.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_preprocess_backward.restype = None
        L.orc_preprocess_backward.argtypes = [C.c_void_p] * 4
        _lib64 = L
    return _lib64


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_preprocess.restype = C.c_int64
        L.orc_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bin.restype = None
        L.orc_bin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_render.restype = None
        L.orc_render.argtypes = [C.c_void_p] * 6
        L.orc_render_backward.restype = None
        L.orc_render_backward.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_preprocess_backward.restype = None
        L.orc_preprocess_backward.argtypes = [C.c_void_p] * 4
        L.orc_mark_visible.restype = None
        L.orc_mark_visible.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_higher_msb.restype = C.c_uint32
        L.orc_higher_msb.argtypes = [C.c_uint32]
        L.orc_set_exp_bias.argtypes = [C.c_int]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data


class Frame:
    """One forward pass through the oracle; keeps every intermediate for stage-wise parity."""

    FT = np.float32        # Frame64 below: np.float64 + the ORC_F64 build
    _camera_t = _Camera
    _lib_fn = staticmethod(lambda: lib())

    @classmethod
    def _f(cls, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=cls.FT))
        return a.reshape(shape) if shape is not None else a

    def __init__(self, *, img_h, img_w, tanfovx, tanfovy, bg, scale_modifier, view_matrix,
                 proj_matrix, sh_degree, campos, means3D, opacities, shs=None,
                 colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        L = self._lib_fn()
        _f32, FT = self._f, self.FT
        self.H, self.W = int(img_h), int(img_w)
        self.means3D = _f32(means3D, (-1, 3))
        P = self.P = self.means3D.shape[0]
        self.opacities = _f32(opacities, (P,))
        self.shs = _f32(shs) if shs is not None and np.size(shs) else None
        self.M = 0 if self.shs is None else self.shs.shape[1]
        self.colors_precomp = _f32(colors_precomp, (P, 3)) if colors_precomp is not None and np.size(colors_precomp) else None
        self.scales = _f32(scales, (P, 3)) if scales is not None and np.size(scales) else None
        self.rotations = _f32(rotations, (P, 4)) if rotations is not None and np.size(rotations) else None
        self.cov3D_precomp = _f32(cov3D_precomp, (P, 6)) if cov3D_precomp is not None and np.size(cov3D_precomp) else None
        self.bg = _f32(bg, (3,))
        self.view = _f32(view_matrix, (16,))
        self.proj = _f32(proj_matrix, (16,))
        self.campos = _f32(campos, (3,))
        self.cam = self._camera_t(self.H, self.W, float(tanfovx), float(tanfovy), float(scale_modifier),
                           int(sh_degree), _ptr(self.bg), _ptr(self.view), _ptr(self.proj),
                           _ptr(self.campos))
        self.g = _Gaussians(P, self.M, _ptr(self.means3D), _ptr(self.opacities), _ptr(self.shs),
                            _ptr(self.colors_precomp), _ptr(self.scales), _ptr(self.rotations),
                            _ptr(self.cov3D_precomp))
        n = max(P, 1)
        self.depths = np.zeros(n, FT)
        self.clamped = np.zeros((n, 3), np.uint8)
        self.radii = np.zeros(n, np.int32)
        self.means2D = np.zeros((n, 2), FT)
        self.cov3D = np.zeros((n, 6), FT)
        self.conic_opacity = np.zeros((n, 4), FT)
        self.rgb = np.zeros((n, 3), FT)
        self.tiles_touched = np.zeros(n, np.uint32)
        self.point_offsets = np.zeros(n, np.uint32)
        self.geo = _Geom(*[_ptr(a) for a in (self.depths, self.clamped, self.radii, self.means2D,
                                              self.cov3D, self.conic_opacity, self.rgb,
                                              self.tiles_touched, self.point_offsets)])
        self.gx, self.gy = (self.W + 15) // 16, (self.H + 15) // 16
        T = self.gx * self.gy
        self.ranges = np.zeros((T, 2), np.uint32)
        self.n_contrib = np.zeros(self.H * self.W, np.uint32)
        self.final_T = np.zeros(self.H * self.W, FT)
        self.img = _Image(_ptr(self.ranges), _ptr(self.n_contrib), _ptr(self.final_T))
        self.out_color = np.zeros((3, self.H, self.W), FT)

        # K1 + K2
        self.R = int(L.orc_preprocess(C.byref(self.cam), C.byref(self.g), C.byref(self.geo))) if P else 0
        R = max(self.R, 1)
        self.keys_unsorted = np.zeros(R, np.uint64)
        self.keys = np.zeros(R, np.uint64)
        self.list_unsorted = np.zeros(R, np.uint32)
        self.point_list = np.zeros(R, np.uint32)
        self.bin = _Binning(_ptr(self.keys_unsorted), _ptr(self.keys), _ptr(self.list_unsorted),
                            _ptr(self.point_list))
        # K3 + K4 + K5
        L.orc_bin(C.byref(self.cam), P, C.byref(self.geo), self.R, C.byref(self.bin),
                  C.byref(self.img))
        # K6
        self.colors = self.colors_precomp if self.colors_precomp is not None else self.rgb
        L.orc_render(C.byref(self.cam), C.byref(self.geo), _ptr(self.colors), C.byref(self.bin),
                     C.byref(self.img), _ptr(self.out_color))

    def backward(self, dL_dpix):
        """K7 + K8; returns dict with the reference's eight gradient tensors (+dL_dconic)."""
        L = self._lib_fn()
        _f32, FT = self._f, self.FT
        P, M = max(self.P, 1), self.M
        dpix = _f32(dL_dpix, (3, self.H, self.W))
        g = dict(
            dL_dmean2D=np.zeros((P, 3), FT), dL_dconic=np.zeros((P, 4), FT),
            dL_dopacity=np.zeros((P, 1), FT), dL_dcolor=np.zeros((P, 3), FT),
            dL_dmean3D=np.zeros((P, 3), FT), dL_dcov3D=np.zeros((P, 6), FT),
            dL_dsh=np.zeros((P, max(M, 0), 3), FT), dL_dscale=np.zeros((P, 3), FT),
            dL_drot=np.zeros((P, 4), FT))
        gs = _Grads(*[_ptr(g[k]) for k in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor",
                                           "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale",
                                           "dL_drot")])
        if self.P:
            L.orc_render_backward(C.byref(self.cam), C.byref(self.geo), _ptr(self.colors),
                                  C.byref(self.bin), C.byref(self.img), self.R, _ptr(dpix),
                                  C.byref(gs))
            L.orc_preprocess_backward(C.byref(self.cam), C.byref(self.g), C.byref(self.geo),
                                      C.byref(gs))
        for k in g:
            g[k] = g[k][: self.P]
        return g

    # R_p of SURVEY.md section 8(d): list entries any pixel of the tile actually consumed.
    def consumed_entries(self):
        nc = self.n_contrib.reshape(self.H, self.W)
        tot = 0
        for ty in range(self.gy):
            for tx in range(self.gx):
                blk = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
                tot += int(blk.max()) if blk.size else 0
        return tot


class Frame64(Frame):
    """The same statements evaluated in binary64 (gcr_oracle.c built with -DORC_F64): how far is the binary32 oracle from
    the exact value of the reference's formulas on THIS scene?  tools/fuzz_f64.py; never a parity target."""
    FT = np.float64
    _camera_t = _Camera64
    _lib_fn = staticmethod(lambda: lib64())


def mark_visible(means3D, view_matrix, proj_matrix):
    m = _f32(means3D, (-1, 3))
    v, p = _f32(view_matrix, (16,)), _f32(proj_matrix, (16,))
    out = np.zeros(max(m.shape[0], 1), np.uint8)
    lib().orc_mark_visible(m.shape[0], _ptr(m), _ptr(v), _ptr(p), _ptr(out))
    return out[: m.shape[0]].astype(bool)


def expf(x):
    return float(lib().orc_expf(C.c_float(x)))


def set_exp_bias(ulps):
    """Census knob: shift every exp() result by `ulps` (0 restores the contract).  See gcr_oracle.c."""
    lib().orc_set_exp_bias(int(ulps))


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
