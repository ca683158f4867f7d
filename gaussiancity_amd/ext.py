"""Native-module surface: the three functions of `diff_gaussian_rasterization_ext`.

Mirrors the reference's pybind module (dgr/bindings.cpp:15-19) and its torch binding
(dgr/rasterize_points.cu:37-173): same names, positional argument order, return tuples and
tensor shapes -- but the work is done by libgcr_hip.so (hand-written gfx950 kernels) through
the C ABI of include/gcr.h.  PyTorch only supplies device memory and the current HIP stream.

Differences, all on the safe side:
  * non-float32 / non-GPU inputs raise RuntimeError instead of being undefined behaviour;
  * kernels run on torch's *current* stream, not the legacy default stream, so DDP's
    side-stream bucket all-reduce orders correctly against them;
  * scratch buffers live on means3D's device (the reference uses "current device").
"""
import collections
import ctypes as C
import threading

import torch

from . import _native as N

NUM_CHANNELS = 3  # cr/config.h:15

# num_rendered of the last frame per (device, P, W, H): the next frame's binning buffer is
# allocated for 1.5x that before the frame starts, which lets gcr_forward enqueue the whole frame
# without a mid-frame host stall.  A wrong guess only costs the staged path for that frame.
# Bounded (least recently used first out): a training loop whose point count changes every step
# would otherwise grow the table without limit.
_CAPACITY_HINT_MAX = 64
_capacity_hint = collections.OrderedDict()


# Debug aid (the GPU tests switch it on): hand gcr_backward NaN-filled outputs instead of whatever the caching
# allocator returns, so that an element the library failed to write cannot pass for a zero gradient.
poison_outputs = False


# "A backward call will follow this forward": set by RasterizeGaussiansFunction around its native call (per host
# thread), read by rasterize_gaussians.  A pure performance hint (gcr_camera.backward): the forward blend leaves its
# checkpoints in the smaller pieces the backward balances best with; results do not depend on it.
_tls = threading.local()


def set_backward_hint(flag):
    _tls.backward = bool(flag)


def _hint_get(key):
    v = _capacity_hint.get(key)
    if v is None:
        return 0, 0
    _capacity_hint.move_to_end(key)
    return v


def _hint_put(key, value):
    _capacity_hint[key] = value
    _capacity_hint.move_to_end(key)
    while len(_capacity_hint) > _CAPACITY_HINT_MAX:
        _capacity_hint.popitem(last=False)


def _dev_f32(t, name, device):
    """Return (tensor_or_None, data_ptr_or_None); numel()==0 is the reference's 'absent'."""
    if t is None or t.numel() == 0:
        return None, None
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    if t.device != device:
        raise RuntimeError("%s must live on %s (got %s)" % (name, device, t.device))
    if not t.is_contiguous():
        t = t.contiguous()
    return t, t.data_ptr()


def _camera(device, bg, view, proj, campos, tan_fovx, tan_fovy, H, W, scale_modifier, degree,
            prefiltered, debug, for_backward=False, flip_x=False, flip_y=False, window=None):
    """gcr_camera.  The four camera tensors are device tensors as in the reference -- or ALL four CPU tensors
    (GaussianRasterizerWrapper(host_camera=True)): then the library copies the 38 floats into its kernels' arguments
    (gcr_camera.host_camera) and no device copy of the camera exists at all."""
    cams = ((bg, "bg", 3), (view, "viewmatrix", 16), (proj, "projmatrix", 16), (campos, "campos", 3))
    on_host = all(isinstance(t, torch.Tensor) and t.device.type == "cpu" and t.numel() != 0 for t, _, _ in cams)
    keep = []
    ptrs = []
    for t, name, n in cams:
        if on_host:
            if t.dtype != torch.float32:
                raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
            tt = t.contiguous()
            p = tt.data_ptr()
        else:
            tt, p = _dev_f32(t, name, device)
        if tt is None or tt.numel() != n:
            raise RuntimeError("%s must have %d elements" % (name, n))
        keep.append(tt)
        ptrs.append(p)
    cam = N.Camera(int(H), int(W), float(tan_fovx), float(tan_fovy), float(scale_modifier),
                   int(degree), int(bool(prefiltered)), int(bool(debug)), *ptrs)
    cam.host_camera = int(on_host)
    cam.flip_x = int(bool(flip_x))
    cam.flip_y = int(bool(flip_y))
    cam.backward = int(bool(for_backward))
    if window is not None:  # (x, y, w, h) of the mirrored image: gcr_camera.win_*
        cam.win_x, cam.win_y, cam.win_w, cam.win_h = (int(v) for v in window)
    return cam, keep


def _gaussians(device, P, means3D, opacity, sh, colors, scales, rotations, cov3D_precomp):
    keep = []
    ptr = {}
    for t, name in ((means3D, "means3D"), (opacity, "opacity"), (sh, "sh"), (colors, "colors"),
                    (scales, "scales"), (rotations, "rotations"),
                    (cov3D_precomp, "cov3D_precomp")):
        tt, p = _dev_f32(t, name, device)
        keep.append(tt)
        ptr[name] = p
    M = 0
    if sh is not None and sh.numel() != 0:  # dgr/rasterize_points.cu:72-75
        if sh.dim() != 3 or sh.size(0) != P or sh.size(2) != 3:
            raise RuntimeError("sh must have dimensions (num_points, M, 3)")
        M = int(sh.size(1))
    for t, name, last in ((opacity, "opacity", None), (colors, "colors", 3), (scales, "scales", 3),
                          (rotations, "rotations", 4), (cov3D_precomp, "cov3D_precomp", 6)):
        if t is not None and t.numel() != 0 and t.numel() != P * (last or 1):
            raise RuntimeError("%s has %d elements, expected %d" % (name, t.numel(), P * (last or 1)))
    g = N.Gaussians(int(P), M, ptr["means3D"], ptr["opacity"], ptr["sh"], ptr["colors"],
                    ptr["scales"], ptr["rotations"], ptr["cov3D_precomp"])
    return g, keep


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    # the raw getter skips the Stream object round trip (several microseconds per frame on the host)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (the guard costs microseconds per frame)."""

    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)
        return False


def _forward(L, device, cam, g, P, H, W):
    """gcr_forward (+ the staged retry) with torch-owned buffers; returns the reference's six-tuple."""
    byte = dict(dtype=torch.uint8, device=device)
    # every pixel / every radius is written by the kernels, so no zero-fill launches are needed
    out_color = torch.empty((NUM_CHANNELS, cam.win_h, cam.win_w) if cam.win_w else (NUM_CHANNELS, H, W),
                            dtype=torch.float32, device=device)
    radii = torch.empty((P,), dtype=torch.int32, device=device)
    stream = _stream(device)
    geom = torch.empty((L.gcr_geometry_bytes(P),), **byte)
    img = torch.empty((L.gcr_image_bytes(W, H),), **byte)
    info = N.FrameInfo()
    key = (device.index, P, W, H)
    capacity, list_cap = _hint_get(key)
    binning = torch.empty((L.gcr_binning_bytes(capacity, W, H) if capacity else 0,), **byte)
    rc = N.check(L.gcr_forward(C.byref(cam), C.byref(g), geom.data_ptr(), geom.numel(),
                               binning.data_ptr() if capacity else None, binning.numel(), capacity,
                               list_cap, img.data_ptr(), img.numel(), radii.data_ptr(), out_color.data_ptr(),
                               C.byref(info), stream),
                 "gcr_forward")
    R = int(info.num_rendered)
    if rc == 1:  # GCR_RETRY_RENDER: no/too small a guess, or a tile list beyond the LDS sort
        binning = torch.empty((L.gcr_binning_bytes(R, W, H),), **byte)
        N.check(L.gcr_forward_render(C.byref(cam), C.byref(g), geom.data_ptr(), geom.numel(),
                                     binning.data_ptr(), binning.numel(), img.data_ptr(),
                                     img.numel(), C.byref(info), out_color.data_ptr(), stream),
                "gcr_forward_render")
    longest = int(info.max_tile_instances)
    _hint_put(key, (R + R // 2 + 4096, longest))  # the library adds its own margin to the longest list
    return R, out_color, radii, geom, binning, img


def _empty_frame(device, H, W):
    e = torch.empty((0,), dtype=torch.uint8, device=device)  # dgr/rasterize_points.cu:71: zero image, nothing rendered
    return (0, torch.zeros((NUM_CHANNELS, H, W), dtype=torch.float32, device=device),
            torch.zeros((0,), dtype=torch.int32, device=device), e, e.clone(), e.clone())


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                        image_width, sh, degree, campos, prefiltered, debug, _for_backward=None):
    """RasterizeGaussiansCUDA (dgr/rasterize_points.cu:37-93, dgr/rasterize_points.h:18-28).

    Returns (num_rendered:int, out_color[3,H,W], radii[P] int32, geomBuffer, binningBuffer,
    imgBuffer) -- the three buffers are opaque uint8 tensors to be handed back to
    rasterize_gaussians_backward.

    `_for_backward` (keyword, not part of the reference's positional signature above) is a pure performance hint:
    None = what set_backward_hint() announced for this thread (RasterizeGaussiansFunction does, when an input requires
    a gradient).  The forward blend then leaves its checkpoints in the smaller pieces the backward balances best with
    (gcr_camera.backward).
    """
    if _for_backward is None:
        _for_backward = getattr(_tls, "backward", False)
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor: this rasterizer has no CPU path")
    L = N.lib()
    device = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if P == 0:
        return _empty_frame(device, H, W)
    with _on_device(device):
        cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx,
                              tan_fovy, H, W, scale_modifier, degree, prefiltered, debug, _for_backward)
        g, keep_g = _gaussians(device, P, means3D, opacity, sh, colors, scales, rotations,
                               cov3D_precomp)
        out = _forward(L, device, cam, g, P, H, W)
        del keep_c, keep_g
    return out


# ---- GaussianCity's own call shape: points [N,14] = xyz(3) opacity(1) scale(3) rotation(4) rgb(3) ---------------------
# dgr/__init__.py:404-426 slices that tensor into five strided views, which the reference's binding copies one by one
# (.contiguous()), and autograd assembles the [N,14] gradient from five slice-backward kernels and four adds.  The C ABI
# takes row strides (gcr_gaussians.stride_*, gcr_grads.stride_* / packed): the kernels read the tensor in place and
# write its gradient in place -- same arithmetic, same bits, about twenty small launches fewer per training step.
_POINT_COLUMNS = dict(means3D=0, opacities=3, scales=4, rotations=7, colors=11)


def _points14_gaussians(points):
    if points.dim() != 2 or points.size(1) != 14:
        raise RuntimeError("points must have dimensions (num_points, 14)")
    if not points.is_cuda:
        raise RuntimeError("points must be a GPU tensor: this rasterizer has no CPU path")
    if points.dtype != torch.float32:
        raise RuntimeError("points must be float32 (got %s)" % points.dtype)
    if points.stride(1) != 1:
        points = points.contiguous()
    base, row = points.data_ptr(), int(points.stride(0))
    col = {k: base + 4 * v for k, v in _POINT_COLUMNS.items()}
    g = N.Gaussians(int(points.size(0)), 0, col["means3D"], col["opacities"], None, col["colors"], col["scales"],
                    col["rotations"], None)
    g.stride_means3D = g.stride_opacities = g.stride_colors = g.stride_scales = g.stride_rotations = row
    return g, points


def rasterize_points14(points, background, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                       image_width, campos, flip_x=False, flip_y=False, for_backward=False, window=None):
    """Forward of GaussianRasterizerWrapper's call shape on the [N,14] tensor in place (precomputed colours, SH degree
    0).  Same six-tuple as rasterize_gaussians; the image comes out already mirrored if flip_x / flip_y ask for it, and
    as the `window` = (x, y, w, h) of that mirrored image if one is given (tiles outside it are not blended)."""
    L = N.lib()
    g, pts = _points14_gaussians(points)
    device = pts.device
    P, H, W = int(pts.size(0)), int(image_height), int(image_width)
    if P == 0:
        return _empty_frame(device, *((window[3], window[2]) if window is not None else (H, W)))
    with _on_device(device):
        cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W,
                              scale_modifier, 0, False, False, for_backward, flip_x, flip_y, window)
        out = _forward(L, device, cam, g, P, H, W)
        del keep_c, pts
    return out


def rasterize_points14_backward(points, radii, background, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                dL_dout_color, campos, geomBuffer, R, binningBuffer, imageBuffer, image_height,
                                image_width, flip_x=False, flip_y=False, window=None):
    """Gradient of rasterize_points14 with respect to `points`, as ONE [N,14] tensor the library fills completely
    (`dL_dout_color` has the shape of the image that was handed out: the window's, if there was one)."""
    L = N.lib()
    g, pts = _points14_gaussians(points)
    device = pts.device
    P = int(pts.size(0))
    H, W = int(image_height), int(image_width)
    want = (NUM_CHANNELS, window[3], window[2]) if window is not None else (NUM_CHANNELS, H, W)
    if tuple(dL_dout_color.shape) != want:
        raise RuntimeError("dL_dout_color has shape %s, expected %s" % (tuple(dL_dout_color.shape), want))
    grad = torch.empty((P, 14), dtype=torch.float32, device=device)
    if P == 0:
        return grad
    if poison_outputs:
        grad.fill_(float("nan"))
    # scratch the API wants besides: accumulation records [P,16] (64-byte aligned), dL_dmeans2D [P,3], dL_dcov3D [P,6]
    nrec = int(L.gcr_grad_record_floats())  # 16, or 32 under option "deterministic_backward"
    flat = torch.empty((P * (nrec + 3 + 6) + 64,), dtype=torch.float32, device=device)
    if poison_outputs:
        flat.fill_(float("nan"))
    rec = flat[:P * nrec]
    m2d = flat[P * nrec:P * (nrec + 3)]
    c3d = flat[P * (nrec + 3):P * (nrec + 9)]
    with _on_device(device):
        cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W,
                              scale_modifier, 0, False, False, False, flip_x, flip_y, window)
        dpix, dpix_ptr = _dev_f32(dL_dout_color, "dL_dout_color", device)
        gb = grad.data_ptr()
        col = {k: gb + 4 * v for k, v in _POINT_COLUMNS.items()}
        grads = N.Grads(m2d.data_ptr(), rec.data_ptr(), col["opacities"], col["colors"], col["means3D"], c3d.data_ptr(),
                        None, col["scales"], col["rotations"])
        grads.stride_means3D = grads.stride_opacity = grads.stride_colors = grads.stride_scales = \
            grads.stride_rotations = 14
        grads.packed = gb
        grads.packed_floats = P * 14
        N.check(L.gcr_backward(C.byref(cam), C.byref(g), radii.data_ptr(), geomBuffer.data_ptr(), geomBuffer.numel(),
                               binningBuffer.data_ptr() if binningBuffer.numel() else None, binningBuffer.numel(),
                               imageBuffer.data_ptr(), imageBuffer.numel(), int(R), dpix_ptr, C.byref(grads),
                               _stream(device)),
                "gcr_backward")
        del keep_c, dpix, pts
    return grad


def _gradient_buffers(P, M, device):
    """The reference allocates nine torch::zeros tensors (dgr/rasterize_points.cu:118-126); here they are nine views
    of ONE uninitialised buffer: gcr_backward writes every element of every output itself (include/gcr.h).
    dL_dconic is the native side's accumulation scratch: one 64-byte record per Gaussian.
    Order: dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations."""
    nrec = int(N.lib().gcr_grad_record_floats())  # 16, or 32 under option "deterministic_backward"
    shapes = ((P, 3), (P, 3), (P, NUM_CHANNELS), (P, nrec), (P, 1), (P, 6), (P, M, 3), (P, 3), (P, 4))
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    starts, off = [], 0
    for n in sizes:  # every view starts 256-byte aligned (dL_dconic and dL_drotations are accessed as float4)
        starts.append(off)
        off += (n + 63) // 64 * 64
    flat = (torch.empty if P != 0 else torch.zeros)((off,), dtype=torch.float32, device=device)
    if poison_outputs and P != 0:
        flat.fill_(float("nan"))
    return [flat[st:st + n].view(sh) for st, n, sh in zip(starts, sizes, shapes)]


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (dgr/rasterize_points.cu:97-155, .h:30-43).

    Returns (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3],
    dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4]) in that order
    (dgr/rasterize_points.cu:153-154).
    """
    L = N.lib()
    device = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    (dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales,
     dL_drotations) = _gradient_buffers(P, M, device)
    if P != 0:
        with _on_device(device):
            cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx,
                                  tan_fovy, H, W, scale_modifier, degree, False, debug)  # (flips: see rasterize_points14)
            # opacity is not an input of the backward (it is read from the geometry state)
            g, keep_g = _gaussians(device, P, means3D, None, sh, colors, scales, rotations,
                                   cov3D_precomp)
            dpix, dpix_ptr = _dev_f32(dL_dout_color, "dL_dout_color", device)
            if radii.dtype != torch.int32:
                raise RuntimeError("radii must be int32")
            radii_c = radii.contiguous()
            grads = N.Grads(dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
                            dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                            dL_dsh.data_ptr() if M else None, dL_dscales.data_ptr(),
                            dL_drotations.data_ptr())
            gb, bb, ib = geomBuffer.contiguous(), binningBuffer.contiguous(), imageBuffer.contiguous()
            N.check(L.gcr_backward(C.byref(cam), C.byref(g), radii_c.data_ptr(), gb.data_ptr(),
                                   gb.numel(), bb.data_ptr() if bb.numel() else None, bb.numel(),
                                   ib.data_ptr(), ib.numel(), int(R), dpix_ptr, C.byref(grads),
                                   _stream(device)),
                    "gcr_backward")
            del keep_c, keep_g, dpix
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
            dL_drotations)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (dgr/rasterize_points.cu:157-173): bool[P], True where view-space z > 0.2."""
    L = N.lib()
    device = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        if not means3D.is_cuda:
            raise RuntimeError("means3D must be a GPU tensor: this rasterizer has no CPU path")
        with torch.cuda.device(device):
            m, mp = _dev_f32(means3D, "means3D", device)
            v, vp = _dev_f32(viewmatrix, "viewmatrix", device)
            p, pp = _dev_f32(projmatrix, "projmatrix", device)
            N.check(L.gcr_mark_visible(P, mp, vp, pp, present.data_ptr(), _stream(device)),
                    "gcr_mark_visible")
            del m, v, p
    return present
