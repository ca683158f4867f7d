"""Python API of the rasterizer -- import-compatible with GaussianCity's
`extensions.diff_gaussian_rasterization` (reference dgr/__init__.py):

    GaussianRasterizationSettings   dgr/__init__.py:203-215   (field names img_h/img_w/
                                                               view_matrix/proj_matrix)
    GaussianRasterizer              dgr/__init__.py:218-273
    RasterizeGaussiansFunction      dgr/__init__.py:19-200
    GaussianRasterizerWrapper       dgr/__init__.py:276-426

so models/generator.py, core/train.py:144, core/test.py:55, scripts/inference.py:638 and
utils/helpers.py:256 can call it unchanged.  The native side is gaussiancity_amd.ext
(libgcr_hip.so, gfx950); this file is autograd glue and camera bookkeeping only.
"""
import math
import typing

import numpy as np
import scipy.spatial.transform
import torch

from . import ext as _ext

_EMPTY = None
_CPU = torch.device("cpu")


def _absent():
    # The reference marks an absent optional input with an empty CPU tensor
    # (dgr/__init__.py:250-259); numel()==0 is what the native side keys on.
    return torch.Tensor([])


# ---- the reference's world-to-camera recipe without the scipy object --------------------------------------------------------
# dgr/__init__.py:349-368 builds the rotation with scipy.spatial.transform.Rotation.from_quat(q).as_matrix(): ~25 us
# of object construction and validation for thirty floating-point operations, in a frame loop that is host-bound.
# _w2c_fast() performs the SAME IEEE binary64 operations in the same order with Python floats and then follows the
# reference's numpy lines verbatim (the 3x3 @ 3x1 product is numpy's: its summation order is not ours to guess).  That
# this gives scipy's bits is checked, not assumed: the first use compares the two routes on 64 fixed poses -- random ones
# and orbit poses looking at a target, where the translation cancels -- and the fast route is only kept if every one
# agrees bit for bit (tests/test_golden_api.py repeats the comparison on thousands).
_fast_w2c_ok = None


def _w2c_reference_ops(cam_position, cam_quaternion):
    """dgr/__init__.py:349-368, operation by operation (numpy / scipy)."""
    cam_position = np.asarray(cam_position)
    rot = scipy.spatial.transform.Rotation.from_quat(cam_quaternion).as_matrix()
    rot = rot[:, [1, 2, 0]]  # [F|R|U] -> [R|U|F]
    w2c = np.zeros((4, 4), dtype=np.float32)
    w2c[:3, :3] = rot.transpose()
    w2c[:3, [3]] = -rot.transpose() @ cam_position[:, None]
    w2c[3, 3] = 1.0
    return w2c


def _w2c_fast(cam_position, cam_quaternion):
    x, y, z, w = (float(v) for v in cam_quaternion)
    p0, p1, p2 = (float(v) for v in cam_position)
    n = math.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    # scipy's as_matrix(), rows of R = [F|R|U]
    m = ((x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)),
         (2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)),
         (2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2))
    rot = np.array(m)[:, [1, 2, 0]]  # [F|R|U] -> [R|U|F]; from here on: the reference's own lines
    cam_position = np.array((p0, p1, p2))
    w2c = np.zeros((4, 4), dtype=np.float32)
    w2c[:3, :3] = rot.transpose()
    w2c[:3, [3]] = -rot.transpose() @ cam_position[:, None]
    w2c[3, 3] = 1.0
    return w2c


def _fast_w2c_available():
    global _fast_w2c_ok
    if _fast_w2c_ok is None:
        rng = np.random.default_rng(20240927)
        ok = True
        from . import synth
        orbit = synth.orbit_poses()
        for i in range(64):
            q = rng.normal(size=4)
            p = rng.normal(size=3) * (1000.0 if i % 2 else 1.0)
            if i % 3 == 0:
                q, p = q.astype(np.float32).astype(np.float64), p.astype(np.float32).astype(np.float64)
            if i >= 40:
                p, q = orbit[i - 40]
            if not np.array_equal(_w2c_fast(p, q).view(np.uint32), _w2c_reference_ops(p, q).view(np.uint32)):
                ok = False
                break
        _fast_w2c_ok = ok
    return _fast_w2c_ok


class GaussianRasterizationSettings(typing.NamedTuple):
    img_h: int
    img_w: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    view_matrix: torch.Tensor
    proj_matrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU deep copy of an argument tuple, taken before a debug-mode native call."""
    return tuple(a.cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_native(fn, args, debug, dump_path, what):
    """Debug mode keeps the reference's behaviour (dgr/__init__.py:65-83,155-175): snapshot
    the inputs first, dump them if the native call raises, then re-raise."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_path)
        print("\nAn error occured in %s. Writing %s for debugging.\n" % (what, dump_path))
        raise


class RasterizeGaussiansFunction(torch.autograd.Function):
    """autograd bridge; positional signature of dgr/__init__.py:29-41."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings):
        rs = raster_settings
        # native argument order: dgr/rasterize_points.h:18-28
        native_args = (
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
            cov3Ds_precomp, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h,
            rs.img_w, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug,
        )
        # This build's native module offers the same call without the host wait: `num_rendered` comes back as a ticket
        # (the reference keeps the number for the backward only, dgr/__init__.py:85-108) and the forward blend is told
        # whether a backward will follow (iff an input requires a gradient).  Any other module with the reference's
        # three functions -- the recording stand-in of the golden tests, the reference's own extension -- is called
        # exactly as upstream calls it.
        ticketed = getattr(_ext, "rasterize_gaussians_ticket", None)
        if ticketed is not None and not rs.debug:
            need = any(ctx.needs_input_grad)
            num_rendered, color, radii, geom_buffer, binning_buffer, img_buffer = ticketed(*native_args, _for_backward=need)
        else:
            num_rendered, color, radii, geom_buffer, binning_buffer, img_buffer = _call_native(
                _ext.rasterize_gaussians, native_args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii,
                              sh, geom_buffer, binning_buffer, img_buffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buffer,
         binning_buffer, img_buffer) = ctx.saved_tensors
        # native argument order: dgr/rasterize_points.h:30-43
        native_args = (
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier,
            cov3Ds_precomp, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy,
            grad_out_color, sh, rs.sh_degree, rs.campos, geom_buffer, ctx.num_rendered,
            binning_buffer, img_buffer, rs.debug,
        )
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp,
         grad_sh, grad_scales, grad_rotations) = _call_native(
            _ext.rasterize_gaussians_backward, native_args, rs.debug, "snapshot_bw.dump",
            "backward")
        # one gradient per forward() input, raster_settings last (dgr/__init__.py:188-198)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities,
                grad_scales, grad_rotations, grad_cov3Ds_precomp, None)


class _RasterizePoints14Function(torch.autograd.Function):
    """GaussianRasterizerWrapper's call shape -- points [N,14], precomputed colours, optional mirrored image -- as ONE
    autograd node on the tensor in place.  The generic route (slice into five views -> GaussianRasterizer ->
    torch.flip) costs, per training step, five .contiguous() copies in each direction, a zeros_like, five
    slice-backward zero-fill + copy pairs, four gradient adds and two flip kernels around six + three launches of
    actual work; here the native side reads the columns with row strides, stores the image mirrored and writes the
    [N,14] gradient itself (gcr_gaussians.stride_*, gcr_grads.packed, gcr_camera.flip_x).  Same kernels, same bits."""

    @staticmethod
    def forward(ctx, points, raster_settings, flip_x, flip_y, window):
        rs = raster_settings
        num_rendered, color, radii, geom_buffer, binning_buffer, img_buffer = _ext.rasterize_points14(
            points, rs.bg, rs.scale_modifier, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h,
            rs.img_w, rs.campos, flip_x, flip_y, for_backward=ctx.needs_input_grad[0], window=window, ticket=True)
        ctx.raster_settings, ctx.num_rendered, ctx.view = rs, num_rendered, (flip_x, flip_y, window)
        ctx.save_for_backward(points, radii, geom_buffer, binning_buffer, img_buffer)
        return color

    @staticmethod
    def backward(ctx, grad_color):
        rs = ctx.raster_settings
        points, radii, geom_buffer, binning_buffer, img_buffer = ctx.saved_tensors
        grad_points = _ext.rasterize_points14_backward(
            points, radii, rs.bg, rs.scale_modifier, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy,
            grad_color, rs.campos, geom_buffer, ctx.num_rendered, binning_buffer, img_buffer, rs.img_h, rs.img_w,
            *ctx.view)
        return grad_points, None, None, None, None


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum (near-plane) test; exported natively as mark_visible (dgr/bindings.cpp:18)."""
        rs = self.raster_settings
        with torch.no_grad():  # (host-side camera matrices, host_camera=True, go to the device for this one call)
            return _ext.mark_visible(positions, rs.view_matrix.to(positions.device), rs.proj_matrix.to(positions.device))

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        # exactly-one-of checks, bare Exception like dgr/__init__.py:236-248
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        has_any_sr = scales is not None or rotations is not None
        has_both_sr = scales is not None and rotations is not None
        if (not has_both_sr and cov3D_precomp is None) or (has_any_sr and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        opt = [shs, colors_precomp, scales, rotations, cov3D_precomp]
        rs = self.raster_settings
        frame = getattr(_ext, "rasterize_gaussians_frame", None)
        if frame is not None and not rs.debug and not (torch.is_grad_enabled() and any(
                t is not None and t.requires_grad for t in (means3D, means2D, opacities, *opt))):
            # Nothing here asks for a gradient: RasterizeGaussiansFunction.apply would build no graph and hand back the
            # same two tensors -- this build's native module renders such a frame without the autograd node, the state
            # buffers a backward would read, or the host wait (ext.rasterize_gaussians_frame: same kernels, same bits).
            return frame(rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3D_precomp,
                         rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, shs, rs.sh_degree,
                         rs.campos, rs.prefiltered)
        shs, colors_precomp, scales, rotations, cov3D_precomp = [
            _absent() if t is None else t for t in opt]
        return RasterizeGaussiansFunction.apply(
            means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
            self.raster_settings)


class GaussianRasterizerWrapper(torch.nn.Module):
    """GaussianCity's camera/point-tensor adapter (dgr/__init__.py:276-426).

    K: 3x3 intrinsics, sensor_size: (W, H).  Points are [N,14] =
    xyz(3) opacity(1) scale(3) rotation(4, r x y z) rgb(3).  Camera pose is a position
    (tx,ty,tz) plus a quaternion (qx,qy,qz,qw) whose rotation columns are [Forward|Right|Up].
    The projection has P[3,2] = -1 (clip w negative in front of the camera), which mirrors
    the image in x and y; flip_lr=True (default) undoes the x mirror.
    """

    def __init__(self, K, sensor_size, flip_lr=True, flip_ud=False, z_near=0.01, z_far=50000.0,
                 device=torch.device("cuda"), host_camera=None):
        """`host_camera` (not in the reference) says where the per-pose camera matrices are computed and live:

        "reference" (default on a GPU device): the reference's own recipe (scipy, matmul, inverse:
            dgr/__init__.py:349-402) evaluated with torch on the HOST -- bit-equal to what the reference computes
            on a CPU device (tests/golden/camera.npz); the settings' tensors are CPU tensors and the native side
            passes the 38 floats to its kernels by value.  No host-to-device copies, no device GEMM, and no
            `view.inverse()` on the device (rocsolver + a synchronisation that serialises the frame loop: the
            inference loop runs 2 490 frames/s with it, 4 915 without, profiles/r03_bench_inference_loop.jsonl).
        True: closed-form host arithmetic (a rigid transform's inverse translation IS the camera position), ~12 us
            instead of ~50; agrees with the recipe to rounding, not bit for bit.
        False (default on a CPU device; "device" semantics of the reference): the recipe on `device`, settings'
            tensors on `device`."""
        super().__init__()
        if host_camera is None:
            host_camera = "reference" if torch.device(device).type == "cuda" else False
        self.host_camera = host_camera if host_camera == "reference" else bool(host_camera)
        self.flip_lr = flip_lr
        self.flip_ud = flip_ud
        self.z_near = z_near
        self.z_far = z_far
        self.device = device
        self.K = K
        self.sensor_size = sensor_size
        self.fov_x, self.fov_y = self._intrinsic_to_fov()
        self.P = self._get_projection_matrix()

    # ---- camera bookkeeping -----------------------------------------------------------
    def _intrinsic_to_fov(self):
        # dgr/__init__.py:326-331
        w, h = self.sensor_size[0], self.sensor_size[1]
        return (2 * np.arctan2(w, 2 * self.K[0, 0]), 2 * np.arctan2(h, 2 * self.K[1, 1]))

    def _get_projection_matrix(self):
        # dgr/__init__.py:333-347
        w, h = self.sensor_size[0], self.sensor_size[1]
        n, f = self.z_near, self.z_far
        P = np.zeros((4, 4), dtype=np.float32)
        P[0, 0] = 2.0 * self.K[0, 0] / w
        P[1, 1] = 2.0 * self.K[1, 1] / h
        P[0, 2] = (2.0 * self.K[0, 2] / w) - 1.0
        P[1, 2] = (2.0 * self.K[1, 2] / h) - 1.0
        P[2, 2] = -(f + n) / (f - n)
        P[2, 3] = -2.0 * f * n / (f - n)
        P[3, 2] = -1.0
        return torch.from_numpy(P).to(self.device)

    def _get_w2c_matrix(self, cam_position, cam_quaternion, device=None):
        # dgr/__init__.py:349-368 (`device`: where the result lives; None = self.device, as upstream)
        if isinstance(cam_position, torch.Tensor):
            cam_position = cam_position.cpu().numpy()
        if isinstance(cam_quaternion, torch.Tensor):
            cam_quaternion = cam_quaternion.cpu().numpy()
        if _fast_w2c_available() and len(cam_quaternion) == 4 and len(cam_position) == 3:
            w2c = _w2c_fast(cam_position, cam_quaternion)  # the same bits, without the library dispatch (see above)
        else:
            w2c = _w2c_reference_ops(cam_position, cam_quaternion)
        return torch.from_numpy(w2c).to(self.device if device is None else device)

    def _get_gaussian_rasterization_settings(self, cam_position, cam_quaternion):
        # dgr/__init__.py:382-402: row-vector (transposed) matrices, black background,
        # sh_degree 0, unit scale modifier.
        # The arithmetic is done on the strided transposes exactly as upstream does it (inverse() of a strided
        # matrix rounds differently from inverse() of its contiguous copy); what is handed out is .contiguous():
        # same values, but the native module would otherwise copy the 16 floats of a strided matrix on every
        # forward and backward call (a 2-4 us kernel in the frame's dependency chain each time).
        if self.host_camera == "reference":
            return self._reference_recipe_on_host(cam_position, cam_quaternion)
        if self.host_camera:
            return self._host_camera_settings(cam_position, cam_quaternion)
        view = self._get_w2c_matrix(cam_position, cam_quaternion).transpose(0, 1)
        proj_t = self.P.transpose(0, 1)
        return GaussianRasterizationSettings(
            img_h=self.sensor_size[1],
            img_w=self.sensor_size[0],
            tanfovx=math.tan(self.fov_x * 0.5),
            tanfovy=math.tan(self.fov_y * 0.5),
            bg=torch.tensor([0.0, 0.0, 0.0], dtype=torch.float32, device=self.device),
            scale_modifier=1.0,
            view_matrix=view.contiguous(),
            proj_matrix=(view @ proj_t).contiguous(),
            sh_degree=0,
            campos=view.inverse()[3, :3].contiguous(),
            prefiltered=False,
            debug=False,
        )

    def _reference_recipe_on_host(self, cam_position, cam_quaternion, need_campos=True):
        """dgr/__init__.py:349-402 operation by operation, on CPU tensors (host_camera="reference").
        need_campos=False (the wrapper's own render path): `campos` -- the one product of `view.inverse()`, 40 us of
        LAPACK dispatch -- is only read by the SH evaluation, and this wrapper always renders precomputed colours
        (sh_degree 0, dgr/__init__.py:404-420): the inverse is skipped and a zero vector stands in, in settings that
        never leave the wrapper.  get_gaussian_rasterizer() hands out complete settings."""
        view = self._get_w2c_matrix(cam_position, cam_quaternion, device=_CPU).transpose(0, 1)  # (no shared state is
        #                                  touched: two threads may share one wrapper)
        if getattr(self, "_P_cpu_t", None) is None:
            self._P_cpu_t = self.P.detach().cpu().transpose(0, 1)
            self._bg_host = torch.zeros(3, dtype=torch.float32)
        return GaussianRasterizationSettings(
            img_h=self.sensor_size[1], img_w=self.sensor_size[0],
            tanfovx=math.tan(self.fov_x * 0.5), tanfovy=math.tan(self.fov_y * 0.5),
            bg=self._bg_host, scale_modifier=1.0, view_matrix=view.contiguous(),
            proj_matrix=(view @ self._P_cpu_t).contiguous(), sh_degree=0,
            campos=view.inverse()[3, :3].contiguous() if need_campos else self._bg_host, prefiltered=False, debug=False)

    def _host_camera_settings(self, cam_position, cam_quaternion):
        """The same settings from host arithmetic only (opt-in, see __init__).  Rotation from the quaternion
        (x, y, z, w) in float64 as scipy does it, columns [F|R|U] -> [R|U|F]; w2c = [R^T | -R^T t]; the tensors
        handed out are CPU tensors, which the native module passes to its kernels by value.  Scalar Python
        arithmetic on purpose: a dozen 3x3 numpy calls cost more host time than the whole frame's launches."""
        tx, ty, tz = (float(v) for v in (cam_position.tolist() if hasattr(cam_position, "tolist") else cam_position))
        qx, qy, qz, qw = (float(v) for v in (cam_quaternion.tolist() if hasattr(cam_quaternion, "tolist")
                                             else cam_quaternion))
        n = math.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
        x, y, z, w = qx / n, qy / n, qz / n, qw / n
        # rotation matrix, columns (forward, right, up)
        f = (1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w))
        r = (2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w))
        u = (2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y))
        rows = (r, u, f)  # w2c[:3, :3] = [R|U|F]^T
        w2c = [c for row in rows for c in (row[0], row[1], row[2], -(row[0] * tx + row[1] * ty + row[2] * tz))]
        view = np.array(w2c + [0.0, 0.0, 0.0, 1.0], dtype=np.float32).reshape(4, 4).T.copy()
        if getattr(self, "_P_host_t", None) is None:
            self._P_host_t = np.ascontiguousarray(self.P.detach().cpu().numpy().T)
            self._bg_host = torch.zeros(3, dtype=torch.float32)
            self._tan_host = (math.tan(self.fov_x * 0.5), math.tan(self.fov_y * 0.5))
        return GaussianRasterizationSettings(
            self.sensor_size[1], self.sensor_size[0], self._tan_host[0], self._tan_host[1], self._bg_host, 1.0,
            torch.from_numpy(view), torch.from_numpy(view @ self._P_host_t), 0,
            torch.tensor((tx, ty, tz), dtype=torch.float32), False, False)

    def get_gaussian_rasterizer(self, cam_position, cam_quaternion):
        return GaussianRasterizer(
            raster_settings=self._get_gaussian_rasterization_settings(cam_position, cam_quaternion))

    # ---- rendering ---------------------------------------------------------------------
    def _get_gaussian_rasterization(self, points, rasterizer, crop=None):
        # dgr/__init__.py:404-426
        if (type(rasterizer) is GaussianRasterizer and points.is_cuda and points.dtype == torch.float32
                and not rasterizer.raster_settings.debug):
            # this build's own rasterizer: the [N,14] tensor in place, flips (and the caller's crop) folded into the
            # image store
            return _RasterizePoints14Function.apply(points, rasterizer.raster_settings, bool(self.flip_lr),
                                                    bool(self.flip_ud), crop)
        xyz, opacity = points[:, 0:3], points[:, 3:4]
        scales, quaternion, rgbs = points[:, 4:7], points[:, 7:11], points[:, 11:]
        image, _radii = rasterizer(
            means3D=xyz,
            means2D=torch.zeros_like(xyz, dtype=torch.float32, device=self.device),
            shs=None,
            colors_precomp=rgbs,
            opacities=opacity,
            scales=scales,
            rotations=quaternion,
            cov3D_precomp=None,
        )
        if self.flip_lr:
            image = torch.flip(image, dims=[2])
        if self.flip_ud:
            image = torch.flip(image, dims=[1])
        if crop is not None:
            x, y, w, h = crop
            image = image[:, y:y + h, x:x + w]
        return image

    def forward(self, points, cam_position=None, cam_quaternion=None, gaussian_rasterizer=None, crop=None,
                as_uint8=False):
        """`crop` = (x, y, w, h) (not in the reference, used by helpers.get_gaussian_rasterization): return only that
        window of the image -- the same values as image[:, y:y+h, x:x+w], but the native side stores just the window,
        skips the tiles outside it in both directions, and the crop's slice-backward kernels disappear.
        `as_uint8` (not in the reference; inference only): return the uint8 [H,W,3] video frame that
        scripts/inference.py:655-667 derives from the image -- (tensor_to_image(img) * 255).to(uint8) -- stored by the
        blend kernel itself: the same bytes, without the float image's round trip and five elementwise kernels."""
        _, n_channels = points.shape
        assert n_channels == 14, "The input tensor should have 14 channels."
        if as_uint8:
            if gaussian_rasterizer is not None or not points.is_cuda or points.dtype != torch.float32:
                raise RuntimeError("as_uint8 needs float32 GPU points and the wrapper's own rasterizer")
            if torch.is_grad_enabled() and points.requires_grad:
                raise RuntimeError("as_uint8 renders video frames: there is no backward through them (use no_grad)")
            rs = (self._reference_recipe_on_host(cam_position, cam_quaternion, need_campos=False)
                  if self.host_camera == "reference" else
                  self._get_gaussian_rasterization_settings(cam_position, cam_quaternion))
            return _ext.rasterize_points14(points, rs.bg, rs.scale_modifier, rs.view_matrix, rs.proj_matrix, rs.tanfovx,
                                           rs.tanfovy, rs.img_h, rs.img_w, rs.campos, bool(self.flip_lr), bool(self.flip_ud),
                                           window=crop, ticket=True, out_uint8=True)[1]
        if gaussian_rasterizer is None:
            if points.is_cuda and points.dtype == torch.float32 and self.host_camera == "reference":
                rs = self._reference_recipe_on_host(cam_position, cam_quaternion, need_campos=False)
            else:
                rs = self._get_gaussian_rasterization_settings(cam_position, cam_quaternion)
            if points.is_cuda and points.dtype == torch.float32:
                # this build's own rasterizer on the [N,14] tensor in place (see _get_gaussian_rasterization): the
                # GaussianRasterizer module the reference constructs per frame (dgr/__init__.py:376-380) would only be
                # unpacked again -- ten microseconds of a host-bound loop
                if not (torch.is_grad_enabled() and points.requires_grad):
                    # an inference frame: the same call without the autograd node (13 us of a host-bound loop) and
                    # without the backward's state
                    return _ext.rasterize_points14(points, rs.bg, rs.scale_modifier, rs.view_matrix, rs.proj_matrix,
                                                   rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, rs.campos,
                                                   bool(self.flip_lr), bool(self.flip_ud), window=crop, ticket=True)[1]
                return _RasterizePoints14Function.apply(points, rs, bool(self.flip_lr), bool(self.flip_ud), crop)
            gaussian_rasterizer = GaussianRasterizer(raster_settings=rs)
        return self._get_gaussian_rasterization(points, gaussian_rasterizer, crop)
