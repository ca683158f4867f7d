"""Host mirror of the reference's hash-grid encoder over libgce_hip.so.

  ext_forward / ext_backward   the native module's two functions, positional signatures of
                               extensions/grid_encoder/bindings.cpp:19-40 (re-exported by grid_encoder_ext.py)
  GridEncoderFunction          extensions/grid_encoder/__init__.py:18-124
  GridEncoder                  extensions/grid_encoder/__init__.py:127-193 (same constructor arguments, buffers,
                               parameter name `embeddings`, init range, output layout)

torch supplies device memory, autograd plumbing and the current stream; the computation is in the HIP
library.  The embeddings' dtype selects the kernels as upstream's AT_DISPATCH_FLOATING_TYPES_AND_HALF does
(grid_encoder_ext.cu:555,597): float32 (GaussianCity's own; the tuned kernels), float16, float64.  `inputs` are float32.
"""
import ctypes as C
import math

import torch

from . import _native_e as E


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_DTYPES = {torch.float32: E.DTYPE_F32, torch.float16: E.DTYPE_F16, torch.float64: E.DTYPE_F64}


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)             # CHECK_CUDA, grid_encoder_ext.cu:466-470
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)       # CHECK_CONTIGUOUS, :472-476
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s must be %s" % (name, dtype))


def _scalar_type(embeddings):
    """gce_dtype of the call: the embeddings' dtype, as upstream dispatches (grid_encoder_ext.cu:555)."""
    code = _DTYPES.get(embeddings.dtype)
    if code is None:
        raise RuntimeError("embeddings must be float32, float16 or float64 (got %s)" % embeddings.dtype)
    return code


def ext_forward(inputs, embeddings, offsets, outputs, B, D, C_, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners):
    """grid_encoder_ext.forward (bindings.cpp:19-24): fills outputs [L,B,C] (and dy_dx [B, L*D*C])."""
    _chk(inputs, "inputs", torch.float32)
    code = _scalar_type(embeddings)
    _chk(embeddings, "embeddings")
    _chk(offsets, "offsets")
    if offsets.dtype != torch.int32:
        raise RuntimeError("offsets must be an int tensor")               # CHECK_IS_INT, :480
    _chk(outputs, "outputs", embeddings.dtype)
    _chk(dy_dx, "dy_dx", embeddings.dtype)
    with torch.cuda.device(inputs.device):
        E.check(E.lib().gce_forward_t(code, inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(), outputs.data_ptr(),
                                      int(B), int(D), int(C_), int(L), float(S), int(H), int(bool(calc_grad_inputs)),
                                      dy_dx.data_ptr(), int(gridtype), int(bool(align_corners)), _stream()),
                "gce_forward")


def ext_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C_, L, S, H, calc_grad_inputs, dy_dx,
                 grad_inputs, gridtype, align_corners):
    """grid_encoder_ext.backward (bindings.cpp:25-33): accumulates into grad_embeddings, fills grad_inputs."""
    code = _scalar_type(embeddings)
    _chk(inputs, "inputs", torch.float32)
    for t, n in ((grad, "grad"), (embeddings, "embeddings"), (grad_embeddings, "grad_embeddings"), (dy_dx, "dy_dx"),
                 (grad_inputs, "grad_inputs")):
        _chk(t, n, embeddings.dtype)
    _chk(offsets, "offsets")
    with torch.cuda.device(inputs.device):
        E.check(E.lib().gce_backward_t(code, grad.data_ptr(), inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(),
                                       grad_embeddings.data_ptr(), int(B), int(D), int(C_), int(L), float(S), int(H),
                                       int(bool(calc_grad_inputs)), dy_dx.data_ptr(), grad_inputs.data_ptr(),
                                       int(gridtype), int(bool(align_corners)), _stream()), "gce_backward")


class _Geometry:
    """Static description of one encode call; what the native side needs besides the tensors."""
    __slots__ = ("n_points", "in_dim", "channels", "levels", "log2_scale", "base", "grid_kind", "corners_aligned",
                 "want_input_grad")

    def __init__(self, points, table, level_rows, per_level_scale, base_resolution, want_input_grad, grid_kind,
                 corners_aligned):
        self.n_points, self.in_dim = points.shape
        self.channels = table.shape[1]
        self.levels = level_rows.shape[0] - 1
        self.log2_scale = math.log2(per_level_scale)  # the kernels take log2 of the growth factor
        self.base = base_resolution
        self.grid_kind = grid_kind
        self.corners_aligned = corners_aligned
        self.want_input_grad = bool(want_input_grad)

    def native(self):
        return (self.n_points, self.in_dim, self.channels, self.levels, self.log2_scale, self.base)


class GridEncoderFunction(torch.autograd.Function):
    """Autograd node of the encoder.  Positional signature, saved state and returned gradients follow
    extensions/grid_encoder/__init__.py:18-124: (inputs [B,D] in [0,1], embeddings [rows,C], offsets int32 [L+1],
    per_level_scale, base_resolution, calc_grad_inputs, gridtype, align_corners) -> [B, L*C]."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False):
        points = inputs.contiguous()
        geo = _Geometry(points, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype,
                        align_corners)
        B, D, Cc, L, S, H = geo.native()
        # the native kernels write level-major [L, B, C]; callers see point-major [B, L*C]
        level_major = embeddings.new_empty((L, B, Cc))
        jacobian = embeddings.new_empty((B, L * D * Cc) if geo.want_input_grad else (1,))
        ext_forward(points, embeddings, offsets, level_major, B, D, Cc, L, S, H, calc_grad_inputs, jacobian, gridtype,
                    align_corners)
        ctx.save_for_backward(points, embeddings, offsets, jacobian)
        ctx.dims = [B, D, Cc, L, S, H, gridtype]
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.align_corners = align_corners
        return level_major.transpose(0, 1).reshape(B, L * Cc)

    @staticmethod
    def backward(ctx, grad):
        points, table, level_rows, jacobian = ctx.saved_tensors
        B, D, Cc, L, S, H, gridtype = ctx.dims
        wants_inputs = ctx.calc_grad_inputs
        upstream = grad.reshape(B, L, Cc).transpose(0, 1).contiguous()      # back to level-major
        d_table = torch.zeros_like(table)                                   # the kernel accumulates with atomics
        d_points = table.new_zeros(points.shape if wants_inputs else (1,))
        ext_backward(upstream, points, table, level_rows, d_table, B, D, Cc, L, S, H, wants_inputs, jacobian, d_points,
                     gridtype, ctx.align_corners)
        d_in = d_points.to(points.dtype) if wants_inputs else None
        return d_in, d_table, None, None, None, None, None, None


def level_offsets(in_channels, n_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """First table row of every level, plus the total (extensions/grid_encoder/__init__.py:158-172): a level holds
    min(2^log2_hashmap_size, side^in_channels) rows rounded up to a multiple of 8, side = resolution (+1 unless
    align_corners).  NOTE: like upstream, the resolution of level i grows with the constructor's `per_level_scale`
    ARGUMENT (default 2), not with the value derived from desired_resolution that the kernels use
    (self.per_level_scale) -- kept bug-for-bug, the table size depends on it."""
    cap = 1 << log2_hashmap_size
    starts = [0]
    for lvl in range(n_levels):
        side = int(math.ceil(base_resolution * per_level_scale ** lvl)) + (0 if align_corners else 1)
        rows = min(cap, side ** in_channels)
        starts.append(starts[-1] + 8 * int(math.ceil(rows / 8)))
    return starts


class GridEncoder(torch.nn.Module):
    """extensions.grid_encoder.GridEncoder (extensions/grid_encoder/__init__.py:127-193): same constructor
    arguments, attributes, `offsets` buffer, `embeddings` parameter and U(-1e-4, 1e-4) initialisation."""

    def __init__(self, in_channels, n_levels, lvl_channels, desired_resolution, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, gridtype="hash", align_corners=False):
        super().__init__()
        growth = (desired_resolution / base_resolution) ** (1.0 / (n_levels - 1)) if n_levels > 1 else 1.0
        self.in_channels, self.n_levels, self.lvl_channels = in_channels, n_levels, lvl_channels
        self.base_resolution, self.log2_hashmap_size = base_resolution, log2_hashmap_size
        self.per_level_scale = 2 ** (math.log2(desired_resolution / base_resolution) / (n_levels - 1)) if n_levels > 1 else growth
        self.gridtype, self.gridtype_id = gridtype, (0 if gridtype == "hash" else 1)
        self.align_corners = align_corners
        self.output_dim = n_levels * lvl_channels
        self.max_params = 1 << log2_hashmap_size
        starts = level_offsets(in_channels, n_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.tensor(starts, dtype=torch.int32))
        self.n_params = self.offsets[-1] * lvl_channels
        self.embeddings = torch.nn.Parameter(torch.empty(starts[-1], lvl_channels))
        self._init_weights()

    def _init_weights(self):
        torch.nn.init.uniform_(self.embeddings, -1e-4, 1e-4)

    def forward(self, inputs, bound=1):
        """inputs [..., in_channels] in [-bound, bound] -> [..., n_levels * lvl_channels]."""
        lead = inputs.shape[:-1]
        unit = ((inputs + bound) / (2 * bound)).view(-1, self.in_channels)
        code = GridEncoderFunction.apply(unit, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                                         unit.requires_grad, self.gridtype_id, self.align_corners)
        return code.view(*lead, self.output_dim)
