"""Host mirror of the reference's hash-grid encoder over libgce_hip.so.

  ext_forward / ext_backward   the native module's two functions, positional signatures of
                               extensions/grid_encoder/bindings.cpp:19-40 (re-exported by grid_encoder_ext.py)
  GridEncoderFunction          extensions/grid_encoder/__init__.py:18-124
  GridEncoder                  extensions/grid_encoder/__init__.py:127-193 (same constructor arguments, buffers,
                               parameter name `embeddings`, init range, output layout)

torch supplies device memory, autograd plumbing and the current stream; the computation is in the HIP
library.  float32 only (upstream also dispatches half/double; GaussianCity keeps float32 embeddings).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _native_e as E


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)             # CHECK_CUDA, grid_encoder_ext.cu:466-470
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)       # CHECK_CONTIGUOUS, :472-476
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s must be %s (this build is float32 only)" % (name, dtype))


def ext_forward(inputs, embeddings, offsets, outputs, B, D, C_, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners):
    """grid_encoder_ext.forward (bindings.cpp:19-24): fills outputs [L,B,C] (and dy_dx [B, L*D*C])."""
    _chk(inputs, "inputs", torch.float32)
    _chk(embeddings, "embeddings", torch.float32)
    _chk(offsets, "offsets")
    if offsets.dtype != torch.int32:
        raise RuntimeError("offsets must be an int tensor")               # CHECK_IS_INT, :480
    _chk(outputs, "outputs", torch.float32)
    _chk(dy_dx, "dy_dx", torch.float32)
    with torch.cuda.device(inputs.device):
        E.check(E.lib().gce_forward(inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(), outputs.data_ptr(),
                                    int(B), int(D), int(C_), int(L), float(S), int(H), int(bool(calc_grad_inputs)),
                                    dy_dx.data_ptr(), int(gridtype), int(bool(align_corners)), _stream()),
                "gce_forward")


def ext_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C_, L, S, H, calc_grad_inputs, dy_dx,
                 grad_inputs, gridtype, align_corners):
    """grid_encoder_ext.backward (bindings.cpp:25-33): accumulates into grad_embeddings, fills grad_inputs."""
    for t, n in ((grad, "grad"), (inputs, "inputs"), (embeddings, "embeddings"), (grad_embeddings, "grad_embeddings"),
                 (dy_dx, "dy_dx"), (grad_inputs, "grad_inputs")):
        _chk(t, n, torch.float32)
    _chk(offsets, "offsets")
    with torch.cuda.device(inputs.device):
        E.check(E.lib().gce_backward(grad.data_ptr(), inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(),
                                     grad_embeddings.data_ptr(), int(B), int(D), int(C_), int(L), float(S), int(H),
                                     int(bool(calc_grad_inputs)), dy_dx.data_ptr(), grad_inputs.data_ptr(),
                                     int(gridtype), int(bool(align_corners)), _stream()), "gce_backward")


class GridEncoderFunction(torch.autograd.Function):
    """inputs [B,D] in [0,1], embeddings [rows,C], offsets int32 [L+1] -> [B, L*C]."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        Cc = embeddings.shape[1]
        S = math.log2(per_level_scale)  # the native side works with log2 of the scale (__init__.py:43-44)
        H = base_resolution
        outputs = torch.empty(L, B, Cc, device=inputs.device, dtype=embeddings.dtype)  # level-major, permuted below
        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * Cc, device=inputs.device, dtype=embeddings.dtype)
        else:
            dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)
        ext_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx, gridtype,
                    align_corners)
        outputs = outputs.permute(1, 0, 2).reshape(B, L * Cc)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, Cc, L, S, H, gridtype]
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cc, L, S, H, gridtype = ctx.dims
        grad = grad.view(B, L, Cc).permute(1, 0, 2).contiguous()  # [B, L*C] -> [L, B, C]
        grad_embeddings = torch.zeros_like(embeddings)
        if ctx.calc_grad_inputs:
            grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype)
        else:
            grad_inputs = torch.zeros(1, device=inputs.device, dtype=embeddings.dtype)
        ext_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, ctx.calc_grad_inputs, dy_dx,
                     grad_inputs, gridtype, ctx.align_corners)
        if ctx.calc_grad_inputs:
            return grad_inputs.to(inputs.dtype), grad_embeddings, None, None, None, None, None, None
        return None, grad_embeddings, None, None, None, None, None, None


def level_offsets(in_channels, n_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Row offsets of the levels (__init__.py:158-172).  NOTE: like upstream, the resolution of level i uses the
    constructor's `per_level_scale` ARGUMENT (default 2), not the value derived from desired_resolution that the
    kernels use (self.per_level_scale) -- kept bug-for-bug, the table size depends on it."""
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(n_levels):
        resolution = int(math.ceil(base_resolution * per_level_scale ** i))
        params = min(max_params, (resolution if align_corners else resolution + 1) ** in_channels)
        params = int(math.ceil(params / 8) * 8)
        offsets.append(offset)
        offset += params
    offsets.append(offset)
    return offsets


class GridEncoder(torch.nn.Module):
    def __init__(self, in_channels, n_levels, lvl_channels, desired_resolution, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, gridtype="hash", align_corners=False):
        super().__init__()
        self.in_channels = in_channels
        self.n_levels = n_levels
        self.lvl_channels = lvl_channels
        self.per_level_scale = 2 ** (math.log2(desired_resolution / base_resolution) / (n_levels - 1))
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = n_levels * lvl_channels
        self.gridtype = gridtype
        self.gridtype_id = 0 if gridtype == "hash" else 1
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size
        offsets = level_offsets(in_channels, n_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = self.offsets[-1] * lvl_channels
        self.embeddings = torch.nn.Parameter(torch.empty(offsets[-1], lvl_channels))
        self._init_weights()

    def _init_weights(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def forward(self, inputs, bound=1):
        """inputs [..., in_channels] in [-bound, bound] -> [..., n_levels * lvl_channels]."""
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.in_channels)
        out = GridEncoderFunction.apply(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                                        inputs.requires_grad, self.gridtype_id, self.align_corners)
        return out.view(prefix + [self.output_dim])
