"""-m gpu: the HIP hash-grid encoder (gaussiancity_amd.grid_encoder -> C ABI include/gce.h -> gfx950 kernels)
against the oracle.  Bars (gce-fp32-v1): outputs, dy_dx and grad_inputs BIT-EXACT; grad_embeddings is a sum of
float atomics (order-dependent) -> max|d| <= 1e-5 * max(1, max|ref|)."""
import math

import numpy as np
import pytest
import torch

import grid_util as GU
from gaussiancity_amd import grid_encoder as GE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def go():
    from oracle import grid_oracle as GO
    GO.lib()
    return GO


def _run(dev, x, emb, offsets, S, H, calc, gridtype, align, grad_lbc=None):
    B, D = x.shape
    L, C = len(offsets) - 1, emb.shape[1]
    xt, et, ot = (torch.from_numpy(a).to(dev) for a in (x, emb, offsets))
    out = torch.empty(L, B, C, device=dev)
    dd = torch.empty(B, L * D * C, device=dev) if calc else torch.empty(1, device=dev)
    GE.ext_forward(xt, et, ot, out, B, D, C, L, S, H, calc, dd, gridtype, align)
    res = [out.cpu().numpy(), dd.cpu().numpy().reshape(B, L, D, C) if calc else None]
    if grad_lbc is not None:
        ge = torch.zeros_like(et)
        gi = torch.zeros(B, D, device=dev) if calc else torch.zeros(1, device=dev)
        GE.ext_backward(torch.from_numpy(grad_lbc).to(dev), xt, et, ot, ge, B, D, C, L, S, H, calc, dd, gi, gridtype, align)
        res += [ge.cpu().numpy(), gi.cpu().numpy() if calc else None]
    return res


@pytest.mark.parametrize("D", [2, 3, 4, 5])
@pytest.mark.parametrize("C", [1, 2, 4, 8])
def test_forward_backward_parity(cuda_device, go, D, C):
    rng = np.random.default_rng(10 * D + C)
    for gridtype, align, lh in ((0, False, 9), (1, True, 13)):
        L, B = 5, 3001  # ragged: not a multiple of the 256-thread block
        x, emb, offsets, S, H = GU.make_case(rng, B, D, C, L, base=3, desired=50, log2_hashmap=lh, align_corners=align)
        grad = rng.normal(size=(L, B, C)).astype(np.float32)
        out_o, dd_o = go.forward(x, emb, offsets, S, H, True, gridtype, align)
        ge_o, gi_o = go.backward(grad, x, emb.shape, offsets, S, H, dd_o, gridtype, align)
        out, dd, ge, gi = _run(cuda_device, x, emb, offsets, S, H, True, gridtype, align, grad)
        assert np.array_equal(out.view(np.uint32), out_o.view(np.uint32)), "outputs not bit-exact"
        assert np.array_equal(dd.view(np.uint32), dd_o.view(np.uint32)), "dy_dx not bit-exact"
        assert np.array_equal(gi.view(np.uint32), gi_o.view(np.uint32)), "grad_inputs not bit-exact"
        assert float(np.abs(ge - ge_o).max()) <= 1e-5 * max(1.0, float(np.abs(ge_o).max()))
        # without input gradients the placeholder buffers are left alone and the results are the same
        out2, _, ge2, _ = _run(cuda_device, x, emb, offsets, S, H, False, gridtype, align, grad)
        assert np.array_equal(out2, out) and float(np.abs(ge2 - ge_o).max()) <= 1e-5 * max(1.0, float(np.abs(ge_o).max()))


def test_gaussiancity_configuration_and_properties(cuda_device, go):
    """models/generator.py:37-42: D=5, 16 levels, 8 channels, 2^19 rows per level (268 MB table), 16384 points
    (config.py:34).  Oracle parity on a slice of the points; linearity in the table at full size."""
    enc = GE.GridEncoder(in_channels=5, n_levels=16, lvl_channels=8, desired_resolution=2048).to(cuda_device)
    assert enc.embeddings.numel() * 4 == 268435456
    torch.manual_seed(3)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.rand(16384, 5, device=cuda_device) * 2 - 1
    y = enc(x)
    assert tuple(y.shape) == (16384, 128) and bool(torch.isfinite(y).all())
    S, H = math.log2(enc.per_level_scale), enc.base_resolution
    xs = ((x[:512] + 1) / 2).cpu().numpy()
    out_o, _ = go.forward(xs, enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(), S, H)
    want = torch.from_numpy(out_o).permute(1, 0, 2).reshape(512, 128)
    assert torch.equal(y[:512].cpu().view(torch.int32), want.view(torch.int32))
    with torch.no_grad():
        enc.embeddings.mul_(2.0)
    assert torch.equal(enc(x), 2.0 * y)  # the encoding is linear in the table (x2 is exact in fp32)


def test_autograd_module_matches_oracle(cuda_device, go):
    enc = GE.GridEncoder(in_channels=3, n_levels=6, lvl_channels=4, desired_resolution=96, base_resolution=4,
                         log2_hashmap_size=11).to(cuda_device)
    torch.manual_seed(5)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = (torch.rand(4, 250, 3, device=cuda_device) * 2.4 - 1.2).requires_grad_(True)  # some points outside [-1, 1]
    y = enc(x, bound=1)
    g = torch.randn_like(y)
    y.backward(g)
    S, H = math.log2(enc.per_level_scale), enc.base_resolution
    xn = ((x.detach() + 1) / 2).reshape(-1, 3).cpu().numpy()
    out_o, dd_o = go.forward(xn, enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(), S, H, True)
    want = torch.from_numpy(out_o).permute(1, 0, 2).reshape(4, 250, 24)
    assert torch.equal(y.detach().cpu().view(torch.int32), want.view(torch.int32))
    grad_lbc = np.ascontiguousarray(g.reshape(1000, 6, 4).permute(1, 0, 2).cpu().numpy())
    ge_o, gi_o = go.backward(grad_lbc, xn, tuple(enc.embeddings.shape), enc.offsets.cpu().numpy(), S, H, dd_o)
    assert float(np.abs(enc.embeddings.grad.cpu().numpy() - ge_o).max()) <= 1e-5 * max(1.0, float(np.abs(ge_o).max()))
    assert np.allclose(x.grad.reshape(-1, 3).cpu().numpy(), gi_o * 0.5, rtol=0, atol=1e-6 * max(1.0, float(np.abs(gi_o).max())))
    outside = ((xn < 0) | (xn > 1)).any(axis=1)
    assert outside.sum() > 50 and not np.any(y.detach().reshape(-1, 24).cpu().numpy()[outside])


def test_argument_errors(cuda_device):
    x = torch.rand(8, 3, device=cuda_device)
    emb = torch.rand(64, 2, device=cuda_device)
    off = torch.tensor([0, 32, 64], dtype=torch.int32, device=cuda_device)
    out = torch.empty(2, 8, 2, device=cuda_device)
    dd = torch.empty(1, device=cuda_device)
    GE.ext_forward(x, emb, off, out, 8, 3, 2, 2, 1.0, 4, False, dd, 0, False)
    with pytest.raises(RuntimeError, match="CUDA"):
        GE.ext_forward(x.cpu(), emb, off, out, 8, 3, 2, 2, 1.0, 4, False, dd, 0, False)
    with pytest.raises(RuntimeError, match="contiguous"):
        GE.ext_forward(x.t().contiguous().t(), emb, off, out, 8, 3, 2, 2, 1.0, 4, False, dd, 0, False)
    with pytest.raises(RuntimeError, match="int tensor"):
        GE.ext_forward(x, emb, off.long(), out, 8, 3, 2, 2, 1.0, 4, False, dd, 0, False)
    with pytest.raises(RuntimeError, match="C must be"):  # grid_encoder_ext.cu:399
        GE.ext_forward(x, torch.rand(64, 3, device=cuda_device), off, torch.empty(2, 8, 3, device=cuda_device), 8, 3, 3, 2,
                       1.0, 4, False, dd, 0, False)
    with pytest.raises(RuntimeError, match="D must be"):  # :455
        GE.ext_forward(torch.rand(8, 6, device=cuda_device), emb, off, out, 8, 6, 2, 2, 1.0, 4, False, dd, 0, False)
    with pytest.raises(RuntimeError, match="outputs must be torch.float16"):   # one dtype per call: the embeddings'
        GE.ext_forward(x, emb.half(), off, out, 8, 3, 2, 2, 1.0, 4, False, dd.half(), 0, False)
    with pytest.raises(RuntimeError, match="float32, float16 or float64"):
        GE.ext_forward(x, emb.bfloat16(), off, out.bfloat16(), 8, 3, 2, 2, 1.0, 4, False, dd.bfloat16(), 0, False)
    with pytest.raises(RuntimeError, match="inputs must be torch.float32"):      # inputs stay float (ge/:97)
        GE.ext_forward(x.double(), emb.double(), off, out.double(), 8, 3, 2, 2, 1.0, 4, False, dd.double(), 0, False)


@pytest.mark.parametrize("dtype", [np.float16, np.float64], ids=["half", "double"])
@pytest.mark.parametrize("D,C,gridtype,align,lh", [(2, 1, 0, False, 8), (3, 2, 0, False, 9), (3, 4, 1, True, 13), (4, 8, 0, True, 10),
                                                    (5, 8, 0, False, 11), (5, 1, 1, False, 9)])
def test_half_and_double_dispatch_matches_the_typed_oracle(cuda_device, dtype, D, C, gridtype, align, lh):
    """Upstream dispatches the three kernels over the embeddings' dtype (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
    grid_encoder_ext.cu:555,597); GaussianCity itself stays in float32, the drop-in does not have to.  Against
    oracle/grid_oracle_typed.py (the reference's arithmetic with scalar_t accumulators: c10::Half rounds after every product
    and every sum): outputs, dy_dx and grad_inputs BIT-EXACT in both dtypes; grad_embeddings is a sum of scalar_t atomics
    (order-dependent roundings): 1e-12 * max in double, and in half within the rounding a sum of that many binary16 addends
    can carry."""
    from oracle import grid_oracle_typed as GT
    rng = np.random.default_rng(7 * D + C)
    L, B = 4, 1777
    x, emb32, offsets, S, H = GU.make_case(rng, B, D, C, L, base=3, desired=40, log2_hashmap=lh, align_corners=align)
    emb = emb32.astype(dtype)
    grad = rng.normal(size=(L, B, C)).astype(dtype)
    out_o, dd_o = GT.forward(x, emb, offsets, S, H, True, gridtype, align)
    ge_o, gi_o = GT.backward(grad, x, emb.shape, offsets, S, H, dd_o, gridtype, align)
    tdt = torch.float16 if dtype == np.float16 else torch.float64
    dev = cuda_device
    xt, et, ot = torch.from_numpy(x).to(dev), torch.from_numpy(emb).to(dev), torch.from_numpy(offsets).to(dev)
    out = torch.empty(L, B, C, device=dev, dtype=tdt)
    dd = torch.empty(B, L * D * C, device=dev, dtype=tdt)
    GE.ext_forward(xt, et, ot, out, B, D, C, L, S, H, True, dd, gridtype, align)
    ge = torch.zeros_like(et)
    gi = torch.zeros(B, D, device=dev, dtype=tdt)
    GE.ext_backward(torch.from_numpy(grad).to(dev), xt, et, ot, ge, B, D, C, L, S, H, True, dd, gi, gridtype, align)
    bits = np.uint16 if dtype == np.float16 else np.uint64
    assert np.array_equal(out.cpu().numpy().view(bits), out_o.view(bits)), "outputs not bit-exact"
    assert np.array_equal(dd.cpu().numpy().reshape(B, L, D, C).view(bits), dd_o.view(bits)), "dy_dx not bit-exact"
    assert np.array_equal(gi.cpu().numpy().view(bits), gi_o.view(bits)), "grad_inputs not bit-exact"
    err = float(np.abs(ge.cpu().numpy().astype(np.float64) - ge_o.astype(np.float64)).max())
    scale = max(1.0, float(np.abs(ge_o.astype(np.float64)).max()))
    assert err <= (1e-12 if dtype == np.float64 else 2e-2) * scale, (err, scale)
    # the module: a half / double table runs through the same autograd node and returns the table's dtype
    enc = GE.GridEncoder(in_channels=D, n_levels=3, lvl_channels=C, desired_resolution=32, base_resolution=4,
                         log2_hashmap_size=9).to(dev).to(tdt)
    xin = (torch.rand(257, D, device=dev) * 2 - 1).requires_grad_(True)
    y = enc(xin)
    assert y.dtype == tdt and tuple(y.shape) == (257, 3 * C)
    y.float().sum().backward()
    assert enc.embeddings.grad.dtype == tdt and xin.grad.dtype == torch.float32 and bool(torch.isfinite(xin.grad).all())
