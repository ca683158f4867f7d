"""CPU tests of the hash-grid encoder: the host mirror against golden vectors captured from the reference's own
Python (tests/golden/grid_encoder_golden.json), and the oracle (oracle/gce_oracle.c) against an independent
float64 PyTorch-autograd formulation."""
import json
import math
import os

import numpy as np
import pytest
import torch

import grid_util as GU
from gaussiancity_amd import grid_encoder as GE

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_encoder_golden.json")))


@pytest.fixture(scope="module")
def go():
    from oracle import grid_oracle as GO
    GO.lib()
    return GO


def _desc(a):
    if isinstance(a, torch.Tensor):
        return {"tensor": list(a.shape), "dtype": str(a.dtype).replace("torch.", "")}
    if isinstance(a, bool):
        return {"bool": a}
    if isinstance(a, int):
        return {"int": a}
    if isinstance(a, float):
        return {"float": a}
    return {"other": repr(a)}


@pytest.mark.parametrize("rec", GOLDEN["configs"], ids=lambda r: "D%d_L%d_C%d" % (r["cfg"]["in_channels"], r["cfg"]["n_levels"], r["cfg"]["lvl_channels"]))
def test_module_and_call_protocol_match_reference(rec, monkeypatch):
    enc = GE.GridEncoder(**rec["cfg"])
    assert [int(v) for v in enc.offsets] == rec["offsets"] and enc.offsets.dtype == torch.int32
    assert enc.per_level_scale == rec["per_level_scale"] and enc.output_dim == rec["output_dim"]
    assert int(enc.n_params) == rec["n_params"] and enc.gridtype_id == rec["gridtype_id"]
    assert list(enc.embeddings.shape) == rec["embeddings_shape"] and sorted(enc.state_dict().keys()) == rec["state_dict_keys"]
    amax = float(enc.embeddings.detach().abs().max())
    assert 0 < amax <= rec["init_abs_max_le"]
    calls = []

    def fwd(*args):
        calls.append(("forward", [_desc(a) for a in args]))
        args[3].fill_(0.25)
        if args[10]:
            args[11].fill_(0.5)

    def bwd(*args):
        calls.append(("backward", [_desc(a) for a in args]))
        args[4].fill_(2.0)
        if args[11]:
            args[13].fill_(3.0)

    monkeypatch.setattr(GE, "ext_forward", fwd)
    monkeypatch.setattr(GE, "ext_backward", bwd)
    torch.manual_seed(1)
    x = (torch.rand(7, 3, rec["cfg"]["in_channels"]) * 2 - 1).requires_grad_(True)
    y = enc(x, bound=1)
    assert list(y.shape) == rec["forward_output_shape"]
    y.sum().backward()
    assert [{"fn": n, "args": a} for n, a in calls] == rec["calls"]          # positional order, shapes, dtypes, scalars
    assert float(x.grad.reshape(-1)[0]) == rec["grad_inputs_value"]
    assert float(enc.embeddings.grad.reshape(-1)[0]) == rec["grad_embeddings_value"]
    calls.clear()
    enc(x.detach(), bound=2).sum().backward()
    assert [{"fn": n, "args": a} for n, a in calls] == rec["calls_no_input_grad"]


CASES = [
    # D, C, L, gridtype, align_corners, log2_hashmap
    (2, 1, 3, 0, False, 8), (2, 8, 4, 1, False, 12), (3, 2, 4, 0, False, 9), (3, 4, 3, 1, True, 14),
    (4, 2, 3, 0, True, 10), (5, 8, 3, 0, False, 11), (5, 1, 2, 1, False, 9),
]


@pytest.mark.parametrize("D,C,L,gridtype,align,lh", CASES)
def test_oracle_matches_float64_autograd(go, D, C, L, gridtype, align, lh):
    rng = np.random.default_rng(100 * D + C)
    B = 200
    x, emb, offsets, S, H = GU.make_case(rng, B, D, C, L, base=3, desired=40, log2_hashmap=lh, align_corners=align)
    out, dy_dx = go.forward(x, emb, offsets, S, H, True, gridtype, align)
    scales = go.level_scales(L, S, H)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    et = torch.from_numpy(emb).double().requires_grad_(True)
    ref = GU.torch_reference(xt, et, offsets, scales, gridtype, align)
    got = torch.from_numpy(out).permute(1, 0, 2).reshape(B, L * C).double()
    assert float((got - ref.detach()).abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max()))
    inside = ((x >= 0) & (x <= 1)).all(axis=1)
    assert np.all(out[:, ~inside] == 0) and np.all(dy_dx[~inside] == 0) and (~inside).sum() >= 10
    g = rng.normal(size=(B, L * C))
    ref.backward(torch.from_numpy(g))
    grad_lbc = np.ascontiguousarray(g.reshape(B, L, C).transpose(1, 0, 2)).astype(np.float32)
    ge, gi = go.backward(grad_lbc, x, emb.shape, offsets, S, H, dy_dx, gridtype, align)
    assert float(np.abs(ge - et.grad.numpy()).max()) <= 2e-5 * max(1.0, float(et.grad.abs().max()))
    assert float(np.abs(gi - xt.grad.numpy()).max()) <= 2e-4 * max(1.0, float(xt.grad.abs().max()))
    assert np.all(gi[~inside] == 0)


def test_level_scales_and_hash_constants(go):
    sc = go.level_scales(16, math.log2(1.381912879967776), 16)
    assert sc[0] == 15.0 and abs(float(sc[15]) - 2047.0) < 0.05 and np.all(np.diff(sc) > 0)
    # tiled level: index = x + y*(res+1), no hashing while the level fits
    x = np.array([[0.5, 0.25]], np.float32)
    offsets = np.array([0, 32], np.int32)  # 5x5 cells used of 32 rows
    emb = np.arange(32, dtype=np.float32)[:, None].copy()
    out, _ = go.forward(x, emb, offsets, 0.0, 4, False, 1, False)  # scale = 3, res = 4
    # pos = (2.0, 1.25): cell (2,1), frac (0, .25) -> rows 7 (w .75) and 12 (w .25)
    assert abs(float(out[0, 0, 0]) - (0.75 * 7 + 0.25 * 12)) < 1e-6


@pytest.mark.parametrize("D,C,L,gridtype,align,lh", CASES)
def test_typed_oracle_agrees_with_the_float32_oracle(go, D, C, L, gridtype, align, lh):
    """oracle/grid_oracle_typed.py (numpy; scalar_t = double / at::Half, what upstream's dtype dispatch instantiates) is
    pinned through the C oracle it restates: in double it agrees with the float32 oracle to float32 rounding (outputs 1e-6,
    dy_dx 1e-4 * max, gradients likewise), in half to binary16 rounding; rows outside [0, 1] encode to exact zeros."""
    from oracle import grid_oracle_typed as GT
    rng = np.random.default_rng(100 * D + C)
    B = 200
    x, emb, offsets, S, H = GU.make_case(rng, B, D, C, L, base=3, desired=40, log2_hashmap=lh, align_corners=align)
    assert np.array_equal(GT.level_scales(L, S, H), go.level_scales(L, S, H))
    o32, d32 = go.forward(x, emb, offsets, S, H, True, gridtype, align)
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    ge32, gi32 = go.backward(g, x, emb.shape, offsets, S, H, d32, gridtype, align)
    inside = ((x >= 0) & (x <= 1)).all(axis=1)
    for T, t_out, t_dd in ((np.float64, 1e-6, 1e-4), (np.float16, 4e-3, 2e-2)):
        o, d = GT.forward(x, emb.astype(T), offsets, S, H, True, gridtype, align)
        assert o.dtype == T and d.dtype == T
        assert float(np.abs(o.astype(np.float64) - o32).max()) <= t_out * max(1.0, float(np.abs(o32).max()))
        assert float(np.abs(d.astype(np.float64) - d32).max()) <= t_dd * max(1.0, float(np.abs(d32).max()))
        assert np.all(o[:, ~inside] == 0) and np.all(d[~inside] == 0)
        ge, gi = GT.backward(g.astype(T), x, emb.shape, offsets, S, H, d, gridtype, align)
        assert float(np.abs(ge.astype(np.float64) - ge32).max()) <= 10 * t_out * max(1.0, float(np.abs(ge32).max()))
        assert float(np.abs(gi.astype(np.float64) - gi32).max()) <= (1e-4 if T == np.float64 else 5e-2) * max(1.0, float(np.abs(gi32).max()))
