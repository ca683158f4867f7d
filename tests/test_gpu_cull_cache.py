"""-m gpu: a static scene's cull cache (gcr_gaussians.cull_cache, gcr_build_cull_cache; gaussiancity_amd/cull_cache.py).

The cache changes what K1's streaming cull READS (one 16-byte record per Gaussian instead of means + scales + rotations),
never what a frame IS: every case below holds the cached frame to the oracle with the bars of test_gpu_parity (radii,
num_rendered, per-Gaussian state, sorted lists, n_contrib, final_T and the image bit-exact), and the Python layer's
bookkeeping to its contract -- an in-place edit of any cached tensor rebuilds, training-type frames never use it."""
import ctypes as C

import numpy as np
import pytest
import torch

import gpu_util as G
import scenes
from test_gpu_parity import _check_forward, _frame

pytestmark = pytest.mark.gpu


@pytest.fixture
def cache():
    from gaussiancity_amd import cull_cache
    cull_cache.invalidate()
    prev = cull_cache.enable(True)
    before = dict(cull_cache.stats)
    yield cull_cache, before
    cull_cache.enable(prev)
    cull_cache.invalidate()


@pytest.mark.parametrize("name,P,W,H,deg,spread,seed", [
    ("mostly_culled", 60000, 256, 160, 2, 150.0, 71),   # a city-like frame: 2.5 % survive the cull
    ("mostly_visible", 40000, 256, 160, 1, 25.0, 57),   # several processing passes per block, queue leftovers
    ("ragged_small", 3000, 200, 150, 3, None, 11),
], ids=lambda v: v if isinstance(v, str) else None)
def test_cached_frame_is_the_oracles_frame(oracle_mod, cuda_device, cache, name, P, W, H, deg, spread, seed):
    cull_cache, before = cache
    rs = scenes.camera(W, H, pose_index=seed % 24)._replace(sh_degree=deg)
    kw = dict(spread=spread) if spread else {}
    if name == "mostly_visible":
        kw["smax"] = 3.0
    sc = scenes.blob_scene(P, seed, deg, **kw)
    fr = _frame(oracle_mod, rs, sc)
    args, out = G.run_forward(rs, sc, cuda_device, for_backward=False)
    assert cull_cache.stats["builds"] == before["builds"] + 1
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    # the same tensors, another pose: a hit, and still the oracle's frame
    rs2 = scenes.camera(W, H, pose_index=(seed + 9) % 24)._replace(sh_degree=deg)
    fr2 = _frame(oracle_mod, rs2, sc)
    from gaussiancity_amd import ext
    a2 = (rs2.bg.to(cuda_device),) + args[1:8] + (rs2.view_matrix.to(cuda_device), rs2.proj_matrix.to(cuda_device)) + \
        args[10:16] + (rs2.campos.to(cuda_device),) + args[17:]
    out2 = ext.rasterize_gaussians(*a2, _for_backward=False)
    torch.cuda.synchronize()
    assert cull_cache.stats["builds"] == before["builds"] + 1 and cull_cache.stats["hits"] == before["hits"] + 1
    _check_forward(fr2, G.decode(P, W, H, out2), P, True)


@pytest.mark.parametrize("variant", ["wild_quats_scales", "precomp_cov_indefinite", "scale_modifier"])
def test_cached_cull_never_changes_a_decision(oracle_mod, cuda_device, cache, variant):
    """test_precull_never_changes_a_decision's adversarial inputs (unnormalised quaternions, scales over six decades,
    indefinite caller-supplied covariances, scale_modifier != 1, NaN / inf rows) through the cached cull."""
    cull_cache, before = cache
    P, W, H = 20000, 208, 120
    rng = np.random.default_rng({"wild_quats_scales": 31, "precomp_cov_indefinite": 33, "scale_modifier": 34}[variant])
    rs = scenes.camera(W, H, pose_index=7)._replace(sh_degree=0)
    sc = scenes.blob_scene(P, 40, 0, spread=120.0)
    use_cov, cov = False, None
    if variant == "wild_quats_scales":
        sc["rotations"] = (sc["rotations"] * rng.uniform(0.05, 3.0, (P, 1))).astype(np.float32)
        sc["scales"] = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (P, 3))).astype(np.float32)
        sc["scales"][::7] *= -1.0
    elif variant == "precomp_cov_indefinite":
        use_cov = True
        A = rng.normal(size=(P, 3, 3)) * np.exp(rng.uniform(-3, 3, (P, 1, 1)))
        S = A + np.transpose(A, (0, 2, 1))
        cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    else:
        rs = rs._replace(scale_modifier=3.7)
    sc["means3D"][5] = np.nan
    sc["means3D"][6, 0] = np.inf
    if not use_cov:
        sc["scales"][8] = np.nan
        sc["rotations"][9] = np.inf
    fr = _frame(oracle_mod, rs, sc, use_sh=False, cov3D=cov)
    args, out = G.run_forward(rs, sc, cuda_device, use_sh=False, use_cov3d=use_cov, cov3D=cov, for_backward=False)
    assert cull_cache.stats["builds"] == before["builds"] + 1
    assert out[0] == fr.R
    np.testing.assert_array_equal(out[2].cpu().numpy(), fr.radii)
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    assert (fr.radii > 0).sum() > 100


def test_points14_rows_and_an_in_place_edit(oracle_mod, cuda_device, cache):
    """GaussianCity's own call shape: one [N,14] tensor read in place (row stride 14).  The cache is keyed on that
    tensor's version: after an in-place edit of its scale columns the next frame rebuilds it and is the EDITED scene's
    frame -- a stale cache would have culled by the old sizes."""
    from gaussiancity_amd import ext
    cull_cache, before = cache
    P, W, H = 50000, 256, 160
    rs = scenes.camera(W, H, pose_index=3)._replace(sh_degree=0)
    sc = scenes.blob_scene(P, 91, 0, spread=300.0, smax=1.5)
    pts = np.concatenate([sc["means3D"], sc["opacities"].reshape(P, 1), sc["scales"], sc["rotations"],
                          sc["colors_precomp"]], 1).astype(np.float32)
    points = torch.from_numpy(pts).to(cuda_device)

    def render():
        o = ext.rasterize_points14(points, rs.bg.to(cuda_device), rs.scale_modifier, rs.view_matrix.to(cuda_device),
                                   rs.proj_matrix.to(cuda_device), rs.tanfovx, rs.tanfovy, H, W, rs.campos.to(cuda_device))
        torch.cuda.synchronize()
        return o

    def expect(scales):
        s2 = dict(sc, scales=scales)
        return _frame(oracle_mod, rs, s2, use_sh=False)

    def same(o, fr):
        assert int(o[0]) == fr.R
        np.testing.assert_array_equal(o[2].cpu().numpy(), fr.radii)
        assert np.array_equal(o[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))

    fr = expect(sc["scales"])
    same(render(), fr)
    same(render(), fr)
    assert cull_cache.stats["builds"] == before["builds"] + 1 and cull_cache.stats["hits"] == before["hits"] + 1
    points[:, 4:7] *= 15.0  # in place: Gaussians far outside the frustum now reach into the image
    fr_big = expect((sc["scales"] * np.float32(15.0)).astype(np.float32))
    assert (fr_big.radii > 0).sum() > (fr.radii > 0).sum() + 50  # the edit matters to the cull
    same(render(), fr_big)
    assert cull_cache.stats["builds"] == before["builds"] + 2


def test_training_frames_and_the_off_switch_do_not_use_it(oracle_mod, cuda_device, cache):
    cull_cache, before = cache
    P, W, H = 3000, 200, 150
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 11, 1)
    fr = _frame(oracle_mod, rs, sc)
    args, out = G.run_forward(rs, sc, cuda_device, for_backward=True)
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    assert cull_cache.stats["builds"] == before["builds"] and cull_cache.stats["hits"] == before["hits"]
    cull_cache.enable(False)
    args, out = G.run_forward(rs, sc, cuda_device, for_backward=False)
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    assert cull_cache.stats["builds"] == before["builds"]


def test_c_abi_build_and_use(oracle_mod, cuda_device):
    """gcr_build_cull_cache + gcr_gaussians.cull_cache through the C ABI alone: the records are (mean, rho) with rho >=
    the largest eigenvalue of the world-space covariance; a misaligned cache pointer is refused."""
    from gaussiancity_amd import _native as N
    L = N.lib()
    P = 5000
    sc = scenes.blob_scene(P, 23, 0, spread=50.0)
    sc["rotations"] = (sc["rotations"] * np.random.default_rng(1).uniform(0.3, 2.0, (P, 1))).astype(np.float32)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda_device) for k, v in sc.items() if isinstance(v, np.ndarray)}
    g = N.Gaussians(P, 0, t["means3D"].data_ptr(), t["opacities"].data_ptr(), None, t["colors_precomp"].data_ptr(),
                    t["scales"].data_ptr(), t["rotations"].data_ptr(), None)
    nbytes = L.gcr_cull_cache_bytes(P)
    assert nbytes == (P * 16 + 127) // 128 * 128 + P * 32
    out = torch.empty((nbytes,), dtype=torch.uint8, device=cuda_device)
    N.check(L.gcr_build_cull_cache(C.byref(g), 1.5, out.data_ptr(), None), "gcr_build_cull_cache")
    torch.cuda.synchronize()
    raw = out.cpu().numpy()
    rec = raw[:P * 16].view(np.float32).reshape(P, 4)
    assert np.array_equal(rec[:, :3].view(np.uint32), sc["means3D"].view(np.uint32))
    shape = raw[nbytes - P * 32:].view(np.float32).reshape(P, 8)   # (scales, opacity, rotation): copies
    want = np.concatenate([sc["scales"], sc["opacities"].reshape(P, 1), sc["rotations"]], 1)
    assert np.array_equal(shape.view(np.uint32), want.view(np.uint32))
    # the exact spectral radius of Sigma = R S S^T R^T with the reference's unnormalised R(q), in float64
    q = sc["rotations"].astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
                  np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
                  np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    M = R * (1.5 * sc["scales"].astype(np.float64))[:, None, :]
    lam = np.linalg.eigvalsh(M @ np.transpose(M, (0, 2, 1)))[:, -1]
    assert (rec[:, 3].astype(np.float64) >= lam).all()
    assert L.gcr_build_cull_cache(C.byref(g), 1.5, out.data_ptr() + 16, None) == -1   # GCR_ERR_INVALID_ARGUMENT
    assert b"128-byte" in L.gcr_last_error()


def test_inference_loop_of_a_static_scene(cuda_device):
    """frames.InferenceLoop(static_scene=True): frames rotate over three streams and all of them read the one cache the
    first frame built -- byte for byte the frames of the stateless loop, float route and uint8 route."""
    from gaussiancity_amd import cull_cache, frames, synth
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    W, H = 160, 96
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=cuda_device)
    sc = scenes.blob_scene(30000, 91, 0, spread=120.0)
    rot_xyzw = sc["rotations"][:, [1, 2, 3, 0]]
    pts = torch.from_numpy(np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], rot_xyzw,
                                           sc["colors_precomp"]], axis=1).astype(np.float32)).to(cuda_device)
    poses = synth.orbit_poses(9, 60.0, 50.0)
    cull_cache.invalidate()
    assert not cull_cache.enabled()
    before = dict(cull_cache.stats)
    with torch.no_grad():
        want = frames.InferenceLoop(wr, device=cuda_device).run(pts, poses)
        assert cull_cache.stats == before                       # off unless asked for
        got = frames.InferenceLoop(wr, device=cuda_device, static_scene=True).run(pts, poses)
        u8 = frames.InferenceLoop(wr, device=cuda_device, static_scene=True,
                                  render_uint8_fn=lambda p, c, q: wr(p, c, q, as_uint8=True)).run(pts, poses)
    assert cull_cache.stats["builds"] == before["builds"] + 1 and cull_cache.stats["hits"] >= before["hits"] + 2 * len(poses) - 1
    assert not cull_cache.enabled()                             # the switch was the loop's, for the loop
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    assert all(np.array_equal(a, b) for a, b in zip(u8, want))
    assert any(f.any() for f in got)
    cull_cache.invalidate()


def test_cached_kernel_with_many_groups_and_passes_per_block(cuda_device):
    """The cached K1's mid-chunk machinery: with the grid forced down to 12 blocks (GCR_K1_BLOCKS, read once per process ->
    a subprocess) every block streams ~3400 Gaussians = three groups of five iterations, most of them candidates: passes
    run at group ends, leftovers move to the front of the queue, a partial pass ends the chunk.  Full state against the
    oracle for a mostly visible and a half-culled scene."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import gpu_util as G, scenes
from gaussiancity_amd import cull_cache
from oracle import oracle as O
from test_gpu_parity import _frame, _check_forward
cull_cache.enable(True)
dev = torch.device("cuda:0")
for spread, smax, floor in ((25.0, 3.0, 30000), (70.0, 3.0, 4000)):
    P, W, H = 40000, 256, 160
    rs = scenes.camera(W, H, pose_index=4)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 57, 1, spread=spread, smax=smax)
    fr = _frame(O, rs, sc)
    assert floor < (fr.radii > 0).sum() < P, (fr.radii > 0).sum()
    b = cull_cache.stats["builds"]
    args, out = G.run_forward(rs, sc, dev, for_backward=False)
    assert cull_cache.stats["builds"] == b + 1
    _check_forward(fr, G.decode(P, W, H, out), P, True)
print("CACHED_MANY_PASSES_OK")
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, GCR_K1_BLOCKS="12")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CACHED_MANY_PASSES_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
