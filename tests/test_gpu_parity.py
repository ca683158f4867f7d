"""-m gpu: HIP path (through the native-module surface -> C ABI -> gfx950 kernels) vs the CPU
oracle on identical seeded inputs.  Bars:
  forward:  radii, num_rendered, sorted list, tile ranges, n_contrib EXACT; per-Gaussian
            state, final_T and the image BIT-EXACT (numerics contract gcr-fp32-v1);
  backward: every gradient tensor max|d| <= 1e-4 * max(1, max|ref|)  (fp32 atomics reorder
            the sums, so bit-exactness is not defined there).
"""
import numpy as np
import pytest
import torch

import gpu_util as G
import scenes

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4
ALL_GRADS = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dsh"]


def _frame(O, rs, sc, use_sh=True, cov3D=None):
    kw = scenes.settings_kwargs(rs)
    extra = dict(shs=sc["shs"]) if use_sh else dict(colors_precomp=sc["colors_precomp"])
    if cov3D is not None:
        extra["cov3D_precomp"] = cov3D
    else:
        extra.update(scales=sc["scales"], rotations=sc["rotations"])
    return O.Frame(**kw, means3D=sc["means3D"], opacities=sc["opacities"], **extra)


def _check_forward(fr, d, P, use_sh, has_cov3d_state=True):
    vis = fr.radii > 0
    assert d["R"] == fr.R
    np.testing.assert_array_equal(d["radii"], fr.radii)
    rect = d["rect"][vis]
    area = ((rect[:, 0] >> 16) - (rect[:, 0] & 0xffff)) * ((rect[:, 1] >> 16) - (rect[:, 1] & 0xffff))
    np.testing.assert_array_equal(area, fr.tiles_touched[:P][vis])
    for name in ("means2D", "conic_opacity", "depths"):
        a, b = d[name][vis], getattr(fr, name)[:P][vis]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name + " not bit-exact"
    if use_sh:
        assert np.array_equal(d["rgb"][vis].view(np.uint32), fr.rgb[:P][vis].view(np.uint32))
        cl = fr.clamped[:P]
        mask = (cl[:, 0] | (cl[:, 1] << 1) | (cl[:, 2] << 2)).astype(np.uint8)
        np.testing.assert_array_equal(d["clamped"][vis], mask[vis])
    # (`has_cov3d_state`: until late in round 6 the geometry buffer held every survivor's covariance and it was compared
    # here; the forward no longer stores it -- K8 derives it again, and the gradients that depend on it are what is checked)
    if fr.R > 0:
        G.assert_point_list(d, fr)
    np.testing.assert_array_equal(d["ranges"], fr.ranges)
    np.testing.assert_array_equal(d["n_contrib"], fr.n_contrib)
    assert np.array_equal(d["final_T"].view(np.uint32), fr.final_T.view(np.uint32)), "final_T not bit-exact"
    assert np.array_equal(d["out_color"].view(np.uint32), fr.out_color.view(np.uint32)), (
        "image not bit-exact, max diff %g" % np.abs(d["out_color"] - fr.out_color).max())


def _check_grads(gref, ggpu, names):
    for n in names:
        ref, got = gref[n], ggpu[n]
        assert ref.shape == got.shape, n
        tol = GRAD_TOL * max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(ref - got).max())
        assert err <= tol, "%s: max|d| %.3e > %.3e" % (n, err, tol)


CASES = [
    # name, P, W, H, sh_degree(None = precomputed colour), bg, seed
    ("sh3_ragged", 3000, 200, 150, 3, (0.0, 0.0, 0.0), 11),
    ("sh0_bg", 2000, 128, 128, 0, (0.3, 0.1, 0.7), 12),
    ("sh1", 1500, 96, 80, 1, (0.0, 0.0, 0.0), 13),
    ("sh2", 1500, 96, 80, 2, (1.0, 1.0, 1.0), 14),
    ("precomp_colour", 4000, 256, 144, None, (0.0, 0.0, 0.0), 15),
    ("tiny_image", 300, 17, 9, 3, (0.5, 0.5, 0.5), 16),
    ("dense_overdraw", 6000, 64, 64, 3, (0.0, 0.0, 0.0), 17),
]


@pytest.mark.parametrize("name,P,W,H,deg,bg,seed", CASES, ids=[c[0] for c in CASES])
def test_forward_backward_parity(oracle_mod, cuda_device, name, P, W, H, deg, bg, seed):
    use_sh = deg is not None
    rs = scenes.camera(W, H, pose_index=seed % 24)._replace(
        sh_degree=deg if use_sh else 0, bg=torch.tensor(bg, dtype=torch.float32))
    sc = scenes.blob_scene(P, seed, deg if use_sh else 0,
                           smax=12.0 if name == "dense_overdraw" else 6.0)
    fr = _frame(oracle_mod, rs, sc, use_sh)
    args, out = G.run_forward(rs, sc, cuda_device, use_sh=use_sh)
    d = G.decode(P, W, H, out)
    _check_forward(fr, d, P, use_sh)
    dpix = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
    gref = fr.backward(dpix)
    ggpu = G.run_backward(args, out, dpix, cuda_device)
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot"]
    if use_sh:
        names.append("dL_dsh")
    _check_grads(gref, ggpu, names)


@pytest.mark.parametrize("hint,piece", [(False, 128), (True, 64), (True, 100), (True, 128), (True, 256)],
                         ids=["inference_frame", "piece64", "piece100", "piece128", "piece256"])
def test_backward_pieces(oracle_mod, cuda_device, hint, piece):
    """The backward blend works on (tile, piece) items and starts a piece from the forward's checkpoint
    (gcr_internal.h "backward pieces"): whatever the piece size -- and whether or not the forward was announced as a
    training frame -- the forward state stays bit-exact and every gradient within tolerance.  Dense overdraw: tile lists
    of several hundred entries, i.e. up to a dozen pieces per tile, pixels that saturate in the middle of a piece,
    and tiles whose walk ends before the last piece."""
    from gaussiancity_amd import _native as N
    P, W, H, seed = 9000, 96, 80, 41
    rs = scenes.camera(W, H, pose_index=seed % 24)._replace(sh_degree=1, bg=torch.tensor((0.2, 0.0, 0.4)))
    sc = scenes.blob_scene(P, seed, 1, smax=14.0)
    fr = _frame(oracle_mod, rs, sc, True)
    assert (fr.ranges[:, 1] - fr.ranges[:, 0]).max() > 3 * 256 and fr.n_contrib.max() > 256
    prev = N.set_option("bwd_piece", piece)
    try:
        args, out = G.run_forward(rs, sc, cuda_device, for_backward=hint)
        _check_forward(fr, G.decode(P, W, H, out), P, True)
        dpix = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
        _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                     ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dsh"])
        # the option may change between the forward and the backward: the backward takes the forward's piece size
        N.set_option("bwd_piece", 64 if piece != 64 else 256)
        _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device), ["dL_dmean3D", "dL_dopacity", "dL_dsh"])
    finally:
        N.set_option("bwd_piece", prev)


@pytest.mark.parametrize("piece", [64, 128, 223])
@pytest.mark.parametrize("name,P,W,H,smax,seed", [("ragged", 3000, 200, 150, 6.0, 11), ("dense", 9000, 96, 80, 14.0, 41)])
def test_both_backward_blend_kernels_hold_the_gradient_bar(oracle_mod, cuda_device, name, P, W, H, smax, seed, piece):
    """The backward blend exists twice (include/gcr.h, option "bwd_wave_units"): one workgroup per (tile, piece) work item
    (the default since round 5) and one wave per (work item, quadrant) (round 4's kernel, the deterministic mode's).  Same
    forward state, same pieces: both hold every gradient tensor to 1e-4 * max against the oracle, at every piece size --
    and to each other well inside that."""
    from gaussiancity_amd import ext
    rs = scenes.camera(W, H, pose_index=seed % 24)._replace(sh_degree=1, bg=torch.tensor((0.2, 0.0, 0.4)))
    sc = scenes.blob_scene(P, seed, 1, smax=smax)
    fr = _frame(oracle_mod, rs, sc, True)
    dpix = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
    gref = fr.backward(dpix)
    got = {}
    for wave_units in (0, 1):
        with ext.options(bwd_wave_units=wave_units, bwd_piece=piece):
            args, out = G.run_forward(rs, sc, cuda_device, for_backward=True)
            _check_forward(fr, G.decode(P, W, H, out), P, True)
            got[wave_units] = G.run_backward(args, out, dpix, cuda_device)
        _check_grads(gref, got[wave_units], ALL_GRADS)
    for n in ALL_GRADS:
        err = float(np.abs(got[0][n] - got[1][n]).max())
        assert err <= GRAD_TOL * max(1.0, float(np.abs(gref[n]).max())), (n, err)


def test_deterministic_backward_mode(oracle_mod, cuda_device):
    """Option "deterministic_backward" (a deterministic debug mode next to the atomics): the
    per-Gaussian blend gradients are accumulated as 64-bit fixed-point sums, so the order in which the tiles' waves
    reach the atomic units no longer matters -- repeated runs are BIT-identical, through both native entry points, and
    still within tolerance of the oracle.  (The default fp32 atomics are not asserted to differ: they usually do, in
    the last bits, but need not.)"""
    from gaussiancity_amd import _native as N, ext
    P, W, H, seed = 9000, 96, 80, 43
    rs = scenes.camera(W, H, pose_index=seed % 24)._replace(sh_degree=1, bg=torch.tensor((0.1, 0.3, 0.0)))
    sc = scenes.blob_scene(P, seed, 1, smax=14.0)   # every Gaussian lands in many tiles: many waves per record
    fr = _frame(oracle_mod, rs, sc, True)
    dpix = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dsh"]
    prev = N.set_option("deterministic_backward", 1)
    try:
        assert N.lib().gcr_grad_record_floats() == 32
        args, out = G.run_forward(rs, sc, cuda_device, for_backward=True)
        runs = [G.run_backward(args, out, dpix, cuda_device) for _ in range(4)]
        for r in runs[1:]:
            for n in names:
                assert np.array_equal(r[n].view(np.uint32), runs[0][n].view(np.uint32)), n + " differs between runs"
        _check_grads(fr.backward(dpix), runs[0], names)
        # the [N,14] entry point shares the mode
        pts = torch.from_numpy(np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], sc["rotations"],
                                               np.clip(sc["shs"][:, 0, :], 0, 1)], axis=1).astype(np.float32)).to(cuda_device)
        cam = (rs.bg.to(cuda_device), rs.scale_modifier, rs.view_matrix.to(cuda_device), rs.proj_matrix.to(cuda_device),
               rs.tanfovx, rs.tanfovy)
        R, _, radii, geom, binning, img = ext.rasterize_points14(pts, *cam, H, W, rs.campos.to(cuda_device), for_backward=True)
        g14 = [ext.rasterize_points14_backward(pts, radii, *cam, torch.from_numpy(dpix).to(cuda_device),
                                               rs.campos.to(cuda_device), geom, R, binning, img, H, W).cpu().numpy()
               for _ in range(3)]
        assert np.array_equal(g14[0].view(np.uint32), g14[1].view(np.uint32)) and np.array_equal(
            g14[0].view(np.uint32), g14[2].view(np.uint32)) and np.abs(g14[0]).max() > 0
    finally:
        N.set_option("deterministic_backward", prev)
    assert N.lib().gcr_grad_record_floats() == 16


def test_precomputed_cov3d(oracle_mod, cuda_device):
    P, W, H = 2500, 160, 96
    rs = scenes.camera(W, H)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 21, 2)
    # take the covariance the oracle derives from scale/rotation and feed it back precomputed
    cov = _frame(oracle_mod, rs, sc).cov3D[:P].copy()
    fr = _frame(oracle_mod, rs, sc, cov3D=cov)
    args, out = G.run_forward(rs, sc, cuda_device, use_cov3d=True, cov3D=cov)
    d = G.decode(P, W, H, out)
    _check_forward(fr, d, P, True, has_cov3d_state=False)
    dpix = np.random.default_rng(3).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                 ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh"])


def test_all_culled_and_empty(oracle_mod, cuda_device):
    """Everything behind the camera -> R == 0, image == background; and P == 0."""
    W, H = 64, 48
    rs = scenes.camera(W, H)._replace(bg=torch.tensor([0.25, 0.5, 0.75]))
    sc = scenes.blob_scene(200, 5, 0)
    sc["means3D"][:, 2] += 10000.0  # far above the camera, behind the near plane
    fr = _frame(oracle_mod, rs, sc, use_sh=False)
    assert fr.R == 0
    args, out = G.run_forward(rs, sc, cuda_device, use_sh=False)
    assert out[0] == 0
    img = out[1].cpu().numpy()
    assert np.array_equal(img, fr.out_color)
    assert np.all(out[2].cpu().numpy() == 0)
    g = G.run_backward(args, out, np.ones((3, H, W), np.float32), cuda_device)
    assert all(np.all(v == 0) for v in g.values())
    # P == 0 short-circuit (dgr/rasterize_points.cu:71)
    sc0 = {k: (v[:0] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    _, out0 = G.run_forward(rs, sc0, cuda_device, use_sh=False)
    assert out0[0] == 0 and out0[1].shape == (3, H, W) and float(out0[1].abs().max()) == 0.0


def test_mark_visible(oracle_mod, cuda_device):
    from gaussiancity_amd import ext
    rs = scenes.camera(64, 64)
    sc = scenes.blob_scene(5000, 9, 0, spread=400.0)
    ref = oracle_mod.mark_visible(sc["means3D"], rs.view_matrix.numpy(), rs.proj_matrix.numpy())
    got = ext.mark_visible(G.to_dev(sc["means3D"], cuda_device), rs.view_matrix.to(cuda_device),
                           rs.proj_matrix.to(cuda_device)).cpu().numpy()
    assert got.dtype == np.bool_ and 0 < ref.sum() < len(ref)
    np.testing.assert_array_equal(got, ref)


def test_sort_is_stable_on_depth_ties(oracle_mod, cuda_device):
    """Many Gaussians at IDENTICAL depth in the same tiles: the blend order must be ascending
    Gaussian index (stable radix sort, cr/rasterizer_impl.cu:255-260)."""
    P, W, H = 1200, 80, 64
    rs = scenes.camera(W, H)
    sc = scenes.blob_scene(P, 31, 0)
    sc["means3D"][:] = sc["means3D"][0]          # same point -> same depth, same tiles
    sc["means3D"][::3, 0] += 3.0                  # a second and third depth class, interleaved
    sc["means3D"][1::3, 0] -= 3.0
    fr = _frame(oracle_mod, rs, sc, use_sh=False)
    args, out = G.run_forward(rs, sc, cuda_device, use_sh=False)
    d = G.decode(P, W, H, out)
    assert fr.R > 1000
    G.assert_point_list(d, fr)
    assert np.array_equal(d["out_color"].view(np.uint32), fr.out_color.view(np.uint32))


@pytest.mark.parametrize("force_radix,force_cursor", [(0, 0), (1, 0), (0, 1)],
                         ids=["lds_tile_table", "global_radix", "global_cursor"])
def test_both_binning_paths(oracle_mod, cuda_device, force_radix, force_cursor):
    """LDS tile-table counting sort, the global radix fallback and the global-cursor variant
    all give the same list."""
    from gaussiancity_amd import _native as N
    P, W, H = 5000, 208, 160
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 51, 1, smax=9.0)
    fr = _frame(oracle_mod, rs, sc)
    N.set_option("force_radix", force_radix)
    N.set_option("force_global_cursor", force_cursor)
    try:
        args, out = G.run_forward(rs, sc, cuda_device)
        d = G.decode(P, W, H, out)   # (while the options still say which path wrote the state)
    finally:
        N.set_option("force_radix", 0)
        N.set_option("force_global_cursor", 0)
    _check_forward(fr, d, P, True)
    dpix = np.random.default_rng(2).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                 ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])


@pytest.mark.parametrize("in_blend", [1, 0], ids=["sort_in_blend", "separate_sort_kernel"])
@pytest.mark.parametrize("P,longest", [(900, 221), (1200, 297)], ids=["lists_within_a_chunk", "some_lists_beyond_256"])
def test_tile_sort_inside_the_blend_and_as_a_kernel(oracle_mod, cuda_device, in_blend, P, longest):
    """K4 has two homes: the per-tile sort kernel (default), or -- option sort_in_blend, for short lists -- the
    forward blend's own workgroups (LDS rank sort up to one 256-entry chunk; a somewhat longer list, the 297-entry
    case, through the same workgroup's global-rank path).  Same sorted list, same image; staged first call,
    speculative second call."""
    from gaussiancity_amd import _native as N
    from gaussiancity_amd import ext
    W = H = 64
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 88, 1, smax=4.0)
    fr = _frame(oracle_mod, rs, sc)
    assert int((fr.ranges[:, 1].astype(np.int64) - fr.ranges[:, 0]).max()) == longest
    ext._capacity_hint.pop((cuda_device.index, P, W, H), None)
    N.set_option("sort_in_blend", in_blend)
    try:
        for _ in range(2):
            args, out = G.run_forward(rs, sc, cuda_device)
            _check_forward(fr, G.decode(P, W, H, out), P, True)
        dpix = np.random.default_rng(9).normal(size=(3, H, W)).astype(np.float32)
        _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                     ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])
    finally:
        N.set_option("sort_in_blend", 0)


@pytest.mark.parametrize("split", [0, 1], ids=["fused_k1", "two_kernel_k1"])
def test_both_preprocess_variants(oracle_mod, cuda_device, split):
    """K1 as one fused kernel (default: workgroups alternate between streaming their chunk and processing the
    candidates waiting in LDS, several passes per chunk here) and as two kernels (option split_preprocess): same
    state, bit for bit.  40 000 Gaussians around the view so that blocks collect more than one pass of candidates."""
    from gaussiancity_amd import _native as N
    P, W, H = 40000, 256, 160
    rs = scenes.camera(W, H, pose_index=4)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 53, 2, spread=90.0)
    fr = _frame(oracle_mod, rs, sc)
    assert (fr.radii > 0).sum() > 3000
    N.set_option("split_preprocess", split)
    try:
        args, out = G.run_forward(rs, sc, cuda_device)
        _check_forward(fr, G.decode(P, W, H, out), P, True)
        dpix = np.random.default_rng(6).normal(size=(3, H, W)).astype(np.float32)
        _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                     ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])
    finally:
        N.set_option("split_preprocess", 0)


def test_backward_fills_unaligned_uninitialised_outputs(oracle_mod, cuda_device, monkeypatch):
    """gcr_backward writes every element of every gradient output itself (include/gcr.h): here the eight arrays are
    separate NaN-filled allocations whose base addresses are only 4-, 8- or 12-byte aligned where the ABI allows it
    (the zero fill's scalar head / dwordx4 body / scalar tail), with NaN guard words on both sides that must survive.
    dL_dconic and dL_drotations keep their float4 alignment.  Also the frame in which nothing is rendered (the fill
    then runs as its own kernel)."""
    from gaussiancity_amd import _native as N, ext
    P, W, H = 5000, 112, 80
    guards = []

    def buffers(P_, M_, device):
        shapes = ((P_, 3), (P_, 3), (P_, 3), (P_, N.GRAD_REC_FLOATS), (P_, 1), (P_, 6), (P_, M_, 3), (P_, 3), (P_, 4))
        shift = (1, 2, 3, 0, 1, 3, 0, 2, 0)  # floats past a 256-byte boundary
        out = []
        for sh, k in zip(shapes, shift):
            n = int(torch.Size(sh).numel())
            raw = torch.full((n + 128,), float("nan"), dtype=torch.float32, device=device)
            guards.append((raw, 64 + k, n))
            out.append(raw[64 + k:64 + k + n].view(sh))
        return out

    monkeypatch.setattr(ext, "_gradient_buffers", buffers)
    for far in (False, True):
        guards.clear()
        rs = scenes.camera(W, H, pose_index=3)._replace(sh_degree=2)
        sc = scenes.blob_scene(P, 33, 2)
        if far:
            sc["means3D"][:, 2] += 10000.0  # behind the camera: R == 0
        fr = _frame(oracle_mod, rs, sc)
        args, out = G.run_forward(rs, sc, cuda_device)
        assert (out[0] == 0) == far
        dpix = np.random.default_rng(4).normal(size=(3, H, W)).astype(np.float32)
        ggpu = G.run_backward(args, out, dpix, cuda_device)
        _check_grads(fr.backward(dpix), ggpu, ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D",
                                               "dL_dscale", "dL_drot", "dL_dsh"])
        for i, (raw, lo, n) in enumerate(guards):
            r = raw.cpu().numpy()
            assert np.isnan(r[:lo]).all() and np.isnan(r[lo + n:]).all(), "output %d: guard words overwritten" % i
            if i != 3:  # dL_dconic is scratch
                assert not np.isnan(r[lo:lo + n]).any(), "output %d: element left unwritten" % i


def test_fused_preprocess_with_many_passes_per_block(cuda_device):
    """The fused K1 kernel's mid-chunk processing passes: with the grid forced down to 12 blocks (GCR_K1_BLOCKS, read
    once per process -> a subprocess) every block streams ~3400 mostly visible Gaussians, so 256 candidates are
    waiting several times per chunk, leftovers are moved to the front of the LDS buffer, and a final partial pass
    runs.  Full state against the oracle, fused and two-kernel."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import gpu_util as G, scenes
from gaussiancity_amd import _native as N
from oracle import oracle as O
from test_gpu_parity import _frame, _check_forward, _check_grads
P, W, H = 40000, 256, 160
rs = scenes.camera(W, H, pose_index=4)._replace(sh_degree=1)
sc = scenes.blob_scene(P, 57, 1, spread=25.0, smax=3.0)
fr = _frame(O, rs, sc)
assert (fr.radii > 0).sum() > 30000, (fr.radii > 0).sum()
dev = torch.device("cuda:0")
for split in (0, 1):
    N.set_option("split_preprocess", split)
    args, out = G.run_forward(rs, sc, dev)
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    dpix = np.random.default_rng(6).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, dev), ["dL_dmean3D", "dL_dsh", "dL_dscale", "dL_dopacity"])
print("MANY_PASSES_OK")
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, GCR_K1_BLOCKS="12")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MANY_PASSES_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("lazy", [0, 1], ids=["sorted_whole", "lazy_sort"])
@pytest.mark.parametrize("P,spread,longest", [(9000, 2.0, 4096), (60000, 4.0, 3 * 4096)],
                         ids=["two_runs", "many_runs_three_merge_passes"])
def test_tile_list_longer_than_lds_capacity(oracle_mod, cuda_device, P, spread, longest, lazy):
    """Lists beyond the LDS sort capacity (4096).  Option "lazy_sort" off: those tiles are sorted by the per-tile
    long-list sort (4096-key runs + merge passes through HBM) while the other tiles stay on the LDS path -- no
    whole-frame fallback.  On (the default): they are sorted segment by segment as far as the blend walks them; the
    Gaussians here are nearly transparent, so the walk goes on through many segments.  Either way everything matches
    the oracle bit for bit.  First call: staged path (the host learns the longest list); second call: speculative
    with a long-list hint."""
    from gaussiancity_amd import _native as N
    from gaussiancity_amd import ext
    W, H = 48, 48
    rs = scenes.camera(W, H)
    sc = scenes.blob_scene(P, 61, 0, spread=spread, smin=0.5, smax=2.0, omin=0.01, omax=0.05)
    fr = _frame(oracle_mod, rs, sc, use_sh=False)
    lens = fr.ranges[:, 1].astype(np.int64) - fr.ranges[:, 0]
    assert lens.max() > longest and (lens[lens > 0] <= 4096).any()   # long AND short tiles in the same frame
    ext._capacity_hint.pop((cuda_device.index, P, W, H), None)
    prev = N.set_option("lazy_sort", lazy)
    try:
        for _ in range(2):
            args, out = G.run_forward(rs, sc, cuda_device, use_sh=False)
            d = G.decode(P, W, H, out)
            _check_forward(fr, d, P, False)
            if lazy:   # more than one segment was sorted somewhere (the blend itself sorted on)
                assert (d["tile_sorted"] > 1024).any()
            else:
                assert "tile_sorted" not in d
        dpix = np.random.default_rng(8).normal(size=(3, H, W)).astype(np.float32)
        _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                     ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dscale", "dL_drot"])
    finally:
        N.set_option("lazy_sort", prev)


@pytest.mark.parametrize("case", ["one_tile_300k", "equal_depths"])
def test_lazy_sort_when_samples_cannot_separate_the_keys(oracle_mod, cuda_device, case):
    """gcr_sort.h "lazy tile sort", the rare branches: (a) 300 000 entries in ONE tile -- the 256 samples lie more than
    a segment apart, so the segment bound is found by bisecting the key interval; (b) thousands of Gaussians at three
    distinct depths -- the keys differ in the index half only, segments end in the middle of a depth class and the
    order inside a class must still be ascending index (cr/rasterizer_impl.cu:255-260, stable sort).  Low opacities
    keep the pixels alive, so the blend sorts on through several segments.  Image, per-pixel state and the final
    prefix of every list against the oracle, gradients within tolerance."""
    if case == "one_tile_300k":
        P, W, H = 300_000, 16, 16
        rs = scenes.camera(W, H)
        sc = scenes.blob_scene(P, 97, 0, spread=0.6, smin=0.3, smax=1.5, omin=0.002, omax=0.01)
    else:
        P, W, H = 6000, 64, 48
        rs = scenes.camera(W, H)
        sc = scenes.blob_scene(P, 99, 0, smin=1.0, smax=4.0, omin=0.002, omax=0.01)
        sc["means3D"][:] = sc["means3D"][0]
        sc["means3D"][::3, 0] += 3.0
        sc["means3D"][1::3, 0] -= 3.0
    fr = _frame(oracle_mod, rs, sc, use_sh=False)
    lens = fr.ranges[:, 1].astype(np.int64) - fr.ranges[:, 0]
    assert lens.max() > (250_000 if case == "one_tile_300k" else 3000)
    assert int(fr.n_contrib.max()) > 2048          # the walk needs more than two segments somewhere
    for _ in range(2):
        args, out = G.run_forward(rs, sc, cuda_device, use_sh=False, for_backward=True)
        d = G.decode(P, W, H, out)
        _check_forward(fr, d, P, False)
        assert (d["tile_sorted"] > 2048).any()
    dpix = np.random.default_rng(9).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                 ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dscale", "dL_drot"])


@pytest.mark.parametrize("global_cursor", [0, 1], ids=["lds_tile_table", "global_cursor"])
def test_fused_forward_speculation_and_retry(oracle_mod, cuda_device, global_cursor):
    """gcr_forward enqueues the whole frame on a capacity guess: first call (no guess) goes
    through the staged path, the second speculates, a deliberately short guess must be vetoed on
    the device and retried -- all three give the oracle's bits.  With the global-cursor binning
    variant too: a vetoed speculative scatter must neither write past the capacity-sized buffer
    nor advance the cursors the retry scatters from."""
    from gaussiancity_amd import _native as N
    from gaussiancity_amd import ext
    N.set_option("force_global_cursor", global_cursor)
    try:
        _speculation_and_retry(oracle_mod, cuda_device)
    finally:
        N.set_option("force_global_cursor", 0)


def _speculation_and_retry(oracle_mod, cuda_device):
    from gaussiancity_amd import ext
    P, W, H = 3500, 176, 128
    rs = scenes.camera(W, H)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 71, 2)
    fr = _frame(oracle_mod, rs, sc)
    key = (cuda_device.index, P, W, H)
    ext._capacity_hint.pop(key, None)
    for mode in ("no_guess", "speculative", "short_guess", "speculative_again"):
        if mode == "short_guess":   # the hint is the last num_rendered seen: the capacity guess becomes 1.5 x 1 + 4096
            assert fr.R > 6000
            ext._capacity_hint[key] = (1, 64)
        args, out = G.run_forward(rs, sc, cuda_device)
        _check_forward(fr, G.decode(P, W, H, out), P, True)
        assert ext._capacity_hint[key][0] == fr.R
    dpix = np.random.default_rng(4).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device),
                 ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])


def test_autograd_api_matches_oracle(oracle_mod, cuda_device):
    """Same check through GaussianRasterizer / autograd (what generator code calls)."""
    from gaussiancity_amd import GaussianRasterizer
    P, W, H = 2000, 112, 80
    rs_cpu = scenes.camera(W, H)._replace(sh_degree=3)
    sc = scenes.blob_scene(P, 41, 3)
    fr = _frame(oracle_mod, rs_cpu, sc)
    rs = rs_cpu._replace(bg=rs_cpu.bg.to(cuda_device), view_matrix=rs_cpu.view_matrix.to(cuda_device),
                         proj_matrix=rs_cpu.proj_matrix.to(cuda_device), campos=rs_cpu.campos.to(cuda_device))
    t = {k: G.to_dev(sc[k], cuda_device).requires_grad_(True)
         for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    means2D = torch.zeros((P, 3), device=cuda_device, requires_grad=True)
    img, radii = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"],
                                        shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    assert np.array_equal(img.detach().cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
    dpix = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    (img * G.to_dev(dpix, cuda_device)).sum().backward()
    gref = fr.backward(dpix)
    got = dict(dL_dmean3D=t["means3D"].grad, dL_dmean2D=means2D.grad, dL_dopacity=t["opacities"].grad,
               dL_dsh=t["shs"].grad, dL_dscale=t["scales"].grad, dL_drot=t["rotations"].grad)
    _check_grads(gref, {k: v.cpu().numpy() for k, v in got.items()}, list(got))


@pytest.mark.parametrize("variant", ["wild_quats_scales", "border_huggers", "precomp_cov_indefinite", "scale_modifier"])
def test_precull_never_changes_a_decision(oracle_mod, cuda_device, variant):
    """K1a drops a Gaussian before the exact math only when its bound says the exact path must give
    radius 0.  Adversarial inputs for that bound: unnormalised quaternions (|q| up to 3), scales over
    six decades, centres just outside every screen border and around the near plane, indefinite
    caller-supplied covariances, scale_modifier != 1, NaN/inf rows.  radii, R and the image must
    still equal the oracle's exactly."""
    P, W, H = 20000, 208, 120
    rng = np.random.default_rng({"wild_quats_scales": 31, "border_huggers": 32,
                                 "precomp_cov_indefinite": 33, "scale_modifier": 34}[variant])
    rs = scenes.camera(W, H, pose_index=7)._replace(sh_degree=0)
    sc = scenes.blob_scene(P, 40, 0, spread=120.0)
    use_cov = False
    cov = None
    if variant == "wild_quats_scales":
        sc["rotations"] = (sc["rotations"] * rng.uniform(0.05, 3.0, (P, 1))).astype(np.float32)
        sc["scales"] = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (P, 3))).astype(np.float32)
        sc["scales"][::7] *= -1.0  # sign is irrelevant to S*S but not to a careless bound
    elif variant == "border_huggers":
        # place centres on rays through pixels just outside the image, at random depths incl. z ~ 0.2
        vm = rs.view_matrix.numpy().astype(np.float64)   # row-vector convention: p_view = p @ vm
        inv = np.linalg.inv(vm)
        u = rng.uniform(-1.6, 1.6, P) * rs.tanfovx
        v = rng.uniform(-1.6, 1.6, P) * rs.tanfovy
        edge = rng.integers(0, 4, P)
        u[edge == 0] = rs.tanfovx * rng.uniform(0.98, 1.25, (edge == 0).sum())
        u[edge == 1] = -rs.tanfovx * rng.uniform(0.98, 1.25, (edge == 1).sum())
        v[edge == 2] = rs.tanfovy * rng.uniform(0.98, 1.25, (edge == 2).sum())
        v[edge == 3] = -rs.tanfovy * rng.uniform(0.98, 1.25, (edge == 3).sum())
        z = np.exp(rng.uniform(np.log(0.15), np.log(400.0), P))
        z[::11] = 0.2 + rng.uniform(-1e-4, 1e-4, z[::11].shape)
        pv = np.stack([u * z, v * z, z, np.ones(P)], 1)
        sc["means3D"] = (pv @ inv)[:, :3].astype(np.float32)
        sc["scales"] = np.exp(rng.uniform(np.log(0.01), np.log(30.0), (P, 3))).astype(np.float32)
    elif variant == "precomp_cov_indefinite":
        use_cov = True
        A = rng.normal(size=(P, 3, 3)) * np.exp(rng.uniform(-3, 3, (P, 1, 1)))
        S = A + np.transpose(A, (0, 2, 1))  # symmetric, indefinite
        cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    else:
        rs = rs._replace(scale_modifier=3.7)
    # a few poisoned rows
    sc["means3D"][5] = np.nan
    sc["means3D"][6, 0] = np.inf
    if not use_cov:
        sc["scales"][8] = np.nan
        sc["rotations"][9] = np.inf
    fr = _frame(oracle_mod, rs, sc, use_sh=False, cov3D=cov)
    args, out = G.run_forward(rs, sc, cuda_device, use_sh=False, use_cov3d=use_cov, cov3D=cov)
    assert out[0] == fr.R
    np.testing.assert_array_equal(out[2].cpu().numpy(), fr.radii)
    img = out[1].cpu().numpy()
    assert np.array_equal(img.view(np.uint32), fr.out_color.view(np.uint32))
    assert (fr.radii > 0).sum() > 100  # the case is not vacuous


def test_full_size_properties(cuda_device):
    """BASELINE.json's headline size (C3: 5M-Gaussian S-city, 1920x1080, SH3) through size-independent
    properties -- the oracle comparison at this size is in bench.py (bit-exact image, same run as the
    timing).  Checked here: per-tile lists partition [0, R) and are depth-sorted with index tie-break;
    n_contrib within the tile's list; idempotence (bit-identical second run); affine background
    (image(bg) - image(0) == final_T * bg up to one rounding); backward linear in dL_dout."""
    from gaussiancity_amd import synth
    cfg, sc = synth.make_scene("C3")
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    rs = scenes.camera(W, H, pose_index=5, radius=512.0, altitude=640.0)._replace(sh_degree=3)
    rs = rs._replace(bg=torch.tensor([0.2, 0.4, 0.6]))
    args, out = G.run_forward(rs, sc, cuda_device)
    R = out[0]
    assert R > 500000
    L = G.N.get_layout(P, W, H, R)
    geom, binning, img = out[3], out[4], out[5]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = img[L.img_ranges:L.img_ranges + 8 * T].view(torch.int32).view(T, 2).cpu().numpy().astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    assert lens.min() >= 0 and lens.sum() == R
    order = np.argsort(ranges[:, 0], kind="stable")
    nz = order[lens[order] > 0]
    assert ranges[nz[0], 0] == 0 and ranges[nz[-1], 1] == R
    assert np.array_equal(ranges[nz[1:], 0], ranges[nz[:-1], 1])  # segments tile [0, R) without gaps
    plist = binning[L.bin_vals[L.bin_sorted]:L.bin_vals[L.bin_sorted] + 4 * R].view(torch.int32).long()
    # option "lazy_sort": a list longer than 1024 entries is in final order only as far as its `n_sorted` says (what
    # lies behind it was never written: mask it out before it is used as an index)
    final = torch.ones(R, dtype=torch.bool, device=plist.device)
    if G.lazy_sort_active():
        ns = img[L.img_tile_lazy:L.img_tile_lazy + 16 * T].view(torch.int32).view(T, 4)[:, 0].long()
        lens_t = torch.from_numpy(lens).to(plist.device)
        assert bool((ns <= lens_t).all()) and bool((ns[lens_t <= 1024] == lens_t[lens_t <= 1024]).all())
        order_t = torch.from_numpy(order).to(plist.device)
        tile_of = torch.repeat_interleave(order_t, lens_t[order_t])   # tile of every list position
        final = torch.arange(R, device=plist.device) - torch.from_numpy(ranges[:, 0]).to(plist.device)[tile_of] < ns[tile_of]
        plist = torch.where(final, plist, plist[0])
    depth = geom[L.geom_rec:L.geom_rec + P * 64].view(torch.float32).view(P, 16)[:, 9]
    dkey = depth[plist].view(torch.int32).long()  # positive floats: the bit pattern orders like the value
    key = (dkey << 32) | plist
    seg_start = torch.zeros(R, dtype=torch.bool, device=key.device)
    seg_start[torch.from_numpy(ranges[nz, 0]).to(key.device)] = True
    assert bool(((key[1:] > key[:-1]) | seg_start[1:] | ~final[1:]).all()), "a tile list is not (depth, index)-sorted"
    radii = out[2]
    assert bool((radii[plist] > 0).all())
    # n_contrib never exceeds the tile's list length
    nc = img[L.img_n_contrib:L.img_n_contrib + 4 * W * H].view(torch.int32).view(H, W)
    gx = (W + 15) // 16
    ty, tx = torch.meshgrid(torch.arange(H, device=nc.device) // 16, torch.arange(W, device=nc.device) // 16, indexing="ij")
    tl = torch.from_numpy(lens).to(nc.device)[ty * gx + tx]
    assert bool((nc <= tl).all()) and bool((nc >= 0).all())
    # idempotence
    _, out2 = G.run_forward(rs, sc, cuda_device)
    assert out2[0] == R and torch.equal(out2[1].view(torch.int32), out[1].view(torch.int32))
    assert torch.equal(out2[2], out[2])
    # affine background: C(bg) = C(0) + final_T * bg
    final_T = img[L.img_final_T:L.img_final_T + 4 * W * H].view(torch.float32).view(H, W)
    rs0 = rs._replace(bg=torch.zeros(3))
    _, out0 = G.run_forward(rs0, sc, cuda_device)
    want = out0[1] + final_T[None] * rs.bg.to(cuda_device)[:, None, None]
    assert float((out[1] - want).abs().max()) <= 1e-6
    # backward is linear in dL_dout (compare g(2*d) with 2*g(d); atomics reorder sums -> tolerance)
    rng = np.random.default_rng(77)
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    g1 = G.run_backward(args, out, dpix, cuda_device)
    g2 = G.run_backward(args, out, 2.0 * dpix, cuda_device)
    for n in g1:
        ref = 2.0 * g1[n]
        tol = GRAD_TOL * max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(g2[n] - ref).max()) <= tol, n
    vis = (radii > 0).cpu().numpy()
    assert not np.any(g1["dL_dmean3D"][~vis]) and not np.any(g1["dL_dsh"][~vis])


def test_inference_loop_matches_sequential_rendering(cuda_device):
    """frames.InferenceLoop (side streams, ring of pinned buffers) returns exactly the frames a plain
    sequential loop over the wrapper produces (scripts/inference.py:655-667)."""
    from gaussiancity_amd import frames, synth
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    W, H = 160, 96
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=cuda_device)
    sc = scenes.blob_scene(3000, 91, 0)
    rot_xyzw = sc["rotations"][:, [1, 2, 3, 0]]
    pts = torch.from_numpy(np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], rot_xyzw,
                                           sc["colors_precomp"]], axis=1).astype(np.float32)).to(cuda_device)
    poses = synth.orbit_poses(7, 60.0, 50.0)
    want = [frames.InferenceLoop.to_uint8_hwc(wr(pts, p, q)).cpu().numpy() for p, q in poses]
    got = frames.InferenceLoop(wr, device=cuda_device).run(pts, poses)
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    assert any(f.any() for f in got)


def test_callback_entry_point_matches_staged_path(oracle_mod, cuda_device):
    """gcr_rasterize_forward: the reference's std::function<char*(size_t)> resize contract as C callbacks
    (cr/rasterizer.h:25-27) -- the binding INTEGRATION.md section C shows.  Buffers are grown through the
    callbacks; image, radii and R equal the oracle's, and the buffers drive gcr_backward."""
    import ctypes as C
    from gaussiancity_amd import _native as N
    from gaussiancity_amd import ext
    P, W, H = 3500, 176, 112
    rs = scenes.camera(W, H, pose_index=9)._replace(sh_degree=2, bg=torch.tensor([0.1, 0.2, 0.3]))
    sc = scenes.blob_scene(P, 61, 2)
    fr = _frame(oracle_mod, rs, sc)
    dev = cuda_device
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    cam_t = [rs.bg.to(dev), rs.view_matrix.to(dev).contiguous(), rs.proj_matrix.to(dev).contiguous(), rs.campos.to(dev)]
    cam = N.Camera(H, W, rs.tanfovx, rs.tanfovy, rs.scale_modifier, 2, 0, 0, *[x.data_ptr() for x in cam_t])
    g = N.Gaussians(P, 9, t["means3D"].data_ptr(), t["opacities"].data_ptr(), t["shs"].data_ptr(), None,
                    t["scales"].data_ptr(), t["rotations"].data_ptr(), None)
    bufs, sizes = {}, {}

    def make_cb(name):
        def cb(user, nbytes):  # == resizeFunctional, dgr/rasterize_points.cu:27-33
            bufs[name] = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            sizes[name] = int(nbytes)
            return bufs[name].data_ptr()
        return N.RESIZE_FN(cb)

    cbs = [make_cb(n) for n in ("geom", "binning", "img")]
    out_color = torch.empty((3, H, W), device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    L = N.lib()
    R = L.gcr_rasterize_forward(cbs[0], None, cbs[1], None, cbs[2], None, C.byref(cam), C.byref(g), out_color.data_ptr(),
                                radii.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert R == fr.R > 0, (R, N.lib().gcr_last_error())
    assert sizes["geom"] == L.gcr_geometry_bytes(P) and sizes["img"] == L.gcr_image_bytes(W, H)
    assert sizes["binning"] == L.gcr_binning_bytes(R, W, H)
    np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
    assert np.array_equal(out_color.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    # the callback-allocated buffers are what the backward consumes
    dpix = np.random.default_rng(4).normal(size=(3, H, W)).astype(np.float32)
    grads = ext.rasterize_gaussians_backward(cam_t[0], t["means3D"], radii, torch.Tensor([]), t["scales"], t["rotations"],
                                             rs.scale_modifier, torch.Tensor([]), cam_t[1], cam_t[2], rs.tanfovx, rs.tanfovy,
                                             torch.from_numpy(dpix).to(dev), t["shs"], 2, cam_t[3], bufs["geom"], R,
                                             bufs["binning"], bufs["img"], False)
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    _check_grads(fr.backward(dpix), {n: x.cpu().numpy() for n, x in zip(names, grads)},
                 ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])
    # a callback that fails to allocate is reported, not dereferenced
    null_cb = N.RESIZE_FN(lambda user, nbytes: None)
    rc = L.gcr_rasterize_forward(null_cb, None, cbs[1], None, cbs[2], None, C.byref(cam), C.byref(g), out_color.data_ptr(),
                                 radii.data_ptr(), None)
    assert rc < 0 and b"callback" in L.gcr_last_error()
