"""Host-side API vs golden vectors captured from the reference's own Python
(tests/golden/make_golden.py; the reference source itself is not in this repo)."""
import json
import os
import types

import numpy as np
import pytest
import torch

import gaussiancity_amd as ga
from gaussiancity_amd import rasterizer as rz

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CPU = torch.device("cpu")


@pytest.fixture()
def recorder(monkeypatch):
    """Recording stand-in for the native module (same contract as make_golden.Recorder)."""
    rec = types.SimpleNamespace(fw_args=None, bw_args=None, ramp=None)

    def fw(*args):
        rec.fw_args = args
        H, W, P = args[12], args[13], args[1].shape[0]
        color = rec.ramp if rec.ramp is not None else torch.zeros(3, H, W)
        return (7, color, torch.zeros(P, dtype=torch.int32), torch.zeros(5, dtype=torch.uint8),
                torch.zeros(6, dtype=torch.uint8), torch.zeros(7, dtype=torch.uint8))

    def bw(*args):
        rec.bw_args = args
        P = args[1].shape[0]
        M = args[13].shape[1] if args[13].numel() else 0
        f = lambda shape, v: torch.full(shape, float(v))  # noqa: E731
        return (f((P, 3), 1), f((P, 3), 2), f((P, 1), 3), f((P, 3), 4), f((P, 6), 5), f((P, M, 3), 6),
                f((P, 3), 7), f((P, 4), 8))

    fake = types.SimpleNamespace(rasterize_gaussians=fw, rasterize_gaussians_backward=bw)
    monkeypatch.setattr(rz, "_ext", fake)
    return rec


def test_reference_recipe_on_host_is_bit_equal_to_the_golden_camera():
    """GaussianRasterizerWrapper(host_camera="reference") evaluates the reference's recipe with torch on the host and
    hands the matrices to the kernels by value: its settings are the golden ones bit for bit (on any `device`), and the
    closed-form variant (host_camera=True) agrees to rounding."""
    g = np.load(os.path.join(GOLD, "camera.npz"))
    for i in range(int(g["n_tuples"])):
        ss = tuple(int(v) for v in g["sensor_%d" % i])
        wr = ga.GaussianRasterizerWrapper(g["K_%d" % i], ss, device=CPU, host_camera="reference")
        pos, q = g["pos_%d" % i], g["quat_%d" % i]
        rs = wr._get_gaussian_rasterization_settings(pos, q)
        assert rs.view_matrix.device.type == "cpu"
        assert np.array_equal(rs.view_matrix.numpy(), g["view_%d" % i])
        assert np.array_equal(rs.proj_matrix.numpy(), g["proj_%d" % i])
        assert np.array_equal(rs.campos.numpy(), g["campos_%d" % i])
        assert np.array_equal(np.array([rs.tanfovx, rs.tanfovy]), g["tanfov_%d" % i])
        fast = ga.GaussianRasterizerWrapper(g["K_%d" % i], ss, device=CPU, host_camera=True)
        rf = fast._get_gaussian_rasterization_settings(pos, q)
        scale = max(1.0, float(np.abs(g["proj_%d" % i]).max()))
        assert np.abs(rf.view_matrix.numpy() - g["view_%d" % i]).max() <= 2e-6 * scale
        assert np.abs(rf.proj_matrix.numpy() - g["proj_%d" % i]).max() <= 2e-6 * scale
        assert np.abs(rf.campos.numpy() - g["campos_%d" % i]).max() <= 1e-6 * max(1.0, float(np.abs(g["campos_%d" % i]).max()))


def test_fast_w2c_route_gives_the_library_route_bits():
    """rasterizer._w2c_fast replaces the scipy Rotation object of dgr/__init__.py:349-368 by the same IEEE operations on
    Python floats (the numpy lines behind it are the reference's): bit-equal to the library route on thousands of poses,
    random and orbit-like (where the translation cancels); and the wrapper's own render path skips `view.inverse()` --
    campos is only read by the SH evaluation, which that path never runs -- without touching view / proj."""
    from gaussiancity_amd import synth
    assert rz._fast_w2c_available()
    rng = np.random.default_rng(7)
    cases = [(rng.normal(size=3) * 10.0 ** rng.integers(0, 4), rng.normal(size=4)) for _ in range(4000)]
    cases += synth.orbit_poses() + synth.orbit_poses(24, 60.0, 50.0)
    for p, q in cases:
        assert np.array_equal(rz._w2c_fast(p, q).view(np.uint32), rz._w2c_reference_ops(p, q).view(np.uint32)), (p, q)
    g = np.load(os.path.join(GOLD, "camera.npz"))
    for i in range(int(g["n_tuples"])):
        ss = tuple(int(v) for v in g["sensor_%d" % i])
        wr = ga.GaussianRasterizerWrapper(g["K_%d" % i], ss, device=CPU, host_camera="reference")
        rs = wr._reference_recipe_on_host(g["pos_%d" % i], g["quat_%d" % i], need_campos=False)
        assert np.array_equal(rs.view_matrix.numpy(), g["view_%d" % i]) and np.array_equal(rs.proj_matrix.numpy(), g["proj_%d" % i])
        assert float(rs.campos.abs().sum()) == 0.0   # the stand-in: never handed out, never read (precomputed colours)


def test_camera_math_matches_reference():
    g = np.load(os.path.join(GOLD, "camera.npz"))
    for i in range(int(g["n_tuples"])):
        ss = tuple(int(v) for v in g["sensor_%d" % i])
        wr = ga.GaussianRasterizerWrapper(g["K_%d" % i], ss, device=CPU)
        assert np.array_equal(np.array([wr.fov_x, wr.fov_y], dtype=np.float64), g["fov_%d" % i])
        assert np.array_equal(wr.P.numpy(), g["P_%d" % i])
        pos, q = g["pos_%d" % i], g["quat_%d" % i]
        assert np.array_equal(wr._get_w2c_matrix(pos, q).numpy(), g["w2c_%d" % i])
        rs = wr._get_gaussian_rasterization_settings(pos, q)
        assert np.array_equal(np.array([rs.tanfovx, rs.tanfovy]), g["tanfov_%d" % i])
        assert np.array_equal(rs.view_matrix.numpy(), g["view_%d" % i])
        assert np.array_equal(rs.proj_matrix.numpy(), g["proj_%d" % i])
        assert np.array_equal(rs.campos.numpy(), g["campos_%d" % i])
        assert [rs.img_h, rs.img_w] == list(g["hw_%d" % i])
        assert rs.sh_degree == 0 and rs.scale_modifier == 1.0 and rs.prefiltered is False and rs.debug is False
        assert float(rs.bg.abs().sum()) == 0.0 and rs.bg.dtype == torch.float32


def test_orbit_poses_match_reference():
    from gaussiancity_amd import synth
    g = np.load(os.path.join(GOLD, "camera.npz"))
    K = np.array([1528.1469407006614, 0, 480, 0, 1528.1469407006614, 270, 0, 0, 1]).reshape(3, 3)
    wr = ga.GaussianRasterizerWrapper(K, (960, 540), device=CPU)
    poses = synth.orbit_poses()
    assert np.array_equal(np.stack([p for p, _ in poses]), g["orbit_pos"])
    assert np.array_equal(np.stack([q for _, q in poses]), g["orbit_quat"])
    for i, (p, q) in enumerate(poses):
        rs = wr._get_gaussian_rasterization_settings(p, q)
        assert np.array_equal(rs.view_matrix.numpy(), g["orbit_view"][i])
        assert np.array_equal(rs.proj_matrix.numpy(), g["orbit_proj"][i])
        assert np.array_equal(rs.campos.numpy(), g["orbit_campos"][i])


def _describe(a):
    if isinstance(a, torch.Tensor):
        return {"kind": "tensor", "shape": list(a.shape), "dtype": str(a.dtype).replace("torch.", ""),
                "tag": (float(a.detach().reshape(-1)[0]) if a.numel() else None)}
    if isinstance(a, bool):
        return {"kind": "bool", "value": a}
    if isinstance(a, int):
        return {"kind": "int", "value": a}
    if isinstance(a, float):
        return {"kind": "float", "value": a}
    return {"kind": type(a).__name__}


def test_native_argument_order_and_gradient_routing(recorder):
    gold = json.load(open(os.path.join(GOLD, "arg_order.json")))
    assert list(ga.GaussianRasterizationSettings._fields) == gold["settings_fields"]
    P, M, H, W = 5, 4, 6, 8
    tag = lambda shape, v: torch.full(shape, float(v), requires_grad=True)  # noqa: E731
    inputs = dict(means3D=tag((P, 3), 101), means2D=tag((P, 3), 102), sh=tag((P, M, 3), 103),
                  colors_precomp=torch.Tensor([]), opacities=tag((P, 1), 105), scales=tag((P, 3), 106),
                  rotations=tag((P, 4), 107), cov3Ds_precomp=torch.Tensor([]))
    rs = ga.GaussianRasterizationSettings(
        img_h=H, img_w=W, tanfovx=0.25, tanfovy=0.125, bg=torch.full((3,), 201.0), scale_modifier=1.5,
        view_matrix=torch.full((4, 4), 202.0), proj_matrix=torch.full((4, 4), 203.0), sh_degree=1,
        campos=torch.full((3,), 204.0), prefiltered=False, debug=False)
    color, radii = ga.RasterizeGaussiansFunction.apply(
        inputs["means3D"], inputs["means2D"], inputs["sh"], inputs["colors_precomp"], inputs["opacities"],
        inputs["scales"], inputs["rotations"], inputs["cov3Ds_precomp"], rs)
    assert color.shape == (3, H, W) and radii.dtype == torch.int32
    color.sum().backward()
    assert [_describe(a) for a in recorder.fw_args] == gold["forward_args"]
    assert [_describe(a) for a in recorder.bw_args] == gold["backward_args"]
    routed = {k: (float(v.grad.reshape(-1)[0]) if (v.requires_grad and v.grad is not None) else None)
              for k, v in inputs.items()}
    assert routed == gold["grad_position_to_input"]


def test_validation_matches_reference(recorder):
    gold = json.load(open(os.path.join(GOLD, "validation.json")))
    P, M = 5, 4
    rs = ga.GaussianRasterizationSettings(6, 8, 0.25, 0.125, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 1,
                                          torch.zeros(3), False, False)
    r = ga.GaussianRasterizer(rs)
    shapes = dict(shs=(P, M, 3), colors_precomp=(P, 3), scales=(P, 3), rotations=(P, 4), cov3D_precomp=(P, 6))
    for case in gold:
        kw = {k: torch.zeros(shapes[k]) for k in case["keys"]}
        call = lambda: r(means3D=torch.zeros(P, 3), means2D=torch.zeros(P, 3), opacities=torch.zeros(P, 1), **kw)  # noqa: E731
        if case["raises"] is None:
            call()
        else:
            with pytest.raises(Exception) as ei:
                call()
            assert type(ei.value).__name__ == case["raises"] and str(ei.value) == case["message"]


def test_point_split_and_flips(recorder):
    g = np.load(os.path.join(GOLD, "split_flip.npz"))
    K = np.array([1528.1469407006614, 0, 480, 0, 1528.1469407006614, 270, 0, 0, 1]).reshape(3, 3)
    pts = torch.from_numpy(g["points"])
    H, W = g["ramp"].shape[1:]
    for flip_lr in (True, False):
        for flip_ud in (True, False):
            recorder.ramp = torch.from_numpy(g["ramp"])
            wr = ga.GaussianRasterizerWrapper(K, (W, H), flip_lr=flip_lr, flip_ud=flip_ud, device=CPU)
            img = wr(pts, np.array([1.0, 2.0, 3.0]), np.array([0.0, 0.0, 0.0, 1.0]))
            assert np.array_equal(img.numpy(), g["img_lr%d_ud%d" % (flip_lr, flip_ud)])
    a = recorder.fw_args
    assert np.array_equal(a[1].numpy(), g["fw_means3D"]) and np.array_equal(a[2].numpy(), g["fw_colors"])
    assert np.array_equal(a[3].numpy(), g["fw_opacity"]) and np.array_equal(a[4].numpy(), g["fw_scales"])
    assert np.array_equal(a[5].numpy(), g["fw_rotations"])
    assert a[14].numel() == int(g["fw_sh_numel"]) and a[7].numel() == int(g["fw_cov_numel"]) and a[15] == int(g["fw_degree"])
    with pytest.raises(AssertionError):
        wr(torch.zeros(4, 13), np.zeros(3), np.array([0.0, 0, 0, 1]))


def test_helpers_match_reference():
    g = np.load(os.path.join(GOLD, "helpers.npz"))
    t = lambda k: torch.from_numpy(g[k].copy())  # noqa: E731
    xyz, scales = t("xyz"), t("scales")
    out = ga.get_gaussian_points(xyz, scales, {"rgb": t("attr_rgb"), "xyz": t("attr_xyz"), "scale": t("attr_scale")})
    assert np.array_equal(out.numpy(), g["points_full"])
    assert np.array_equal(xyz.numpy(), g["xyz_after"]) and np.array_equal(scales.numpy(), g["scales_after"])
    out2 = ga.get_gaussian_points(t("xyz2"), t("scales2"), {"rgb": t("rgb2")})
    assert np.array_equal(out2.numpy(), g["points_default"])

    def fake_rasterizer(pts, pos, quat):
        return float(pts.sum()) + torch.arange(3 * 10 * 12, dtype=torch.float32).reshape(3, 10, 12)

    pts = torch.from_numpy(g["points_full"])
    boxes = [dict(x=int(b[0]), y=int(b[1]), w=int(b[2]), h=int(b[3])) for b in g["boxes"]]
    a = ga.get_gaussian_rasterization(pts, fake_rasterizer, t("cam_pos"), t("cam_quat"))
    b = ga.get_gaussian_rasterization(pts, fake_rasterizer, t("cam_pos"), t("cam_quat"), boxes)
    assert np.array_equal(a.numpy(), g["raster_nocrop"]) and np.array_equal(b.numpy(), g["raster_crop"])


def test_import_paths():
    import extensions.diff_gaussian_rasterization as dgr
    import diff_gaussian_rasterization_ext as e
    assert dgr.GaussianRasterizerWrapper is ga.GaussianRasterizerWrapper
    assert all(hasattr(e, n) for n in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"))


def test_product_has_no_cpu_path():
    """The product must fail loudly without a GPU tensor -- never fall back to the oracle."""
    from gaussiancity_amd import ext
    with pytest.raises(RuntimeError):
        ext.rasterize_gaussians(torch.zeros(3), torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 1),
                                torch.zeros(4, 3), torch.zeros(4, 4), 1.0, torch.Tensor([]), torch.eye(4),
                                torch.eye(4), 0.3, 0.3, 16, 16, torch.Tensor([]), 0, torch.zeros(3), False, False)
    with pytest.raises(RuntimeError):
        ext.rasterize_gaussians(torch.zeros(3), torch.zeros(4, 2), *([torch.Tensor([])] * 4), 1.0, torch.Tensor([]),
                                torch.eye(4), torch.eye(4), 0.3, 0.3, 16, 16, torch.Tensor([]), 0, torch.zeros(3),
                                False, False)
    src = open(os.path.join(os.path.dirname(ga.__file__), "ext.py")).read() + \
        open(os.path.join(os.path.dirname(ga.__file__), "rasterizer.py")).read()
    assert "oracle" not in src.replace("bit-exact vs oracle", "")
