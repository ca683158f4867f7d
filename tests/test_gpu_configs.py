"""-m gpu: every BASELINE.json config (C1..C5, SURVEY.md section 8d "Config mapping") at its FULL size, HIP path
vs the CPU oracle on the same `gcity-synth-v1` inputs.

  C1  10k S-rand, 256x256, degree 0, forward (precomputed colours AND the SH run)       -- full state, bit-exact
  C2  500k S-rand, 640x448, SH3, forward + backward                                      -- full state bit-exact,
                                                                                            all 8 gradient tensors <= 1e-4*max
  C3  5M S-city, 1920x1080, SH3, forward, 3 orbit poses                                  -- full state (1 pose), image/radii/
                                                                                            n_contrib (2 more), bit-exact
  C4  16384 points, 960x540 -> 640x448 crop, through helpers.get_gaussian_points /
      get_gaussian_rasterization / GaussianRasterizerWrapper / autograd (core/train.py:263-295,
      utils/helpers.py:226-270)                                                           -- image bit-exact, leaf gradients
  C5  20M S-city, 3840x2160, SH3, forward                                                 -- image/radii/R/n_contrib bit-exact
                                                                                            (1 pose) + list properties
Bars as in test_gpu_parity.py: integer state exact, float forward state and image BIT-exact, gradients
max|d| <= 1e-4 * max(1, max|ref|).
"""
import numpy as np
import pytest
import torch

import gpu_util as G
import scenes
from gaussiancity_amd import synth
from test_gpu_parity import GRAD_TOL, _check_forward, _check_grads, _frame

pytestmark = pytest.mark.gpu

ALL_GRADS = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"]


def _orbit_camera(cfg, pose):
    return scenes.camera(cfg["W"], cfg["H"], pose_index=pose, radius=512.0, altitude=640.0)._replace(
        sh_degree=cfg["sh_degree"])


def _light_check(fr, out, W, H):
    """R, radii, n_contrib, final_T and the image against the oracle without decoding the geometry buffer."""
    R, color, radii, geom, binning, img = out
    assert R == fr.R
    assert np.array_equal(radii.cpu().numpy(), fr.radii)
    L = G.N.get_layout(fr.P, W, H, R)
    nc = img[L.img_n_contrib:L.img_n_contrib + 4 * W * H].view(torch.int32).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(nc, fr.n_contrib)
    ft = img[L.img_final_T:L.img_final_T + 4 * W * H].view(torch.int32).cpu().numpy().view(np.uint32)
    assert np.array_equal(ft, fr.final_T.view(np.uint32)), "final_T not bit-exact"
    got = color.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), fr.out_color.view(np.uint32)), (
        "image not bit-exact: %d pixels differ, max |d| %g" % (int((got != fr.out_color).any(0).sum()),
                                                            float(np.abs(got - fr.out_color).max())))


@pytest.mark.parametrize("variant", ["precomputed_colour", "sh_degree0"])
def test_c1_10k_256x256_forward(oracle_mod, cuda_device, variant):
    cfg, sc = synth.make_scene("C1")
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    use_sh = variant == "sh_degree0"
    for pose in (0, 7, 19):
        rs = _orbit_camera(cfg, pose)
        fr = _frame(oracle_mod, rs, sc, use_sh)
        assert fr.R > 0
        args, out = G.run_forward(rs, sc, cuda_device, use_sh=use_sh)
        _check_forward(fr, G.decode(P, W, H, out), P, use_sh)


def test_c2_500k_640x448_sh3_forward_backward(oracle_mod, cuda_device):
    cfg, sc = synth.make_scene("C2")
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    rs = _orbit_camera(cfg, 3)
    fr = _frame(oracle_mod, rs, sc)
    assert fr.R > 100000 and (fr.radii > 0).sum() > 10000
    args, out = G.run_forward(rs, sc, cuda_device)
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    dpix = synth.grad_image(W, H, cfg["seed"])
    gref = fr.backward(dpix)
    ggpu = G.run_backward(args, out, dpix, cuda_device)
    _check_grads(gref, ggpu, ALL_GRADS)
    # nothing leaks into Gaussians that were not rendered
    hidden = fr.radii <= 0
    for n in ("dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dopacity"):
        assert not np.any(ggpu[n][hidden]), n
    # a second pose, forward only, image-level (different tile lists)
    rs2 = _orbit_camera(cfg, 14)
    _light_check(_frame(oracle_mod, rs2, sc), G.run_forward(rs2, sc, cuda_device)[1], W, H)


def test_c2_reference_faithful_960x540_then_crop(oracle_mod, cuda_device):
    """the reference renders the full 960x540 sensor and crops 640x448 (utils/helpers.py:255-260); the
    loss only sees the crop, so dL/dpixel is zero outside it."""
    cfg, sc = synth.make_scene("C2")
    P, W, H = cfg["P"], 960, 540
    rs = scenes.camera(W, H, pose_index=5, radius=512.0, altitude=640.0)._replace(sh_degree=3)
    fr = _frame(oracle_mod, rs, sc)
    args, out = G.run_forward(rs, sc, cuda_device)
    _light_check(fr, out, W, H)
    dfull = np.zeros((3, H, W), np.float32)
    y0, x0 = (H - 448) // 2, (W - 640) // 2
    dfull[:, y0:y0 + 448, x0:x0 + 640] = synth.grad_image(640, 448, cfg["seed"])
    _check_grads(fr.backward(dfull), G.run_backward(args, out, dfull, cuda_device), ALL_GRADS)


def test_c3_5m_1080p_three_poses_vs_oracle(oracle_mod, cuda_device):
    cfg, sc = synth.make_scene("C3")
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    for k, pose in enumerate((0, 9, 17)):
        rs = _orbit_camera(cfg, pose)
        fr = _frame(oracle_mod, rs, sc)
        assert fr.R > 500000
        args, out = G.run_forward(rs, sc, cuda_device)
        if k == 0:
            _check_forward(fr, G.decode(P, W, H, out), P, True)   # every sub-array of the three buffers
        else:
            _light_check(fr, out, W, H)
        # the same frame as GaussianRasterizer.forward renders it when nothing asks for a gradient (the bench's timed entry
        # point: no autograd node, one scratch block, num_rendered not waited for) -- twice, the second from cached records
        from gaussiancity_amd import ext
        for _ in range(2):
            img, radii = ext.rasterize_gaussians_frame(*args[:18])
            assert np.array_equal(img.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
            np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)


def test_c4_training_step_through_helpers_wrapper_and_autograd(oracle_mod, cuda_device):
    """The G-step's rasterizer leg exactly as the reference calls it: get_gaussian_points -> [B,N,14] ->
    get_gaussian_rasterization(wrapper, crop) -> loss -> autograd back to the generator's outputs."""
    from gaussiancity_amd import helpers
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    cfg, sc = synth.make_scene("C4")
    N, W, H = cfg["P"], cfg["W"], cfg["H"]
    cw, ch = cfg["crop"]
    dev = cuda_device
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
    pos, quat = synth.orbit_poses()[5]   # the bench's C4 camera (bench.py --train-step)
    xyz_leaf = torch.from_numpy(sc["means3D"][None]).to(dev).requires_grad_(True)
    scl_leaf = torch.from_numpy(sc["scales"][None]).to(dev).requires_grad_(True)
    rgb_leaf = torch.from_numpy(sc["colors_precomp"][None]).to(dev).requires_grad_(True)
    # get_gaussian_points updates xyz / scales in place (utils/helpers.py:232-234): hand it non-leaf views
    offset = torch.full((1, N, 3), 0.25, device=dev)
    gs = helpers.get_gaussian_points(xyz_leaf * 1.0, scl_leaf * 1.0, {"rgb": rgb_leaf, "xyz": offset, "scale": 2.0})
    assert gs.shape == (1, N, 14)
    box = {"x": (W - cw) // 2, "y": (H - ch) // 2, "w": cw, "h": ch}
    img = helpers.get_gaussian_rasterization(gs, wr, [pos], [quat], crop_bboxes=[box])
    assert img.shape == (1, 3, ch, cw)
    dpix = synth.grad_image(cw, ch, cfg["seed"])
    (img[0] * torch.from_numpy(dpix).to(dev)).sum().backward()
    # oracle on what the wrapper feeds the native module (dgr/__init__.py:382-426)
    rs = wr._get_gaussian_rasterization_settings(pos, quat)
    rs_cpu = rs._replace(bg=rs.bg.cpu(), view_matrix=rs.view_matrix.cpu(), proj_matrix=rs.proj_matrix.cpu(),
                         campos=rs.campos.cpu())
    rot = np.zeros((N, 4), np.float32)
    rot[:, 0] = 1.0
    sco = dict(means3D=sc["means3D"] + np.float32(0.25), scales=sc["scales"] * np.float32(2.0), rotations=rot,
               opacities=np.ones((N, 1), np.float32), colors_precomp=sc["colors_precomp"])
    fr = _frame(oracle_mod, rs_cpu, sco, use_sh=False)
    assert fr.R > 1000 and (fr.radii > 0).sum() > 500
    want = fr.out_color[:, :, ::-1][:, box["y"]:box["y"] + ch, box["x"]:box["x"] + cw]       # flip_lr, then crop
    got = img[0].detach().cpu().numpy()
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32)), "C4 image not bit-exact"
    dfull = np.zeros((3, H, W), np.float32)
    dfull[:, box["y"]:box["y"] + ch, box["x"]:box["x"] + cw] = dpix
    gref = fr.backward(np.ascontiguousarray(dfull[:, :, ::-1]))
    got_g = {"dL_dmean3D": xyz_leaf.grad[0].cpu().numpy(), "dL_dcolor": rgb_leaf.grad[0].cpu().numpy(),
             "dL_dscale": scl_leaf.grad[0].cpu().numpy()}
    gref = dict(gref, dL_dscale=gref["dL_dscale"] * np.float32(2.0))  # d(scale*2)/d(scale)
    _check_grads(gref, got_g, list(got_g))
    assert np.abs(got_g["dL_dmean3D"]).max() > 0


def test_c4_step_with_the_discriminators_half(cuda_device):
    """frames.DDPTrainStep(discriminator=...) on the device, single process (round 6; core/train.py:227-295): the D-step's
    render runs under no_grad -- an inference frame for the rasterizer, its image the same bits as the G-step's render of the
    same points --, the discriminator stand-in gets dense gradients of its full size, the generator none from the D-step, and
    the G-step that follows produces the gradients of a G-step-only object plus the GAN term's."""
    from gaussiancity_amd import frames
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    cfg, sc = synth.make_scene("C4")
    N, W, H = cfg["P"], cfg["W"], cfg["H"]
    cw, ch = cfg["crop"]
    dev = cuda_device
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
    pos, quat = synth.orbit_poses()[5]
    rot = sc["rotations"][:, [1, 2, 3, 0]]
    base = torch.from_numpy(np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], rot, sc["colors_precomp"]],
                                           axis=1).astype(np.float32)).to(dev)
    crop = ((W - cw) // 2, (H - ch) // 2, cw, ch)
    target = torch.full((3, ch, cw), 0.25, device=dev)

    def generator():
        g = frames.StandInGenerator(n_param=2 * 4000, n_layers=2, device=dev)
        with torch.no_grad():
            for i, p in enumerate(g.layers):
                p[:28] = 1e-3 * (i + 1)
        return g

    gen, dis = generator(), frames.StandInDiscriminator(n_param=5001, device=dev)
    with torch.no_grad():
        dis.p[0], dis.p[1] = 0.5, -0.2
    step = frames.DDPTrainStep(wr, gen, crop=crop, discriminator=dis)
    d_loss = step.d_step(base, pos, quat, target)
    assert float(d_loss) > 0 and all(p.grad is None for p in gen.parameters())
    assert dis.p.grad.shape == (5001,) and float(dis.p.grad[:2].abs().max()) > 0 and float(dis.p.grad[2:].abs().max()) == 0.0
    with torch.no_grad():
        fake = step._render(gen(base), pos, quat)
    loss, img = step.step(base, pos, quat, target)
    assert np.array_equal(fake.cpu().numpy().view(np.uint32), img.cpu().numpy().view(np.uint32))  # no_grad frame = training frame
    # the G-step alone on a twin generator, plus the GAN term by hand
    gen2 = generator()
    step2 = frames.DDPTrainStep(wr, gen2, crop=crop)
    pts = gen2(base)
    img2 = step2._render(pts, pos, quat)
    want = (img2 - target).abs().mean() + 0.5 * torch.relu(1.0 - dis(img2))
    for p in gen2.parameters():
        p.grad = None
    want.backward()
    assert abs(float(want.detach()) - float(loss)) <= 1e-5 * max(1.0, abs(float(want.detach())))
    for a, b in zip(gen.parameters(), gen2.parameters()):
        tol = 2e-4 * max(1.0, float(b.grad.abs().max()))  # (float atomics in the backward blend: two runs differ in the last bits)
        assert float((a.grad - b.grad).abs().max()) <= tol


@pytest.mark.parametrize("flip_lr,flip_ud", [(True, False), (False, False), (True, True), (False, True)])
def test_points14_node_equals_the_generic_route(oracle_mod, cuda_device, flip_lr, flip_ud):
    """GaussianRasterizerWrapper on this build's own rasterizer runs as ONE autograd node on the [N,14] tensor in place
    (row strides, mirrored image store, one [N,14] gradient: rasterizer._RasterizePoints14Function).  Against the
    generic route of the reference (five slices -> GaussianRasterizer -> torch.flip), forced here by a subclass
    instance: image bit-equal for every flip combination, gradient equal within the atomics' tolerance -- with general
    opacities and rotations, and with `points` a column window of a wider tensor (row stride 20)."""
    from gaussiancity_amd.rasterizer import GaussianRasterizer, GaussianRasterizerWrapper
    cfg, sc = synth.make_scene("C4")
    N, W, H = cfg["P"], cfg["W"], cfg["H"]
    dev = cuda_device
    rng = np.random.default_rng(77)
    rot = rng.normal(size=(N, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opa = rng.uniform(0.1, 1.0, size=(N, 1)).astype(np.float32)
    pts_np = np.concatenate([sc["means3D"], opa, sc["scales"] * np.float32(3.0), rot, sc["colors_precomp"]], axis=1)
    wide = torch.zeros((N, 20), dtype=torch.float32, device=dev)
    wide[:, 3:17] = torch.from_numpy(pts_np).to(dev)
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), flip_lr=flip_lr, flip_ud=flip_ud, device=dev)
    pos, quat = synth.orbit_poses()[7]
    dpix = torch.from_numpy(synth.grad_image(W, H, 5)).to(dev)

    class Foreign(GaussianRasterizer):  # not `type(...) is GaussianRasterizer`: takes the reference's route
        pass

    res = {}
    for name in ("node", "generic"):
        leaf = wide.clone().requires_grad_(True)
        points = leaf[:, 3:17]
        assert points.stride(0) == 20
        rz = wr.get_gaussian_rasterizer(pos, quat)
        if name == "generic":
            rz = Foreign(rz.raster_settings)
        img = wr(points, gaussian_rasterizer=rz)
        (img * dpix).sum().backward()
        res[name] = (img.detach().cpu().numpy(), leaf.grad.cpu().numpy())
    assert np.array_equal(res["node"][0].view(np.uint32), res["generic"][0].view(np.uint32)), "image differs"
    gn, gg = res["node"][1], res["generic"][1]
    assert np.all(gn[:, :3] == 0) and np.all(gn[:, 17:] == 0) and np.abs(gn[:, 3:17]).max() > 0
    for lo, hi, nme in ((3, 6, "xyz"), (6, 7, "opacity"), (7, 10, "scale"), (10, 14, "rotation"), (14, 17, "rgb")):
        a, b = gn[:, lo:hi], gg[:, lo:hi]
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max()), nme
    # and against the oracle (image of the un-mirrored frame, then the wrapper's flips)
    rs = wr._get_gaussian_rasterization_settings(pos, quat)
    rs_cpu = rs._replace(bg=rs.bg.cpu(), view_matrix=rs.view_matrix.cpu(), proj_matrix=rs.proj_matrix.cpu(),
                         campos=rs.campos.cpu())
    sco = dict(means3D=sc["means3D"], scales=sc["scales"] * np.float32(3.0), rotations=rot, opacities=opa,
               colors_precomp=sc["colors_precomp"])
    fr = _frame(oracle_mod, rs_cpu, sco, use_sh=False)
    want = fr.out_color
    if flip_lr:
        want = want[:, :, ::-1]
    if flip_ud:
        want = want[:, ::-1, :]
    assert np.array_equal(res["node"][0].view(np.uint32), np.ascontiguousarray(want).view(np.uint32))


@pytest.mark.parametrize("flip_lr,flip_ud,crop", [(True, False, (37, 21, 301, 173)), (False, True, (0, 0, 17, 9)),
                                                  (True, True, (640, 400, 320, 140)), (False, False, (5, 530, 950, 10))])
def test_windowed_render_equals_cropping_the_full_frame(oracle_mod, cuda_device, flip_lr, flip_ud, crop):
    """GaussianRasterizerWrapper.forward(crop=(x, y, w, h)) stores and differentiates only that window of the (mirrored)
    image (gcr_camera.win_*; tiles outside it are skipped in both directions).  Windows that cut tiles, touch the image
    border or are a few pixels wide: the image equals the oracle's full frame flipped and cropped, bit for bit, and the
    [N,14] gradient equals the oracle's for dL/dpixel zero outside the window."""
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    cfg, sc = synth.make_scene("C4")
    N, W, H = cfg["P"], cfg["W"], cfg["H"]
    dev = cuda_device
    pts_np, rot = _c4_points(sc)
    pos, quat = synth.orbit_poses()[11]
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), flip_lr=flip_lr, flip_ud=flip_ud, device=dev)
    points = torch.from_numpy(pts_np).to(dev).requires_grad_(True)
    x, y, w, h = crop
    img = wr(points, pos, quat, crop=crop)
    assert tuple(img.shape) == (3, h, w)
    dwin = np.random.default_rng(3).normal(size=(3, h, w)).astype(np.float32)
    (img * torch.from_numpy(dwin).to(dev)).sum().backward()
    rs = wr._get_gaussian_rasterization_settings(pos, quat)
    rs_cpu = rs._replace(bg=rs.bg.cpu(), view_matrix=rs.view_matrix.cpu(), proj_matrix=rs.proj_matrix.cpu(),
                         campos=rs.campos.cpu())
    sco = dict(means3D=sc["means3D"], scales=sc["scales"], rotations=rot, opacities=np.ones((N, 1), np.float32),
               colors_precomp=sc["colors_precomp"])
    fr = _frame(oracle_mod, rs_cpu, sco, use_sh=False)

    def view(a):  # the wrapper's flips (each is its own inverse)
        a = a[:, :, ::-1] if flip_lr else a
        return a[:, ::-1, :] if flip_ud else a

    want = np.ascontiguousarray(view(fr.out_color)[:, y:y + h, x:x + w])
    assert np.array_equal(img.detach().cpu().numpy().view(np.uint32), want.view(np.uint32))
    dfull = np.zeros((3, H, W), np.float32)
    dfull[:, y:y + h, x:x + w] = dwin
    gref = fr.backward(np.ascontiguousarray(view(dfull)))
    g = points.grad.cpu().numpy()
    got = {"dL_dmean3D": g[:, 0:3], "dL_dopacity": g[:, 3:4], "dL_dscale": g[:, 4:7], "dL_drot": g[:, 7:11],
           "dL_dcolor": g[:, 11:14]}
    _check_grads(gref, got, list(got))


def test_camera_modes_of_the_wrapper(oracle_mod, cuda_device):
    """Where GaussianRasterizerWrapper computes its camera (`host_camera`): the DEFAULT on a GPU is the reference's
    recipe evaluated with torch on the host -- bit-equal to a CPU-device wrapper's settings (what the golden vectors
    pin), handed to the kernels by value (gcr_camera.host_camera); True = closed form on the host; False = the recipe on
    the device.  Every mode renders bit-exactly against the oracle fed with ITS matrices, the matrices agree to
    rounding, the images agree within 1e-4 but for a counted handful of threshold pixels, and gradients flow."""
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    cfg, sc = synth.make_scene("C4")
    N, W, H = cfg["P"], cfg["W"], cfg["H"]
    dev = cuda_device
    pts_np, rot = _c4_points(sc)
    pos, quat = synth.orbit_poses()[9]
    K = synth.intrinsics(W, H)
    cpu_rs = GaussianRasterizerWrapper(K, (W, H), device=torch.device("cpu"))._get_gaussian_rasterization_settings(pos, quat)
    sco = dict(means3D=sc["means3D"], scales=sc["scales"], rotations=rot, opacities=np.ones((N, 1), np.float32),
               colors_precomp=sc["colors_precomp"])
    images = {}
    for mode in (None, True, False):
        wr = GaussianRasterizerWrapper(K, (W, H), device=dev, host_camera=mode)
        rs = wr._get_gaussian_rasterization_settings(pos, quat)
        on_host = mode is not False
        assert (rs.view_matrix.device.type == "cpu") == on_host and (rs.campos.device.type == "cpu") == on_host
        for a, b in ((rs.view_matrix, cpu_rs.view_matrix), (rs.proj_matrix, cpu_rs.proj_matrix), (rs.campos, cpu_rs.campos)):
            if mode is None:   # the default: the recipe on the host = the CPU reference, bit for bit
                assert np.array_equal(a.numpy(), b.numpy())
            else:
                assert np.allclose(a.cpu().numpy(), b.numpy(), rtol=2e-6, atol=3e-4)
        points = torch.from_numpy(pts_np).to(dev).requires_grad_(True)
        img = wr(points, pos, quat)
        img.abs().sum().backward()
        assert points.grad is not None and float(points.grad.abs().max()) > 0
        rs_cpu = rs._replace(bg=rs.bg.cpu(), view_matrix=rs.view_matrix.cpu(), proj_matrix=rs.proj_matrix.cpu(),
                             campos=rs.campos.cpu())
        fr = _frame(oracle_mod, rs_cpu, sco, use_sh=False)
        got = img.detach().cpu().numpy()
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(fr.out_color[:, :, ::-1]).view(np.uint32)), mode
        images[mode] = got
        assert bool(wr.get_gaussian_rasterizer(pos, quat).markVisible(points.detach()[:, :3].contiguous()).any())
    for mode in (True, False):
        off = np.abs(images[mode] - images[None]).max(axis=0) > 1e-4 * max(1.0, float(np.abs(images[None]).max()))
        assert int(off.sum()) <= max(2, int(2e-5 * W * H)), (mode, int(off.sum()))


def test_c5_20m_4k_forward_vs_oracle_and_list_properties(oracle_mod, cuda_device):
    cfg, sc = synth.make_scene("C5")
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    rs = _orbit_camera(cfg, 11)
    fr = _frame(oracle_mod, rs, sc)
    assert fr.R > 2000000
    args, out = G.run_forward(rs, sc, cuda_device)
    _light_check(fr, out, W, H)
    R, _, radii, geom, binning, img = out
    # the sorted list itself and the tile ranges, against the oracle's
    L = G.N.get_layout(P, W, H, R)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    plist = binning[L.bin_vals[L.bin_sorted]:L.bin_vals[L.bin_sorted] + 4 * R].view(torch.int32).cpu().numpy().view(np.uint32)
    d = {"point_list": plist}
    if G.lazy_sort_active():   # lists beyond 1024 entries are final only as far as the blend walked them
        d["tile_sorted"] = img[L.img_tile_lazy:L.img_tile_lazy + 16 * T].view(torch.int32).view(T, 4)[:, 0].cpu().numpy().view(np.uint32)
    G.assert_point_list(d, fr)
    ranges = img[L.img_ranges:L.img_ranges + 8 * T].view(torch.int32).view(T, 2).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(ranges, fr.ranges)
    # idempotence at this size (second run takes the speculative path)
    _, out2 = G.run_forward(rs, sc, cuda_device)
    assert out2[0] == R and torch.equal(out2[1].view(torch.int32), out[1].view(torch.int32))


def _c4_points(sc):
    """[N,14] = xyz, opacity, scale3, rot4 (r,x,y,z), rgb3 as get_gaussian_points builds it (identity rotation)."""
    N = sc["means3D"].shape[0]
    rot = np.zeros((N, 4), np.float32)
    rot[:, 0] = 1.0
    return np.concatenate([sc["means3D"], np.ones((N, 1), np.float32), sc["scales"], rot, sc["colors_precomp"]],
                          axis=1).astype(np.float32), rot


def test_c4_train_step_harness_with_the_real_rasterizer(oracle_mod, cuda_device):
    """frames.TrainStepHarness (core/train.py:263-295) on the GPU with the real wrapper: the image of the
    step and d(loss)/d(points) equal the oracle's, stale gradients do not accumulate, Adam moves the points."""
    from gaussiancity_amd import frames
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    cfg, sc = synth.make_scene("C4")
    N, W, H = cfg["P"], cfg["W"], cfg["H"]
    cw, ch = cfg["crop"]
    dev = cuda_device
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
    pts_np, rot = _c4_points(sc)
    points = torch.from_numpy(pts_np).to(dev).requires_grad_(True)
    crop = ((W - cw) // 2, (H - ch) // 2, cw, ch)
    target = torch.zeros((3, ch, cw), device=dev)
    h = frames.TrainStepHarness(wr, n_param=4096, crop=crop, device=dev)
    pos, quat = synth.orbit_poses()[2]
    points.grad = torch.full_like(points, 123.0)   # stale gradient: step() must not add to it
    loss, img, _ = h.step(points, pos, quat, target)
    rs = wr._get_gaussian_rasterization_settings(pos, quat)
    rs_cpu = rs._replace(bg=rs.bg.cpu(), view_matrix=rs.view_matrix.cpu(), proj_matrix=rs.proj_matrix.cpu(),
                         campos=rs.campos.cpu())
    sco = dict(means3D=sc["means3D"], scales=sc["scales"], rotations=rot, opacities=np.ones((N, 1), np.float32),
               colors_precomp=sc["colors_precomp"])
    fr = _frame(oracle_mod, rs_cpu, sco, use_sh=False)
    x, y, w, hh = crop
    want = np.ascontiguousarray(fr.out_color[:, :, ::-1][:, y:y + hh, x:x + w])
    assert np.array_equal(img.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # d|img|.mean()/d img = sign(img)/numel on the crop, zero elsewhere; un-flip for the oracle
    dcrop = np.sign(want) / np.float32(want.size)
    dfull = np.zeros((3, H, W), np.float32)
    dfull[:, y:y + hh, x:x + w] = dcrop
    gref = fr.backward(np.ascontiguousarray(dfull[:, :, ::-1]))
    g = points.grad.cpu().numpy()
    got = {"dL_dmean3D": g[:, 0:3], "dL_dopacity": g[:, 3:4], "dL_dscale": g[:, 4:7], "dL_drot": g[:, 7:11],
           "dL_dcolor": g[:, 11:14]}
    for n in got:   # gradients of a mean over 860k values are ~1e-6: the bar is relative to the tensor's own maximum
        ref = gref[n]
        assert float(np.abs(ref - got[n]).max()) <= GRAD_TOL * float(np.abs(ref).max()) + 1e-12, n
    assert float(h.param_grad[0]) == float(h.param_grad[-1]) != 0.0
    # with a learning rate the harness applies Adam (replicas stay equal through the rank-averaged gradient)
    h2 = frames.TrainStepHarness(wr, n_param=4096, crop=crop, device=dev, lr=1e-2)
    before = points.detach().clone()
    l0 = float(h2.step(points, pos, quat, target)[0])
    for _ in range(4):
        l1 = float(h2.step(points, pos, quat, target)[0])
    assert not torch.equal(points.detach(), before) and l1 < l0


def test_inference_loop_frames_equal_the_oracle(oracle_mod, cuda_device):
    """frames.InferenceLoop (scripts/inference.py:655-667 shape: pose list -> wrapper -> uint8 HWC frames, three
    side streams, ring of pinned buffers) against the ORACLE's frames, not against the HIP path itself; `points` is
    produced on the caller's stream immediately before run() (the ordering ADVICE r01 flagged)."""
    from gaussiancity_amd import frames
    from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
    W, H = 256, 144
    dev = cuda_device
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
    sc = scenes.blob_scene(6000, 93, 0)
    rot_xyzw = sc["rotations"]   # the wrapper passes columns 7:11 straight through as (r,x,y,z)
    base = np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], rot_xyzw, sc["colors_precomp"]], axis=1)
    poses = synth.orbit_poses(9, 60.0, 50.0)
    staging = torch.from_numpy(base.astype(np.float32)).to(dev)
    big = torch.randn(64 << 20, device=dev)          # keeps the caller's stream busy ...
    big = big * 1.0001 + 0.5
    pts = staging * 1.0                               # ... so `pts` is written late on that stream
    got = frames.InferenceLoop(wr, device=dev).run(pts, poses)
    assert len(got) == len(poses)
    for (pos, quat), frame in zip(poses, got):
        rs = wr._get_gaussian_rasterization_settings(pos, quat)
        rs_cpu = rs._replace(bg=rs.bg.cpu(), view_matrix=rs.view_matrix.cpu(), proj_matrix=rs.proj_matrix.cpu(),
                             campos=rs.campos.cpu())
        fr = _frame(oracle_mod, rs_cpu, sc, use_sh=False)
        want = frames.InferenceLoop.to_uint8_hwc(torch.from_numpy(np.ascontiguousarray(fr.out_color[:, :, ::-1]))).numpy()
        assert np.array_equal(frame, want)
    assert any(f.any() for f in got)


def test_dense_3dgs_like_scene_long_lists(oracle_mod, cuda_device):
    """General-3DGS-like stress scene (synth.s_dense; not a GaussianCity workload, the API must still support it):
    ~2000 list entries per tile on average, the central tiles beyond the 4096-entry LDS sort capacity, early
    termination active.  Forward state, image and gradients against the oracle."""
    P, W, H = 300_000, 640, 360
    sc = synth.s_dense(P, 7)
    rs = scenes.camera(W, H, pose_index=5, radius=512.0, altitude=640.0)._replace(sh_degree=3)
    fr = _frame(oracle_mod, rs, sc)
    lens = fr.ranges[:, 1].astype(np.int64) - fr.ranges[:, 0]
    assert lens.mean() > 2000 and (lens > 4096).sum() >= 3 and fr.consumed_entries() < 0.8 * fr.R
    for _ in range(2):   # staged, then speculative with the long-list hint
        args, out = G.run_forward(rs, sc, cuda_device)
        d = G.decode(P, W, H, out)
        _check_forward(fr, d, P, True)
    # option "lazy_sort" (default): this is the scene it is for -- most of what was emitted is never sorted, and some
    # tiles were sorted on by the blend behind their first segment
    ns = d["tile_sorted"].astype(np.int64)
    assert ns[lens > 1024].sum() < 0.6 * lens[lens > 1024].sum() and (ns > 1024).any()
    dpix = synth.grad_image(W, H, 7)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device), ALL_GRADS)
    # ... and sorting every list whole gives the same image and the same per-pixel state
    prev = G.N.set_option("lazy_sort", 0)
    try:
        _, out_full = G.run_forward(rs, sc, cuda_device)
        _check_forward(fr, G.decode(P, W, H, out_full), P, True)
    finally:
        G.N.set_option("lazy_sort", prev)
    assert torch.equal(out_full[1].view(torch.int32), out[1].view(torch.int32))
