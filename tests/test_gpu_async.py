"""-m gpu: ABI v6 / v7 -- frames without the host wait (gcr_forward_async / FrameTicket), the overflow rescue behind the
frame gate, frames rendered without the backward's state, per-call options from several host threads, and the global
radix path on a scene with empty tiles.  Everything against the CPU oracle on identical seeded inputs, through the
native-module surface (ext -> C ABI -> gfx950 kernels)."""
import threading

import numpy as np
import pytest
import torch

import gpu_util as G
import scenes
from test_gpu_parity import GRAD_TOL, _check_forward, _check_grads, _frame

pytestmark = pytest.mark.gpu

ALL_GRADS = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dsh"]


def _args(rs, sc, device, sh=True):
    e = torch.Tensor([])
    return (rs.bg.to(device), G.to_dev(sc["means3D"], device), e if sh else G.to_dev(sc["colors_precomp"], device),
            G.to_dev(sc["opacities"], device), G.to_dev(sc["scales"], device), G.to_dev(sc["rotations"], device),
            rs.scale_modifier, e, rs.view_matrix.to(device), rs.proj_matrix.to(device), rs.tanfovx, rs.tanfovy,
            rs.img_h, rs.img_w, G.to_dev(sc["shs"], device) if sh else e, rs.sh_degree, rs.campos.to(device), False, False)


def test_frames_without_the_host_wait_match_the_oracle(oracle_mod, cuda_device):
    """rasterize_gaussians_ticket: the call returns with the frame enqueued and num_rendered as a FrameTicket.  Twelve
    frames of four poses back to back over three streams, nothing synchronised in between: every image bit-exact,
    every ticket resolves to the oracle's num_rendered, the frames after the first of the key really were asynchronous,
    and none of them needed the rescue."""
    from gaussiancity_amd import _native as N, ext
    P, W, H = 3500, 176, 128
    sc = scenes.blob_scene(P, 71, 2)
    cams = [scenes.camera(W, H, pose_index=i)._replace(sh_degree=2) for i in (1, 5, 9, 13)]
    frames = [_frame(oracle_mod, rs, sc) for rs in cams]
    ext._capacity_hint.pop((cuda_device.index, P, W, H), None)
    rescued0 = N.lib().gcr_rescue_count()
    streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]
    argsets = [_args(rs, sc, cuda_device) for rs in cams]
    torch.cuda.synchronize()
    outs = []
    for i in range(12):
        with torch.cuda.stream(streams[i % 3]):
            outs.append(ext.rasterize_gaussians_ticket(*argsets[i % 4]))
    assert all(isinstance(o[0], ext.FrameTicket) for o in outs)
    assert sum(1 for o in outs if o[0].seq != 0) >= 10, "the frames after the key's first must take the asynchronous path"
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        fr = frames[i % 4]
        assert int(o[0]) == fr.R and o[0].done() and not o[0].rescued
        assert np.array_equal(o[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32)), "frame %d" % i
        np.testing.assert_array_equal(o[2].cpu().numpy(), fr.radii)
    assert N.lib().gcr_rescue_count() == rescued0
    # the whole state of an asynchronous inference frame, decoded (lean binning buffer: sorted list + keys only)
    o = outs[-1]
    _check_forward(frames[11 % 4], G.decode(P, W, H, (int(o[0]),) + tuple(o[1:])), P, True)
    assert o[4].numel() == N.lib().gcr_binning_bytes_lean(o[0].capacity, W, H) < N.lib().gcr_binning_bytes(o[0].capacity, W, H)


@pytest.mark.parametrize("train", [False, True], ids=["inference_frame", "training_frame"])
def test_a_frame_that_overflows_its_capacity_guess_is_rescued_in_stream_order(oracle_mod, cuda_device, train, monkeypatch):
    """The capacity guess of an asynchronous frame can be short (the camera jumped).  The frame's gate kernel then
    holds the stream while the library's rescue thread renders the frame with an exactly sized buffer: work enqueued
    BEHIND the frame -- here a clone of the image, enqueued before anything is synchronised -- sees the right pixels.
    A training frame's backward bins the frame again (its state was not in the caller's buffer) and still matches."""
    from gaussiancity_amd import _native as N, ext
    P, W, H = 3500, 176, 128
    rs = scenes.camera(W, H)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 71, 2)
    fr = _frame(oracle_mod, rs, sc)
    key = (cuda_device.index, P, W, H)
    monkeypatch.setattr(ext, "_ASYNC_MARGIN", 16)
    ext._capacity_hint[key] = (10, 64)     # "the last frame rendered ten instances": capacity guess 36
    rescued0 = N.lib().gcr_rescue_count()
    a = _args(rs, sc, cuda_device)
    torch.cuda.synchronize()
    out = ext.rasterize_gaussians_ticket(*a, _for_backward=train)
    behind = out[1].clone()                # enqueued behind the gate, long before the rescue has run
    t = out[0]
    assert t.seq != 0 and t.capacity == 36
    assert int(t) == fr.R and t.rescued
    torch.cuda.synchronize()
    assert N.lib().gcr_rescue_count() == rescued0 + 1
    assert np.array_equal(behind.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    assert ext._capacity_hint[key][0] == fr.R   # the next guess starts from the truth
    # the same with the host thread blocked in a device-wide synchronisation instead of the ticket's wait: only the
    # rescue thread can release the gate then
    ext._capacity_hint[key] = (10, 64)
    out_s = ext.rasterize_gaussians_ticket(*a, _for_backward=train)
    torch.cuda.synchronize()
    assert out_s[0].done() and out_s[0].rescued and N.lib().gcr_rescue_count() == rescued0 + 2
    assert np.array_equal(out_s[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    # the frame after it fits again
    out2 = ext.rasterize_gaussians_ticket(*a, _for_backward=train)
    assert int(out2[0]) == fr.R and not out2[0].rescued
    assert np.array_equal(out2[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    if train:
        dpix = np.random.default_rng(4).normal(size=(3, H, W)).astype(np.float32)
        gref = fr.backward(dpix)
        for o in (out, out2):   # the rescued frame (binned again by the backward) and the ordinary one
            (bg, m3, col, opa, scl, rot, smod, cov, view, proj, tfx, tfy, h, w, sh, deg, campos, _, _) = a
            g = ext.rasterize_gaussians_backward(bg, m3, o[2], col, scl, rot, smod, cov, view, proj, tfx, tfy,
                                                 G.to_dev(dpix, cuda_device), sh, deg, campos, o[3], o[0], o[4], o[5], False)
            names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
            _check_grads(gref, {n: x.cpu().numpy() for n, x in zip(names, g)},
                         ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])


def test_autograd_through_tickets_and_the_wrapper_under_no_grad(oracle_mod, cuda_device):
    """GaussianRasterizer (autograd) keeps a ticket where the reference keeps num_rendered: forward + backward through
    it match the oracle, several steps in a row without a synchronisation in between; under no_grad the same module
    renders inference frames (no backward state) bit-exactly."""
    from gaussiancity_amd import GaussianRasterizer
    P, W, H = 2000, 112, 80
    rs_cpu = scenes.camera(W, H)._replace(sh_degree=3)
    sc = scenes.blob_scene(P, 41, 3)
    fr = _frame(oracle_mod, rs_cpu, sc)
    rs = rs_cpu._replace(bg=rs_cpu.bg.to(cuda_device), view_matrix=rs_cpu.view_matrix.to(cuda_device),
                         proj_matrix=rs_cpu.proj_matrix.to(cuda_device), campos=rs_cpu.campos.to(cuda_device))
    dpix = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    gref = fr.backward(dpix)
    dp = G.to_dev(dpix, cuda_device)
    r = GaussianRasterizer(rs)
    results = []
    for _ in range(4):
        t = {k: G.to_dev(sc[k], cuda_device).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        m2 = torch.zeros((P, 3), device=cuda_device, requires_grad=True)
        img, radii = r(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                       rotations=t["rotations"])
        (img * dp).sum().backward()
        results.append((img.detach(), t, m2))
    torch.cuda.synchronize()
    for img, t, m2 in results:
        assert np.array_equal(img.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
        for name, key in (("dL_dmean3D", "means3D"), ("dL_dopacity", "opacities"), ("dL_dsh", "shs"),
                          ("dL_dscale", "scales"), ("dL_drot", "rotations")):
            ref = gref[name]
            got = t[key].grad.cpu().numpy().reshape(ref.shape)
            assert np.abs(ref - got).max() <= GRAD_TOL * max(1.0, float(np.abs(ref).max())), name
    with torch.no_grad():
        t = {k: G.to_dev(sc[k], cuda_device) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        imgs = [r(means3D=t["means3D"], means2D=torch.zeros((P, 3), device=cuda_device), opacities=t["opacities"],
                  shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0] for _ in range(5)]
    for img in imgs:
        assert np.array_equal(img.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))


def test_backward_on_a_frame_without_state_is_binned_again_or_poisoned_never_silently_zero(oracle_mod, cuda_device):
    """An inference frame (gcr_camera.backward == 0) carries no backward state.  rasterize_gaussians_backward knows such
    a frame by its geometry buffer and bins it again, state only: the gradients match.  When it CANNOT know (the
    bookkeeping entry is gone: here it is deleted), the C ABI's own guards speak: the lean binning buffer of an
    inference frame is refused as too small, and -- should the buffer be large enough all the same (here: the lean
    buffer's contents copied into a full-size one) -- every rendered Gaussian's gradient is NaN, never a plausible zero."""
    from gaussiancity_amd import _native as N
    from gaussiancity_amd import ext
    P, W, H = 3000, 200, 150
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 11, 1)
    fr = _frame(oracle_mod, rs, sc)
    dpix = np.random.default_rng(2).normal(size=(3, H, W)).astype(np.float32)
    args, out = G.run_forward(rs, sc, cuda_device, for_backward=False)
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device), ALL_GRADS)
    # (that backward rendered the frame's state into a buffer of its own and stamped the frame with it: a fresh frame)
    args, out = G.run_forward(rs, sc, cuda_device, for_backward=False)
    with ext._meta_lock:
        ext._frame_meta.clear()
    with pytest.raises(RuntimeError, match="binning buffer too small"):
        G.run_backward(args, out, dpix, cuda_device)
    R, color, radii, geom, binning, img = out
    big = torch.zeros((N.lib().gcr_binning_bytes(R, W, H),), dtype=torch.uint8, device=cuda_device)
    big[:binning.numel()] = binning
    g = G.run_backward(args, (R, color, radii, geom, big, img), dpix, cuda_device)
    vis = (fr.radii > 0) & (fr.tiles_touched[:P] > 0)   # K1's survivors: the Gaussians the backward visits
    assert vis.sum() > 100
    for n in ("dL_dmean3D", "dL_dopacity", "dL_dscale", "dL_drot", "dL_dmean2D"):
        a = g[n].reshape(P, -1)
        a = a[:, :2] if n == "dL_dmean2D" else a   # (the reference's [P,3] slot: x and y are used)
        assert np.isnan(a[vis]).all(), n + ": a frame without backward state must poison its gradients"
        assert (a[~vis] == 0).all(), n


def test_three_host_threads_with_different_per_call_options(oracle_mod, cuda_device):
    """gcr_options travel with the call (ABI v6): three host threads render and differentiate the same scene at the same
    time, one with the defaults (bit-exact image), one with the wave-per-quadrant backward kernel + 64-entry pieces, one with the deterministic
    backward (bit-identical gradients run to run) -- and the process-wide defaults are what they were."""
    from gaussiancity_amd import _native as N, ext
    P, W, H, seed = 9000, 96, 80, 41
    rs = scenes.camera(W, H, pose_index=seed % 24)._replace(sh_degree=1, bg=torch.tensor((0.2, 0.0, 0.4)))
    sc = scenes.blob_scene(P, seed, 1, smax=14.0)
    fr = _frame(oracle_mod, rs, sc, True)
    dpix = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
    gref = fr.backward(dpix)
    res, errs = {}, []

    def worker(name, opts, rounds):
        try:
            torch.cuda.set_device(cuda_device)
            with torch.cuda.stream(torch.cuda.Stream(device=cuda_device)):
                with ext.options(**opts):
                    got = []
                    for _ in range(rounds):
                        args, out = G.run_forward(rs, sc, cuda_device, for_backward=True)
                        got.append((out[1].cpu().numpy(), G.run_backward(args, out, dpix, cuda_device)))
                    res[name] = got
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))

    th = [threading.Thread(target=worker, args=a) for a in (("default", {}, 4), ("waves", dict(bwd_wave_units=1, bwd_piece=64), 4),
                                                            ("det", dict(deterministic_backward=1), 4))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for name in res:
        for img, _ in res[name]:
            assert np.array_equal(img.view(np.uint32), fr.out_color.view(np.uint32)), name
    for name in res:
        for _, g in res[name]:
            _check_grads(gref, g, ALL_GRADS)
    d0 = res["det"][0][1]
    for _, g in res["det"][1:]:
        for n in ALL_GRADS:
            assert np.array_equal(g[n].view(np.uint32), d0[n].view(np.uint32)), n + " differs between deterministic runs"
    assert N.lib().gcr_grad_record_floats() == 16 and N.get_option("bwd_wave_units") == 0 and N.get_option("bwd_piece") == 160


def test_global_radix_path_on_a_scene_with_empty_tiles(oracle_mod, cuda_device):
    """the reference's own binning scheme (option force_radix) leaves an untouched tile's range at (0, 0);
    the backward pieces address their slots by the range START, so empty tiles must sit where the previous list ends.
    A sparse scene -- most tiles empty, a few long lists -- forward + backward on that path."""
    from gaussiancity_amd import ext
    P, W, H = 800, 208, 160
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 52, 1, spread=6.0, smin=0.5, smax=2.5)   # 37 of 130 tiles empty, longest list 707
    fr = _frame(oracle_mod, rs, sc)
    lens = fr.ranges[:, 1].astype(np.int64) - fr.ranges[:, 0]
    assert (lens == 0).sum() >= 20 and lens.max() > 128, (int((lens == 0).sum()), int(lens.max()))
    dpix = np.random.default_rng(5).normal(size=(3, H, W)).astype(np.float32)
    gref = fr.backward(dpix)
    for piece in (64, 128):
        with ext.options(force_radix=1, bwd_piece=piece):
            args, out = G.run_forward(rs, sc, cuda_device, for_backward=True)
            R, color, radii, geom, binning, img = out
            assert R == fr.R
            assert np.array_equal(color.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
            from gaussiancity_amd import _native as N
            L = N.get_layout(P, W, H, R)
            T = fr.ranges.shape[0]
            got = img.cpu().numpy()[L.img_ranges:L.img_ranges + 8 * T].view(np.uint32).reshape(T, 2)
            ne = lens > 0
            np.testing.assert_array_equal(got[ne], fr.ranges[ne])
            assert (got[~ne, 0] == got[~ne, 1]).all()
            ends = np.maximum.accumulate(np.concatenate([[0], fr.ranges[:, 1]]))[:-1]  # end of the last list before each tile
            np.testing.assert_array_equal(got[~ne, 0], ends[~ne])
            _check_grads(gref, G.run_backward(args, out, dpix, cuda_device), ALL_GRADS)


@pytest.mark.parametrize("flip_lr,flip_ud,crop", [(True, False, None), (False, True, (100, 37, 301, 200)), (True, True, (0, 0, 17, 9))])
def test_video_frames_stored_by_the_blend_equal_the_torch_conversion(cuda_device, flip_lr, flip_ud, crop):
    """GaussianRasterizerWrapper(as_uint8=True): the forward blend stores the uint8 [H,W,3] video frame that
    scripts/inference.py:655-667 derives from the float image (tensor_to_image * 255 -> uint8, five torch kernels).
    Same bytes -- values beyond [-1, 1] included (the colours are scaled so that the clamp matters) -- for mirrored and
    windowed frames, and through frames.InferenceLoop."""
    from gaussiancity_amd import GaussianRasterizerWrapper, synth
    from gaussiancity_amd.frames import InferenceLoop
    W, H, n = 480, 270, 40000
    cfg, sc = synth.make_scene("C4", n)
    rot = np.zeros((n, 4), np.float32)
    rot[:, 0] = 1.0
    pts = np.concatenate([sc["means3D"], np.full((n, 1), 0.7, np.float32), sc["scales"] * 3.0, rot, sc["colors_precomp"] * 1.7], axis=1)
    points = torch.from_numpy(pts.astype(np.float32)).to(cuda_device)
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), flip_lr=flip_lr, flip_ud=flip_ud, device=cuda_device)
    poses = synth.orbit_poses()[:6]
    with torch.no_grad():
        for pos, quat in poses:
            ref = InferenceLoop.to_uint8_hwc(wr(points, pos, quat, crop=crop))
            got = wr(points, pos, quat, crop=crop, as_uint8=True)
            assert got.dtype == torch.uint8 and got.shape == ref.shape
            assert torch.equal(got, ref)
            if crop is None or crop[2] * crop[3] > 10000:   # (the frame exercises the whole range, both clamps included)
                assert int(ref.max()) == 255 and int(ref.min()) == 0 and 20 < float(ref.float().mean()) < 235
        a = InferenceLoop(lambda p, cp, cq: wr(p, cp, cq, crop=crop), device=cuda_device).run(points, poses)
        b = InferenceLoop(None, device=cuda_device, render_uint8_fn=lambda p, cp, cq: wr(p, cp, cq, crop=crop, as_uint8=True)).run(points, poses)
    assert len(a) == len(b) == 6 and all(np.array_equal(x, y) for x, y in zip(a, b))
    with pytest.raises(RuntimeError, match="no backward"):
        wr(points.clone().requires_grad_(True), *poses[0], as_uint8=True)


def test_long_lazily_sorted_lists_in_asynchronous_frames(oracle_mod, cuda_device, monkeypatch):
    """Tile lists beyond 4096 entries of nearly transparent Gaussians: the forward blend sorts them on, segment by
    segment, as far as it walks -- here inside ASYNCHRONOUS frames: inference frames on the lean binning carve (the
    unsorted keys the extension reads sit at the capacity-carved offset), a training frame whose backward follows, and a
    training frame that overflows its capacity guess (rescued, then binned again by the backward)."""
    from gaussiancity_amd import ext
    P, W, H = 9000, 48, 48
    rs = scenes.camera(W, H)
    sc = scenes.blob_scene(P, 61, 0, spread=2.0, smin=0.5, smax=2.0, omin=0.01, omax=0.05)
    fr = _frame(oracle_mod, rs, sc, use_sh=False)
    lens = fr.ranges[:, 1].astype(np.int64) - fr.ranges[:, 0]
    assert lens.max() > 4096
    key = (cuda_device.index, P, W, H)
    ext._capacity_hint.pop(key, None)
    a = _args(rs, sc, cuda_device, sh=False)
    outs = [ext.rasterize_gaussians_ticket(*a) for _ in range(4)]                      # 1 synchronous + 3 asynchronous
    outs.append(ext.rasterize_gaussians_ticket(*a, _for_backward=True))                 # training frame
    monkeypatch.setattr(ext, "_ASYNC_MARGIN", 16)
    ext._capacity_hint[key] = (100, 64)
    outs.append(ext.rasterize_gaussians_ticket(*a, _for_backward=True))                 # overflows: rescued
    torch.cuda.synchronize()
    assert [o[0].seq != 0 for o in outs] == [False, True, True, True, True, True] and outs[-1][0].rescued
    for o in outs:
        assert int(o[0]) == fr.R
        assert np.array_equal(o[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    d = G.decode(P, W, H, (fr.R,) + tuple(outs[2][1:]))
    _check_forward(fr, d, P, False)
    assert (d["tile_sorted"] > 1024).any()
    dpix = np.random.default_rng(8).normal(size=(3, H, W)).astype(np.float32)
    gref = fr.backward(dpix)
    (bg, m3, col, opa, scl, rot, smod, cov, view, proj, tfx, tfy, h, w, sh, deg, campos, _, _) = a
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    for o in outs[-2:]:
        g = ext.rasterize_gaussians_backward(bg, m3, o[2], col, scl, rot, smod, cov, view, proj, tfx, tfy,
                                             G.to_dev(dpix, cuda_device), sh, deg, campos, o[3], o[0], o[4], o[5], False)
        _check_grads(gref, {n: x.cpu().numpy() for n, x in zip(names, g)},
                     ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dscale", "dL_drot"])


def test_soak_threads_streams_mixed_frames_with_forced_overflows(oracle_mod, cuda_device, monkeypatch):
    """VERDICT r04 item 7 / ADVICE r04: three host threads x three streams each (nine caller streams + the rescue's) x
    GCR_SOAK_FRAMES (default 2 000) mixed inference / training frames per thread, the capacity guess of about one frame
    in twenty cut to 36 instances so that it overflows and needs the rescue -- several at once, from different threads.
    Every image is compared on the device with the oracle's (bit-exact), a sample of the training frames is
    differentiated and compared with the oracle's gradients (rescued frames among them: their backward bins again), every
    ticket resolves, nobody times out."""
    import os
    import random
    from gaussiancity_amd import _native as N, ext
    nframes = int(os.environ.get("GCR_SOAK_FRAMES", "2000"))
    W, H = 160, 112
    cut = {}

    def hook(key, capacity):
        return 36 if cut.pop(threading.get_ident(), False) else capacity

    monkeypatch.setattr(ext, "_guess_hook", hook)
    L = N.lib()
    rescued0, dropped0 = L.gcr_rescue_count(), L.gcr_rescue_dropped_count()
    scenes_t = []
    for ti in range(3):
        P = 3000 + 300 * ti
        sc = scenes.blob_scene(P, 90 + ti, 1)
        cams = [scenes.camera(W, H, pose_index=i)._replace(sh_degree=1) for i in (2 + ti, 8 + ti, 14 + ti)]
        scenes_t.append((P, sc, cams, [_frame(oracle_mod, rs, sc) for rs in cams]))
    dpix = np.random.default_rng(5).normal(size=(3, H, W)).astype(np.float32)
    errs, stats = [], []

    def worker(ti):
        try:
            torch.cuda.set_device(cuda_device)
            P, sc, cams, frames = scenes_t[ti]
            rng = random.Random(1000 + ti)
            streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]
            argsets = [_args(rs, sc, cuda_device) for rs in cams]
            want = [torch.from_numpy(fr.out_color.view(np.int32)).to(cuda_device) for fr in frames]
            dp = G.to_dev(dpix, cuda_device)
            grefs = [None] * len(cams)
            bad = torch.zeros((), dtype=torch.int64, device=cuda_device)
            torch.cuda.synchronize()
            tickets, n_over, n_bwd = [], 0, 0
            for f in range(nframes):
                k = rng.randrange(len(cams))
                train = rng.random() < 0.25
                over = f > 3 and rng.random() < 0.05
                if over:
                    cut[threading.get_ident()] = True
                st = streams[rng.randrange(3)]
                with torch.cuda.stream(st):
                    out = ext.rasterize_gaussians_ticket(*argsets[k], _for_backward=train)
                    bad += (out[1].view(torch.int32) != want[k]).any()   # enqueued behind the frame (and its gate)
                    n_over += int(over and out[0].seq != 0)
                    tickets.append((out[0], k))
                    if train and rng.random() < 0.04:   # a sample of the backward passes, rescued frames included
                        (bg, m3, col, opa, scl, rot, smod, cov, view, proj, tfx, tfy, h, w, sh, deg, campos, _, _) = argsets[k]
                        g = ext.rasterize_gaussians_backward(bg, m3, out[2], col, scl, rot, smod, cov, view, proj, tfx, tfy,
                                                             dp, sh, deg, campos, out[3], out[0], out[4], out[5], False)
                        if grefs[k] is None:
                            grefs[k] = frames[k].backward(dpix)
                        names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
                        _check_grads(grefs[k], {n: x.cpu().numpy() for n, x in zip(names, g)},
                                     ["dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"])
                        n_bwd += 1
                if len(tickets) > 64:   # resolve in the background, as a frame loop that logs num_rendered would
                    t, kk = tickets.pop(0)
                    assert int(t) == frames[kk].R
            for t, kk in tickets:
                assert int(t) == frames[kk].R
            for st in streams:
                st.synchronize()
            assert int(bad.item()) == 0, "%d frames of thread %d differ from the oracle" % (int(bad.item()), ti)
            stats.append((ti, n_over, n_bwd))
        except Exception as e:  # noqa: BLE001
            import traceback
            errs.append((ti, repr(e), traceback.format_exc()))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    forced = sum(s[1] for s in stats)
    assert L.gcr_rescue_count() - rescued0 == forced, (L.gcr_rescue_count() - rescued0, forced)
    assert L.gcr_rescue_dropped_count() == dropped0
    assert forced > (50 if nframes >= 2000 else 0), stats
    print("soak: %d frames, %d rescued, %d sampled backward passes" % (3 * nframes, forced, sum(s[2] for s in stats)))


def test_a_gate_that_times_out_fails_its_own_ticket_and_nothing_else(oracle_mod, cuda_device, monkeypatch):
    """The gate's timeout path: with the rescue thread held, a frame that overflows is
    NOT rendered, its gate gives up after `gate_polls`, the stream goes on, and the frame's own ticket -- nobody else's --
    raises.  The rescue that would come afterwards must not touch the frame's buffers (the stream has long handed them
    on): it is counted as dropped.  The thread's next frames render as if nothing had happened."""
    import time
    from gaussiancity_amd import _native as N, ext
    P, W, H = 3500, 176, 128
    rs = scenes.camera(W, H)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 71, 2)
    fr = _frame(oracle_mod, rs, sc)
    key = (cuda_device.index, P, W, H)
    a = _args(rs, sc, cuda_device)
    L = N.lib()
    out0 = ext.rasterize_gaussians_ticket(*a)   # the key's first frame: synchronous, sets the hint
    assert int(out0[0]) == fr.R
    monkeypatch.setattr(ext, "_guess_hook", lambda k, c: 36)
    rescued0, dropped0 = L.gcr_rescue_count(), L.gcr_rescue_dropped_count()
    prev_polls = N.set_option("gate_polls", 20000)   # ~0.1-0.2 s instead of ~2 s
    N.set_option("rescue_hold", 1)
    try:
        out = ext.rasterize_gaussians_ticket(*a)
        t = out[0]
        assert t.seq != 0 and t.capacity == 36
        t0 = time.time()
        torch.cuda.synchronize()                 # returns: the gate gave up, the device is not hung
        assert time.time() - t0 < 30.0
        with pytest.raises(RuntimeError, match="not rescued in time"):
            int(t)
        assert t.done() and t.failed is not None
        out[1].fill_(7.0)                        # "the allocator handed the block to somebody else"
        torch.cuda.synchronize()
    finally:
        N.set_option("rescue_hold", 0)
        N.set_option("gate_polls", prev_polls)
    monkeypatch.setattr(ext, "_guess_hook", None)
    deadline = time.time() + 10.0
    while L.gcr_rescue_dropped_count() == dropped0 and time.time() < deadline:
        time.sleep(0.01)
    assert L.gcr_rescue_dropped_count() == dropped0 + 1 and L.gcr_rescue_count() == rescued0
    time.sleep(0.05)
    assert bool((out[1] == 7.0).all()), "a rescue that came after its gate had given up wrote into the frame's buffers"
    with pytest.raises(RuntimeError, match="not rescued in time"):   # the failure stays with its ticket ...
        t.wait()
    for _ in range(40):                                               # ... and only there: the ring goes round past it
        o = ext.rasterize_gaussians_ticket(*a)
        assert int(o[0]) == fr.R
    assert np.array_equal(o[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    assert ext._capacity_hint[key][0] == fr.R


def test_the_ticket_ring_is_given_back_when_its_thread_ends(cuda_device):
    """Pinned host words are a per-thread ring (ext._TicketRing): a thread that rendered asynchronously and ended must
    not leave its block mapped -- the frames it left unresolved are waited for first, the library's
    rescue thread is told to forget the block (gcr_host_words_free), and other threads' frames go on."""
    import gc
    import weakref
    from gaussiancity_amd import ext
    P, W, H = 2000, 96, 64
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 3, 1)
    a = _args(rs, sc, cuda_device)
    first = ext.rasterize_gaussians_ticket(*a)
    R = int(first[0])
    seen = {}

    def worker():
        torch.cuda.set_device(cuda_device)
        outs = [ext.rasterize_gaussians_ticket(*a) for _ in range(6)]   # the last ones stay unresolved
        seen["ring"] = weakref.ref(ext._tls.ring)
        seen["tickets"] = [o[0] for o in outs]

    for _ in range(3):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        gc.collect()
        assert seen["ring"]() is None, "the thread's ring outlived the thread"
        assert all(x.done() and int(x) == R for x in seen["tickets"])   # resolved by the ring's close()
    o = ext.rasterize_gaussians_ticket(*a)
    assert int(o[0]) == R
    torch.cuda.synchronize()
    ring = ext._tls.ring   # explicit close is idempotent, a closed ring is replaced on the next frame
    ring.close()
    ring.close()
    ext._tls.ring = None
    assert int(ext.rasterize_gaussians_ticket(*a)[0]) == R


def test_prefiltered_with_a_gaussian_behind_the_near_plane_fails_the_frame_not_the_device(oracle_mod, cuda_device):
    """gcr_camera.prefiltered (cr/auxiliary.h:135-156): upstream a point that fails the near-plane test although the caller
    declared the cloud pre-filtered hits a device printf + __trap().  Here the frame fails with the reference's text -- the
    synchronous entry point at once, an asynchronous frame through ITS ticket --, nothing is rendered, and the device, the
    stream and the thread's next frames are fine.  With every Gaussian in front of the camera the flag changes nothing;
    fused and two-kernel K1, LDS tables and the global-cursor path."""
    from gaussiancity_amd import ext
    P, W, H = 3000, 128, 96
    rs = scenes.camera(W, H)._replace(sh_degree=1)
    sc = scenes.blob_scene(P, 17, 1)
    fr = _frame(oracle_mod, rs, sc)
    a = list(_args(rs, sc, cuda_device))
    ok = ext.rasterize_gaussians(*a)
    assert ok[0] == fr.R
    a_pre = list(a)
    a_pre[17] = True   # prefiltered
    view = rs.view_matrix.numpy().reshape(4, 4)
    tz = np.concatenate([sc["means3D"], np.ones((P, 1), np.float32)], 1) @ view[:, 2]
    if (tz > 0.2).all():   # this cloud is in front of the camera: the flag must change nothing
        out = ext.rasterize_gaussians(*a_pre)
        assert out[0] == fr.R and np.array_equal(out[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    # one Gaussian moved behind the camera (its view-space z is negated through the camera's position)
    campos = rs.campos.numpy()
    means = sc["means3D"].copy()
    means[7] = campos - 3.0 * (means[7] - campos)   # mirrored through the camera centre: behind it
    tz7 = float(np.append(means[7], 1.0) @ view[:, 2])
    assert tz7 <= 0.2
    a_bad = list(a_pre)
    a_bad[1] = G.to_dev(means, cuda_device)
    a_fine = list(a_bad)
    a_fine[17] = False
    fr_bad = _frame(oracle_mod, rs, dict(sc, means3D=means))
    for opts in ({}, dict(split_preprocess=1), dict(force_global_cursor=1)):
        with ext.options(**opts):
            with pytest.raises(RuntimeError, match="Point is filtered although prefiltered is set"):
                ext.rasterize_gaussians(*a_bad)
            out = ext.rasterize_gaussians(*a_fine)    # the same cloud without the declaration: culled silently, as upstream
            assert out[0] == fr_bad.R
            assert np.array_equal(out[1].cpu().numpy().view(np.uint32), fr_bad.out_color.view(np.uint32)), opts
    # asynchronous frames: the error is the frame's own ticket's
    first = ext.rasterize_gaussians_ticket(*a_fine)
    assert int(first[0]) == fr_bad.R
    t_bad = ext.rasterize_gaussians_ticket(*a_bad)[0]
    t_ok = ext.rasterize_gaussians_ticket(*a_fine)
    assert t_bad.seq != 0 and t_ok[0].seq != 0
    with pytest.raises(RuntimeError, match="Point is filtered although prefiltered is set"):
        int(t_bad)
    assert int(t_ok[0]) == fr_bad.R
    torch.cuda.synchronize()
    assert np.array_equal(t_ok[1].cpu().numpy().view(np.uint32), fr_bad.out_color.view(np.uint32))
