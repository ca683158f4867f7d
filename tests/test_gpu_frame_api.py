"""-m gpu: inference frames through the Python API without the autograd node (ext.rasterize_gaussians_frame, what
GaussianRasterizer.forward calls when nothing asks for a gradient): the image and the radii of the reference's call,
bit for bit against the CPU oracle; the per-thread record cache keyed on the identity of the call's tensors; one scratch
allocation for the three state buffers; the overflow rescue behind such a frame."""
import numpy as np
import pytest
import torch

import gpu_util as G
import scenes
from test_gpu_async import _args
from test_gpu_parity import _frame

pytestmark = pytest.mark.gpu


def _frame_args(a):
    """positional arguments of rasterize_gaussians -> those of rasterize_gaussians_frame (no `debug`)"""
    return a[:18]


def _same(img, fr):
    return np.array_equal(img.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))


def test_frame_entry_point_matches_the_oracle_and_reuses_its_records(oracle_mod, cuda_device):
    """Four poses, three streams, sixteen frames back to back: images and radii bit-exact; the second frame of a pose
    finds the records of the first (same tensor objects); a tensor modified IN PLACE is read again by the kernels (the
    cache holds addresses, not values); a NEW tensor object -- even with equal contents -- is a miss."""
    from gaussiancity_amd import ext
    P, W, H = 3500, 176, 128
    sc = scenes.blob_scene(P, 71, 2)
    cams = [scenes.camera(W, H, pose_index=i)._replace(sh_degree=2) for i in (1, 5, 9, 13)]
    frames = [_frame(oracle_mod, rs, sc) for rs in cams]
    argsets = [_frame_args(_args(rs, sc, cuda_device)) for rs in cams]
    streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]
    ext._prep_cache().clear()
    torch.cuda.synchronize()
    outs = []
    for i in range(16):
        with torch.cuda.stream(streams[i % 3]):
            outs.append(ext.rasterize_gaussians_frame(*argsets[i % 4]))
    torch.cuda.synchronize()
    assert len(ext._prep_cache()) == 4
    for i, (img, radii) in enumerate(outs):
        assert _same(img, frames[i % 4]), "frame %d" % i
        np.testing.assert_array_equal(radii.cpu().numpy(), frames[i % 4].radii)
    ents = [id(e) for e in ext._prep_cache().values()]
    ext.rasterize_gaussians_frame(*argsets[0])
    assert [id(e) for e in ext._prep_cache().values()] == ents, "a repeated call must hit"
    # in place: the opacities halve, the SAME records render the new frame
    a0 = argsets[0]
    a0[3].mul_(0.5)
    sc2 = dict(sc, opacities=sc["opacities"] * np.float32(0.5))
    fr2 = _frame(oracle_mod, cams[0], sc2)
    img, _ = ext.rasterize_gaussians_frame(*a0)
    assert _same(img, fr2) and [id(e) for e in ext._prep_cache().values()] == ents
    # a new object with the old contents: a miss (its key is new), and the right image
    a_new = a0[:3] + (G.to_dev(sc["opacities"], cuda_device),) + a0[4:]
    img, _ = ext.rasterize_gaussians_frame(*a_new)
    assert _same(img, frames[0]) and len(ext._prep_cache()) == 5


def test_frame_entry_point_with_strided_inputs_and_absent_optionals(oracle_mod, cuda_device):
    """A non-contiguous input is copied by the binding (as the reference's .contiguous() does) and such a call is never
    cached -- the copy would be a snapshot; None and empty tensors both mean "absent"; precomputed colours instead of SH."""
    from gaussiancity_amd import ext
    P, W, H = 2500, 160, 112
    rs = scenes.camera(W, H)._replace(sh_degree=0)
    sc = scenes.blob_scene(P, 5, 0)
    sc["colors_precomp"] = np.random.default_rng(3).uniform(0, 1, (P, 3)).astype(np.float32)
    fr = _frame(oracle_mod, rs, sc, use_sh=False)
    a = list(_frame_args(_args(rs, sc, cuda_device, sh=False)))
    wide = torch.zeros((P, 6), dtype=torch.float32, device=cuda_device)
    wide[:, :3] = a[1]
    a[1] = wide[:, :3]  # means3D as a strided view
    a[7] = None         # cov3D_precomp absent as None, sh absent as an empty tensor
    ext._prep_cache().clear()
    for _ in range(3):
        img, radii = ext.rasterize_gaussians_frame(*a)
        assert _same(img, fr)
        np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
    assert len(ext._prep_cache()) == 0
    wide[:, :3] += 0.25  # the view's contents change: the next call must see them (it copies again)
    sc2 = dict(sc, means3D=sc["means3D"] + np.float32(0.25))
    img, _ = ext.rasterize_gaussians_frame(*a)
    assert _same(img, _frame(oracle_mod, rs, sc2, use_sh=False))


def test_module_call_without_gradients_takes_the_frame_entry_point(oracle_mod, cuda_device, monkeypatch):
    """GaussianRasterizer.forward: no_grad, or no input that requires a gradient -> rasterize_gaussians_frame (no autograd
    node); an input that requires one -> RasterizeGaussiansFunction as upstream, and its backward still matches."""
    from gaussiancity_amd import ext
    from gaussiancity_amd.rasterizer import GaussianRasterizer
    P, W, H = 3000, 160, 112
    rs = scenes.camera(W, H)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 9, 2)
    fr = _frame(oracle_mod, rs, sc)
    dev = cuda_device
    rs_dev = rs._replace(bg=rs.bg.to(dev), view_matrix=rs.view_matrix.to(dev), proj_matrix=rs.proj_matrix.to(dev),
                         campos=rs.campos.to(dev))
    t = {k: G.to_dev(sc[k], dev) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    calls = []
    real = ext.rasterize_gaussians_frame
    monkeypatch.setattr(ext, "rasterize_gaussians_frame", lambda *a: (calls.append(1), real(*a))[1])
    ras = GaussianRasterizer(rs_dev)
    kw = dict(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], shs=t["shs"],
              scales=t["scales"], rotations=t["rotations"])
    img, radii = ras(**kw)  # grad mode on, nothing requires a gradient
    assert len(calls) == 1 and not img.requires_grad and _same(img, fr)
    with torch.no_grad():
        t["means3D"].requires_grad_(True)
        img, radii = ras(**kw)
    assert len(calls) == 2 and _same(img, fr)
    np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
    img, radii = ras(**kw)  # means3D requires a gradient: the autograd function
    assert len(calls) == 2 and img.requires_grad and _same(img.detach(), fr)
    dpix = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    img.backward(torch.from_numpy(dpix).to(dev))
    gref = fr.backward(dpix)
    err = float(np.abs(gref["dL_dmean3D"] - t["means3D"].grad.cpu().numpy()).max())
    assert err <= 1e-4 * max(1.0, float(np.abs(gref["dL_dmean3D"]).max())), err


def test_frame_entry_point_overflow_is_rescued(oracle_mod, cuda_device, monkeypatch):
    """A capacity guess that is too short (the camera jumped): the library's rescue renders the frame, work enqueued
    behind it sees the right pixels -- with the state buffers carved from one allocation that the call has already
    given back to the allocator."""
    from gaussiancity_amd import _native as N, ext
    P, W, H = 3500, 176, 128
    rs = scenes.camera(W, H)._replace(sh_degree=2)
    sc = scenes.blob_scene(P, 71, 2)
    fr = _frame(oracle_mod, rs, sc)
    key = (cuda_device.index, P, W, H)
    monkeypatch.setattr(ext, "_ASYNC_MARGIN", 16)
    ext._capacity_hint[key] = (10, 64)  # "the last frame rendered ten instances"
    rescued0 = N.lib().gcr_rescue_count()
    a = _frame_args(_args(rs, sc, cuda_device))
    torch.cuda.synchronize()
    img, radii = ext.rasterize_gaussians_frame(*a)
    behind = img.clone()
    torch.cuda.synchronize()
    assert N.lib().gcr_rescue_count() == rescued0 + 1
    assert _same(behind, fr) and _same(img, fr)
    np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
    ext._ring(N.lib()).harvest()
    assert ext._hint_get(key)[0] == fr.R  # the ticket nobody holds still refreshes the hint


def test_frame_entry_point_soak_with_short_guesses(oracle_mod, cuda_device, monkeypatch):
    """240 inference frames of four poses back to back over three streams, nothing synchronised in between, every twelfth
    one with its capacity guess cut to 36 instances (the rescue renders it into the scratch block the call has long given
    back to the allocator -- in stream order, behind the frame's gate): every image compared ON THE DEVICE with the oracle's
    as soon as it is enqueued (the comparison kernels queue behind the frame), the rescues counted."""
    from gaussiancity_amd import _native as N, ext
    P, W, H = 3500, 176, 128
    sc = scenes.blob_scene(P, 71, 2)
    cams = [scenes.camera(W, H, pose_index=i)._replace(sh_degree=2) for i in (1, 5, 9, 13)]
    want = [torch.from_numpy(_frame(oracle_mod, rs, sc).out_color).to(cuda_device).view(torch.int32) for rs in cams]
    argsets = [_frame_args(_args(rs, sc, cuda_device)) for rs in cams]
    key = (cuda_device.index, P, W, H)
    monkeypatch.setattr(ext, "_ASYNC_MARGIN", 16)
    for a in argsets:  # learn the hints (synchronous first frame of the key)
        ext.rasterize_gaussians_frame(*a)
    torch.cuda.synchronize()
    monkeypatch.setattr(ext, "_ASYNC_MARGIN", 65536)
    rescued0 = N.lib().gcr_rescue_count()
    streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]
    bad = torch.zeros(240, dtype=torch.int32, device=cuda_device)
    cut = 0
    for i in range(240):
        with torch.cuda.stream(streams[i % 3]):
            if i % 12 == 5:
                ext._ring(N.lib()).harvest()
                monkeypatch.setattr(ext, "_ASYNC_MARGIN", 16)
                ext._capacity_hint[key] = (10, 64)
                cut += 1
            img, _ = ext.rasterize_gaussians_frame(*argsets[i % 4])
            if i % 12 == 5:
                monkeypatch.setattr(ext, "_ASYNC_MARGIN", 65536)
            bad[i] = (img.view(torch.int32) != want[i % 4]).sum()
            del img
    torch.cuda.synchronize()
    assert int(bad.sum()) == 0, bad.nonzero().flatten().tolist()[:10]
    assert N.lib().gcr_rescue_count() - rescued0 == cut == 20
