"""-m gpu: the binning front end of round 6 -- K1's compact survivor records (gcr_layout.geom_vis_rec, ABI v9), the
tile-table kernels' EQUAL cut of the survivor lists (whatever the index order of the scene leaves in each K1 block),
and the band sort of frames with many instances (option "band_sort_min": survivors renumbered by the band of tiles
their rectangle starts in, as 16-byte records while a quarter of P of them fit, as 4-byte positions beyond).  None of
it may change a bit of the frame: everything against the CPU oracle through the native-module surface."""
import numpy as np
import pytest
import torch

import gpu_util as G
import scenes
from test_gpu_async import _args
from test_gpu_parity import _check_forward, _check_grads, _frame

pytestmark = pytest.mark.gpu

ALL_GRADS = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dsh"]


@pytest.fixture
def band_sort(request):
    """band_sort_min for the test: 0 = every frame is band-sorted, -1 = none; the shipping default afterwards."""
    from gaussiancity_amd import _native as N
    prev = N.set_option("band_sort_min", request.param)
    yield request.param
    N.set_option("band_sort_min", prev)


def _city(P, seed, W, H, pose):
    """S-city cut down to P Gaussians: 95 % of them off screen from the orbit, identity rotations, opacity 1."""
    from gaussiancity_amd import synth
    sc = synth.s_city(P, seed, 1)
    wr = scenes.GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=torch.device("cpu"))
    pos, quat = synth.orbit_poses()[pose]
    return sc, wr._get_gaussian_rasterization_settings(pos, quat)._replace(sh_degree=1)


def _k1_grid(P, max_blocks=2048):
    """gcr_preprocess_grid (gcr_internal.h): K1's blocks and the Gaussians per block (= the distance between two lists)."""
    nb = max(1, min((P + 255) // 256, max_blocks))
    c = max(256, (-(-P // nb) + 255) // 256 * 256)
    if c < 512 and P >= 512 * 512:
        c = 512
    return max(1, -(-P // c)), c


def _vis_rec(P, W, H, out):
    """K1's survivor lists and records, decoded: {index: (depth bits, rect_x, rect_y)} over every K1 block."""
    from gaussiancity_amd import _native as N
    R, _, radii, geom, _, _ = out
    L = N.get_layout(P, W, H, R)
    gb = geom.cpu().numpy()
    nblk = (P + 255) // 256 + 1
    counts = gb[L.geom_vis_count:L.geom_vis_count + 4 * nblk].view(np.uint32)
    lists = gb[L.geom_vis_list:L.geom_vis_list + 4 * P].view(np.uint32)
    recs = gb[L.geom_vis_rec:L.geom_vis_rec + 16 * P].view(np.uint32).reshape(P, 4)
    return counts, lists, recs


@pytest.mark.parametrize("band_sort", [0, -1], indirect=True, ids=["band_sorted", "k1_order"])
@pytest.mark.parametrize("order", ["seeded", "morton"])
def test_mostly_culled_scene_in_both_index_orders(oracle_mod, cuda_device, band_sort, order):
    """A city seen from the orbit (a few per cent of the Gaussians survive: the copy holds RECORDS), in the seeded order
    and renumbered along a Morton curve (most K1 blocks empty, a few full: the equal cut crosses hundreds of empty
    lists).  Whole forward state + gradients."""
    P, W, H = 120_000, 640, 368
    sc, rs = _city(P, 301, W, H, 5)
    if order == "morton":
        xy = np.clip(sc["means3D"][:, :2] * 8, 0, 65535).astype(np.uint64)

        def spread(v):
            for sh, m in ((8, 0x00ff00ff), (4, 0x0f0f0f0f), (2, 0x33333333), (1, 0x55555555)):
                v = (v | (v << np.uint64(sh))) & np.uint64(m)
            return v
        perm = np.argsort(spread(xy[:, 0]) | (spread(xy[:, 1]) << np.uint64(1)), kind="stable")
        sc = {k: (np.ascontiguousarray(v[perm]) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    fr = _frame(oracle_mod, rs, sc)
    nvis = int((fr.radii > 0).sum())
    assert 0 < nvis * 4 <= P and fr.R > nvis
    args, out = G.run_forward(rs, sc, cuda_device)
    _check_forward(fr, G.decode(P, W, H, out), P, True)
    # the survivor records say what the 64-byte records say, at the positions of the survivor lists
    counts, lists, recs = _vis_rec(P, W, H, out)
    d = G.decode(P, W, H, out)
    assert int(counts.sum()) == nvis
    nblocks, chunk = _k1_grid(P)
    starts = [b for b in range(nblocks) if counts[b]]
    for b in starts:
        sl = slice(b * chunk, b * chunk + counts[b])
        np.testing.assert_array_equal(recs[sl, 0], lists[sl])
        i = lists[sl]
        np.testing.assert_array_equal(recs[sl, 1], d["depths"][i].view(np.uint32))
        np.testing.assert_array_equal(recs[sl, 2:4], d["rect"][i])
    dpix = np.random.default_rng(3).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device), ALL_GRADS)


@pytest.mark.parametrize("band_sort", [0], indirect=True, ids=["band_sorted"])
def test_mostly_visible_scene_takes_the_position_copy(oracle_mod, cuda_device, band_sort):
    """Every Gaussian on screen: more survivors than a quarter of P, the renumbered copy holds positions.  Long lists
    (the lazy tile sort runs behind the scatter), an inference frame and a training frame."""
    P, W, H = 9000, 320, 272
    rs = scenes.camera(W, H, pose_index=7)._replace(sh_degree=2, bg=torch.tensor((0.1, 0.2, 0.3)))
    # (scales up to 7: lists of up to 1 858 entries, and the gradients stay a factor three inside the 1e-4 bar in every
    # run -- with scales up to 10 this scene is in the fuzz sweep's ill-conditioned tier, tools/fuzz_parity.py, and
    # dL_dcov3D lands between 0.4 and 1.4 times the bar depending on the order of the float atomics)
    sc = scenes.blob_scene(P, 77, 2, smax=7.0)
    fr = _frame(oracle_mod, rs, sc)
    assert int((fr.radii > 0).sum()) * 4 > P
    for train in (False, True):
        args, out = G.run_forward(rs, sc, cuda_device, for_backward=train)
        _check_forward(fr, G.decode(P, W, H, out), P, True)
    dpix = np.random.default_rng(5).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device), ALL_GRADS)


@pytest.mark.parametrize("band_sort", [0], indirect=True, ids=["band_sorted"])
def test_band_sorted_frames_without_the_host_wait_and_their_rescue(oracle_mod, cuda_device, band_sort, monkeypatch):
    """Asynchronous frames: the count kernel records which numbering the tile table was counted in and the scatter --
    the frame's own or the rescue thread's, from another buffer on another stream -- follows it."""
    from gaussiancity_amd import _native as N, ext
    P, W, H = 60_000, 512, 288
    sc, rs = _city(P, 302, W, H, 11)
    fr = _frame(oracle_mod, rs, sc)
    key = (cuda_device.index, P, W, H)
    ext._capacity_hint.pop(key, None)
    a = _args(rs, sc, cuda_device)
    outs = [ext.rasterize_gaussians_ticket(*a) for _ in range(4)]
    torch.cuda.synchronize()
    assert sum(1 for o in outs if o[0].seq != 0) >= 3
    for o in outs:
        assert int(o[0]) == fr.R and not o[0].rescued
        assert np.array_equal(o[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))
    o = outs[-1]
    _check_forward(fr, G.decode(P, W, H, (int(o[0]),) + tuple(o[1:])), P, True)
    monkeypatch.setattr(ext, "_ASYNC_MARGIN", 16)
    ext._capacity_hint[key] = (10, 64)   # a guess of 36 instances: the gate calls the rescue
    rescued0 = N.lib().gcr_rescue_count()
    out = ext.rasterize_gaussians_ticket(*a)
    assert int(out[0]) == fr.R and out[0].rescued
    torch.cuda.synchronize()
    assert N.lib().gcr_rescue_count() == rescued0 + 1
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32))


def test_the_default_threshold_sorts_big_frames_only(cuda_device):
    """The shipping default: frames whose capacity guess stays below "band_sort_min" instances are not band-sorted
    (two launches that only pay when a (group, tile) pair holds several instances)."""
    from gaussiancity_amd import _native as N
    assert N.get_option("band_sort_min") == 6_000_000
    assert N.set_option("band_sort_min", -7) == 6_000_000 and N.get_option("band_sort_min") == -1
    N.set_option("band_sort_min", 6_000_000)
