"""-m gpu: option "stream_policy" (include/gcr.h) -- the cache policy of a frame's one-pass streams.  The non-temporal
instantiation of the streaming cull (gcr_preprocess.hip "NT"), its non-temporal zeros for culled Gaussians and the
non-temporal per-pixel stores of an inference frame (gcr_blend.hip `nt_out`) are HINTS to the memory system: every
policy must leave the same bits.  Forced on (1), forced off (0) and automatic (-1: on for the entry points that wait
for num_rendered, off for asynchronous frames), each against the CPU oracle."""
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
def stream_policy(request):
    from gaussiancity_amd import _native as N
    prev = N.set_option("stream_policy", request.param)
    yield request.param
    N.set_option("stream_policy", prev)


def _city(P, seed, W, H, pose, deg=1):
    """S-city cut down to P Gaussians seen from the orbit: most of them leave through the streaming cull (whose stores
    of radii = 0 are the non-temporal ones), a few per cent through the exact pass."""
    from gaussiancity_amd import synth
    sc = synth.s_city(P, seed, deg)
    wr = scenes.GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=torch.device("cpu"))
    pos, quat = synth.orbit_poses()[pose]
    return sc, wr._get_gaussian_rasterization_settings(pos, quat)._replace(sh_degree=deg)


def test_the_option_is_process_wide_and_three_valued():
    from gaussiancity_amd import _native as N
    assert N.get_option("stream_policy") == -1
    assert N.set_option("stream_policy", 7) == -1 and N.get_option("stream_policy") == 1
    assert N.set_option("stream_policy", 0) == 1 and N.get_option("stream_policy") == 0
    assert N.set_option("stream_policy", -5) == 0 and N.get_option("stream_policy") == -1


@pytest.mark.parametrize("stream_policy", [1, 0, -1], indirect=True, ids=["non_temporal", "default_policy", "automatic"])
def test_synchronous_frames_inference_and_training(oracle_mod, cuda_device, stream_policy):
    """ext.rasterize_gaussians (gcr_forward: the caller waits for num_rendered): a mostly culled scene, the whole forward
    state of an inference frame (non-temporal per-pixel stores when the policy says so) and of a training frame (never),
    then the gradients."""
    P, W, H = 150_000, 640, 368
    sc, rs = _city(P, 411, W, H, 7)
    fr = _frame(oracle_mod, rs, sc)
    nvis = int((fr.radii > 0).sum())
    assert 0 < nvis * 4 <= P and fr.R > nvis
    for train in (False, True):
        args, out = G.run_forward(rs, sc, cuda_device, for_backward=train)
        _check_forward(fr, G.decode(P, W, H, out), P, True)
    dpix = np.random.default_rng(9).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(fr.backward(dpix), G.run_backward(args, out, dpix, cuda_device), ALL_GRADS)


@pytest.mark.parametrize("stream_policy", [1, 0], indirect=True, ids=["non_temporal", "default_policy"])
def test_precomputed_covariance_and_colours(oracle_mod, cuda_device, stream_policy):
    """The other instantiation of the stateless fused kernel (six covariance floats per Gaussian in the stream)."""
    P, W, H = 20_000, 333, 128
    rs = scenes.camera(W, H, pose_index=3)._replace(bg=torch.tensor((0.2, 0.4, 0.1)))
    sc = scenes.blob_scene(P, 83, 0, spread=70.0)
    # the covariance the oracle derives from scale / rotation, fed back precomputed (tests/test_gpu_parity.py)
    cov3D = _frame(oracle_mod, rs, sc, use_sh=False).cov3D[:P].copy()
    fr = _frame(oracle_mod, rs, sc, use_sh=False, cov3D=cov3D)
    assert 0 < int((fr.radii > 0).sum()) < P
    args, out = G.run_forward(rs, sc, cuda_device, use_sh=False, use_cov3d=True, cov3D=cov3D, for_backward=False)
    _check_forward(fr, G.decode(P, W, H, out), P, False, has_cov3d_state=False)


@pytest.mark.parametrize("stream_policy", [1, -1], indirect=True, ids=["non_temporal", "automatic"])
def test_asynchronous_frames_over_three_streams(oracle_mod, cuda_device, stream_policy):
    """rasterize_gaussians_ticket (gcr_forward_async): nine frames of three poses on three streams, nothing synchronised
    in between -- forced on, the culls of neighbouring frames read the same arrays with the non-temporal policy side by
    side; automatic leaves asynchronous frames on the default policy.  Images and num_rendered against the oracle."""
    from gaussiancity_amd import ext
    P, W, H = 60_000, 480, 272
    sc, rs0 = _city(P, 523, W, H, 2, deg=2)
    from gaussiancity_amd import synth
    wr = scenes.GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=torch.device("cpu"))
    cams = [wr._get_gaussian_rasterization_settings(*synth.orbit_poses()[p])._replace(sh_degree=2) for p in (2, 10, 18)]
    frames = [_frame(oracle_mod, rs, sc) for rs in cams]
    streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]
    argsets = [_args(rs, sc, cuda_device) for rs in cams]
    torch.cuda.synchronize()
    outs = []
    for i in range(9):
        with torch.cuda.stream(streams[i % 3]):
            outs.append(ext.rasterize_gaussians_ticket(*argsets[i % 3], _for_backward=False))
    torch.cuda.synchronize()
    for i, (ticket, color, radii, _, _, _) in enumerate(outs):
        fr = frames[i % 3]
        assert int(ticket) == fr.R
        np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
        assert np.array_equal(color.cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32)), "frame %d" % i
