"""CPU: the bookkeeping of gaussiancity_amd/cull_cache.py (when a static scene's cull cache is reused, rebuilt, dropped).
The build itself is a HIP kernel (tests/test_gpu_cull_cache.py); here a hook stands in for it."""
import pytest
import torch

from gaussiancity_amd import _native as N
from gaussiancity_amd import cull_cache as cc


@pytest.fixture
def builds(monkeypatch):
    calls = []
    monkeypatch.setattr(cc, "_build_hook", lambda g, mod, out, stream: calls.append((int(g.P), mod, out)))
    monkeypatch.setattr(cc, "_CACHE_BYTES_HOOK", lambda P: P * 48 + 128)
    cc.invalidate()
    prev = cc.enable(True)
    yield calls
    cc.enable(prev)
    cc.invalidate()


def _scene(P=64):
    return torch.randn(P, 3), torch.rand(P, 3), torch.randn(P, 4), torch.rand(P, 1)


def _attach(m, s, r, o, mod=1.0, cov=None):
    g = N.Gaussians(int(m.size(0)), 0, m.data_ptr(), o.data_ptr(), None, None, s.data_ptr() if s is not None else None,
                    r.data_ptr() if r is not None else None, cov.data_ptr() if cov is not None else None)
    c = cc.attach(g, mod, (m, s, r, cov, o), torch.device("cpu"), None)
    return g, c


def test_hit_rebuild_on_edit_and_on_another_modifier(builds):
    m, s, r, o = _scene()
    g, c = _attach(m, s, r, o)
    assert len(builds) == 1 and g.cull_cache == c.data_ptr() and c.numel() == 64 * 48 + 128
    g2, c2 = _attach(m, s, r, o)
    assert len(builds) == 1 and c2 is c                       # same tensors, same versions: a hit
    s.mul_(2.0)                                                # in-place edit: the version counter moves
    g3, c3 = _attach(m, s, r, o)
    assert len(builds) == 2 and c3 is not c
    _attach(m, s, r, o, mod=2.0)                               # rho holds scale_modifier
    assert len(builds) == 3
    m[0, 0] = 5.0                                              # ... the means are part of the record
    _attach(m, s, r, o, mod=2.0)
    assert len(builds) == 4
    o.clamp_(0.1, 0.9)                                         # ... and so are the opacities
    _attach(m, s, r, o, mod=2.0)
    assert len(builds) == 5


def test_views_of_one_point_tensor_share_its_version(builds):
    pts = torch.randn(128, 14)
    views = lambda: (pts[:, :3], pts[:, 4:7], pts[:, 7:11], pts[:, 3:4])  # new view objects every frame, as the wrapper makes them
    _attach(*views())
    _attach(*views())
    assert len(builds) == 1
    pts[:, 4:7] *= 3.0                                         # an edit through ANY view of the storage
    _attach(*views())
    assert len(builds) == 2


def test_entries_keep_their_tensors_alive_and_are_bounded(builds):
    import weakref
    scenes = [_scene() for _ in range(3)]
    refs = [weakref.ref(sc[1]) for sc in scenes]
    for sc in scenes:
        _attach(*sc)
    assert len(builds) == 3 and len(cc._entries) == cc._MAX_ENTRIES
    del scenes, sc
    alive = [r() is not None for r in refs]
    assert alive == [False, True, True]                        # the evicted scene is released, the cached ones are pinned
    cc.invalidate()
    assert [r() is not None for r in refs] == [False, False, False]


def test_a_scene_that_changes_every_frame_stops_being_cached(builds):
    for i in range(cc._THRASH_LIMIT + 5):
        g, c = _attach(*_scene())
    assert len(builds) == cc._THRASH_LIMIT and c is None and not g.cull_cache
    assert cc.stats["thrashing"] >= 5
    cc.invalidate()
    g, c = _attach(*_scene())
    assert c is not None


def test_a_thrashing_thread_does_not_switch_the_cache_off_for_the_others(builds):
    """the builds-in-a-row counter is the calling thread's.  One thread whose scene is new every frame stops being
    served; a thread with a static scene keeps its hits."""
    import threading
    done = []

    def thrasher():
        for i in range(cc._THRASH_LIMIT + 3):
            g, c = _attach(*_scene())
        done.append(c is None)

    th = threading.Thread(target=thrasher)
    th.start(); th.join()
    assert done == [True] and cc.stats["thrashing"] >= 3
    scene = _scene()
    g, c1 = _attach(*scene)
    g, c2 = _attach(*scene)
    assert c1 is not None and c2 is c1  # this thread: a build, then a hit


def test_inference_tensors_are_never_cached(builds):
    with torch.inference_mode():
        m, s, r, o = _scene()
    g, c = _attach(m, s, r, o)
    assert c is None and not g.cull_cache and builds == []


def test_scoped_switch_is_the_calling_threads():
    import threading
    prev = cc.enable(False)
    try:
        seen = []
        with cc.scoped(True):
            assert cc.enabled()
            th = threading.Thread(target=lambda: seen.append(cc.enabled()))
            th.start(); th.join()
            with cc.scoped(False):
                assert not cc.enabled()
            assert cc.enabled()
        assert not cc.enabled() and seen == [False]
    finally:
        cc.enable(prev)
