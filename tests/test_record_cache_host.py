"""Host logic of ext._records (no GPU, no kernel call): the per-thread cache of the gcr_camera / gcr_gaussians records of a
frame -- reused only while every tensor of the call is the same object at the same address, never for a call that needed
a .contiguous() copy, never keeping a tensor alive, separate per thread."""
import gc
import threading
import weakref

import torch

from gaussiancity_amd import ext

CPU = torch.device("cpu")


def _call(t, **over):
    kw = dict(device=CPU, P=t["means3D"].size(0), H=48, W=64, background=t["bg"], means3D=t["means3D"], colors=None,
              opacity=t["opacity"], scales=t["scales"], rotations=t["rotations"], scale_modifier=1.0, cov3D_precomp=None,
              viewmatrix=t["view"], projmatrix=t["proj"], tan_fovx=0.5, tan_fovy=0.4, sh=t["sh"], degree=1, campos=t["campos"],
              prefiltered=False, debug=False, for_backward=False, opt=None)
    kw.update(over)
    return ext._records(kw["device"], kw["P"], kw["H"], kw["W"], kw["background"], kw["means3D"], kw["colors"], kw["opacity"],
                        kw["scales"], kw["rotations"], kw["scale_modifier"], kw["cov3D_precomp"], kw["viewmatrix"],
                        kw["projmatrix"], kw["tan_fovx"], kw["tan_fovy"], kw["sh"], kw["degree"], kw["campos"],
                        kw["prefiltered"], kw["debug"], kw["for_backward"], kw["opt"])


def _scene(P=10):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.rand(*s, generator=g)
    return dict(bg=r(3), means3D=r(P, 3), opacity=r(P, 1), scales=r(P, 3), rotations=r(P, 4), view=r(4, 4), proj=r(4, 4),
                sh=r(P, 4, 3), campos=r(3))


def test_records_are_reused_for_the_same_objects_only():
    ext._prep_cache().clear()
    t = _scene()
    cam, g, keep = _call(t)
    assert keep is not None and len(ext._prep_cache()) == 1
    assert g.P == 10 and g.M == 4 and g.means3D == t["means3D"].data_ptr() and cam.img_w == 64 and cam.host_camera == 1
    cam2, g2, keep2 = _call(t)
    assert cam2 is cam and g2 is g and keep2 is None  # a hit: the very records
    # an in-place edit keeps object and address: still a hit (the records hold addresses, the kernels read the values)
    t["opacity"].mul_(0.5)
    assert _call(t)[0] is cam
    # another object with the same values: a miss with records of its own
    t2 = dict(t, opacity=t["opacity"].clone())
    cam3, g3, _ = _call(t2)
    assert cam3 is not cam and g3.opacities == t2["opacity"].data_ptr() and len(ext._prep_cache()) == 2
    # a scalar of the key, the backward flag, absent-as-empty versus absent-as-None
    assert _call(t, tan_fovx=0.51)[0] is not cam
    assert _call(t, for_backward=True)[0].backward == 1 and _call(t)[0] is cam
    assert _call(t, colors=torch.Tensor([]), cov3D_precomp=torch.Tensor([]))[0] is cam  # numel() == 0 is "absent" as well
    # the same object at a NEW address (set_ / resize): rebuilt, never the stale pointer
    old_ptr = t["scales"].data_ptr()
    t["scales"].set_(torch.rand(10, 3))
    assert t["scales"].data_ptr() != old_ptr
    cam4, g4, _ = _call(t)
    assert g4.scales == t["scales"].data_ptr() and cam4 is not cam


def test_a_call_that_needed_a_copy_is_never_cached_and_nothing_is_kept_alive():
    ext._prep_cache().clear()
    t = _scene()
    wide = torch.rand(10, 6)
    t["means3D"] = wide[:, :3]  # strided: the binding copies it, as the reference's .contiguous() does
    cam, g, keep = _call(t)
    assert keep is not None and len(ext._prep_cache()) == 0
    assert g.means3D != t["means3D"].data_ptr()  # the copy's address; `keep` holds the copy for the duration of the call
    # the cache holds weak references only
    t = _scene()
    _call(t)
    assert len(ext._prep_cache()) == 1
    ref = weakref.ref(t["sh"])
    del t
    gc.collect()
    assert ref() is None, "the record cache must not keep a caller's tensor alive"
    # a dead entry is simply never hit again: fresh objects are a new key
    t = _scene()
    assert _call(t)[2] is not None


def test_the_cache_is_per_thread_and_bounded():
    ext._prep_cache().clear()
    t = _scene()
    cam = _call(t)[0]
    seen = []
    th = threading.Thread(target=lambda: seen.append((_call(t)[0] is cam, len(ext._prep_cache()))))
    th.start()
    th.join()
    assert seen == [(False, 1)]  # the other thread built its own records in its own cache
    views = [torch.rand(4, 4) for _ in range(ext._PREP_MAX + 8)]
    for v in views:
        _call(t, viewmatrix=v)
    assert len(ext._prep_cache()) == ext._PREP_MAX
