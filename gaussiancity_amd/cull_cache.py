"""A static scene's cull cache (gcr_gaussians.cull_cache, include/gcr.h ABI v8) kept current for the Python API.

When ONE set of Gaussians is rendered from many poses and most of it is off screen in each of them (BASELINE's C3 / C5
workloads: 5 M / 20 M Gaussians of a city, 95 % culled per frame), every frame's streaming cull reads the mean, the
scales and the rotation of all of them to reject that 95 %.  With a cache -- (mean, rho) as one 16-byte record per
Gaussian, rho = the camera-independent factor of the cull's screen bound, plus (scales, opacity, rotation) as one 32-byte
record -- the cull streams 16 bytes per Gaussian instead of 40 (56 when the rows are [N,14]) and the few per cent that
survive it fetch one 32-byte record instead of a 128-byte line of each of three arrays.  Every output is the same bits
(tests/test_gpu_cull_cache.py); what changes is K1's time (C3 73-79 -> 51-53 us).

Where it does NOT pay.  GaussianCity's own inference loop (scripts/inference.py:208-263, 655-667) makes a NEW point tensor for
every pose out of the points its visibility step found for that pose: nothing is static (the cache would be rebuilt per
frame and steps aside), and everything is on screen.  Forced onto such a mostly-visible [N,14] set the cache costs bytes
instead of saving them -- the stateless kernel reads each 56-byte row once, the cached one reads 16 + 32 bytes of cache
AND the row for the colour: 9 565 -> 8 380 frames/s in bench.py --inference-loop (profiles/r05_bench_static_scene_other.jsonl).

OFF by default -- the reference's API has no notion of a static scene, and the headline numbers are the stateless
path's.  On: `gaussiancity_amd.cull_cache.enable(True)` (or GCR_STATIC_SCENE=1 in the environment, or
InferenceLoop(static_scene=True)).  Only frames that are rendered WITHOUT the backward's state use it (no_grad /
inference frames): a training step changes its Gaussians every iteration, the cache would be rebuilt each time.

Validity.  An entry is keyed on (data_ptr, _version, shape, strides) of every tensor it was built from plus
scale_modifier, and it keeps those tensors alive -- so the address cannot be handed to another allocation while the entry
exists, and any in-place edit through torch (which bumps `_version`) misses the key and rebuilds.  Writes torch's version
counter does not see are not detected -- a raw kernel scribbling into the storage, an in-place operation on `tensor.data`
(which carries a version counter of its own): call invalidate() after such a write.  Tensors without a version counter
(inference-mode tensors) are never cached.  A scene whose tensors are new objects with new storage every frame is
rebuilt every frame; after eight such frames in a row FROM ONE HOST THREAD the cache steps aside for that thread until
invalidate() (the counter is per thread: one caller whose scene is not static does not switch the cache off for the others).

Lifetime across streams.  A cache buffer is read by frames on whatever stream the caller renders on; every use marks
the buffer for that stream (`Tensor.record_stream`), so when an entry is dropped the allocator itself holds the block
back until those streams have passed the frames that read it -- no device-wide wait, nothing is held under the lock.

Memory: 48 bytes per Gaussian per cached scene (240 MB at 5 M, 960 MB at 20 M), at most two scenes, plus the scene's own
tensors, which an entry keeps alive.
"""
import collections
import ctypes as C
import os
import threading

import torch

from . import _native as N

_enabled = os.environ.get("GCR_STATIC_SCENE", "0") not in ("", "0")
_MAX_ENTRIES = 2  # scenes per process that stay cached (a second one: A/B renders of two cities)
_entries = collections.OrderedDict()  # key -> (cache buffer, uint8 [gcr_cull_cache_bytes(P)], tensors kept alive)
_CACHE_BYTES_HOOK = None  # tests without the library: fn(P) -> bytes
_lock = threading.Lock()
stats = {"hits": 0, "builds": 0, "uncacheable": 0, "thrashing": 0}
_THRASH_LIMIT = 8  # that many builds in a row without a hit BY ONE THREAD: its scene is not static; it stops until invalidate()
_epoch = 0  # bumped by invalidate(): a thread's thrash counter of an older epoch starts again

# tests replace this to count / fake builds without a GPU: fn(g: N.Gaussians, scale_modifier, out_ptr, stream) -> None
_build_hook = None


def enable(on=True):
    """Process-wide switch; returns the previous value."""
    global _enabled
    prev, _enabled = _enabled, bool(on)
    if not on:
        invalidate()
    return prev


_tls = threading.local()  # .on: this host thread's override of the process-wide switch (scoped)


def enabled():
    """Does the calling thread's next inference frame use the cache?  (A scoped() block of this thread, else the
    process-wide switch.)"""
    return getattr(_tls, "on", _enabled)


class scoped:
    """`with cull_cache.scoped(True): ...` -- the switch for a block of frames of THIS host thread (other threads keep
    rendering with the process-wide setting); entries stay cached afterwards, a later block on the same scene hits them."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = getattr(_tls, "on", None)
        _tls.on = self.on
        return self

    def __exit__(self, *exc):
        if self.prev is None:
            del _tls.on
        else:
            _tls.on = self.prev
        return False


def invalidate():
    """Drop every entry (e.g. after a write into a cached tensor's storage that torch's version counter cannot see)."""
    with _lock:
        _drop_all()


def _mark_used(cache, device):
    """The frame about to be enqueued on the current stream reads `cache`: the allocator may not hand the block out again
    before that stream has passed it (replaces a device-wide synchronize on every eviction)."""
    if cache.is_cuda:
        cache.record_stream(torch.cuda.current_stream(device))


def _drop_all():
    global _epoch
    _epoch += 1
    _entries.clear()


def _thrash_counter():
    """[count] of this thread's builds in a row without a hit (reset by a hit, by invalidate())."""
    c = getattr(_tls, "builds", None)
    if c is None or c[1] != _epoch:
        c = _tls.builds = [0, _epoch]
    return c


def _key_of(tensors, scale_modifier):
    key = [float(scale_modifier)]
    for t in tensors:
        if t is None or t.numel() == 0:
            key.append(None)
            continue
        try:
            ver = t._version
        except RuntimeError:  # inference tensors do not track versions: cannot tell whether they were edited
            return None
        key.append((t.data_ptr(), ver, tuple(t.shape), tuple(t.stride()), t.device.index))
    return tuple(key)


def attach(g, scale_modifier, tensors, device, stream):
    """Point g.cull_cache (N.Gaussians, already filled for this frame) at the current cache of `tensors` = the tensors g's
    means3D / scales / rotations / cov3D_precomp / opacities pointers came from, building it on `stream` if there is
    none.  Returns the cache buffer (keep it alive until the frame is enqueued) or None when the tensors cannot be keyed."""
    key = _key_of(tensors, scale_modifier)
    if key is None:
        stats["uncacheable"] += 1
        return None
    with _lock:
        hit = _entries.get(key)
        if hit is not None:
            _entries.move_to_end(key)
            stats["hits"] += 1
            _thrash_counter()[0] = 0
            _mark_used(hit[0], device)
            g.cull_cache = hit[0].data_ptr()
            return hit[0]
        builds = _thrash_counter()
        if builds[0] >= _THRASH_LIMIT:  # every frame of this thread brings new tensors: a build per frame costs more than it saves
            stats["thrashing"] += 1
            return None
        builds[0] += 1
        nbytes = _CACHE_BYTES_HOOK(int(g.P)) if _CACHE_BYTES_HOOK is not None else N.lib().gcr_cull_cache_bytes(int(g.P))
        cache = torch.empty((int(nbytes),), dtype=torch.uint8, device=device)  # (torch's blocks are 512-byte aligned)
        if _build_hook is not None:
            _build_hook(g, float(scale_modifier), cache.data_ptr(), stream)
        else:
            N.check(N.lib().gcr_build_cull_cache(C.byref(g), float(scale_modifier), cache.data_ptr(), stream),
                    "gcr_build_cull_cache")
            # frames on OTHER streams will read it (InferenceLoop rotates three): the build is a one-off, wait for it
            torch.cuda.current_stream(device).synchronize()
        stats["builds"] += 1
        while len(_entries) >= _MAX_ENTRIES:
            _entries.popitem(last=False)  # (its block waits in the allocator for the streams that were marked on it)
        _mark_used(cache, device)
        _entries[key] = (cache, tuple(t for t in tensors if t is not None))
        g.cull_cache = cache.data_ptr()
        return cache
