"""Native-module surface: the three functions of `diff_gaussian_rasterization_ext`.

Mirrors the reference's pybind module (dgr/bindings.cpp:15-19) and its torch binding
(dgr/rasterize_points.cu:37-173): same names, positional argument order, return tuples and
tensor shapes -- but the work is done by libgcr_hip.so (hand-written gfx950 kernels) through
the C ABI of include/gcr.h.  PyTorch only supplies device memory and the current HIP stream.

Differences, all on the safe side:
  * non-float32 / non-GPU inputs raise RuntimeError instead of being undefined behaviour;
  * kernels run on torch's *current* stream, not the legacy default stream, so DDP's
    side-stream bucket all-reduce orders correctly against them;
  * scratch buffers live on means3D's device (the reference uses "current device").
"""
import collections
import ctypes as C
import os
import threading
import weakref

import torch

from . import _native as N
from . import cull_cache as _cull

NUM_CHANNELS = 3  # cr/config.h:15

# num_rendered of the last frame per (device, P, W, H): the next frame's binning buffer is
# allocated for 1.5x that before the frame starts, which lets gcr_forward enqueue the whole frame
# without a mid-frame host stall.  A wrong guess only costs the staged path for that frame.
# Bounded (least recently used first out): a training loop whose point count changes every step
# would otherwise grow the table without limit.  One table per process, read and written under a lock: the frame
# loop may be driven by several host threads (bench.py --host-threads, nn.DataParallel-style callers, core/test.py:41).
_CAPACITY_HINT_MAX = 64
_capacity_hint = collections.OrderedDict()
_hint_lock = threading.Lock()


# Debug aid (the GPU tests switch it on): hand gcr_backward NaN-filled outputs instead of whatever the caching
# allocator returns, so that an element the library failed to write cannot pass for a zero gradient.
poison_outputs = False


# Per host thread: "a backward call will follow the next rasterize_gaussians()" for callers that cannot pass the keyword
# (set_backward_hint; None / never set = True, the reference's contract), the per-call options, the ticket ring.
# gcr_camera.backward never changes an image: it says whether the forward blend leaves the backward's per-piece state.
_tls = threading.local()


def set_backward_hint(flag):
    _tls.backward = bool(flag)


def _hint_get(key):
    """(num_rendered, longest tile list) of the most recent resolved frame with this key, or (0, 0)."""
    with _hint_lock:
        v = _capacity_hint.get(key)
        if v is None:
            return 0, 0
        _capacity_hint.move_to_end(key)
        return v


def _hint_put(key, value):
    with _hint_lock:
        _capacity_hint[key] = value
        _capacity_hint.move_to_end(key)
        while len(_capacity_hint) > _CAPACITY_HINT_MAX:
            _capacity_hint.popitem(last=False)


# ---- per-call options (gcr_options; ABI v6, fields as of v7) ---------------------------------------------------------------------------
class options:
    """`with ext.options(bwd_wave_units=1, bwd_piece=64): ...` -- the native calls of THIS thread inside the block carry these
    gcr_options (include/gcr.h); other threads, and the process-wide defaults of gcr_set_option, are untouched.  The
    forward and the backward of a frame must run under the same options."""

    def __init__(self, **kw):
        self.opt = N.Options(**kw)
        self.prev = None

    def __enter__(self):
        self.prev = getattr(_tls, "options", None)
        _tls.options = self.opt
        return self

    def __exit__(self, *a):
        _tls.options = self.prev
        return False


def _current_options():
    return getattr(_tls, "options", None)


# ---- asynchronous frames: num_rendered as a ticket (gcr_forward_async) ------------------------------------------------
# The reference returns num_rendered as a Python int, which blocks its caller in every frame until K1 + the scan have
# run (cr/rasterizer_impl.cu:236-238) -- although nothing in GaussianCity's Python looks at the number except the
# backward of the same frame (dgr/__init__.py:404-420).  The autograd functions of gaussiancity_amd.rasterizer therefore
# take a FrameTicket instead: the whole frame is enqueued, the call returns, and the number is read from a pinned host
# word when (if) somebody asks for it -- int(ticket), or the backward.  rasterize_gaussians() itself keeps the reference's
# contract and returns an int.
_RING = 32  # asynchronous frames a host thread may have outstanding (the ring's oldest ticket is waited for first)
# capacity guess of an asynchronous frame from the newest num_rendered this thread has seen for the same (device, P, W,
# H): the host runs ahead of the device, so that number may be a few frames old -- twice it, plus a constant
_ASYNC_FACTOR, _ASYNC_MARGIN = 2, 65536
_SYNC_ONLY = os.environ.get("GCR_EXT_SYNC") == "1"  # measurement aid: keep the reference's host wait in every frame
_guess_hook = None  # fault injection (tests/test_gpu_async.py): callable(key, capacity) -> capacity of the next asynchronous frame


class FrameTicket:
    """Lazy num_rendered of one frame.  int(ticket) / ticket.wait() block until the device has published it (and, for a
    frame that overflowed its capacity guess, until the library's rescue has re-rendered it); .done() never blocks.
    A frame that FAILED (overflow of the 32-bit instance index, a rescue that failed or was not there in time) raises
    from ITS OWN wait() / int() -- every time -- and from nowhere else: the ring and the hint table treat it as
    resolved, so one lost frame does not take the thread's later frames with it."""

    __slots__ = ("words", "addr", "seq", "capacity", "stream", "key", "stateful", "_R", "_longest", "_lib", "_err")

    def __init__(self, lib, words, addr, seq, capacity, stream, key, stateful, R=None, longest=0):
        self._lib, self.words, self.addr, self.seq, self.capacity = lib, words, addr, seq, capacity
        self.stream, self.key, self.stateful = stream, key, stateful
        self._R, self._longest, self._err = R, longest, None

    def _resolve(self, block):
        """True once the ticket is resolved (successfully or not); never raises."""
        if self._R is not None or self._err is not None:
            return True
        try:
            v = self.words[0]
            if (v >> 32) != self.seq or (v & 0xFFFFFFFF) > self.capacity:  # not yet / overflow: the library's protocol
                info = N.FrameInfo()
                if block:
                    rc = N.check(self._lib.gcr_ticket_wait(self.addr, self.seq, self.capacity, self.stream, C.byref(info)),
                                 "gcr_ticket_wait")
                else:
                    rc = N.check(self._lib.gcr_ticket_poll(self.addr, self.seq, self.capacity, C.byref(info)),
                                 "gcr_ticket_poll")
                if rc != 0:
                    return False
                self._R, self._longest = int(info.num_rendered), int(info.max_tile_instances)
            else:
                if (v & 0xFFFFFFFF) > 0x7FFFFFFF:
                    raise RuntimeError("num_rendered exceeds 2^31-1 (32-bit instance index, as in the reference)")
                self._R, self._longest = int(v & 0xFFFFFFFF), int(self.words[1])
        except RuntimeError as e:
            self._err = e
        if self._err is None and self.key is not None and self._R > 0:
            _hint_put(self.key, (self._R, self._longest))
        self.words = None  # the ring slot may be reused
        return True

    def done(self):
        return self._resolve(False)

    @property
    def failed(self):
        """The frame's error once the ticket is resolved (None: rendered, or not resolved yet)."""
        return self._err

    def wait(self):
        self._resolve(True)
        if self._err is not None:
            raise RuntimeError(str(self._err))
        return self._R

    @property
    def rescued(self):
        """True when the frame did not fit its binning buffer: the image is right (the library re-rendered it), the
        binning state in the caller's buffer is not -- a backward re-bins first (rasterize_*_backward do)."""
        return self.wait() > self.capacity

    __int__ = wait
    __index__ = wait

    def __repr__(self):
        return "FrameTicket(%s)" % ("failed" if self._err is not None else self._R if self._R is not None else "pending")


class _TicketRing:
    """Per host thread: _RING sets of eight pinned host words (gcr_host_words_alloc) handed out round-robin.  The block
    is given back when the thread ends (the ring lives in a threading.local): what is still unresolved then is waited
    for first, so neither a device store nor the library's rescue thread meets unmapped memory."""

    def __init__(self, lib):
        self.lib = lib
        self.base = lib.gcr_host_words_alloc(N.TICKET_WORDS * _RING)
        if not self.base:
            raise RuntimeError("gcr_host_words_alloc failed (pinned host memory for the frame tickets)")
        self.words = [(C.c_uint64 * N.TICKET_WORDS).from_address(self.base + 8 * N.TICKET_WORDS * i) for i in range(_RING)]
        self.tickets = [None] * _RING
        self.pending = collections.deque()  # unresolved tickets, oldest first
        self.next = 0
        self.seq = 0

    def take(self):
        i = self.next
        self.next = (i + 1) % _RING
        old = self.tickets[i]
        if old is not None:
            old._resolve(True)  # back-pressure: at most _RING frames of this thread are unresolved (its error stays its own)
        self.seq = (self.seq % 0xFFFFFFFE) + 1  # never 0
        return i, self.words[i], self.base + 8 * N.TICKET_WORDS * i, self.seq

    def harvest(self):
        """Resolve what has been published, oldest first, without blocking: keeps the capacity hints fresh."""
        p = self.pending
        while p and p[0].done():
            p.popleft()

    def close(self):
        base, self.base = self.base, None
        if not base:
            return
        try:
            for t in self.tickets:
                if t is not None:
                    t._resolve(True)
            self.lib.gcr_host_words_free(base)
        except Exception:  # noqa: BLE001  (interpreter shutdown: the process is going away with the block)
            pass

    def __del__(self):
        self.close()


def _ring(lib):
    r = getattr(_tls, "ring", None)
    if r is None or r.base is None:
        r = _tls.ring = _TicketRing(lib)
    return r


def _dev_f32(t, name, device):
    """Return (tensor_or_None, data_ptr_or_None); numel()==0 is the reference's 'absent'."""
    if t is None or t.numel() == 0:
        return None, None
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    if t.device != device:
        raise RuntimeError("%s must live on %s (got %s)" % (name, device, t.device))
    if not t.is_contiguous():
        t = t.contiguous()
    return t, t.data_ptr()


def _camera(device, bg, view, proj, campos, tan_fovx, tan_fovy, H, W, scale_modifier, degree,
            prefiltered, debug, for_backward=False, flip_x=False, flip_y=False, window=None, out_u8=False):
    """gcr_camera.  The four camera tensors are device tensors as in the reference -- or ALL four CPU tensors
    (GaussianRasterizerWrapper(host_camera=True)): then the library copies the 38 floats into its kernels' arguments
    (gcr_camera.host_camera) and no device copy of the camera exists at all."""
    cams = ((bg, "bg", 3), (view, "viewmatrix", 16), (proj, "projmatrix", 16), (campos, "campos", 3))
    on_host = all(isinstance(t, torch.Tensor) and t.device.type == "cpu" and t.numel() != 0 for t, _, _ in cams)
    keep = []
    ptrs = []
    for t, name, n in cams:
        if on_host:
            if t.dtype != torch.float32:
                raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
            tt = t.contiguous()
            p = tt.data_ptr()
        else:
            tt, p = _dev_f32(t, name, device)
        if tt is None or tt.numel() != n:
            raise RuntimeError("%s must have %d elements" % (name, n))
        keep.append(tt)
        ptrs.append(p)
    cam = N.Camera(int(H), int(W), float(tan_fovx), float(tan_fovy), float(scale_modifier),
                   int(degree), int(bool(prefiltered)), int(bool(debug)), *ptrs)
    cam.host_camera = int(on_host)
    cam.flip_x = int(bool(flip_x))
    cam.flip_y = int(bool(flip_y))
    cam.backward = int(bool(for_backward))
    cam.out_u8 = int(bool(out_u8))
    opt = _current_options()
    if opt is not None:
        cam.options = C.pointer(opt)
        keep.append(opt)
    if window is not None:  # (x, y, w, h) of the mirrored image: gcr_camera.win_*
        cam.win_x, cam.win_y, cam.win_w, cam.win_h = (int(v) for v in window)
    return cam, keep


def _gaussians(device, P, means3D, opacity, sh, colors, scales, rotations, cov3D_precomp):
    keep = []
    ptr = {}
    for t, name in ((means3D, "means3D"), (opacity, "opacity"), (sh, "sh"), (colors, "colors"),
                    (scales, "scales"), (rotations, "rotations"),
                    (cov3D_precomp, "cov3D_precomp")):
        tt, p = _dev_f32(t, name, device)
        keep.append(tt)
        ptr[name] = p
    M = 0
    if sh is not None and sh.numel() != 0:  # dgr/rasterize_points.cu:72-75
        if sh.dim() != 3 or sh.size(0) != P or sh.size(2) != 3:
            raise RuntimeError("sh must have dimensions (num_points, M, 3)")
        M = int(sh.size(1))
    for t, name, last in ((opacity, "opacity", None), (colors, "colors", 3), (scales, "scales", 3),
                          (rotations, "rotations", 4), (cov3D_precomp, "cov3D_precomp", 6)):
        if t is not None and t.numel() != 0 and t.numel() != P * (last or 1):
            raise RuntimeError("%s has %d elements, expected %d" % (name, t.numel(), P * (last or 1)))
    g = N.Gaussians(int(P), M, ptr["means3D"], ptr["opacity"], ptr["sh"], ptr["colors"],
                    ptr["scales"], ptr["rotations"], ptr["cov3D_precomp"])
    return g, keep


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    # the raw getter skips the Stream object round trip (several microseconds per frame on the host)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (the guard costs microseconds per frame)."""

    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)
        return False


_size_cache = {}  # (P,) / (W, H) -> gcr_geometry_bytes / gcr_image_bytes: pure functions, two ctypes calls saved per frame

# What the backward must know about the frame it is handed and the reference's signature has no room for: was the
# forward rendered WITH the backward's state (gcr_camera.backward), and did it fit its binning buffer?  Keyed by the
# geometry buffer's address -- the buffer is alive from the forward to the backward of a frame, so its entry is this
# frame's; a later forward that gets the same block overwrites it.  Bounded, locked.
_FRAME_META_MAX = 512
_frame_meta = collections.OrderedDict()
_meta_lock = threading.Lock()


def _meta_put(geom, stateful, ticket_or_R, longest):
    with _meta_lock:
        _frame_meta[geom.data_ptr()] = (stateful, ticket_or_R, longest)
        _frame_meta.move_to_end(geom.data_ptr())
        while len(_frame_meta) > _FRAME_META_MAX:
            _frame_meta.popitem(last=False)


def _meta_get(geom):
    with _meta_lock:
        return _frame_meta.get(geom.data_ptr())


def _forward(L, device, cam, g, P, H, W, ticket=False):
    """One frame with torch-owned buffers; returns the reference's six-tuple.

    ticket=False: gcr_forward (+ the staged retry) -- the first element is num_rendered as an int, which costs the one
    host wait of the frame (the reference's contract).  ticket=True: gcr_forward_async -- the first element is a
    FrameTicket and the call returns as soon as the frame is enqueued; the first frames of a (device, P, W, H) key, whose
    num_rendered cannot be guessed yet, and the debug / force_radix modes take the synchronous path and come back with
    a resolved ticket."""
    byte = dict(dtype=torch.uint8, device=device)
    stateful = cam.backward == 1
    # every pixel / every radius is written by the kernels, so no zero-fill launches are needed
    oh, ow = (cam.win_h, cam.win_w) if cam.win_w else (H, W)
    if cam.out_u8:  # the video frame itself (gcr_camera.out_u8)
        out_color = torch.empty((oh, ow, NUM_CHANNELS), dtype=torch.uint8, device=device)
    else:
        out_color = torch.empty((NUM_CHANNELS, oh, ow), dtype=torch.float32, device=device)
    radii = torch.empty((P,), dtype=torch.int32, device=device)
    stream = _stream(device)
    gbytes = _size_cache.get(P)
    if gbytes is None:
        gbytes = _size_cache[P] = L.gcr_geometry_bytes(P)
    ibytes = _size_cache.get((W, H))
    if ibytes is None:
        ibytes = _size_cache[(W, H)] = L.gcr_image_bytes(W, H)
    if len(_size_cache) > 256:
        _size_cache.clear()
    geom = torch.empty((gbytes,), **byte)
    img = torch.empty((ibytes,), **byte)
    key = (device.index, P, W, H)
    ring = None
    if ticket and _SYNC_ONLY:
        ticket = None  # A/B switch: every frame through the synchronous entry point, tickets resolved on return
    if ticket:
        ring = _ring(L)
        ring.harvest()
    R_seen, list_cap = _hint_get(key)
    nbytes = L.gcr_binning_bytes if stateful else L.gcr_binning_bytes_lean
    opt = _current_options()
    radix = opt.force_radix if (opt is not None and opt.force_radix >= 0) else N.get_option("force_radix")
    if ticket and R_seen > 0 and not cam.debug and not radix:
        # The guess is made from a frame that may be several frames old (the host runs ahead of the device): twice its
        # num_rendered.  A frame that still does not fit is rendered correctly by the library's rescue (include/gcr.h).
        capacity = _ASYNC_FACTOR * R_seen + _ASYNC_MARGIN
        if _guess_hook is not None:
            capacity = max(1, int(_guess_hook(key, capacity)))
        binning = torch.empty((nbytes(capacity, W, H),), **byte)
        slot, words, addr, seq = ring.take()
        N.check(L.gcr_forward_async(C.byref(cam), C.byref(g), geom.data_ptr(), gbytes, binning.data_ptr(), binning.numel(),
                                    capacity, list_cap, img.data_ptr(), ibytes, radii.data_ptr(), out_color.data_ptr(),
                                    addr, seq, stream),
                "gcr_forward_async")
        t = FrameTicket(L, words, addr, seq, capacity, stream, key, stateful)
        ring.tickets[slot] = t
        ring.pending.append(t)
        _meta_put(geom, stateful, t, list_cap)
        return t, out_color, radii, geom, binning, img
    info = N.FrameInfo()
    capacity = R_seen + R_seen // 2 + 4096 if R_seen > 0 else 0
    binning = torch.empty((nbytes(capacity, W, H) if capacity else 0,), **byte)
    rc = N.check(L.gcr_forward(C.byref(cam), C.byref(g), geom.data_ptr(), gbytes,
                               binning.data_ptr() if capacity else None, binning.numel(), capacity,
                               list_cap, img.data_ptr(), ibytes, radii.data_ptr(), out_color.data_ptr(),
                               C.byref(info), stream),
                 "gcr_forward")
    R = int(info.num_rendered)
    if rc == 1:  # GCR_RETRY_RENDER: no/too small a guess, or a tile list beyond the LDS sort
        capacity = R
        binning = torch.empty((nbytes(R, W, H),), **byte)
        N.check(L.gcr_forward_render(C.byref(cam), C.byref(g), geom.data_ptr(), gbytes,
                                     binning.data_ptr(), binning.numel(), img.data_ptr(),
                                     ibytes, C.byref(info), out_color.data_ptr(), stream),
                "gcr_forward_render")
    longest = int(info.max_tile_instances)
    _hint_put(key, (R, longest))  # the library adds its own margin to the longest list
    _meta_put(geom, stateful, R, longest)
    if ticket is not False:
        R = FrameTicket(L, None, None, 0, max(capacity, R), stream, None, stateful, R=R, longest=longest)
    return R, out_color, radii, geom, binning, img


def _state_for_backward(L, device, cam, g, geom, binning, img, R, W, H):
    """What gcr_backward needs besides the arguments of the reference: num_rendered as an int, and a binning buffer that
    holds the forward blend's per-piece state.  A frame that was rendered as an inference frame (gcr_camera.backward
    == 0) or that overflowed its capacity guess (FrameTicket.rescued) is binned again, state only, into an exactly
    sized buffer -- rare paths, both pinned by tests.  Returns (R, binning)."""
    meta = _meta_get(geom)
    stateful, longest = True, 0
    if isinstance(R, FrameTicket):
        t, R = R, R.wait()
        stateful, longest = t.stateful and not t.rescued, t._longest
    elif meta is not None and not isinstance(meta[1], FrameTicket) and int(meta[1]) == int(R):
        stateful, longest = meta[0], meta[2]
    elif meta is not None and isinstance(meta[1], FrameTicket) and meta[1].wait() == int(R):
        stateful, longest = meta[0] and not meta[1].rescued, meta[1]._longest
    R = int(R)
    if stateful or R == 0:
        return R, binning
    binning = torch.empty((L.gcr_binning_bytes(R, W, H),), dtype=torch.uint8, device=device)
    info = N.FrameInfo(R, int(longest))
    N.check(L.gcr_forward_render(C.byref(cam), C.byref(g), geom.data_ptr(), geom.numel(), binning.data_ptr(),
                                 binning.numel(), img.data_ptr(), img.numel(), C.byref(info), None, _stream(device)),
            "gcr_forward_render (state only)")
    return R, binning



def _empty_frame(device, H, W):
    e = torch.empty((0,), dtype=torch.uint8, device=device)  # dgr/rasterize_points.cu:71: zero image, nothing rendered
    return (0, torch.zeros((NUM_CHANNELS, H, W), dtype=torch.float32, device=device),
            torch.zeros((0,), dtype=torch.int32, device=device), e, e.clone(), e.clone())


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                        image_width, sh, degree, campos, prefiltered, debug, _for_backward=None, _ticket=False):
    """RasterizeGaussiansCUDA (dgr/rasterize_points.cu:37-93, dgr/rasterize_points.h:18-28).

    Returns (num_rendered:int, out_color[3,H,W], radii[P] int32, geomBuffer, binningBuffer,
    imgBuffer) -- the three buffers are opaque uint8 tensors to be handed back to
    rasterize_gaussians_backward.

    `_for_backward` (keyword, not part of the reference's positional signature above) never changes a result: True
    (the default -- whoever calls this function may call the backward next, as the reference's autograd function does) =
    the forward blend leaves the backward's per-piece state behind (gcr_camera.backward); False = an inference frame,
    which does not pay for a backward it never runs (rasterize_gaussians_backward still works on such a frame: it bins
    it again, state only).  None = what set_backward_hint() announced for this thread, else True.
    """
    if _for_backward is None:  # the reference's contract: whoever calls this may call the backward next
        _for_backward = getattr(_tls, "backward", None)
        _for_backward = True if _for_backward is None else _for_backward
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor: this rasterizer has no CPU path")
    L = N.lib()
    device = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if P == 0:
        e = _empty_frame(device, H, W)
        return ((FrameTicket(L, None, None, 0, 0, None, None, False, R=0),) + e[1:]) if _ticket else e
    with _on_device(device):
        if _cull.enabled() and not _for_backward:  # a static scene's cull cache: same bits, fewer bytes (cull_cache.py)
            cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx,
                                  tan_fovy, H, W, scale_modifier, degree, prefiltered, debug, _for_backward)
            g, keep_g = _gaussians(device, P, means3D, opacity, sh, colors, scales, rotations,
                                   cov3D_precomp)
            keep = (keep_c, keep_g, _cull.attach(g, scale_modifier, (means3D, scales, rotations, cov3D_precomp, opacity),
                                                 device, _stream(device)))
        else:
            cam, g, keep = _records(device, P, H, W, background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                    cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, sh, degree, campos,
                                    bool(prefiltered), bool(debug), bool(_for_backward), _current_options())
        out = _forward(L, device, cam, g, P, H, W, _ticket)
        del keep
    return out


def rasterize_gaussians_ticket(*args, _for_backward=False):
    """rasterize_gaussians() without the host wait: same arguments, same six-tuple, but the first element is a
    FrameTicket (int(ticket) blocks until the number is known) and the call returns as soon as the frame is enqueued.
    What gaussiancity_amd.rasterizer's autograd functions call: the reference's Python never looks at num_rendered
    except in the backward of the same frame (dgr/__init__.py:85-108)."""
    return rasterize_gaussians(*args, _for_backward=_for_backward, _ticket=True)


# ---- inference frames through the Python API: (image, radii) and nothing else ------------------------------------------
# GaussianRasterizer.forward hands its caller the image and the radii (dgr/__init__.py:262-273); when no input asks for
# a gradient the three state buffers never leave the call.  Such a frame is a host matter as much as a device one --
# the calling thread needs ~63 us to get a C3 frame from the module call to the six launches (tools/host_profile_api.py),
# a third of the frame's period, and the first frames behind every synchronize are spaced by exactly that -- so this
# entry point does what rasterize_gaussians_ticket does with less of it:
#   * the gcr_camera / gcr_gaussians records are kept per thread while a call's tensors are THE SAME OBJECTS at the same
#     addresses as the previous call's with that key (weak references: nothing is kept alive; a copy made by
#     .contiguous() is never cached -- it would be a snapshot);
#   * geometry, image and binning state are carved from ONE allocation;
#   * no autograd node (rasterizer.GaussianRasterizer.forward decides that), no frame-meta entry for a backward that
#     cannot come.
# Same library call, same kernels, same bits.
_PREP_MAX = 64


class _Prepared:
    __slots__ = ("refs", "ptrs", "cam", "g", "device", "P")


def _prep_cache():
    c = getattr(_tls, "prep", None)
    if c is None:
        c = _tls.prep = collections.OrderedDict()
    return c


def _a512(n):
    return (int(n) + 511) & ~511


def _records(device, P, H, W, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
             viewmatrix, projmatrix, tan_fovx, tan_fovy, sh, degree, campos, prefiltered, debug, for_backward, opt):
    """(gcr_camera, gcr_gaussians, what must stay alive during the call) of a frame -- from this thread's cache when every
    tensor of the call is the same object at the same address as when the records were built."""
    tensors = (background, means3D, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, sh, campos)
    key = (tuple(0 if (t is None or t.numel() == 0) else id(t) for t in tensors), scale_modifier, tan_fovx, tan_fovy, H, W,
           degree, prefiltered, debug, for_backward, id(opt))
    cache = _prep_cache()
    ent = cache.get(key)
    if ent is not None and ent.P == P:
        for r, p, t in zip(ent.refs, ent.ptrs, tensors):
            if r is not None and (r() is not t or t.data_ptr() != p):
                break
        else:
            return ent.cam, ent.g, None
    cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W,
                          scale_modifier, degree, prefiltered, debug, for_backward)
    g, keep_g = _gaussians(device, P, means3D, opacity, sh, colors, scales, rotations, cov3D_precomp)
    kept = [t for t in keep_c + keep_g if isinstance(t, torch.Tensor)]
    live = [t for t in tensors if t is not None and t.numel() != 0]
    if len(kept) == len(live) and all(any(k is t for t in live) for k in kept):  # no .contiguous() copy was made
        ent = _Prepared()
        ent.refs = tuple(None if (t is None or t.numel() == 0) else weakref.ref(t) for t in tensors)
        ent.ptrs = tuple(0 if (t is None or t.numel() == 0) else t.data_ptr() for t in tensors)
        ent.cam, ent.g, ent.device, ent.P = cam, g, device, P
        cache[key] = ent
        while len(cache) > _PREP_MAX:
            cache.popitem(last=False)
    return cam, g, (keep_c, keep_g)


def rasterize_gaussians_frame(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                              prefiltered):
    """One inference frame: rasterize_gaussians_ticket()'s image and radii -- (out_color[3,H,W], radii[P]) -- without
    the host wait, the state buffers, or a backward.  Absent optional inputs are None (or empty tensors)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor: this rasterizer has no CPU path")
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    opt = _current_options()
    if P == 0 or _cull.enabled() or _SYNC_ONLY or _guess_hook is not None:
        return rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                   viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered,
                                   False, _for_backward=False, _ticket=True)[1:3]
    L = N.lib()
    device = means3D.device
    with _on_device(device):
        cam, g, keep = _records(device, P, H, W, background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, sh, degree, campos, prefiltered,
                                False, False, opt)
        out = _forward_lean(L, device, cam, g, P, H, W, opt)
        del keep
    return out


def _forward_lean(L, device, cam, g, P, H, W, opt):
    """_forward(ticket=True) for a frame whose state nobody will ask for: one scratch allocation, (out_color, radii)."""
    key = (device.index, P, W, H)
    R_seen, list_cap = _hint_get(key)
    radix = opt.force_radix if (opt is not None and opt.force_radix >= 0) else N.get_option("force_radix")
    if R_seen <= 0 or radix:  # the first frames of a key (no guess yet) and the radix mode take the synchronous path
        return _forward(L, device, cam, g, P, H, W, True)[1:3]
    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=device)
    radii = torch.empty((P,), dtype=torch.int32, device=device)
    gbytes = _size_cache.get(P)
    if gbytes is None:
        gbytes = _size_cache[P] = L.gcr_geometry_bytes(P)
    ibytes = _size_cache.get((W, H))
    if ibytes is None:
        ibytes = _size_cache[(W, H)] = L.gcr_image_bytes(W, H)
    ring = _ring(L)
    ring.harvest()
    capacity = _ASYNC_FACTOR * R_seen + _ASYNC_MARGIN
    bbytes = L.gcr_binning_bytes_lean(capacity, W, H)
    off_i = _a512(gbytes)
    off_b = off_i + _a512(ibytes)
    scratch = torch.empty((off_b + bbytes,), dtype=torch.uint8, device=device)
    base = scratch.data_ptr()
    stream = _stream(device)
    slot, words, addr, seq = ring.take()
    N.check(L.gcr_forward_async(C.byref(cam), C.byref(g), base, gbytes, base + off_b, bbytes, capacity, list_cap,
                                base + off_i, ibytes, radii.data_ptr(), out_color.data_ptr(), addr, seq, stream),
            "gcr_forward_async")
    t = FrameTicket(L, words, addr, seq, capacity, stream, key, False)
    ring.tickets[slot] = t
    ring.pending.append(t)
    return out_color, radii


# ---- GaussianCity's own call shape: points [N,14] = xyz(3) opacity(1) scale(3) rotation(4) rgb(3) ---------------------
# dgr/__init__.py:404-426 slices that tensor into five strided views, which the reference's binding copies one by one
# (.contiguous()), and autograd assembles the [N,14] gradient from five slice-backward kernels and four adds.  The C ABI
# takes row strides (gcr_gaussians.stride_*, gcr_grads.stride_* / packed): the kernels read the tensor in place and
# write its gradient in place -- same arithmetic, same bits, about twenty small launches fewer per training step.
_POINT_COLUMNS = dict(means3D=0, opacities=3, scales=4, rotations=7, colors=11)


def _points14_gaussians(points):
    if points.dim() != 2 or points.size(1) != 14:
        raise RuntimeError("points must have dimensions (num_points, 14)")
    if not points.is_cuda:
        raise RuntimeError("points must be a GPU tensor: this rasterizer has no CPU path")
    if points.dtype != torch.float32:
        raise RuntimeError("points must be float32 (got %s)" % points.dtype)
    if points.stride(1) != 1:
        points = points.contiguous()
    base, row = points.data_ptr(), int(points.stride(0))
    col = {k: base + 4 * v for k, v in _POINT_COLUMNS.items()}
    g = N.Gaussians(int(points.size(0)), 0, col["means3D"], col["opacities"], None, col["colors"], col["scales"],
                    col["rotations"], None)
    g.stride_means3D = g.stride_opacities = g.stride_colors = g.stride_scales = g.stride_rotations = row
    return g, points


def rasterize_points14(points, background, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                       image_width, campos, flip_x=False, flip_y=False, for_backward=False, window=None, ticket=False,
                       out_uint8=False):
    """Forward of GaussianRasterizerWrapper's call shape on the [N,14] tensor in place (precomputed colours, SH degree
    0).  Same six-tuple as rasterize_gaussians; the image comes out already mirrored if flip_x / flip_y ask for it, and
    as the `window` = (x, y, w, h) of that mirrored image if one is given (tiles outside it are not blended).
    out_uint8 (inference frames): the second element is the uint8 [h,w,3] video frame of scripts/inference.py:655-667
    instead of the float [3,h,w] image -- the same bytes its five elementwise kernels produce, stored by the blend."""
    if out_uint8 and for_backward:
        raise RuntimeError("out_uint8 renders video frames: there is no backward through them")
    L = N.lib()
    g, pts = _points14_gaussians(points)
    device = pts.device
    P, H, W = int(pts.size(0)), int(image_height), int(image_width)
    if P == 0:
        e = _empty_frame(device, *((window[3], window[2]) if window is not None else (H, W)))
        return ((FrameTicket(L, None, None, 0, 0, None, None, False, R=0),) + e[1:]) if ticket else e
    with _on_device(device):
        cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W,
                              scale_modifier, 0, False, False, for_backward, flip_x, flip_y, window, out_uint8)
        keep_cc = None
        if _cull.enabled() and not for_backward:  # (see rasterize_gaussians)
            keep_cc = _cull.attach(g, scale_modifier, (points,), device, _stream(device))
        out = _forward(L, device, cam, g, P, H, W, ticket)
        del keep_c, pts, keep_cc
    return out


def rasterize_points14_backward(points, radii, background, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                dL_dout_color, campos, geomBuffer, R, binningBuffer, imageBuffer, image_height,
                                image_width, flip_x=False, flip_y=False, window=None):
    """Gradient of rasterize_points14 with respect to `points`, as ONE [N,14] tensor the library fills completely
    (`dL_dout_color` has the shape of the image that was handed out: the window's, if there was one)."""
    L = N.lib()
    g, pts = _points14_gaussians(points)
    device = pts.device
    P = int(pts.size(0))
    H, W = int(image_height), int(image_width)
    want = (NUM_CHANNELS, window[3], window[2]) if window is not None else (NUM_CHANNELS, H, W)
    if tuple(dL_dout_color.shape) != want:
        raise RuntimeError("dL_dout_color has shape %s, expected %s" % (tuple(dL_dout_color.shape), want))
    grad = torch.empty((P, 14), dtype=torch.float32, device=device)
    if P == 0:
        return grad
    if poison_outputs:
        grad.fill_(float("nan"))
    # scratch the API wants besides: accumulation records [P,16] (64-byte aligned), dL_dmeans2D [P,3], dL_dcov3D [P,6]
    opt = _current_options()
    nrec = int(L.gcr_grad_record_floats_opt(C.byref(opt) if opt is not None else None))  # 16, or 32 in the deterministic mode
    flat = torch.empty((P * (nrec + 3 + 6) + 64,), dtype=torch.float32, device=device)
    if poison_outputs:
        flat.fill_(float("nan"))
    rec = flat[:P * nrec]
    m2d = flat[P * nrec:P * (nrec + 3)]
    c3d = flat[P * (nrec + 3):P * (nrec + 9)]
    with _on_device(device):
        cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W,
                              scale_modifier, 0, False, False, True, flip_x, flip_y, window)
        dpix, dpix_ptr = _dev_f32(dL_dout_color, "dL_dout_color", device)
        R, binningBuffer = _state_for_backward(L, device, cam, g, geomBuffer, binningBuffer, imageBuffer, R, W, H)
        gb = grad.data_ptr()
        col = {k: gb + 4 * v for k, v in _POINT_COLUMNS.items()}
        grads = N.Grads(m2d.data_ptr(), rec.data_ptr(), col["opacities"], col["colors"], col["means3D"], c3d.data_ptr(),
                        None, col["scales"], col["rotations"])
        grads.stride_means3D = grads.stride_opacity = grads.stride_colors = grads.stride_scales = \
            grads.stride_rotations = 14
        grads.packed = gb
        grads.packed_floats = P * 14
        N.check(L.gcr_backward(C.byref(cam), C.byref(g), radii.data_ptr(), geomBuffer.data_ptr(), geomBuffer.numel(),
                               binningBuffer.data_ptr() if binningBuffer.numel() else None, binningBuffer.numel(),
                               imageBuffer.data_ptr(), imageBuffer.numel(), int(R), dpix_ptr, C.byref(grads),
                               _stream(device)),
                "gcr_backward")
        del keep_c, dpix, pts
    return grad


def _gradient_buffers(P, M, device):
    """The reference allocates nine torch::zeros tensors (dgr/rasterize_points.cu:118-126); here they are nine views
    of ONE uninitialised buffer: gcr_backward writes every element of every output itself (include/gcr.h).
    dL_dconic is the native side's accumulation scratch: one 64-byte record per Gaussian.
    Order: dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations."""
    opt = _current_options()
    nrec = int(N.lib().gcr_grad_record_floats_opt(C.byref(opt) if opt is not None else None))  # 16, or 32 (deterministic)
    shapes = ((P, 3), (P, 3), (P, NUM_CHANNELS), (P, nrec), (P, 1), (P, 6), (P, M, 3), (P, 3), (P, 4))
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    starts, off = [], 0
    for n in sizes:  # every view starts 256-byte aligned (dL_dconic and dL_drotations are accessed as float4)
        starts.append(off)
        off += (n + 63) // 64 * 64
    flat = (torch.empty if P != 0 else torch.zeros)((off,), dtype=torch.float32, device=device)
    if poison_outputs and P != 0:
        flat.fill_(float("nan"))
    return [flat[st:st + n].view(sh) for st, n, sh in zip(starts, sizes, shapes)]


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (dgr/rasterize_points.cu:97-155, .h:30-43).

    Returns (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3],
    dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4]) in that order
    (dgr/rasterize_points.cu:153-154).
    """
    L = N.lib()
    device = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    (dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales,
     dL_drotations) = _gradient_buffers(P, M, device)
    if P != 0:
        with _on_device(device):
            cam, keep_c = _camera(device, background, viewmatrix, projmatrix, campos, tan_fovx,
                                  tan_fovy, H, W, scale_modifier, degree, False, debug, True)  # (flips: see rasterize_points14)
            # opacity is not an input of the backward (it is read from the geometry state)
            g, keep_g = _gaussians(device, P, means3D, None, sh, colors, scales, rotations,
                                   cov3D_precomp)
            dpix, dpix_ptr = _dev_f32(dL_dout_color, "dL_dout_color", device)
            if radii.dtype != torch.int32:
                raise RuntimeError("radii must be int32")
            radii_c = radii.contiguous()
            grads = N.Grads(dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
                            dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                            dL_dsh.data_ptr() if M else None, dL_dscales.data_ptr(),
                            dL_drotations.data_ptr())
            gb, bb, ib = geomBuffer.contiguous(), binningBuffer.contiguous(), imageBuffer.contiguous()
            R, bb = _state_for_backward(L, device, cam, g, gb, bb, ib, R, W, H)
            N.check(L.gcr_backward(C.byref(cam), C.byref(g), radii_c.data_ptr(), gb.data_ptr(),
                                   gb.numel(), bb.data_ptr() if bb.numel() else None, bb.numel(),
                                   ib.data_ptr(), ib.numel(), int(R), dpix_ptr, C.byref(grads),
                                   _stream(device)),
                    "gcr_backward")
            del keep_c, keep_g, dpix
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
            dL_drotations)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (dgr/rasterize_points.cu:157-173): bool[P], True where view-space z > 0.2."""
    L = N.lib()
    device = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        if not means3D.is_cuda:
            raise RuntimeError("means3D must be a GPU tensor: this rasterizer has no CPU path")
        with torch.cuda.device(device):
            m, mp = _dev_f32(means3D, "means3D", device)
            v, vp = _dev_f32(viewmatrix, "viewmatrix", device)
            p, pp = _dev_f32(projmatrix, "projmatrix", device)
            N.check(L.gcr_mark_visible(P, mp, vp, pp, present.data_ptr(), _stream(device)),
                    "gcr_mark_visible")
            del m, v, p
    return present
