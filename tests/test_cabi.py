"""C-ABI library (not gpu): it loads without a GPU, exports every symbol include/gcr.h declares,
its host-only entry points work, and argument errors are reported through the status/last-error
convention -- no compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from gaussiancity_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gcr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gcr_[a-z_]+)\s*\(", src)) - {"gcr_resize_fn"})


def test_library_exports_every_declared_symbol():
    lib = N.lib()
    declared = _header_functions()
    assert set(declared) == set(N.EXPORTED_SYMBOLS), (declared, N.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", N.LIB_PATH]).decode()
    exported = set(re.findall(r" T (gcr_[a-z_]+)", out))
    assert set(declared) <= exported
    assert lib.gcr_abi_version() == N.ABI_VERSION == int(re.search(r"#define GCR_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "gcr.h")).read()).group(1))


def test_library_contains_gfx950_code_object():
    blob = open(N.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and all(k in blob for k in (b"k_blend_fwd", b"k_radix_scatter", b"k_tile_sort"))


def test_scratch_sizes_and_layout():
    lib = N.lib()
    assert lib.gcr_geometry_bytes(0) >= 0
    g1, g2 = lib.gcr_geometry_bytes(1000), lib.gcr_geometry_bytes(2000)
    assert 1000 * (64 + 32 + 4) <= g1 < g2   # 64-byte records, 32-byte covariance slots, survivor list (ABI v8)
    assert lib.gcr_image_bytes(1920, 1080) >= 1920 * 1080 * 8 + 8160 * 8
    b0, b1 = lib.gcr_binning_bytes(0, 640, 448), lib.gcr_binning_bytes(1_000_000, 640, 448)
    assert b1 >= 1_000_000 * 24 and b0 < b1
    L = N.get_layout(5000, 640, 448, 123456)
    offs = [L.geom_rec, L.geom_cov3D, L.geom_clamped, L.geom_tiles_touched, L.geom_block_sums,
            L.geom_num_rendered, L.geom_total]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert L.geom_cov3D - L.geom_rec >= 5000 * 64 and L.geom_clamped - L.geom_cov3D >= 5000 * 32   # whole sectors per Gaussian
    assert L.geom_total == lib.gcr_geometry_bytes(5000)
    assert L.img_total == lib.gcr_image_bytes(640, 448) and L.img_ranges < L.img_tile_cursor < L.img_total
    assert L.bin_total == lib.gcr_binning_bytes(123456, 640, 448)
    # 1120 tiles -> 11 tile bits -> 43 key bits -> 6 radix passes -> result in half 0
    assert L.bin_sorted == 0
    assert N.get_layout(10, 16, 16, 10).bin_sorted == 1  # 1 tile -> 33 bits -> 5 passes


def test_argument_errors_are_reported_without_touching_the_gpu():
    lib = N.lib()
    R = N.FrameInfo(-1, -1)
    cam = N.Camera(16, 16, 0.3, 0.3, 1.0, 0, 0, 0, None, None, None, None)
    g = N.Gaussians(4, 0, None, None, None, None, None, None, None)
    rc = lib.gcr_forward_preprocess(C.byref(cam), C.byref(g), None, 0, None, 0, None, C.byref(R), None)
    assert rc == -1 and b"non-null" in lib.gcr_last_error()
    with pytest.raises(RuntimeError, match="gcr_status -1"):
        N.check(rc, "gcr_forward_preprocess")
    buf = (C.c_float * 64)()
    p = C.addressof(buf)
    cam = N.Camera(16, 16, 0.3, 0.3, 1.0, 0, 0, 0, p, p, p, p)
    # exactly one of SH / precomputed colour
    g = N.Gaussians(4, 0, p, p, None, None, p, p, None)
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(g), None, 0, None, 0, None, C.byref(R), None) == -1
    assert b"exactly one of SHs" in lib.gcr_last_error()
    # scale without rotation
    g = N.Gaussians(4, 0, p, p, None, p, p, None, None)
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(g), None, 0, None, 0, None, C.byref(R), None) == -1
    assert b"scale/rotation" in lib.gcr_last_error()
    # SH table too small for the requested degree
    cam3 = N.Camera(16, 16, 0.3, 0.3, 1.0, 3, 0, 0, p, p, p, p)
    g = N.Gaussians(4, 4, p, p, p, None, p, p, None)
    assert lib.gcr_forward_preprocess(C.byref(cam3), C.byref(g), None, 0, None, 0, None, C.byref(R), None) == -1
    # geometry buffer too small -> -2, before any launch
    g = N.Gaussians(4, 0, p, p, None, p, p, p, None)
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(g), p, 8, p, 8, p, C.byref(R), None) == -2
    # P beyond what 32-bit index arithmetic covers is refused, not wrapped
    gbig = N.Gaussians(800_000_000, 0, p, p, None, p, p, p, None)
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(gbig), p, 8, p, 8, p, C.byref(R), None) == -1
    assert b"700 000 000" in lib.gcr_last_error()
    # P == 0 short-circuits successfully (dgr/rasterize_points.cu:71)
    g0 = N.Gaussians(0, 0, None, None, None, None, None, None, None)
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(g0), None, 0, None, 0, None, C.byref(R), None) == 0
    assert R.num_rendered == 0 and R.max_tile_instances == 0
    assert lib.gcr_mark_visible(-1, None, None, None, None, None) == -1
    assert lib.gcr_set_option(b"no_such_option", 1) < 0


def test_backward_rejects_misaligned_gradient_records_and_experiment_options_are_compiled_out():
    """dL_dconic is accessed as 64-byte records, dL_drotations as float4 -- a misaligned pointer must come
    back as GCR_ERR_INVALID_ARGUMENT, not as a memory fault.  And the shipping library must not know the
    results-are-wrong timing option `k7_skip_flush`."""
    lib = N.lib()
    buf = (C.c_float * 256)()
    p = (C.addressof(buf) + 63) // 64 * 64
    cam = N.Camera(16, 16, 0.3, 0.3, 1.0, 0, 0, 0, p, p, p, p)
    g = N.Gaussians(1, 0, p, None, None, p, p, p, None)

    def call(conic, rot):
        gr = N.Grads(p, conic, p, p, p, p, None, p, rot)
        return lib.gcr_backward(C.byref(cam), C.byref(g), p, p, 1 << 30, None, 0, p, 1 << 30, 0, p, C.byref(gr), None)

    assert call(p + 4, p) == -1 and b"64-byte" in lib.gcr_last_error()
    assert call(p, p + 4) == -1 and b"16-byte" in lib.gcr_last_error()
    assert lib.gcr_set_option(b"k7_skip_flush", 1) < 0


def test_abi_v5_argument_checks_without_a_gpu():
    """The round-3 additions to the structs are validated before any launch: an output window must lie inside the
    image; strided gradient outputs need the block they live in; dense rotations must be 16-byte aligned."""
    lib = N.lib()
    buf = (C.c_float * 256)()
    p = (C.addressof(buf) + 63) // 64 * 64
    R = N.FrameInfo(-1, -1)
    cam = N.Camera(32, 32, 0.3, 0.3, 1.0, 0, 0, 0, p, p, p, p)
    cam.win_x, cam.win_y, cam.win_w, cam.win_h = 20, 0, 16, 8           # 20 + 16 > 32
    g = N.Gaussians(4, 0, p, p, None, p, p, p, None)
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(g), p, 8, p, 8, p, C.byref(R), None) == -1
    assert b"window" in lib.gcr_last_error()
    cam.win_x = 0
    assert lib.gcr_forward_preprocess(C.byref(cam), C.byref(g), p, 8, p, 8, p, C.byref(R), None) == -2   # passes the check
    cam = N.Camera(16, 16, 0.3, 0.3, 1.0, 0, 0, 0, p, p, p, p)
    g = N.Gaussians(1, 0, p, None, None, p, p, p, None)
    gr = N.Grads(p, p, p, p, p, p, None, p, p)
    gr.stride_means3D = 14                                                 # strided, but no `packed` block
    rc = lib.gcr_backward(C.byref(cam), C.byref(g), p, p, 1 << 30, None, 0, p, 1 << 30, 0, p, C.byref(gr), None)
    assert rc == -1 and b"packed" in lib.gcr_last_error()
    g2 = N.Gaussians(1, 0, p, None, None, p, p, p + 4, None)               # dense rotations, misaligned
    gr2 = N.Grads(p, p, p, p, p, p, None, p, p)
    rc = lib.gcr_backward(C.byref(cam), C.byref(g2), p, p, 1 << 30, None, 0, p, 1 << 30, 0, p, C.byref(gr2), None)
    assert rc == -1 and b"rotations must be 16-byte aligned" in lib.gcr_last_error()
    default_piece = lib.gcr_get_option(b"bwd_piece")
    assert default_piece == 160 and lib.gcr_get_option(b"no_such_option") == -2 ** 31     # (ABI v7: a getter)
    assert lib.gcr_set_option(b"bwd_piece", 10) == default_piece and lib.gcr_set_option(b"bwd_piece", 128) == 64     # clamped to 64..223
    assert lib.gcr_set_option(b"bwd_piece", 4096) == 128 and lib.gcr_set_option(b"bwd_piece", default_piece) == 223  # (upper clamp)
    assert lib.gcr_get_option(b"bwd_piece") == default_piece
    # "stream_policy" (round 6): -1 automatic / 0 never / 1 always, process-wide, no struct field
    assert lib.gcr_get_option(b"stream_policy") == -1 and lib.gcr_set_option(b"stream_policy", 5) == -1
    assert lib.gcr_set_option(b"stream_policy", -3) == 1 and lib.gcr_get_option(b"stream_policy") == -1
    assert lib.gcr_set_option(b"deterministic_backward", 1) == 0 and lib.gcr_grad_record_floats() == 32
    assert lib.gcr_set_option(b"deterministic_backward", 0) == 1 and lib.gcr_grad_record_floats() == 16


def test_abi_v6_per_call_options_async_arguments_and_ticket_protocol_without_a_gpu():
    """ABI v6: per-call gcr_options override the process-wide defaults for one call only; the lean binning carve; the
    argument checks of gcr_forward_async / gcr_backward / the state-only gcr_forward_render; and the ticket protocol of
    an asynchronous frame, driven here by hand on ordinary host memory (no device is touched)."""
    lib = N.lib()
    # options: one call's view, the process-wide default untouched
    opt = N.Options(deterministic_backward=1)
    assert lib.gcr_grad_record_floats_opt(C.byref(opt)) == 32 and lib.gcr_grad_record_floats() == 16
    assert lib.gcr_grad_record_floats_opt(None) == 16 and lib.gcr_grad_record_floats_opt(C.byref(N.Options())) == 16
    with pytest.raises(TypeError):
        N.Options(no_such_option=1)
    # lean carve: everything in front of the backward's state
    L = N.get_layout(1000, 640, 448, 200000)
    assert L.bin_lean_total == lib.gcr_binning_bytes_lean(200000, 640, 448) == L.bin_work < L.bin_mask < L.bin_ckpt < L.bin_staged < L.bin_total and L.bin_total - L.bin_staged >= 48 * 200000
    assert 24 * 200000 <= L.bin_lean_total <= 40 * 200000
    buf = (C.c_float * 256)()
    p = (C.addressof(buf) + 63) // 64 * 64
    words = (C.c_uint64 * N.TICKET_WORDS)()
    w = C.addressof(words)
    cam = N.Camera(16, 16, 0.3, 0.3, 1.0, 0, 0, 0, p, p, p, p)
    g = N.Gaussians(4, 0, p, p, None, p, p, p, None)
    big = 1 << 30
    call = lambda capacity, wp, seq: lib.gcr_forward_async(C.byref(cam), C.byref(g), p, big, p, big, capacity, 0, p, big, p, p,  # noqa: E731
                                                           wp, seq, None)
    assert call(100, None, 1) == -1 and b"words_host" in lib.gcr_last_error()
    assert call(100, w, 0) == -1
    assert call(0, w, 1) == -1 and b"capacity guess" in lib.gcr_last_error()
    cam.debug = 1
    assert call(100, w, 1) == -1 and b"debug" in lib.gcr_last_error()
    cam.debug = 0
    opt_r = N.Options(force_radix=1)
    cam.options = C.pointer(opt_r)
    assert call(100, w, 1) == -1          # the reference's global radix scheme has no speculative form
    cam.options = None
    assert lib.gcr_forward_async(C.byref(cam), C.byref(g), p, big, p, 64, 100, 0, p, big, p, p, w, 1, None) == -2  # binning too small
    # P == 0: nothing to enqueue, the ticket resolves at once
    g0 = N.Gaussians(0, 0, None, None, None, None, None, None, None)
    info = N.FrameInfo(-1, -1)
    assert lib.gcr_forward_async(C.byref(cam), C.byref(g0), None, 0, None, 0, 0, 0, None, 0, None, None, w, 7, None) == 0
    assert lib.gcr_ticket_poll(w, 7, 0, C.byref(info)) == 0 and info.num_rendered == 0
    # ticket protocol by hand: pending -> resolved; overflow -> pending until the rescue says done; failure words
    assert lib.gcr_ticket_poll(w, 9, 1000, C.byref(info)) == 1                      # tag of another frame
    words[0], words[1] = (9 << 32) | 700, 33
    assert lib.gcr_ticket_poll(w, 9, 1000, C.byref(info)) == 0 and (info.num_rendered, info.max_tile_instances) == (700, 33)
    assert lib.gcr_ticket_wait(w, 9, 1000, None, C.byref(info)) == 0
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == 1                       # 700 > 500: the rescue is still to come
    words[2] = 9
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == 0 and info.num_rendered == 700
    words[5] = 9
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == -4 and b"rescue" in lib.gcr_last_error()
    words[5], words[2], words[4] = 0, 0, 9
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == -4 and b"not rescued in time" in lib.gcr_last_error()
    words[0] = (9 << 32) | 0xFFFFFFFF
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == -3                      # num_rendered beyond 2^31 - 1
    assert lib.gcr_ticket_poll(None, 9, 500, C.byref(info)) == -1 and lib.gcr_ticket_poll(w, 0, 500, C.byref(info)) == -1
    # the backward wants a frame that was rendered for it, and says so; a state-only render needs backward == 1
    gb = N.Gaussians(1, 0, p, None, None, p, p, p, None)
    gr = N.Grads(p, p, p, p, p, p, None, p, p)
    rc = lib.gcr_backward(C.byref(cam), C.byref(gb), p, p, big, None, 0, p, big, 0, p, C.byref(gr), None)
    assert rc == -1 and b"backward == 1" in lib.gcr_last_error()
    fi = N.FrameInfo(10, 10)
    assert lib.gcr_forward_render(C.byref(cam), C.byref(gb), p, big, p, big, p, big, C.byref(fi), None, None) == -1
    assert b"state-only" in lib.gcr_last_error()
    assert lib.gcr_rescue_count() == 0
    # ABI v7 handshake: a gate that gave up (word 4) while a rescue had already started (word 7) is still waiting for
    # it -- the ticket stays pending until the rescue's release (word 2), it does not report "not rescued in time"
    words[0] = (9 << 32) | 700
    words[5], words[2], words[4], words[7] = 0, 0, 9, 9
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == 1
    words[2] = 9
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == 0 and info.num_rendered == 700
    words[5] = 9   # ... and a rescue that saw the give-up word and touched nothing says so through the failure word
    assert lib.gcr_ticket_poll(w, 9, 500, C.byref(info)) == -4
    assert lib.gcr_rescue_dropped_count() == 0 and lib.gcr_get_option(b"gate_polls") == 400000


def test_a_failed_frame_fails_its_own_ticket_and_nothing_else_without_a_gpu():
    """ext.FrameTicket keeps a frame's error to itself.  Driven here on ordinary host memory: a ticket whose
    words say "overflowed and not rescued in time" resolves (done() is True, the ring's harvest pops it, take() goes past
    it), raises from its own wait() / int() -- every time --, leaves the capacity hints alone, and the next ticket on the
    ring resolves normally."""
    import collections

    import pytest
    from gaussiancity_amd import ext
    lib = N.lib()
    words = [(C.c_uint64 * N.TICKET_WORDS)() for _ in range(2)]
    key = ("cpu-test", 1, 2, 3)
    ext._capacity_hint.pop(key, None)
    bad = ext.FrameTicket(lib, words[0], C.addressof(words[0]), 5, 100, None, key, True)
    good = ext.FrameTicket(lib, words[1], C.addressof(words[1]), 6, 100, None, key, True)
    assert not bad.done() and not good.done()                 # nothing published yet
    words[0][0], words[0][4] = (5 << 32) | 4000, 5            # R = 4000 > capacity 100, the gate gave up
    words[1][0], words[1][1] = (6 << 32) | 80, 17
    ring = ext._TicketRing.__new__(ext._TicketRing)           # (no pinned block: only the bookkeeping is exercised)
    ring.base, ring.pending, ring.tickets = None, collections.deque([bad, good]), [bad, good]
    ring.harvest()
    assert not ring.pending and bad.done() and good.done()
    assert bad.failed is not None and good.failed is None and int(good) == 80
    for _ in range(2):
        with pytest.raises(RuntimeError, match="not rescued in time"):
            int(bad)
    with pytest.raises(RuntimeError, match="not rescued in time"):
        bad.rescued
    assert ext._capacity_hint[key] == (80, 17)                # the failed frame left no hint behind
    ext._capacity_hint.pop(key, None)



def _gcv_header_functions():
    src = open(os.path.join(ROOT, "include", "gcv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gcv_[a-z_]+)\s*\(", src)))


def test_gcv_library_exports_every_declared_symbol():
    """libgcv_hip.so (point generation / visibility, include/gcv.h): loads without a GPU and exports
    exactly what the header declares; argument errors come back without touching the device."""
    from gaussiancity_amd import _native_v as V
    lib = V.lib()
    declared = _gcv_header_functions()
    assert set(declared) == set(V.EXPORTED_SYMBOLS), (declared, V.EXPORTED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", V.LIB_PATH]).decode()
    assert set(declared) <= set(re.findall(r" T (gcv_[a-z_]+)", out))
    assert lib.gcv_abi_version() == V.ABI_VERSION == int(
        re.search(r"#define GCV_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "gcv.h")).read()).group(1))
    code = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", V.LIB_PATH]).decode() \
        if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") else "gfx950"
    assert "gfx950" in code
    assert lib.gcv_extrude_scratch_bytes(2048, 2048) >= 8 * (2 + 2048 * 2048 // 256)
    assert lib.gcv_occupancy_bytes(16, 16, 16) == 4 and lib.gcv_occupancy_bytes(0, 1, 1) == 0
    n = C.c_int64(0)
    m = V.SegIns(100, 32767, 32767, 2, 1)
    assert lib.gcv_extrude_count(1, None, C.byref(m), 8, 8, None, None, None, None, None, 0, C.byref(n), None) < 0
    assert b"null" in lib.gcv_last_error()
    assert lib.gcv_points_to_volume(0, None, None, None, 0, 4, 4, None, None, None) < 0
    assert lib.gcv_ray_voxel_intersection(None, None, None, None, None, None, None, 1.0, None, None, 1, None, None,
                                          None, None) < 0


def test_gce_library_exports_every_declared_symbol():
    """libgce_hip.so (hash-grid encoder, include/gce.h)."""
    from gaussiancity_amd import _native_e as E
    lib = E.lib()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gce.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(gce_[a-z_]+)\s*\(", src)))
    assert set(declared) == set(E.EXPORTED_SYMBOLS), (declared, E.EXPORTED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", E.LIB_PATH]).decode()
    assert set(declared) <= set(re.findall(r" T (gce_[a-z_]+)", out))
    assert lib.gce_abi_version() == E.ABI_VERSION == int(re.search(r"#define GCE_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "gce.h")).read()).group(1))
    sc = (C.c_float * 4)()
    assert lib.gce_level_scales(4, 1.0, 16, sc) == 0 and list(sc) == [15.0, 31.0, 63.0, 127.0]
    assert lib.gce_forward(None, None, None, None, 8, 3, 3, 2, 1.0, 4, 0, None, 0, 0, None) < 0   # C = 3
    assert b"C must be" in lib.gce_last_error()
    assert lib.gce_forward(None, None, None, None, 8, 3, 2, 2, 1.0, 4, 0, None, 0, 0, None) < 0   # null tensors
    assert lib.gce_backward(None, None, None, None, None, 8, 9, 2, 2, 1.0, 4, 0, None, None, 0, 0, None) < 0
