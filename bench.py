#!/usr/bin/env python
"""bench.py -- hot-path benchmark of the MI355X-native Gaussian rasterizer.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one frame: one forward pass of the rasterizer over the resident synthetic scene for
the next pose of the 24-pose inference orbit (reference scripts/inference.py:655-667 renders
one frame per loop iteration).  Default workload = BASELINE.json's headline config C3
(5M-Gaussian S-city scene, 1920x1080, SH degree 3, forward only).  Frames are independent, so
with N GPUs rank r renders poses r, r+N, ... (no collective on the data path; weak scaling:
every rank renders K frames).  Scene tensors and camera matrices are resident in HBM before
the timed region starts.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  "roofline":     dominant kernel's ALGORITHMIC bytes (SURVEY.md 8d / DESIGN.md 6) / its mean
                  duration, measured with HIP events on the launch stream inside the timed region;
  "cpu_baseline": the CPU oracle (oracle/, a restatement -- the reference has no CPU path)
                  timed on a bounded sample on this host's cores;
  "stages_ms":    every stage's mean device time and achieved algorithmic GB/s;
  "secondary":    C2 (500k Gaussians, 640x448, SH3) forward+backward ms/frame, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 measured copy)
NUMERICS = "gcr-fp32-v2"  # the numerics contract shared by oracle/ and the HIP kernels


def higher_msb(n):
    """cr/rasterizer_impl.cu:35-48 (getHigherMsb): bits needed for the tile id."""
    msb, step = 16, 16
    while step > 1:
        step //= 2
        msb = msb + step if (n >> msb) else msb - step
    return msb + 1 if (n >> msb) else msb


def algorithmic_bytes(P, P_v, R, R_p, W, H, M, sh):
    """Per-frame algorithmic HBM bytes of every stage."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    c3 = 12 * M if sh else 0
    w1 = 75 if sh else 60
    P_c = P - P_v
    return {
        "preprocess": P_v * (44 + c3 + w1) + 20 * P_c,
        "scan": 8 * P,
        "emit": 4 * P + 16 * P_v + 12 * R,
        "sort": 24 * R,  # lower bound: one read + one write of 12-byte pairs
        "ranges": 8 * R + 8 * T,
        "blend_fwd": 40 * R_p + 20 * W * H + 8 * T,
        "blend_bwd": 112 * R_p + 20 * W * H + 8 * T,
        "preprocess_bwd": P_v * (96 + (15 + c3 if sh else 0)) + P_v * (64 + c3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5", "D1"],
                    help="BASELINE configs C1..C5 (default C3 = the headline); D1 = dense general-3DGS-like stress scene "
                         "(long tile lists; informational, not a BASELINE config)")
    ap.add_argument("--points", type=int, default=None, help="override P (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--bwd-wave-units", action="store_true", help="backward blend as one wave per (work item, quadrant): round 4's kernel (A/B)")
    ap.add_argument("--sort-whole", action="store_true",
                    help="A/B: option lazy_sort off -- every tile list is sorted whole, as the reference does "
                         "(default: lists beyond 1024 entries are sorted segment by segment as far as the blend walks)")
    ap.add_argument("--sort-in-blend", action="store_true",
                    help="the forward blend sorts its own tiles (lower frame latency, lower throughput; A/B)")
    ap.add_argument("--static-scene", action="store_true",
                    help="second line, never the headline: ONE mostly off-screen set of Gaussians rendered from many poses "
                         "(C3 / C5), so the rasterizer keeps its cull cache (gaussiancity_amd/cull_cache.py: same frames, K1 "
                         "streams 16 B per Gaussian); a mostly visible set (--inference-loop) is slower with it")
    ap.add_argument("--split-preprocess", action="store_true",
                    help="K1 as two kernels (streaming cull, then exact pass) instead of the fused one (A/B only)")
    ap.add_argument("--backward", action="store_true",
                    help="one step = forward + backward of the frame (with --config C2: the BASELINE metric's fwd+bwd half "
                         "as the main loop, e.g. under rocprofv3)")
    ap.add_argument("--path", default="raster", choices=["raster", "visibility", "grid-encoder"],
                    help="raster (default): the rasterizer hot path; visibility: SURVEY 8 row f2, BEV maps -> points -> "
                         "volume -> per-pixel first hit (one step = one camera pose)")
    ap.add_argument("--layout-size", type=int, default=2048, help="--path visibility: BEV map edge in pixels")
    ap.add_argument("--encoder-points", type=int, default=16384,
                    help="--path grid-encoder: points per step (16384 = TRAIN_MAX_POINTS, config.py:34)")
    ap.add_argument("--resident-volume", type=int, default=1,
                    help="--path visibility: keep the volume resident and erase only the written voxels (0 = clear it)")
    ap.add_argument("--jumps", type=int, default=0,
                    help="--path visibility: empty-space jumps in the traversal (A/B knob; same outputs)")
    ap.add_argument("--train-step", action="store_true",
                    help="C4 training-step harness (fwd + loss + bwd + gradient all-reduce) instead of the frame loop")
    ap.add_argument("--inference-loop", action="store_true",
                    help="the render loop of scripts/inference.py:655-667 through the wrapper: N points [N,14] -> "
                         "image -> uint8 HWC frame on the host, per orbit pose (frames.InferenceLoop)")
    ap.add_argument("--host-camera", nargs="?", const="closed-form", default="reference",
                    choices=["closed-form", "reference", "device"],
                    help="--train-step / --inference-loop: GaussianRasterizerWrapper(host_camera=...): the reference's "
                         "recipe with torch on the host and the matrices by value (the wrapper's default), closed-form host "
                         "arithmetic, or the recipe on the device (H2D copies + GEMM + inverse with a sync)")
    ap.add_argument("--host-threads", type=int, default=1,
                    help="host threads driving the frame loop, one stream each (frames are independent)")
    ap.add_argument("--native-int-api", action="store_true",
                    help="time the native module's positional rasterize_gaussians() (returns num_rendered as an int: one "
                         "host wait per frame, the reference's contract) instead of the Python API GaussianRasterizer "
                         "(returns image and radii; this build does not wait for num_rendered there)")
    ap.add_argument("--float-frames", action="store_true",
                    help="--inference-loop: render float images and convert them to uint8 frames with torch kernels, as "
                         "scripts/inference.py does (default: the blend kernel stores the uint8 frame, same bytes)")
    ap.add_argument("--d-step", action="store_true",
                    help="--train-step: the step WITH the discriminator's half (core/train.py:227-257): a forward-only render "
                         "under no_grad, an 18 262 793-parameter discriminator stand-in under DDP (73.1 MB all-reduce), then "
                         "the G-step with the GAN term -- the reference's step when DISCRIMINATOR.ENABLED")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B only: a library option by name (gcr_set_option), e.g. --opt pipeline=0 --opt blend_lds_pad=4096; "
                         "the line's config.options lists what was set")
    ap.add_argument("--scene-order", default="seeded", choices=["seeded", "morton"],
                    help="A/B only, never the workload: `morton` renumbers the synthetic scene's Gaussians along a Morton "
                         "curve of their (x, y) -- the index order real GaussianCity points have (a raster scan of the "
                         "BEV maps) and the seeded S-city, drawn uniformly at random, has not")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the frame loop alternates over (frames are independent; 1 = serial); "
                         "default 3 (measured optimum for both paths: C3 forward 4 520 / 4 700 / 4 690 frames/s "
                         "with 2 / 3 / 4 streams on one box)")
    args = ap.parse_args()
    if args.streams is None:
        args.streams = 3

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: re-launch this command as N ranks (one per GPU) under
        # torch.distributed.run on this node; the children see WORLD_SIZE and take the normal path
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    # Rank -> CPU affinity (counterpart of utils/distributed.py:19-62) BEFORE anything initialises HIP: the runtime's
    # helper threads and the first pinned allocations then come up on the GPU's NUMA node.  From sysfs alone
    # (affinity.bind_rank_early); GCR_NO_AFFINITY=1 disables it, GCR_AFFINITY=late reproduces round 3's binding after
    # torch.cuda.set_device for the A/B (profiles/r04_affinity_ab.json).
    import importlib.util
    _spec = importlib.util.spec_from_file_location("_gcr_affinity", os.path.join(ROOT, "gaussiancity_amd", "affinity.py"))
    _aff = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_aff)
    args.full_cpu_mask = os.sched_getaffinity(0)  # the CPU baseline legs run on all host cores again
    # N = 1 is not bound unless GCR_AFFINITY says so (one process has the whole host; round 4's A/B of none / early /
    # late binding at N = 1 was inside the noise of the host-bound lines, profiles/r04_affinity_ab.jsonl)
    want_affinity = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("GCR_AFFINITY") in ("early", "late", "1")
    late_affinity = os.environ.get("GCR_AFFINITY") == "late"
    share_gpu_early = os.environ.get("GCR_BENCH_SHARE_GPU") == "1"
    args.affinity = None
    if want_affinity and not late_affinity and not share_gpu_early:
        args.affinity = _aff.bind_rank_early(int(os.environ.get("LOCAL_RANK", "0")))
        late_affinity = not args.affinity.get("bound") and os.environ.get("GCR_NO_AFFINITY") != "1"  # no KFD sysfs: late

    import torch
    import torch.distributed as dist
    from gaussiancity_amd import _native as N
    from gaussiancity_amd import ext, synth
    from gaussiancity_amd.rasterizer import GaussianRasterizer, GaussianRasterizerWrapper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Debug aid for boxes with ONE GPU: GCR_BENCH_SHARE_GPU=1 lets every rank of a torchrun launch use cuda:0 and
    # rendezvous over gloo, so the N>1 control flow (sharding, barriers, max-over-ranks reduction, rank-0 report)
    # can be exercised where RCCL would refuse two ranks on one device.  Never set by the driver.
    share_gpu = os.environ.get("GCR_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if want_affinity and late_affinity and not share_gpu:
        from gaussiancity_amd.affinity import bind_rank_to_gpu
        early_why = (args.affinity or {}).get("why")
        args.affinity = bind_rank_to_gpu(local_rank)
        args.affinity["when"] = "after torch.cuda.set_device" + (" (early binding: %s)" % early_why if early_why else "")
    args.rccl_log = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            # RCCL's own account of the communicator (rings / trees / transports) goes to a FILE per rank -- stdout
            # carries exactly one JSON line -- and is summarised into the report (affinity.summarize_rccl_log)
            if "NCCL_DEBUG" not in os.environ:
                os.environ["NCCL_DEBUG"] = "INFO"
                os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING")
                os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/gcr_rccl_%d_rank%d.log" % (os.getppid(), rank))
            args.rccl_log = os.environ.get("NCCL_DEBUG_FILE")
            dist.init_process_group(backend="nccl", device_id=dev)

    def collect_ranks():
        """rank 0: [{rank, affinity, rccl}] of every rank (all_gather_object); other ranks: None."""
        from gaussiancity_amd.affinity import summarize_rccl_log
        mine = {"rank": rank, "affinity": getattr(args, "affinity", None), "rccl": None}
        if args.rccl_log:
            try:
                mine["rccl"] = summarize_rccl_log(open(args.rccl_log.replace("%h", os.uname().nodename)
                                                       .replace("%p", str(os.getpid()))).read())
            except OSError as e:
                mine["rccl"] = {"log": "unreadable: %s" % e}
        if world == 1:
            return [mine]
        got = [None] * world
        dist.all_gather_object(got, mine)
        return got if rank == 0 else None
    args.collect_ranks = collect_ranks
    N.lib()
    N.set_option("bwd_wave_units", 1 if args.bwd_wave_units else 0)
    N.set_option("split_preprocess", 1 if args.split_preprocess else 0)
    N.set_option("sort_in_blend", 1 if args.sort_in_blend else 0)
    N.set_option("lazy_sort", 0 if args.sort_whole else 1)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        if N.set_option(name, int(val)) == -1 and N.get_option(name) != int(val):
            raise SystemExit("unknown library option %r" % name)
    if args.static_scene:
        from gaussiancity_amd import cull_cache
        cull_cache.enable(True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def copy_ceiling():
        """On-box device-copy ceiling: 1 GiB -> 1 GiB D2D copies, read + write bytes."""
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        for _ in range(2):
            b.copy_(a)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(10):
            b.copy_(a)
        torch.cuda.synchronize()
        return 10 * 2 * a.numel() * 4 / 1e9 / (time.perf_counter() - tc)

    if args.path == "grid-encoder":
        return grid_encoder_bench(args, torch, dist, dev, world, rank, barrier, copy_ceiling)
    if args.path == "visibility":
        return visibility_bench(args, torch, dist, synth, dev, world, rank, barrier, copy_ceiling)
    if args.train_step:
        return train_step_bench(args, torch, dist, N, synth, GaussianRasterizerWrapper, dev, world, rank, barrier)
    if args.inference_loop:
        return inference_loop_bench(args, torch, dist, synth, GaussianRasterizerWrapper, dev, world, rank, barrier)

    def load_scene(cfg_name, points=None, size=None):
        cfg, sc = synth.make_scene(cfg_name, points)
        if args.scene_order == "morton":
            def spread(v):
                v = v.astype(np.uint64) & np.uint64(0xffff)
                for sh, m in ((8, 0x00ff00ff), (4, 0x0f0f0f0f), (2, 0x33333333), (1, 0x55555555)):
                    v = (v | (v << np.uint64(sh))) & np.uint64(m)
                return v
            xy = np.clip(sc["means3D"][:, :2] * 8.0, 0, 65535)
            order = np.argsort(spread(xy[:, 0]) | (spread(xy[:, 1]) << np.uint64(1)), kind="stable")
            sc = {k: (np.ascontiguousarray(v[order]) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
        if size is not None:
            cfg["W"], cfg["H"] = size
        W, H = cfg["W"], cfg["H"]
        wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
        cams = []
        for pos, quat in synth.orbit_poses():
            rs = wr._get_gaussian_rasterization_settings(pos, quat)
            cams.append(rs._replace(sh_degree=cfg["sh_degree"]))
        use_sh = not cfg.get("precomp_color", False)
        t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
        empty = torch.Tensor([])

        def fwd(pose, for_backward=False):
            """One frame through the NATIVE MODULE's positional function (dgr/rasterize_points.h:18-28): returns
            num_rendered as an int, i.e. pays the reference's one host wait per frame."""
            rs = cams[pose % len(cams)]
            a = (rs.bg, t["means3D"], empty if use_sh else t["colors_precomp"], t["opacities"],
                 t["scales"], t["rotations"], rs.scale_modifier, empty, rs.view_matrix, rs.proj_matrix,
                 rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"] if use_sh else empty,
                 rs.sh_degree, rs.campos, False, False)
            # `for_backward`: what RasterizeGaussiansFunction tells the native side when an input requires a gradient
            return a, ext.rasterize_gaussians(*a, _for_backward=for_backward)

        # The same frame through the PYTHON API the reference's callers use (GaussianRasterizer.forward ->
        # RasterizeGaussiansFunction, dgr/__init__.py:223-273): returns (image, radii) -- num_rendered is not part of it,
        # so this build does not wait for it (FrameTicket, gaussiancity_amd/ext.py) and the host enqueues frame after
        # frame.  Every call is still one complete forward of one pose.
        rasters = [GaussianRasterizer(rs) for rs in cams]
        means2D = torch.zeros_like(t["means3D"])  # the reference's gradient slot (dgr/__init__.py:231), never read
        kw = dict(shs=t["shs"]) if use_sh else dict(colors_precomp=t["colors_precomp"])

        def fwd_api(pose):
            with torch.no_grad():
                return rasters[pose % len(cams)](means3D=t["means3D"], means2D=means2D, opacities=t["opacities"],
                                                 scales=t["scales"], rotations=t["rotations"], **kw)

        fwd.api = fwd_api
        return cfg, sc, cams, use_sh, fwd

    def make_fwd_bwd(fwd_fn, dpix):
        def fb(pose):
            a, o = fwd_fn(pose, True)
            (bg, m3, col, opa, scl, rot, smod, cov, view, proj, tfx, tfy, h, w, sh, deg, campos, _, _) = a
            R, color, radii, geom, binning, img = o
            g = ext.rasterize_gaussians_backward(bg, m3, radii, col, scl, rot, smod, cov, view, proj, tfx, tfy,
                                                 dpix, sh, deg, campos, geom, R, binning, img, False)
            return a, o, g
        return fb

    def frame_statistics(fwd_fn, pose_list, P, W, H):
        """(R, R_p, P_v) means over the given poses (untimed): R_p = sum over tiles of the largest n_contrib."""
        stats = []
        for ps in pose_list:
            _, o = fwd_fn(ps)
            R, _, radii, _, _, img = o
            L = N.get_layout(P, W, H, R)
            nc = img[L.img_n_contrib:L.img_n_contrib + 4 * W * H].view(torch.int32).view(H, W)
            Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
            pad = torch.zeros((Hp, Wp), dtype=torch.int32, device=dev)
            pad[:H, :W] = nc
            R_p = int(pad.view(Hp // 16, 16, Wp // 16, 16).amax(dim=(1, 3)).sum().item())
            stats.append((R, R_p, int((radii > 0).sum().item())))
        return tuple(float(np.mean([x[i] for x in stats])) for i in range(3))

    def oracle_kwargs(rs, sc_np, use_sh_):
        kw = dict(img_h=rs.img_h, img_w=rs.img_w, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                  bg=rs.bg.cpu().numpy(), scale_modifier=rs.scale_modifier,
                  view_matrix=rs.view_matrix.cpu().numpy(), proj_matrix=rs.proj_matrix.cpu().numpy(),
                  sh_degree=rs.sh_degree, campos=rs.campos.cpu().numpy(), means3D=sc_np["means3D"],
                  opacities=sc_np["opacities"], scales=sc_np["scales"], rotations=sc_np["rotations"])
        kw.update(dict(shs=sc_np["shs"]) if use_sh_ else dict(colors_precomp=sc_np["colors_precomp"]))
        return kw

    def traffic_doc(tag):
        """Newest committed rocprofv3 --pmc summary profiles/rNN_traffic[_<tag>].json, or None."""
        import glob
        pat = "r[0-9][0-9]_traffic%s.json" % ("_" + tag if tag else "")
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        return (json.load(open(files[-1])), os.path.basename(files[-1])) if files else (None, None)

    def traffic_is_current(doc, sources):
        """True / False: do the kernel sources the committed counters were taken from (tools/make_traffic.py stores their
        hashes) equal this tree's?  None when the file predates the hashes."""
        import hashlib
        have = (doc or {}).get("kernel_source_sha16")
        if not have:
            return None
        csrc = os.path.join(ROOT, "gaussiancity_amd", "csrc")
        return all(hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16] == have.get(f) for f in sources)

    VALU_PEAK_GINSTR = 1024 * 2.4 / 2.0  # 256 CUs x 4 SIMDs, 2.4 GHz, 2 cycles per wave64 VALU instruction

    cfg, sc, cams, use_sh, fwd = load_scene(args.config, args.points)
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    M = sc["shs"].shape[1] if use_sh else 0
    class _Poses:  # frame i of this rank (frames shard round-robin: rank r renders frames r, r + world, ...)
        def __getitem__(self, i):
            return rank + i * world
    poses = _Poses()
    # the timed step: one frame through the reference's Python API (--native-int-api: through the native module's
    # positional function, which returns num_rendered as an int and therefore waits for it in every frame)
    step_fn = fwd if args.native_int_api else fwd.api
    if args.backward:  # one step = forward + backward of the same frame (BASELINE metric's second half)
        dpix_main = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)
        step_fn = make_fwd_bwd(fwd, dpix_main)
    step_main = step_fn

    # Frames are independent units, so consecutive frames go to alternating HIP streams: frame
    # f+1's preprocess/binning (latency-bound, low occupancy) overlaps frame f's blend.  Every
    # step is still one complete forward; each call still blocks the host until it knows R.
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams, args.host_threads))]

    # HIP creates a stream's hardware queue at its first use (about a millisecond): touch every stream once so that
    # this one-off cost of the plumbing is not billed to the first timed frames when W is small
    for st in streams:
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    default_stream = torch.cuda.current_stream(dev)

    def run_frames(lo, hi, step_fn=None):
        step_fn = step_fn or step_main
        if args.host_threads <= 1:
            # torch.cuda.set_stream, not the `with torch.cuda.stream(...)` context: the context manager costs the calling
            # thread 6 us per frame (tools/host_profile_api.py), a tenth of what it needs to enqueue a frame
            try:
                for i in range(lo, hi):
                    torch.cuda.set_stream(streams[i % len(streams)])
                    step_fn(poses[i])
            finally:
                torch.cuda.set_stream(default_stream)
            return
        # One host thread per stream.  The reference API returns num_rendered as a Python int, so every
        # call blocks its caller until the frame's scan has run; with a single host thread that wait
        # (plus the enqueue) is the frame period.  The library is re-entrant (per-thread read-back
        # slot, no global state) and ctypes drops the GIL during the call, so thread t renders frames
        # lo+t, lo+t+n, ... on its own stream while the others are blocked in their waits.
        import threading

        def worker(t):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[t % len(streams)]):
                for i in range(lo + t, hi, args.host_threads):
                    step_fn(poses[i])

        th = [threading.Thread(target=worker, args=(t,)) for t in range(args.host_threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    # ---- per-stage pass FIRST: the same frame loop with HIP events on the launch streams (the event pairs are
    # barrier packets and cost a few us per frame, so they stay out of `value`).  (1) exactly as the timed region
    # -- frames alternating over the streams, kernels of neighbouring frames overlapping -- which is what a
    # rocprofv3 --kernel-trace of this command averages over; (2) one stream, every kernel alone on the GPU.
    # Running it before the timed region also means the clocks are up and the capacity hints are learnt whatever
    # W the caller chose (measured: with W = 5 and K = 20 straight after start-up a frame costs 0.29 ms, not 0.255).
    # These untimed frames are reported as `pre_frames` next to `warmup`.
    n_burn, n_inst, n_alone = 240, 96, 48
    poses_pre = [rank + i * world for i in range(4 + n_inst)]
    for i in range(n_burn):  # learn the capacity hints (first frames take the staged path) and bring the clocks up
        with torch.cuda.stream(streams[i % len(streams)]):
            step_fn(poses_pre[i % 24])
    N.set_option("timing", 1)
    N.stage_ms()  # reset accumulators
    barrier()
    t1 = time.perf_counter()
    for i in range(4, 4 + n_inst):
        with torch.cuda.stream(streams[i % len(streams)]):
            step_fn(poses_pre[i])
    barrier()
    elapsed_instrumented = (time.perf_counter() - t1) * args.steps / n_inst
    stage = N.stage_ms()
    for i in range(4, 4 + n_alone):
        step_fn(poses_pre[i])
    barrier()
    stage_alone = N.stage_ms()
    N.set_option("timing", 0)
    # one frame alone on the GPU, enqueue -> finished (what a caller that needs THIS frame waits for); the
    # headline `value` is a throughput with len(streams) frames in flight
    barrier()
    t1 = time.perf_counter()
    for i in range(4, 4 + 24):
        step_fn(poses_pre[i])
        torch.cuda.synchronize()
    frame_latency_ms = 1e3 * (time.perf_counter() - t1) / 24
    pre_frames = n_burn + n_inst + n_alone + 24

    # ---- W untimed warm-up frames, then the timed region: EXACTLY K frames, barrier + synchronize on both sides
    # A block of K frames can be a few milliseconds (K = 20 at C3: 4.5 ms), which a single clock reading does not
    # resolve reliably (round 2: the driver's K = 20 line read 12 % below the K = 1000 one).  So the timed block of
    # EXACTLY K frames is repeated, each repetition bracketed the same way, until the blocks add up to >= 0.25 s
    # (at most 64 of them); `value` is K frames over the MEDIAN block, min / max are reported beside it.
    def timed_blocks(step, budget_s=0.25):
        run_frames(0, args.warmup, step)
        blocks, nxt = [], args.warmup
        while True:
            barrier()
            t0 = time.perf_counter()
            run_frames(nxt, nxt + args.steps, step)
            barrier()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            blocks.append(el)
            nxt += args.steps
            if sum(blocks) >= budget_s or len(blocks) >= 64:  # (the same decision on every rank: `el` is the max over ranks)
                return blocks

    block_s = timed_blocks(step_main)
    elapsed = float(np.median(block_s))
    total_frames = args.steps * world
    fps = total_frames / elapsed
    # the same K-frame blocks through the other entry point (see step_main above), reported beside `value`
    other_blocks = None
    if not args.backward:
        other_blocks = timed_blocks(fwd.api if args.native_int_api else fwd, 0.15)

    out = None
    ranks_info = args.collect_ranks()  # (a collective for N > 1: every rank calls it)
    if rank == 0:
        # ---- per-frame workload statistics over the same poses (untimed) --------------------
        R_mean, Rp_mean, Pv_mean = frame_statistics(
            fwd, [poses[i] for i in range(args.warmup, args.warmup + min(args.steps, len(cams)))], P, W, H)
        ab = algorithmic_bytes(P, Pv_mean, R_mean, Rp_mean, W, H, M, use_sh)
        stage_names = ("preprocess", "scan", "emit", "sort", "ranges", "blend_fwd")
        if args.backward:
            stage_names += ("blend_bwd", "preprocess_bwd")
        stages = {}
        for k in stage_names:
            ms = stage.get(k, 0.0)
            stages[k] = {"ms": round(ms, 4), "ms_single_stream": round(stage_alone.get(k, 0.0), 4),
                         "alg_MB": round(ab[k] / 1e6, 3),
                         "alg_GBps": round(ab[k] / 1e9 / (ms / 1e3), 1) if ms > 0 else None}
        # The roofline prices a kernel that has the GPU to itself (that is the premise of a roofline): the dominant
        # kernel and its launch duration come from the one-stream pass.  With several frames in flight the same
        # launch is stretched by the other frames' kernels sharing the CUs (`launch_ms_in_flight`, what a rocprofv3
        # trace of the default command averages); the one-stream figure is what a trace of `--streams 1` shows.
        dom = max(stage_names, key=lambda k: stage_alone.get(k, 0.0) or stage.get(k, 0.0))
        dom_flight_ms = stage[dom]
        dom_ms = stage_alone.get(dom, 0.0) or dom_flight_ms
        dom_alone_ms = dom_ms
        achieved = ab[dom] / 1e9 / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        blend_ms = stage_alone.get("blend_fwd", 0.0) or stage["blend_fwd"]
        blend_ach = ab["blend_fwd"] / 1e9 / (blend_ms / 1e3) if blend_ms > 0 else 0.0
        # HBM bytes / VALU instructions per launch from the committed rocprofv3 --pmc passes of this workload
        tag = {"C3": "", "C2": "c2", "C5": "c5"}.get(args.config) if args.points is None else None
        if args.static_scene:  # the cached K1 moves other bytes: its own counter passes, or none
            tag = "static_scene" if tag == "" else None
        tdoc, tfile = traffic_doc(tag) if tag is not None else (None, None)
        tk = tdoc["kernels"].get(dom) if tdoc else None
        traffic = int((2 * tk["FETCH_SIZE_KB"] + tk["WRITE_SIZE_KB"]) * 1024) if tk else None  # gfx950: FETCH_SIZE
        #                                   counts half of a 16-B/lane read (MI355X_MICROARCH.md, HBM)
        valu_instr = tk.get("SQ_INSTS_VALU") if tk else None
        ceiling = copy_ceiling()
        hbm = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
               "alg_bytes_per_launch": int(ab[dom]), "copy_ceiling_GBps": round(ceiling, 1),
               "frac_of_copy_ceiling": round(achieved / ceiling, 4)}
        valu = None
        if valu_instr and dom_ms > 0:
            # wave64 VALU instructions per launch (committed SQ_INSTS_VALU) over the live launch duration, against
            # what 1024 SIMDs issue at 2 cycles per instruction and 2.4 GHz (MI355X_MICROARCH.md "Wave scheduling")
            g = valu_instr / 1e9 / (dom_ms / 1e3)
            valu = {"achieved": round(g, 1), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                    "frac": round(g / VALU_PEAK_GINSTR, 4), "instr_per_launch": int(valu_instr),
                    # a VALU-issue fraction rises when a kernel executes MORE instructions in the same time: the count
                    # per unit of useful work beside it (consumed list entries R_p) says which way it moved
                    "valu_instr_per_consumed_entry": round(valu_instr / max(Rp_mean, 1.0), 2),
                    "frac_in_flight": round(valu_instr / 1e9 / (dom_flight_ms / 1e3) / VALU_PEAK_GINSTR, 4),
                    "source": "profiles/" + tfile,
                    "source_matches_this_tree": traffic_is_current(tdoc, ("gcr_blend.hip", "gcr_device.h", "gcr_cull.h"))}
        blend_like = dom in ("blend_fwd", "blend_bwd")
        if blend_like and valu:
            # the alpha-blend kernels are bound by VALU issue, not by HBM: the roofline object prices the launch
            # against the instruction roofline and carries the HBM accounting of SURVEY 8d beside it
            roofline = {"kernel": dom, "bound": "valu", "achieved": valu["achieved"], "peak": valu["peak"],
                        "unit": valu["unit"], "frac": valu["frac"], "traffic": traffic, "valu": valu, "hbm": hbm}
        else:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": hbm["frac"], "traffic": traffic, "valu": valu, "hbm": hbm}
        roofline.update({"launch_ms": round(dom_ms, 4), "launch_ms_in_flight": round(dom_flight_ms, 4),
                         "frames_in_flight": len(streams),
                         "alg_bytes_per_launch": int(ab[dom]),
                         "alpha_blend": {"kernel": "blend_fwd", "hbm_achieved_GBps": round(blend_ach, 1),
                                         "hbm_frac": round(blend_ach / HBM_PEAK_GBS, 4),
                                         "launch_ms": round(blend_ms, 4)}})
        # the frame's HBM-bound kernel (K1: streaming cull + exact pass), priced the same way: alone on the GPU,
        # algorithmic bytes of SURVEY 8d and the counted traffic of the committed --pmc passes
        k1_ms = stage_alone.get("preprocess", 0.0) or stage.get("preprocess", 0.0)
        tk1 = tdoc["kernels"].get("preprocess") if tdoc else None
        if k1_ms > 0:
            k1_traffic = int((2 * tk1["FETCH_SIZE_KB"] + tk1["WRITE_SIZE_KB"]) * 1024) if tk1 else None
            k1_alg = ab["preprocess"] / 1e9 / (k1_ms / 1e3)
            roofline["hbm_kernel"] = {
                "kernel": "preprocess", "bound": "hbm", "launch_ms": round(k1_ms, 4),
                "alg_bytes_per_launch": int(ab["preprocess"]), "achieved": round(k1_alg, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(k1_alg / HBM_PEAK_GBS, 4), "traffic": k1_traffic,
                "traffic_GBps": round(k1_traffic / 1e9 / (k1_ms / 1e3), 1) if k1_traffic else None,
                "traffic_frac_of_copy_ceiling": round(k1_traffic / 1e9 / (k1_ms / 1e3) / ceiling, 4) if k1_traffic else None}
        # The whole frame against its two floors (round 6): every kernel's VALU instructions and counted HBM bytes from the
        # same committed counter passes, priced as if they could be packed perfectly -- instruction issue at the blend's
        # measured mix (27 of its 40 step instructions at 2.5 cycles, 13 at 4.2: 3.05 cycles per wave64 instruction,
        # profiles/r04_valu_probe.jsonl) and at the 2-cycle peak; bytes at the chip's achievable 6.3 TB/s (three K1 in flight
        # reach it; MI355X_MICROARCH.md) -- beside the measured period.  No schedule of these kernels goes below the larger.
        if tdoc and not args.backward:
            kk = [v for k, v in tdoc["kernels"].items() if k in stage_names]
            f_valu = sum(v.get("SQ_INSTS_VALU", 0) for v in kk)
            f_bytes = sum((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024 for v in kk)
            period_us = 1e6 * elapsed / args.steps
            floor_mix = f_valu * 3.05 / (1024 * 2.4e9) * 1e6
            floor_hbm = f_bytes / 6.3e12 * 1e6
            roofline["frame"] = {
                "valu_instr_per_frame": int(f_valu), "hbm_bytes_per_frame": int(f_bytes),
                "floor_us_valu_issue_at_measured_mix": round(floor_mix, 1),
                "floor_us_valu_issue_at_2_cycle_peak": round(f_valu * 2.0 / (1024 * 2.4e9) * 1e6, 1),
                "floor_us_hbm_at_6.3_TBps": round(floor_hbm, 1), "period_us": round(period_us, 1),
                "floor_over_period": round(max(floor_mix, floor_hbm) / period_us, 4),
                "sum_of_kernels_alone_us": round(1e3 * sum(stage_alone.get(k, 0.0) for k in stage_names), 1),
                "source": "profiles/" + tfile}
        T_tiles = ((W + 15) // 16) * ((H + 15) // 16)
        sort_passes = (32 + higher_msb(T_tiles) + 7) // 8
        if blend_like:
            roofline["note"] = ("the alpha blend evaluates exp() for every (pixel, Gaussian) pair that survives culling: "
                                "its limiter is VALU issue (`valu`), its HBM traffic (`hbm`, SURVEY 8d accounting) is a few "
                                "percent of peak by construction; the HBM-bound kernel of the frame is the streaming cull "
                                "inside `preprocess` (DESIGN.md section 5)")

        mode = "forward+backward" if args.backward else "forward"
        out = {
            "metric": ("rendered frames/sec (fwd+bwd) @ %d Gaussians, %dx%d" if args.backward else
                       "rendered frames/sec (fwd) @ %d Gaussians, %dx%d") % (P, W, H),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "pre_frames": pre_frames,
            "repeats": len(block_s), "value_min": round(total_frames / max(block_s), 3),
            "value_max": round(total_frames / min(block_s), 3),
            "value_is": "K frames / the median of `repeats` timed blocks of exactly K frames each",
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "ms_per_step_with_stage_events": round(1e3 * elapsed_instrumented / args.steps, 4),
            "frame_latency_ms": round(frame_latency_ms, 4),
            "entry_point": ("native module, positional rasterize_gaussians() -> (num_rendered:int, ...): one host wait "
                            "per frame" if (args.native_int_api or args.backward) else
                            "Python API GaussianRasterizer.forward -> (image, radii), dgr/__init__.py:223-273: num_rendered "
                            "is not part of it and is not waited for"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (gcity-synth-v1 %s, seed %d)" % (cfg["scene"], cfg["seed"])
                    + ("; A/B: Gaussians RENUMBERED along a Morton curve -- not the workload" if args.scene_order != "seeded" else ""),
            "config": {"workload": "%s: %s %d Gaussians, %dx%d, SH degree %d, %s, 24-pose orbit"
                                   % (args.config, cfg["scene"], P, W, H, cfg["sh_degree"], mode),
                       "parallelism": "frames sharded round-robin, one frame per GPU, no data-path collective; "
                                      "%d HIP streams per GPU alternate over consecutive frames (`value` is a throughput "
                                      "with that many frames in flight; `frame_latency_ms` is one frame alone)" % len(streams),
                       "exp": "%s (bit-exact vs oracle)" % NUMERICS,
                       "tile_sort": "every list sorted whole (--sort-whole)" if args.sort_whole else
                                    "lists beyond 1024 entries sorted lazily, as far as the blend walks (default)",
                       "cull": ("STATIC SCENE (--static-scene): a second line, not the headline -- the cull streams a cached "
                                "16-byte (mean, bound) record per Gaussian built once for the scene; same frames bit for bit")
                               if args.static_scene else "stateless: every frame reads means, scales, rotations (default)",
                       "options_set_on_the_command_line": list(args.opt)},
            "frame_stats": {"num_rendered": R_mean, "consumed_entries_Rp": Rp_mean, "visible": Pv_mean,
                            "tiles": T_tiles, "entries_per_tile": round(R_mean / T_tiles, 1),
                            "ns_per_instance": round(1e9 * elapsed / args.steps / max(R_mean, 1.0), 4)},
            "stages_ms": stages,
            "roofline": roofline,
            "ranks": ranks_info,
        }
        if other_blocks:
            oel = float(np.median(other_blocks))
            out["other_entry_point"] = {
                "entry_point": "Python API GaussianRasterizer.forward" if args.native_int_api else
                               "native module, positional rasterize_gaussians() -> int num_rendered (the host waits for it "
                               "in every frame, as the reference does)",
                "value": round(total_frames / oel, 3), "unit": "frames/s", "ms_per_step": round(1e3 * oel / args.steps, 4),
                "repeats": len(other_blocks)}
            # which line is like for like with the reference.  Its binding returns num_rendered as an int
            # (cr/rasterizer_impl.cu:236-238: a blocking read-back in every frame); a speed-up against a reference number
            # is to be taken from the line whose entry point keeps that contract
            out["like_for_like_with_the_reference"] = {
                "line": "value" if args.native_int_api else "other_entry_point.value",
                "why": "the positional rasterize_gaussians() of the native module returns num_rendered as an int, one host "
                       "wait per frame, as the reference's binding does; the Python API line is faster because that API "
                       "never hands the number out and this build does not wait for it"}

        # ---- CPU baseline: the oracle on a bounded sample of the same workload --------------
        if world == 1 and getattr(args, "full_cpu_mask", None):
            os.sched_setaffinity(0, args.full_cpu_mask)  # (the timed GPU region is over; the oracle wants every core)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            kw = oracle_kwargs(cams[0], sc, use_sh)
            O.lib()
            O.Frame(**kw)  # warm-up (page-in, OpenMP pool)
            n_cpu, cpu_s = 0, 0.0
            while n_cpu < len(cams) and cpu_s < 12.0:  # bounded sample: <= 24 poses or ~12 s
                rs = cams[n_cpu]
                kw.update(view_matrix=rs.view_matrix.cpu().numpy(), proj_matrix=rs.proj_matrix.cpu().numpy(),
                          campos=rs.campos.cpu().numpy())
                tc = time.perf_counter()
                fr = O.Frame(**kw)
                if args.backward:
                    fr.backward(dpix_main.cpu().numpy())
                cpu_s += time.perf_counter() - tc
                n_cpu += 1
            _, o = fwd(n_cpu - 1)
            same = bool(np.array_equal(o[1].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32)))
            if not args.backward:  # ... and the same frame through the timed entry point (the Python API's inference frame)
                same = same and bool(np.array_equal(fwd.api(n_cpu - 1)[0].cpu().numpy().view(np.uint32), fr.out_color.view(np.uint32)))
            threads = O.num_threads()
            O.set_num_threads(1)
            tc = time.perf_counter()
            fr1 = O.Frame(**kw)
            if args.backward:
                fr1.backward(dpix_main.cpu().numpy())
            one_s = time.perf_counter() - tc
            O.set_num_threads(threads)
            out["cpu_baseline"] = {"value": round(n_cpu / cpu_s, 4), "unit": "frames/s", "cores": threads,
                                   "kind": "port",
                                   "sample": "%d orbit frames (%s) of the same workload in %.1f s (oracle/, OpenMP)"
                                             % (n_cpu, mode, cpu_s),
                                   "single_thread_frames_per_s": round(1.0 / one_s, 4),
                                   "host_cpus": os.cpu_count(), "gpu_image_bit_exact_vs_cpu": same}
            if args.config != "C1":
                out["cpu_baseline"]["C1"] = cpu_baseline_c1(O, synth, GaussianRasterizerWrapper, torch, oracle_kwargs)

        # ---- secondary metric: C2 forward+backward ms/frame ----------------------------------
        if world == 1 and not args.no_secondary and args.config != "C2" and not args.backward:
            del fwd, step_fn
            torch.cuda.empty_cache()
            cfg2, sc2, cams2, use_sh2, fwd2 = load_scene("C2", None if args.points is None else min(args.points, 500000))
            W2, H2, P2 = cfg2["W"], cfg2["H"], cfg2["P"]
            dpix = torch.from_numpy(synth.grad_image(W2, H2, cfg2["seed"])).to(dev)
            fb = make_fwd_bwd(fwd2, dpix)
            n2 = 24
            for i in range(3 + n2):  # capacity hints, clocks, allocator pools
                fb(i)
            # value: n2 frames back to back on ONE stream (a training iteration is serial: its forward needs the
            # previous step's update), no instrumentation
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n2):
                fb(3 + i)
            torch.cuda.synchronize()
            ms2 = 1e3 * (time.perf_counter() - t1) / n2
            # per-stage times: the same frames again with HIP events around every stage (a few us per frame)
            # (two passes, the smaller average per stage: one preempted frame in 24 otherwise doubles a 10 us stage)
            N.set_option("timing", 1)
            st2 = None
            for _ in range(2):
                N.stage_ms()
                for i in range(n2):
                    fb(3 + i)
                torch.cuda.synchronize()
                st = N.stage_ms()
                st2 = st if st2 is None else {k: min(v, st[k]) for k, v in st2.items()}
            N.set_option("timing", 0)
            R2, Rp2, Pv2 = frame_statistics(fwd2, list(range(3, 3 + n2)), P2, W2, H2)
            ab2 = algorithmic_bytes(P2, Pv2, R2, Rp2, W2, H2, sc2["shs"].shape[1], True)
            sec_stages = {k: {"ms": round(v, 4), "alg_MB": round(ab2[k] / 1e6, 3),
                              "alg_GBps": round(ab2[k] / 1e9 / (v / 1e3), 1) if v > 0 else None}
                          for k, v in st2.items() if k in ab2}
            bwd_ms = st2.get("blend_bwd", 0.0)
            bwd_ach = ab2["blend_bwd"] / 1e9 / (bwd_ms / 1e3) if bwd_ms > 0 else 0.0
            t2doc, t2file = traffic_doc("c2") if args.points is None else (None, None)
            tk2 = t2doc["kernels"].get("blend_bwd") if t2doc else None
            sec_roof = {"kernel": "blend_bwd", "launch_ms": round(bwd_ms, 4), "alg_bytes_per_launch": int(ab2["blend_bwd"]),
                        "hbm": {"achieved": round(bwd_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(bwd_ach / HBM_PEAK_GBS, 4),
                                "traffic": int((2 * tk2["FETCH_SIZE_KB"] + tk2["WRITE_SIZE_KB"]) * 1024) if tk2 else None}}
            if tk2 and tk2.get("SQ_INSTS_VALU") and bwd_ms > 0:
                g2 = tk2["SQ_INSTS_VALU"] / 1e9 / (bwd_ms / 1e3)
                sec_roof.update({"bound": "valu", "achieved": round(g2, 1), "peak": VALU_PEAK_GINSTR,
                                 "unit": "G wave-instr/s", "frac": round(g2 / VALU_PEAK_GINSTR, 4),
                                 "instr_per_launch": int(tk2["SQ_INSTS_VALU"]), "source": "profiles/" + t2file,
                                 "source_matches_this_tree": traffic_is_current(
                                     t2doc, ("gcr_blend.hip", "gcr_device.h", "gcr_cull.h"))})
            out["secondary"] = {"metric": "fwd+bwd ms/frame @ %d Gaussians, %dx%d, SH3" % (P2, W2, H2),
                                "value": round(ms2, 4), "unit": "ms/frame", "higher_is_better": False,
                                "frame_stats": {"num_rendered": R2, "consumed_entries_Rp": Rp2, "visible": Pv2},
                                "stages_ms": sec_stages, "roofline": sec_roof}
            if not args.no_cpu_baseline:
                # C2 CPU baseline (SURVEY 8d "must"): oracle forward + backward on the same frames, with the
                # parity gate (image bit-exact, every gradient within 1e-4 * max) in the same run
                from oracle import oracle as O
                kw2 = oracle_kwargs(cams2[3], sc2, True)
                dnp = dpix.cpu().numpy()
                n_cpu, cpu_s = 0, 0.0
                while n_cpu < 12 and cpu_s < 8.0:
                    rs = cams2[(3 + n_cpu) % len(cams2)]
                    kw2.update(view_matrix=rs.view_matrix.cpu().numpy(), proj_matrix=rs.proj_matrix.cpu().numpy(),
                               campos=rs.campos.cpu().numpy())
                    tc = time.perf_counter()
                    fr2 = O.Frame(**kw2)
                    gref = fr2.backward(dnp)
                    cpu_s += time.perf_counter() - tc
                    n_cpu += 1
                _, o2, g2t = fb(3 + n_cpu - 1)
                same2 = bool(np.array_equal(o2[1].cpu().numpy().view(np.uint32), fr2.out_color.view(np.uint32)))
                names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
                worst = 0.0
                for nme, tg in zip(names, g2t):
                    ref = gref[nme]
                    worst = max(worst, float(np.abs(ref - tg.cpu().numpy().reshape(ref.shape)).max())
                                / max(1.0, float(np.abs(ref).max())))
                out["secondary"]["cpu_baseline"] = {
                    "value": round(1e3 * cpu_s / n_cpu, 3), "unit": "ms/frame", "cores": O.num_threads(), "kind": "port",
                    "sample": "%d frames forward+backward in %.1f s (oracle/, OpenMP)" % (n_cpu, cpu_s),
                    "gpu_image_bit_exact_vs_cpu": same2, "worst_gradient_error_over_max": float("%.3g" % worst)}
            # the reference-faithful variant: render the full 960x540 sensor, keep a 640x448 crop
            # (utils/helpers.py:255-260) -- the loss only sees the crop, so dL/dpixel is zero outside it
            del fwd2, fb
            torch.cuda.empty_cache()
            _, _, _, _, fwd3 = load_scene("C2", None if args.points is None else min(args.points, 500000), size=(960, 540))
            dfull = torch.zeros((3, 540, 960), dtype=torch.float32, device=dev)
            y0, x0 = (540 - H2) // 2, (960 - W2) // 2
            dfull[:, y0:y0 + H2, x0:x0 + W2] = dpix
            fb3 = make_fwd_bwd(fwd3, dfull)
            for i in range(3):
                fb3(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n2):
                fb3(3 + i)
            torch.cuda.synchronize()
            out["secondary"]["render_960x540_then_crop_ms"] = round(1e3 * (time.perf_counter() - t1) / n2, 4)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_c1(O, synth, Wrapper, torch, oracle_kwargs):
    """C1 (BASELINE configs[0]: 10k random Gaussians, 256x256, SH degree 0, forward only, the CPU-runnable plumbing
    case -- SURVEY 8d "must"): the oracle on the host cores, all 24 orbit poses."""
    cfg, sc = synth.make_scene("C1")
    W, H = cfg["W"], cfg["H"]
    wr = Wrapper(synth.intrinsics(W, H), (W, H), device=torch.device("cpu"))
    cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=0) for p, q in synth.orbit_poses()]
    O.Frame(**oracle_kwargs(cams[0], sc, True))
    tc = time.perf_counter()
    for rs in cams:
        O.Frame(**oracle_kwargs(rs, sc, True))
    dt = time.perf_counter() - tc
    return {"value": round(len(cams) / dt, 2), "unit": "frames/s", "cores": O.num_threads(), "kind": "port",
            "sample": "24 orbit frames, 10000 Gaussians, 256x256, SH degree 0, forward (oracle/, OpenMP)"}


def committed_traffic(tag, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r01_traffic_<tag>.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic_%s.json" % tag)))
    if not files:
        return None
    k = json.load(open(files[-1]))["kernels"].get(kernel)
    return int((2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024) if k else None


def grid_encoder_bench(args, torch, dist, dev, world, rank, barrier, copy_ceiling):
    """Row f3: GaussianCity's positional encoder (models/generator.py:37-42): D=5 inputs, 16 levels x 8 channels,
    2^19 rows per level (268 MB fp32 table).  One step = forward + backward (table gradient + input gradient)
    over B points -- what one G-step does (core/train.py:263-295); every rank encodes its own B points against
    its replica of the table (data parallel; the table gradient is part of the DDP all-reduce, not timed here)."""
    import math
    from gaussiancity_amd import _native_e as E
    from gaussiancity_amd.grid_encoder import GridEncoder
    B = args.encoder_points
    enc = GridEncoder(in_channels=5, n_levels=16, lvl_channels=8, desired_resolution=2048).to(dev)
    torch.manual_seed(1234 + rank)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = (torch.rand(B, 5, device=dev) * 2 - 1).requires_grad_(True)
    g = torch.randn(B, 128, device=dev)

    def step():
        enc.embeddings.grad = None
        x.grad = None
        y = enc(x)
        y.backward(g)
        return y

    for _ in range(60):  # allocator pools and clocks up before the contract's W warm-up steps
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    E.set_option("timing", 1)
    E.stage_ms()
    for _ in range(min(args.steps, 20)):
        y = step()
    torch.cuda.synchronize()
    st = E.stage_ms()
    E.set_option("timing", 0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        D, L, Cc = 5, 16, 8
        # algorithmic bytes per point and level: forward gathers 2^D rows of 4*C B, reads D*4 B of input, writes
        # 4*C B and (with input gradients) D*C*4 B of dy_dx; backward adds 2^D rows (read-modify-write = 2 x 4*C B)
        ab = {"forward": B * L * ((1 << D) * 4 * Cc + 4 * D + 4 * Cc + 4 * D * Cc),
              "backward_embeddings": B * L * ((1 << D) * 8 * Cc + 4 * D + 4 * Cc),
              "backward_inputs": B * L * (4 * D * Cc + 4 * Cc) + 4 * B * D}
        stages = {k: {"ms": round(st[k], 4), "alg_MB": round(ab[k] / 1e6, 2),
                      "alg_GBps": round(ab[k] / 1e9 / (st[k] / 1e3), 1) if st[k] > 0 else None} for k in ab}
        dom = max(ab, key=lambda k: st[k])
        ceiling = copy_ceiling()
        ach = ab[dom] / 1e9 / (st[dom] / 1e3)
        out = {"metric": "hash-grid encoder points/sec (forward + backward) @ D=5, 16 levels x 8 ch, 2^19 rows/level",
               "value": round(B * args.steps * world / elapsed, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (uniform points, uniform table)",
               "config": {"workload": "E1: %d points x 16 levels, 268 MB table, forward + table gradient + input gradient" % B,
                          "parallelism": "data parallel: every rank encodes its own points against its replica of the table"},
               "stages_ms": stages,
               "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4),
                            "traffic": committed_traffic("grid_encoder", dom) if B == 16384 else None,
                            "copy_ceiling_GBps": round(ceiling, 1),
                            "frac_of_copy_ceiling": round(ach / ceiling, 4), "launch_ms": round(st[dom], 4),
                            "alg_bytes_per_launch": int(ab[dom])}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import grid_oracle as GO
            GO.lib()
            xn = ((x.detach() + 1) / 2).cpu().numpy()
            emb = enc.embeddings.detach().cpu().numpy()
            off = enc.offsets.cpu().numpy()
            S, H = math.log2(enc.per_level_scale), enc.base_resolution
            grad_lbc = np.ascontiguousarray(g.view(B, 16, 8).permute(1, 0, 2).cpu().numpy())
            GO.forward(xn[:256], emb, off, S, H, True)  # warm-up (OpenMP pool, page-in)
            n_rep, tc = 0, time.perf_counter()
            while time.perf_counter() - tc < 10.0:
                out_o, dd_o = GO.forward(xn, emb, off, S, H, True)
                GO.backward(grad_lbc, xn, emb.shape, off, S, H, dd_o)
                n_rep += 1
            cpu_s = time.perf_counter() - tc
            same = bool(np.array_equal(torch.from_numpy(out_o).permute(1, 0, 2).reshape(B, 128).numpy().view(np.uint32),
                                       y.detach().cpu().numpy().view(np.uint32)))
            out["cpu_baseline"] = {"value": round(B * n_rep / cpu_s, 1), "unit": "points/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "%d forward+backward passes over the same %d points in %.1f s (oracle/, OpenMP)"
                                             % (n_rep, B, cpu_s), "gpu_encoding_bit_exact_vs_cpu": same}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def visibility_bench(args, torch, dist, synth, dev, world, rank, barrier, copy_ceiling):
    """Row f2 (scripts/dataset_generator.py:1251-1461 as called per frame by scripts/inference.py:296-333):
    BEV maps -> footprint extruder -> point-id volume -> perspective traversal -> vp_map / ins_map, everything
    resident in HBM.  One step = one pose of the 24-pose orbit over the same 2048 x 2048 layout (upstream
    re-extrudes and re-voxelises per frame too); frames are independent, so rank r takes poses r, r+N, ...
    (weak scaling, no collective)."""
    from gaussiancity_amd import _native_v as V
    from gaussiancity_amd import points as PT
    size = args.layout_size
    L = synth.s_layout(size, 2001)
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    maps = [torch.from_numpy(L[k]).to(dev) for k in ("INS", "TD_HF", "BU_HF", "PTS")]
    Wimg, Himg = 960, 540
    rig = synth.layout_camera(size, Wimg, Himg)[0]
    poses = [synth.layout_camera(size, Wimg, Himg, pose=i)[1:] for i in range(24)]
    V.lib()

    workspaces = {}

    def frame(i):
        cam_pos, cam_quat = poses[i % len(poses)]
        rows = PT.extrude_points(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, *maps)
        ws = None
        if args.resident_volume:  # one resident all-zero volume per stream (restored after every traversal)
            key = torch.cuda.current_stream().cuda_stream
            ws = workspaces.setdefault(key, PT.VolumeWorkspace(dev))
        vp, ins = PT.visible_point_map(rows, rig, cam_pos, cam_quat, 0, use_jumps=bool(args.jumps), workspace=ws)
        return rows, vp, ins

    # frames are independent: consecutive frames alternate over HIP streams, so frame f+1's volume clear
    # (HBM-bound) and scatter overlap frame f's traversal (instruction-bound); the two host round trips of a
    # frame (point count, bounding box) only wait for their own stream
    vstreams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    def run(lo, hi):
        for i in range(lo, hi):
            with torch.cuda.stream(vstreams[i % len(vstreams)]):
                frame(rank + i * world)

    run(0, 36)  # stream queues, allocator pools and clocks up before the contract's W warm-up frames
    run(0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.warmup, args.warmup + args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    V.set_option("timing", 1)
    V.stage_ms()
    run(0, min(args.steps, 36))          # as the timed region: frames alternating over the streams
    torch.cuda.synchronize()
    st = V.stage_ms()
    for i in range(min(args.steps, 12)):  # every kernel alone on the GPU
        rows, vp, ins = frame(rank + i * world)
    torch.cuda.synchronize()
    st_alone = V.stage_ms()
    V.set_option("timing", 0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        n_pts = int(rows.shape[0])
        mn, mx = rows[:, :3].min(dim=0).values.cpu().numpy(), rows[:, :3].max(dim=0).values.cpu().numpy()
        w, h, d = int(mx[0]) - int(mn[0]) + 1, int(mx[1]) - int(mn[1]) + 1, int(mx[2]) - int(mn[2]) + 2
        npix_map = size * size
        # algorithmic bytes per stage: maps 7 B/pixel per pass (count, emit) + 10 B/point
        # written; volume clear 4 B/voxel; scatter 10 B/point read + 4 B per written voxel; traversal: outputs
        # only (24 B/pixel) -- what it reads depends on the scene, so it is reported as time, not GB/s
        voxels_written = int((rows[:, 3].long() ** 3).sum().item())
        ab = {"extrude_count": 7 * npix_map, "extrude_emit": 7 * npix_map + 10 * n_pts,
              "volume_clear": (10 * n_pts + 4 * voxels_written) if args.resident_volume else 4 * h * w * d,
              "volume_scatter": 10 * n_pts + 4 * voxels_written,
              "traversal": 24 * Himg * Wimg}
        stages = {k: {"ms": round(st[k], 4), "ms_single_stream": round(st_alone[k], 4), "alg_MB": round(ab[k] / 1e6, 2),
                      "alg_GBps": round(ab[k] / 1e9 / (st[k] / 1e3), 1) if st.get(k, 0) > 0 and k != "traversal" else None}
                  for k in ab}
        dom = max(ab, key=lambda k: st.get(k, 0.0))
        hbm_dom = max((k for k in ab if k != "traversal"), key=lambda k: st.get(k, 0.0))
        ceiling = copy_ceiling()
        ach = ab[hbm_dom] / 1e9 / (st[hbm_dom] / 1e3)
        out = {
            "metric": "visibility frames/sec (BEV maps -> points -> volume -> first-hit map) @ %dx%d layout, %dx%d image"
                      % (size, size, Wimg, Himg),
            "value": round(args.steps * world / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int16/int32 (+ f32 ray setup)",
            "data": "synthetic (gcity-layout-v1, seed 2001)",
            "config": {"workload": "V1: %dx%d BEV layout, %d extruded points, %dx%dx%d int32 volume (%.2f GB), %dx%d rays, "
                                   "24-pose orbit" % (size, size, n_pts, h, w, d, 4 * h * w * d / 1e9, Wimg, Himg),
                       "parallelism": "frames sharded round-robin, one frame per GPU, no data-path collective; %d HIP "
                                      "streams per GPU alternate over consecutive frames" % len(vstreams),
                       "empty_space_jumps": bool(args.jumps), "resident_volume": bool(args.resident_volume)},
            "frame_stats": {"points": n_pts, "voxels_written": voxels_written,
                            "hit_fraction": round(float((vp >= 0).float().mean().item()), 4)},
            "stages_ms": stages, "longest_stage": dom,
            "roofline": {"kernel": hbm_dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": committed_traffic("visibility", hbm_dom) if size == 2048 else None,
                         "copy_ceiling_GBps": round(ceiling, 1),
                         "frac_of_copy_ceiling": round(ach / ceiling, 4), "launch_ms": round(st[hbm_dom], 4),
                         "launch_ms_single_stream": round(st_alone[hbm_dom], 4),
                         "alg_bytes_per_launch": int(ab[hbm_dom]),
                         "note": "the traversal is latency-bound pointer chasing through the volume (no byte model): "
                                 "reported as time; the roofline object describes the slowest streaming stage"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = visibility_cpu_baseline(L, inv, synth, rows, (h, w, d), mn, poses[0], rig, vp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def visibility_cpu_baseline(L, inv, synth, rows_gpu, dims, mn, pose, rig, vp_gpu_last):
    """CPU side of one frame of the same workload: the REFERENCE's own extruder (oracle/_ref, compiled from the
    reference source, single-threaded as upstream) when present -- kind "reference" -- else the oracle port;
    volume + traversal = oracle port (the reference has no CPU path for them; traversal on all cores)."""
    from gaussiancity_amd import points as PT
    from oracle import points_oracle as PO
    PO.lib()
    ref = PO.reference_extruder()
    a = (True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, L["INS"], L["TD_HF"], L["BU_HF"], L["PTS"])
    tc = time.perf_counter()
    cpu_pts = ref.get_points_from_projection(*a) if ref is not None else PO.extrude(*a)
    t_ext = time.perf_counter() - tc
    same_pts = bool(np.array_equal(cpu_pts, rows_gpu.cpu().numpy().view(np.uint16)))
    p16 = cpu_pts.astype(np.int16)
    loc = p16[:, :3] - np.array([mn[0], mn[1], mn[2] - 1], np.int16)
    h, w, d = dims
    tc = time.perf_counter()
    vol = PO.points_to_volume(loc, np.arange(1, len(loc) + 1, dtype=np.int32), np.repeat(p16[:, [3]], 3, axis=1), h, w, d)
    t_vol = time.perf_counter() - tc
    cp = np.array(pose[0], np.float64) - np.array(mn, np.float64)
    look = PT.get_camera_look_at(cp, pose[1])
    K, sensor = rig["intrinsics"], rig["sensor_size"]
    tc = time.perf_counter()
    vid, _, _ = PO.ray_voxel_intersection_perspective(
        vol, np.array([cp[1], cp[0], cp[2]], np.float32),
        np.array([look[1] - cp[1], look[0] - cp[0], look[2] - cp[2]], np.float32), np.array([0, 0, 1], np.float32),
        K[0], [K[5], K[2]], [sensor[1], sensor[0]], 1)
    t_rv = time.perf_counter() - tc
    vp0, _ = PT.visible_point_map(rows_gpu, rig, pose[0], pose[1], 0)
    same_vp = bool(np.array_equal(vp0.cpu().numpy(), vid.squeeze().astype(np.int64) - 1))
    total = t_ext + t_vol + t_rv
    return {"value": round(1.0 / total, 4), "unit": "frames/s", "cores": 1, "kind": "reference" if ref is not None else "port",
            "sample": "1 frame of the same workload: extruder %.2f s (%s), volume %.2f s + traversal %.2f s (oracle port, "
                      "traversal on all cores)" % (t_ext, "the reference's footprint_extruder.cpp" if ref is not None
                                                   else "oracle port", t_vol, t_rv),
            "gpu_points_bit_exact_vs_cpu": same_pts, "gpu_first_hit_map_bit_exact_vs_cpu": same_vp}


def inference_loop_bench(args, torch, dist, synth, Wrapper, dev, world, rank, barrier):
    """The product's inference loop around the rasterizer (scripts/inference.py:614-669): for every pose of the orbit,
    points [N,14] -> GaussianRasterizerWrapper -> [3,H,W] -> uint8 [H,W,3] frame in host memory.  N = 518 400 points
    (one per pixel of a 960x540 view: what the reference's visible-point pipeline feeds it), precomputed colours,
    identity rotations, opacity 1.  frames.InferenceLoop alternates the frames over three side streams and pinned
    frame buffers; ranks take frames round-robin (no collective)."""
    from gaussiancity_amd.frames import InferenceLoop
    n_pts = args.points or 518400
    cfg, sc = synth.make_scene("C4", n_pts)
    W, H = cfg["W"], cfg["H"]
    wr = Wrapper(synth.intrinsics(W, H), (W, H), device=dev, host_camera={"device": False, "closed-form": True, "reference": "reference"}[args.host_camera])
    rot = np.zeros((n_pts, 4), np.float32)
    rot[:, 0] = 1.0
    pts_np = np.concatenate([sc["means3D"], np.ones((n_pts, 1), np.float32), sc["scales"], rot, sc["colors_precomp"]], axis=1)
    points = torch.from_numpy(pts_np.astype(np.float32)).to(dev)
    orbit = synth.orbit_poses()
    # --float-frames: the reference's route (float image, then tensor_to_image * 255 -> uint8 with torch kernels);
    # default: the blend kernel stores the uint8 frame itself (GaussianRasterizerWrapper(as_uint8=True): the same bytes)
    loop = InferenceLoop(lambda p, cp, cq: wr(p, cp, cq), device=dev, n_streams=args.streams,
                         render_uint8_fn=None if args.float_frames else (lambda p, cp, cq: wr(p, cp, cq, as_uint8=True)))
    sink = [0]

    def consume(i, frame):  # what a video writer would do with the frame: touch it
        sink[0] += int(frame[0, 0, 0])

    def run(n):
        poses = [orbit[(rank + i * world) % len(orbit)] for i in range(n)]
        with torch.no_grad():
            loop.run(points, poses, consume=consume)

    run(max(args.warmup, 24))
    block_s = []
    while True:
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        block_s.append(el)
        if sum(block_s) >= 0.5 or len(block_s) >= 16:
            break
    elapsed = float(np.median(block_s))
    if rank == 0:
        print(json.dumps({
            "metric": "inference-loop frames/sec (points [N,14] -> wrapper -> uint8 frame on the host)",
            "value": round(args.steps * world / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "repeats": len(block_s), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "value_min": round(args.steps * world / max(block_s), 3), "value_max": round(args.steps * world / min(block_s), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (gcity-synth-v1 %s, seed %d)" % (cfg["scene"], cfg["seed"]),
            "config": {"workload": "%d points [N,14] with precomputed colours, %dx%d, 24-pose orbit, uint8 HWC frames "
                                   "copied to pinned host memory (scripts/inference.py:655-667)" % (n_pts, W, H),
                       "parallelism": "frames round-robin over ranks; %d HIP streams per GPU" % loop.n,
                       "camera": args.host_camera,
                       "frames": "float image + torch conversion kernels" if args.float_frames else
                                 "uint8 frame stored by the blend kernel"},
            "frame_bytes_to_host": 3 * W * H}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def train_step_bench(args, torch, dist, N, synth, Wrapper, dev, world, rank, barrier):
    """C4: the reference's G-step shape around the rasterizer -- a 69,809,101-parameter generator
    stand-in under torch's DistributedDataParallel (buckets all-reduced over RCCL while the backward runs, core/train.py:
    78-87), 16 384 points with precomputed colours through helpers.get_gaussian_rasterization (wrapper -> one autograd
    node on the [N,14] tensor), render 960x540, crop 640x448, L1 loss, backward (core/train.py:263-295).  One frame per
    rank per step; the DDP all-reduce is the path's only collective.  No optimizer step (not part of SURVEY's C4)."""
    from gaussiancity_amd import helpers
    from gaussiancity_amd import frames as frames_mod
    from gaussiancity_amd.frames import DDPTrainStep, StandInGenerator, allreduce_gradients
    cfg, sc = synth.make_scene("C4", args.points)
    W, H = cfg["W"], cfg["H"]
    cw, ch = cfg["crop"]
    wr = Wrapper(synth.intrinsics(W, H), (W, H), device=dev, host_camera={"device": False, "closed-form": True, "reference": "reference"}[args.host_camera])
    # [N,14] = xyz, opacity, scale3, rot4, rgb3  (dgr/__init__.py:404-409)
    rot = sc["rotations"][:, [1, 2, 3, 0]]
    pts_np = np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], rot, sc["colors_precomp"]], axis=1)
    base_points = torch.from_numpy(pts_np.astype(np.float32)).to(dev)
    target = torch.zeros((3, ch, cw), dtype=torch.float32, device=dev)
    crop = ((W - cw) // 2, (H - ch) // 2, cw, ch)
    gen = StandInGenerator(device=dev)
    n_param = sum(p.numel() for p in gen.parameters())
    dis = frames_mod.StandInDiscriminator(device=dev) if args.d_step else None
    n_param_d = sum(p.numel() for p in dis.parameters()) if dis is not None else 0
    h = DDPTrainStep(wr, gen, crop=crop, discriminator=dis)
    poses = synth.orbit_poses()

    def step(i):
        pos, quat = poses[(rank + i * world) % len(poses)]
        return h.step(base_points, pos, quat, target)

    for i in range(max(args.warmup, 12)):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    host_s = time.perf_counter() - t0  # the host has enqueued every step; the GPU may still be working
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- the rasterizer leg alone (what the product calls per frame): helpers -> wrapper -> autograd, loss, backward
    leaf = base_points.clone().requires_grad_(True)
    box = [{"x": crop[0], "y": crop[1], "w": cw, "h": ch}]

    def leg(i):
        pos, quat = poses[(rank + i * world) % len(poses)]
        leaf.grad = None
        img = helpers.get_gaussian_rasterization(leaf[None], wr, [pos], [quat], crop_bboxes=box)[0]
        (img - target).abs().mean().backward()

    for i in range(12):
        leg(i)
    # the leg is host-bound (0.05 ms of GPU work per frame) and host timing on a shared box is noisy: five blocks,
    # the median is reported and the fastest beside it
    leg_blocks, leg_host_s = [], 0.0
    for _ in range(5):
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            leg(i)
        leg_host_s += time.perf_counter() - t1
        barrier()
        leg_blocks.append(time.perf_counter() - t1)
    leg_s, leg_host_s = float(np.median(leg_blocks)), leg_host_s / 5

    # ---- the same leg as an UNCHANGED caller would run it: utils/helpers.py:250-270 knows no `crop`
    # keyword -- it renders the full 960x540 frame through rasterizator(points, pos, quat), slices the crop out of the
    # image in Python and stacks the batch.  Same wrapper object, same autograd node underneath; what differs is that the
    # tiles outside the crop are blended and walked, and torch runs the slice / stack (and their backward) kernels.
    def unchanged_caller(gs_points, rasterizator, cam_pos, cam_quat, crop_bboxes):
        frames_out = []
        for b in range(gs_points.size(0)):
            full = rasterizator(gs_points[b], cam_pos[b], cam_quat[b])
            c = crop_bboxes[b]
            frames_out.append(full[:, c["y"]:c["y"] + c["h"], c["x"]:c["x"] + c["w"]])
        return torch.stack(frames_out, dim=0)

    def leg_unchanged(i):
        pos, quat = poses[(rank + i * world) % len(poses)]
        leaf.grad = None
        img = unchanged_caller(leaf[None], wr, [pos], [quat], box)[0]
        (img - target).abs().mean().backward()

    def blocks_of(fn, nblocks, nsteps):
        out = []
        for i in range(12):
            fn(i)
        for _ in range(nblocks):
            barrier()
            t1 = time.perf_counter()
            for i in range(nsteps):
                fn(i)
            barrier()
            out.append(1e3 * (time.perf_counter() - t1) / nsteps)
        return out

    def spread(v):
        v = sorted(v)
        return {"median": round(float(np.median(v)), 4), "min": round(v[0], 4), "p10": round(v[len(v) // 10], 4),
                "p90": round(v[(9 * len(v)) // 10], 4), "max": round(v[-1], 4), "blocks": len(v)}

    nsteps_leg = max(16, min(args.steps, 64))
    leg_windowed_blocks = blocks_of(leg, 30, nsteps_leg)
    leg_unchanged_blocks = blocks_of(leg_unchanged, 30, nsteps_leg)
    N.set_option("timing", 1)
    N.stage_ms()
    for i in range(48):
        leg(i)
    torch.cuda.synchronize()
    st = N.stage_ms()
    N.set_option("timing", 0)
    raster_ms = sum(st.values())

    # ---- the collective alone, same bytes, the in-place bucketed exchange of frames.allreduce_gradients
    ar_ms, n_msg, flat, ar_d_ms = None, None, None, None
    if world > 1:
        def time_allreduce(n):
            buf = torch.zeros(n, dtype=torch.float32, device=dev)
            for _ in range(2):
                allreduce_gradients([buf])
            barrier()
            t2 = time.perf_counter()
            for _ in range(5):
                msgs = allreduce_gradients([buf])
            barrier()
            return 1e3 * (time.perf_counter() - t2) / 5, msgs
        ar_ms, n_msg = time_allreduce(n_param)
        ar_d_ms = time_allreduce(n_param_d)[0] if n_param_d else 0.0  # the discriminator's message set (--d-step)
        tt = torch.tensor([elapsed, ar_ms, leg_s, ar_d_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, ar_ms, leg_s, ar_d_ms = (float(tt[k].item()) for k in range(4))
    ranks_info = args.collect_ranks()
    if rank == 0:
        nbytes = n_param * 4
        bus = (2.0 * (world - 1) / world) * nbytes / 1e9 / (ar_ms / 1e3) if world > 1 else None
        print(json.dumps({
            "metric": "training steps/sec (C4: generator stand-in under DDP + rasterizer fwd + L1 + bwd)",
            "value": round(args.steps * world / elapsed, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "host_ms": round(1e3 * host_s / args.steps, 4),
            "rasterizer_leg": {
                "what": "helpers.get_gaussian_rasterization -> wrapper -> autograd + crop + L1 + backward to the [N,14] leaf",
                "ms_per_frame": round(1e3 * leg_s / args.steps, 4), "host_ms": round(1e3 * leg_host_s / args.steps, 4),
                "ms_per_frame_fastest_block": round(1e3 * min(leg_blocks) / args.steps, 4),
                "raster_ms": round(raster_ms, 4), "stages_ms": {k: round(v, 4) for k, v in st.items() if v > 0},
                "ms_per_frame_over_30_blocks": spread(leg_windowed_blocks)},
            "rasterizer_leg_unchanged_caller": {
                "what": "the loop of utils/helpers.py:250-270 as an unchanged GaussianCity would run it: full 960x540 render "
                        "through rasterizator(points, pos, quat), Python slice of the crop, torch.stack -- then the same L1 "
                        "and backward (no `crop` keyword: every tile is blended and walked, torch runs the slice / stack kernels)",
                "ms_per_frame_over_30_blocks": spread(leg_unchanged_blocks), "frames_per_block": nsteps_leg},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (gcity-synth-v1 %s, seed %d)" % (cfg["scene"], cfg["seed"]),
            "config": {"workload": "C4: %d points, precomputed colours, %dx%d render, %dx%d crop, %d fp32 stand-in "
                                   "parameters (the BG generator's size; dense gradients produced layer by layer, 28 "
                                   "non-zero values per layer -- message sizes and readiness order of a generator, "
                                   "not its arithmetic); no optimizer step" % (cfg["P"], W, H, cw, ch, n_param),
                       "parallelism": "torch DistributedDataParallel(find_unused_parameters=True, bucket_cap_mb=64) over "
                                      "RCCL: buckets all-reduced while the backward runs; one frame per rank per step",
                       "camera": {"device": "the reference's recipe on the device (scipy + H2D + GEMM + inverse with a sync)",
                                  "closed-form": "closed-form host arithmetic, by value (opt-in)",
                                  "reference": "the reference's recipe with torch on the host, by value (the wrapper's "
                                               "default on a GPU: bit-equal to the golden camera)"}[args.host_camera]},
            "allreduce_ms": round(ar_ms, 4) if ar_ms else None, "allreduce_bytes": nbytes,
            "allreduce_messages": n_msg, "allreduce_bus_GBps": round(bus, 1) if bus else None,
            "d_step": ({"what": "the step includes the discriminator's half (core/train.py:227-257): a forward-only render of "
                                "the frame under no_grad, the discriminator stand-in on fake and real image under DDP, its "
                                "backward and all-reduce; the G-step carries the GAN term",
                        "discriminator_parameters": n_param_d, "allreduce_bytes": 4 * n_param_d,
                        "allreduce_ms": round(ar_d_ms, 4) if ar_d_ms else None,
                        "allreduce_bus_GBps": (round((2.0 * (world - 1) / world) * 4 * n_param_d / 1e9 / (ar_d_ms / 1e3), 1)
                                               if ar_d_ms else None)} if args.d_step else None),
            "rccl": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_DEBUG") if os.environ.get(k)} or None,
            "expected_allreduce": {"bytes": nbytes, "ring_ms_estimate": 3.2, "direct_all_links_ms_estimate": 0.46,
                                   "note": "SURVEY.md section 5: 279 MB fp32 over xGMI, 7 links x ~153 GB/s per GPU; a ring "
                                           "is per-link bound"},
            "ranks": ranks_info}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
