"""Randomised parity sweep: many small seeded frames with random sizes, Gaussian counts, extents, opacities, SH degrees,
camera poses, backgrounds, scale modifiers, training / inference frames and library options, each compared with the
CPU oracle -- image, per-pixel state and radii bit for bit, the final prefix of every tile list, the eight gradient
tensors within tolerance.  One JSON line per failing case, a summary at the end.

    python tools/fuzz_parity.py [--cases 300] [--wrapper-cases 150] [--seed 1] [--max-seconds 240]

A second sweep calls GaussianRasterizerWrapper on random [N,14] tensors (row strides, column offsets, flips, crop
windows, the three camera modes) and compares its single autograd node with the reference's route on the same GPU.

(tests/test_gpu_fuzz.py runs a fixed 40-case slice of the same generator under pytest -m gpu.)

Gradient tolerance.  1e-4 * max (the north-star bar) for scenes whose largest scale stays within 8 -- anisotropy up to
40:1, harsher than any BASELINE scene.  Beyond that (scales up to 14 against 0.2: near-degenerate 2D covariances,
1 / det^2 in the conic gradient) fp32 itself is the limit: in the first 400-case sweep 8 such cases exceeded 1e-4 * max
by up to 12x in dL_drot / dL_dscale / dL_dmean3D, and for the three of them that were re-computed in float64
(oracle/dense_f64.py; cases 58, 202, 317 of seed 1) the fp32 ORACLE is just as far from the float64 gradient (58: oracle
vs float64 0.0648 on dL_drot, GPU vs oracle 0.0625; 317: 0.00116 vs 0.00161; 202: 0.00235 vs 0.00253) -- different
summation orders of an ill-conditioned sum, not a defect of either.  Those cases are held to 5e-3 * max.  Dense frames
(thousands of overlapping Gaussians per tile) with scales within 8: 3 of 1 200 cases of seed 2 exceeded 1e-4 * max, by
at most 2.3x (dL_dscale / dL_drot); case 373 in float64: the oracle is 20x further from the float64 gradient (0.0057
on dL_dscale) than the GPU is from the oracle (0.0003) -- held to 3e-4 * max.  The forward is bit-exact everywhere
(1 600 cases of seeds 1 and 2).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

GRAD_TOL = 1e-4
GRAD_TOL_DENSE = 3e-4             # thousands of overlapping Gaussians per tile (see the module docstring)
GRAD_TOL_ILL_CONDITIONED = 5e-3   # scenes with scales beyond 8
GRADS = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")


def gradient_tolerance(c):
    """The tier a case's gradients are held to (see the module docstring): relative to max(1, max|ref|)."""
    if c["smax"] > 8.0:
        return GRAD_TOL_ILL_CONDITIONED
    return GRAD_TOL_DENSE if c["regime"] in ("dense", "translucent_pile", "opaque_pile") else GRAD_TOL


def draw_case(rng):
    """One random frame description (plain dict, JSON-able)."""
    W = int(rng.choice([16, 17, 31, 48, 64, 97, 160, 208, 333]))
    H = int(rng.choice([16, 23, 32, 48, 75, 128, 176]))
    regime = str(rng.choice(["sparse", "medium", "dense", "translucent_pile", "opaque_pile"]))
    P = {"sparse": int(rng.integers(0, 200)), "medium": int(rng.integers(200, 4000)),
         "dense": int(rng.integers(4000, 20000)), "translucent_pile": int(rng.integers(3000, 40000)),
         "opaque_pile": int(rng.integers(3000, 40000))}[regime]
    pile = regime.endswith("pile")
    in_blend = int(rng.random() < 0.15)   # (the blend's own sort of short lists leaves no lazy-sort state to check against)
    return dict(
        W=W, H=H, P=P, regime=regime, seed=int(rng.integers(1, 1 << 30)),
        deg=int(rng.integers(0, 4)), use_sh=bool(rng.random() < 0.7),
        spread=float(rng.uniform(1.0, 6.0) if pile else rng.uniform(10.0, 60.0)),
        smin=float(rng.uniform(0.2, 1.0)), smax=float(rng.uniform(1.5, 14.0)),
        omin=float(0.003 if regime == "translucent_pile" else rng.uniform(0.01, 0.5)),
        omax=float(0.03 if regime == "translucent_pile" else rng.uniform(0.5, 1.0)),
        pose=int(rng.integers(0, 24)), radius=float(rng.uniform(30.0, 90.0)), altitude=float(rng.uniform(10.0, 80.0)),
        bg=[float(x) for x in rng.uniform(0, 1, 3)], scale_modifier=float(rng.choice([1.0, 1.0, 0.5, 1.7])),
        backward=bool(rng.random() < 0.6), train_frame=bool(rng.random() < 0.5),
        piece=int(rng.choice([64, 96, 128, 192, 256])), lazy=0 if in_blend else int(rng.random() < 0.8),
        sort_in_blend=in_blend, split_preprocess=int(rng.random() < 0.15),
        deterministic=int(rng.random() < 0.15))


def with_round5_options(c):
    """Options added after the generator's draw order was fixed (the case sequence of a seed must not move): derived from
    the case's own seed.  wave_units: the backward blend as one wave per (work item, quadrant) -- a quarter of the cases."""
    c.setdefault("wave_units", int(c["seed"] % 4 == 0))
    # round 6: the survivors renumbered by tile band in front of the tile-table kernels (option band_sort_min = 0) -- a
    # third of the cases; the others keep the default (frames of this size are not band-sorted)
    c.setdefault("band_sort", int(c["seed"] % 3 == 0))
    return c


def run_case(c, O, G, scenes, N, dev):
    """Returns a list of failure strings (empty = the case agrees with the oracle)."""
    rs = scenes.camera(c["W"], c["H"], pose_index=c["pose"], radius=c["radius"], altitude=c["altitude"])
    rs = rs._replace(sh_degree=c["deg"], bg=torch.tensor(c["bg"], dtype=torch.float32), scale_modifier=c["scale_modifier"])
    sc = scenes.blob_scene(c["P"], c["seed"], c["deg"], spread=c["spread"], smin=c["smin"], smax=c["smax"],
                           omin=c["omin"], omax=c["omax"])
    kw = scenes.settings_kwargs(rs)
    extra = dict(shs=sc["shs"]) if c["use_sh"] else dict(colors_precomp=sc["colors_precomp"])
    fr = O.Frame(**kw, means3D=sc["means3D"], opacities=sc["opacities"], scales=sc["scales"], rotations=sc["rotations"],
                 **extra)
    fails = []
    with_round5_options(c)
    opts = dict(bwd_piece=c["piece"], lazy_sort=c["lazy"], sort_in_blend=c["sort_in_blend"],
                split_preprocess=c["split_preprocess"], deterministic_backward=c["deterministic"],
                bwd_wave_units=c["wave_units"])
    if c["band_sort"]:
        opts["band_sort_min"] = 0
    prev = {k: N.set_option(k, v) for k, v in opts.items()}
    try:
        args, out = G.run_forward(rs, sc, dev, use_sh=c["use_sh"], for_backward=c["train_frame"])
        if c["P"] == 0:   # the binding's short-circuit (dgr/rasterize_points.cu:71): no state, R = 0, an image of ZEROS
            if out[0] != 0 or tuple(out[1].shape) != (3, c["H"], c["W"]) or float(out[1].abs().max()) != 0.0:
                fails.append("empty frame: R or image")
            return fails
        d = G.decode(c["P"], c["W"], c["H"], out)
        if d["R"] != fr.R:
            fails.append("R %d != %d" % (d["R"], fr.R))
        if not np.array_equal(d["radii"], fr.radii):
            fails.append("radii")
        if not fails and fr.R > 0:
            try:
                G.assert_point_list(d, fr)
            except AssertionError as e:
                fails.append("point_list: " + str(e)[:120])
        if not np.array_equal(d["n_contrib"], fr.n_contrib):
            fails.append("n_contrib (%d pixels)" % int((d["n_contrib"] != fr.n_contrib).sum()))
        if not np.array_equal(d["final_T"].view(np.uint32), fr.final_T.view(np.uint32)):
            fails.append("final_T")
        if not np.array_equal(d["out_color"].view(np.uint32), fr.out_color.view(np.uint32)):
            fails.append("image (max diff %g)" % float(np.abs(d["out_color"] - fr.out_color).max()))
        if c["backward"] and not fails:
            dpix = np.random.default_rng(c["seed"] + 5).normal(size=(3, c["H"], c["W"])).astype(np.float32)
            gref, ggpu = fr.backward(dpix), G.run_backward(args, out, dpix, dev)
            for n in GRADS:
                if n == "dL_dsh" and not c["use_sh"]:
                    continue
                if n == "dL_dcolor" and c["use_sh"]:
                    continue
                ref, got = gref[n], ggpu[n].reshape(gref[n].shape)
                tol = gradient_tolerance(c) * max(1.0, float(np.abs(ref).max()))
                err = float(np.abs(ref - got).max()) if ref.size else 0.0
                if not (err <= tol) or not np.isfinite(got).all():
                    fails.append("%s err %.3g > %.3g" % (n, err, tol))
    finally:
        for k, v in prev.items():
            N.set_option(k, v)
    return fails


def draw_wrapper_case(rng):
    """One random call of GaussianRasterizerWrapper on an [N,14] tensor: size, pose, flips, crop window, row stride."""
    W = int(rng.choice([33, 64, 100, 160, 251, 320]))
    H = int(rng.choice([17, 48, 75, 128, 180]))
    crop = None
    if rng.random() < 0.6:
        w, h = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
        crop = [int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h]
    return dict(W=W, H=H, N=int(rng.integers(1, 6000)), seed=int(rng.integers(1, 1 << 30)), pose=int(rng.integers(0, 24)),
                flip_lr=bool(rng.random() < 0.5), flip_ud=bool(rng.random() < 0.5), crop=crop,
                stride=int(rng.choice([14, 14, 16, 20, 31])), col0=int(rng.integers(0, 3)),
                spread=float(rng.uniform(8.0, 50.0)), smax=float(rng.uniform(1.5, 6.0)),
                backward=bool(rng.random() < 0.7), host_camera=[None, True, False][int(rng.integers(0, 3))])


def run_wrapper_case(c, dev):
    """The wrapper's single autograd node (strided [N,14] input in place, mirrored / windowed store, packed gradient)
    against the reference's route on the same GPU (five slices -> GaussianRasterizer -> torch.flip -> slice crop): the
    image must be bit-equal, the [N,14] gradient equal within the atomics' tolerance."""
    import scenes
    from gaussiancity_amd import synth
    from gaussiancity_amd.rasterizer import GaussianRasterizer, GaussianRasterizerWrapper
    sc = scenes.blob_scene(c["N"], c["seed"], 0, spread=c["spread"], smin=0.4, smax=c["smax"])
    pts = np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], sc["rotations"], sc["colors_precomp"]], axis=1)
    stride = max(c["stride"], c["col0"] + 14)
    wide = torch.zeros((c["N"], stride), dtype=torch.float32, device=dev)
    wide[:, c["col0"]:c["col0"] + 14] = torch.from_numpy(pts.astype(np.float32)).to(dev)
    wr = GaussianRasterizerWrapper(synth.intrinsics(c["W"], c["H"]), (c["W"], c["H"]), flip_lr=c["flip_lr"],
                                   flip_ud=c["flip_ud"], device=dev, host_camera=c["host_camera"])
    pos, quat = synth.orbit_poses(24, 60.0, 50.0)[c["pose"]]
    oh, ow = (c["crop"][3], c["crop"][2]) if c["crop"] else (c["H"], c["W"])
    dpix = torch.from_numpy(np.random.default_rng(c["seed"] + 1).normal(size=(3, oh, ow)).astype(np.float32)).to(dev)

    class Foreign(GaussianRasterizer):  # not `type(...) is GaussianRasterizer`: takes the reference's route
        pass

    res = {}
    for name in ("node", "generic"):
        leaf = wide.clone().requires_grad_(c["backward"])
        points = leaf[:, c["col0"]:c["col0"] + 14]
        rz = wr.get_gaussian_rasterizer(pos, quat)
        if name == "generic":
            img = wr(points, gaussian_rasterizer=Foreign(rz.raster_settings))
            if c["crop"]:
                x, y, w, h = c["crop"]
                img = img[:, y:y + h, x:x + w]
        else:
            img = wr(points, gaussian_rasterizer=rz, crop=tuple(c["crop"]) if c["crop"] else None)
        if c["backward"]:
            (img * dpix).sum().backward()
        res[name] = (img.detach().cpu().numpy(), leaf.grad.cpu().numpy() if c["backward"] else None)
    fails = []
    a, b = res["node"][0], res["generic"][0]
    if a.shape != b.shape or not np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)):
        fails.append("image differs (shapes %s %s)" % (a.shape, b.shape))
    if c["backward"] and not fails:
        gn, gg = res["node"][1], res["generic"][1]
        err = float(np.abs(gn - gg).max())
        if not err <= GRAD_TOL_DENSE * max(1.0, float(np.abs(gg).max())):
            fails.append("gradient err %.3g (max %.3g)" % (err, float(np.abs(gg).max())))
        outside = np.ones(gn.shape[1], bool)
        outside[c["col0"]:c["col0"] + 14] = False
        if np.abs(gn[:, outside]).max(initial=0.0) != 0.0:
            fails.append("gradient written outside the 14 columns")
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-seconds", type=float, default=240.0)
    ap.add_argument("--wrapper-cases", type=int, default=150)
    ap.add_argument("--forward-only", action="store_true", help="skip the gradient comparison (same cases, same order)")
    ap.add_argument("--static-scene", action="store_true",
                    help="every case as an inference frame through the fused K1 with a static scene's cull cache "
                         "(gaussiancity_amd/cull_cache.py), built per case; forward only.  GCR_K1_BLOCKS=12 in the "
                         "environment makes every block stream several groups and run mid-chunk passes")
    a = ap.parse_args()
    import gpu_util as G
    import scenes
    from gaussiancity_amd import _native as N
    from oracle import oracle as O
    O.build()
    dev = torch.device("cuda:0")
    from gaussiancity_amd import cull_cache
    if a.static_scene:
        cull_cache.enable(True)
        a.wrapper_cases = 0
    rng = np.random.default_rng(a.seed)
    t0, bad, done, by_regime = time.time(), 0, 0, {}
    for i in range(a.cases):
        if time.time() - t0 > a.max_seconds:
            break
        c = draw_case(rng)
        if a.forward_only:
            c["backward"] = False
        if a.static_scene:
            c["backward"], c["train_frame"], c["split_preprocess"] = False, False, 0
            cull_cache.invalidate()  # (a new scene per case: keep the cache from stepping aside)
        try:
            fails = run_case(c, O, G, scenes, N, dev)
        except Exception as e:  # a crash is a failure of the case, not of the sweep
            fails = ["exception %s: %s" % (type(e).__name__, str(e)[:200])]
        done += 1
        by_regime[c["regime"]] = by_regime.get(c["regime"], 0) + 1
        if fails:
            bad += 1
            print(json.dumps({"case": i, "fails": fails, "desc": c}), flush=True)
    summary = {"cases": done, "failed": bad, "seconds": round(time.time() - t0, 1), "seed": a.seed, "by_regime": by_regime}
    if a.static_scene:
        summary.update(static_scene=True, cull_cache=dict(cull_cache.stats), k1_blocks_env=os.environ.get("GCR_K1_BLOCKS"))
    print(json.dumps(summary), flush=True)
    # second sweep: the wrapper's own node (strides, flips, windows, camera modes) against the reference's route
    t1, wbad, wdone = time.time(), 0, 0
    for i in range(a.wrapper_cases):
        if time.time() - t1 > a.max_seconds:
            break
        c = draw_wrapper_case(rng)
        try:
            fails = run_wrapper_case(c, dev)
        except Exception as e:
            fails = ["exception %s: %s" % (type(e).__name__, str(e)[:200])]
        wdone += 1
        if fails:
            wbad += 1
            print(json.dumps({"wrapper_case": i, "fails": fails, "desc": c}), flush=True)
    print(json.dumps({"wrapper_cases": wdone, "failed": wbad, "seconds": round(time.time() - t1, 1)}), flush=True)
    return 1 if bad or wbad else 0


if __name__ == "__main__":
    raise SystemExit(main())
