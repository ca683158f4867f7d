"""Pins the recorded fuzz exceedances (tests/golden/fuzz_exceedances.json) to the NORTH-STAR bar instead of to the
oracle's own noise: for every golden case and every gradient tensor, the worst

    ratio = max|GPU - oracle32| / (tier * max(1, max|oracle32|))        (1.0 = exactly at the 1e-4 / 3e-4 / 5e-3 bar)

-- exactly, for the deterministic mode (fixed-point sums: the same bits in every run), and as median / 90th percentile / maximum
over RUNS backward passes for the two float-atomic kernels (the order of their atomics moves the difference between a
handful of values, up to 4x apart, so one run says little).  The file gets a `ratio_to_tier` record per case;
tests/test_gpu_fuzz.py then fails when the deterministic result of a tensor lands above max(1, 1.1 x its recorded ratio) -- a
regression of a recorded case by more than 10 % -- or when the best of three float-atomic runs lands above max(1, 1.1 x the
recorded 90th percentile), and still holds the older max(tier, 8 x oracle noise) bar beside both.

    python tools/fuzz_exceedance_ratios.py [RUNS]          (needs a GPU; rewrites the golden file in place)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import fuzz_parity as F

GOLDEN = os.path.join(ROOT, "tests", "golden", "fuzz_exceedances.json")


def build_case(c, oracle_mod, scenes):
    """(camera settings, scene, oracle32 gradients, dL/dpixel, tensor names, tier) of a golden case description."""
    rs = scenes.camera(c["W"], c["H"], pose_index=c["pose"], radius=c["radius"], altitude=c["altitude"])
    rs = rs._replace(sh_degree=c["deg"], bg=torch.tensor(c["bg"], dtype=torch.float32), scale_modifier=c["scale_modifier"])
    sc = scenes.blob_scene(c["P"], c["seed"], c["deg"], spread=c["spread"], smin=c["smin"], smax=c["smax"],
                           omin=c["omin"], omax=c["omax"])
    kw = scenes.settings_kwargs(rs)
    extra = dict(shs=sc["shs"]) if c["use_sh"] else dict(colors_precomp=sc["colors_precomp"])
    kw.update(means3D=sc["means3D"], opacities=sc["opacities"], scales=sc["scales"], rotations=sc["rotations"], **extra)
    dpix = np.random.default_rng(c["seed"] + 5).normal(size=(3, c["H"], c["W"])).astype(np.float32)
    names = [n for n in F.GRADS if not (n == "dL_dsh" and not c["use_sh"]) and not (n == "dL_dcolor" and c["use_sh"])]
    return rs, sc, kw, dpix, names, F.gradient_tolerance(c)


def ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, wave_units, G, ext, dev, deterministic=0):
    with ext.options(bwd_wave_units=wave_units, bwd_piece=min(c["piece"], 223), lazy_sort=c["lazy"],
                     split_preprocess=c["split_preprocess"], deterministic_backward=deterministic):
        args, out = G.run_forward(rs, sc, dev, use_sh=c["use_sh"], for_backward=True)
        assert np.array_equal(out[1].cpu().numpy().view(np.uint32), f32.out_color.view(np.uint32)), "forward is not bit-exact"
        gg = G.run_backward(args, out, dpix, dev)
    r = {}
    for n in names:
        err = float(np.abs(gg[n].reshape(g32[n].shape) - g32[n]).max())
        r[n] = err / (tier * max(1.0, float(np.abs(g32[n]).max())))
    return r


KERNELS = ((0, "workgroup_per_item"), (1, "wave_per_item_quadrant"))


def main():
    import gpu_util as G
    import scenes
    from gaussiancity_amd import ext
    from oracle import oracle as O
    O.build()
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    doc = json.load(open(GOLDEN))
    worst_all = 0.0
    for rec in doc["cases"]:
        c = rec["desc"]
        rs, sc, kw, dpix, names, tier = build_case(c, O, scenes)
        f32 = O.Frame(**kw)
        g32 = f32.backward(dpix)
        out = {"runs_per_kernel": runs}
        for wave_units, key in KERNELS:
            v = {n: [] for n in names}
            for _ in range(runs):
                r = ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, wave_units, G, ext, dev)
                for n in names:
                    v[n].append(r[n])
            out[key] = {n: {"median": round(float(np.median(x)), 4), "p90": round(float(np.percentile(x, 90)), 4),
                            "max": round(float(np.max(x)), 4)} for n, x in v.items()}
            worst_all = max(worst_all, max(float(np.max(x)) for x in v.values()))
        # the deterministic mode (fixed-point record sums, bit-identical from run to run): one exact number per tensor
        d1 = ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, 1, G, ext, dev, deterministic=1)
        d2 = ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, 1, G, ext, dev, deterministic=1)
        assert d1 == d2, "the deterministic mode gave two different results"
        out["deterministic"] = {n: round(v, 6) for n, v in d1.items()}
        worst_all = max(worst_all, max(d1.values()))
        rec["ratio_to_tier"] = out
        print(rec["case"], json.dumps(out), flush=True)
    doc["ratio_bar"] = (
        "ratio_to_tier: per case and gradient tensor, ratio = |GPU - oracle32| / (tier * max(1, max|oracle32|)) (1.0 = at the "
        "north-star bar of the case's tier; tools/fuzz_exceedance_ratios.py).  `deterministic`: the fixed-point mode, whose "
        "result is the same bits in every run -- the test holds it to max(1, 1.1 x this number): a recorded case may not get "
        "more than 10 %% worse.  `workgroup_per_item` / `wave_per_item_quadrant`: median / 90th percentile / maximum over "
        "`runs_per_kernel` backward passes of the two float-atomic kernels, whose difference to the oracle takes a handful of "
        "values depending on the order of the atomics -- the test takes the best of three runs and holds it to max(1, 1.1 x the "
        "recorded 90th percentile).  Worst ratio on record: %.2f." % worst_all)
    doc["worst_ratio_to_tier"] = round(worst_all, 3)
    json.dump(doc, open(GOLDEN, "w"), indent=1)
    print("worst ratio", worst_all)


if __name__ == "__main__":
    main()
