"""Pins the recorded fuzz exceedances (tests/golden/fuzz_exceedances.json) to the NORTH-STAR bar instead of to the
oracle's own noise: for every golden case and every gradient tensor, the worst

    ratio = max|GPU - oracle32| / (tier * max(1, max|oracle32|))        (1.0 = exactly at the 1e-4 / 3e-4 / 5e-3 bar)

over RUNS backward passes of each backward blend kernel (the float atomics' order moves the difference by 2-4x from run to
run, so one run says little).  The file gets a `ratio_to_tier` record per case; tests/test_gpu_fuzz.py then fails when a
tensor lands above max(1, 1.1 x its recorded worst ratio) -- a regression of a recorded case by more than 10 % -- and still
holds the older max(tier, 8 x oracle noise) bar beside it.

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


def ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, wave_units, G, ext, dev):
    with ext.options(bwd_wave_units=wave_units, bwd_piece=min(c["piece"], 223), lazy_sort=c["lazy"],
                     split_preprocess=c["split_preprocess"]):
        args, out = G.run_forward(rs, sc, dev, use_sh=c["use_sh"], for_backward=True)
        assert np.array_equal(out[1].cpu().numpy().view(np.uint32), f32.out_color.view(np.uint32)), "forward is not bit-exact"
        gg = G.run_backward(args, out, dpix, dev)
    r = {}
    for n in names:
        err = float(np.abs(gg[n].reshape(g32[n].shape) - g32[n]).max())
        r[n] = err / (tier * max(1.0, float(np.abs(g32[n]).max())))
    return r


def main():
    import gpu_util as G
    import scenes
    from gaussiancity_amd import ext
    from oracle import oracle as O
    O.build()
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    dev = torch.device("cuda:0")
    doc = json.load(open(GOLDEN))
    worst_all = 0.0
    for rec in doc["cases"]:
        c = rec["desc"]
        rs, sc, kw, dpix, names, tier = build_case(c, O, scenes)
        f32 = O.Frame(**kw)
        g32 = f32.backward(dpix)
        out = {"runs_per_kernel": runs}
        for wave_units, key in ((0, "workgroup_per_item"), (1, "wave_per_item_quadrant")):
            worst = {n: 0.0 for n in names}
            for _ in range(runs):
                r = ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, wave_units, G, ext, dev)
                for n in names:
                    worst[n] = max(worst[n], r[n])
            out[key] = {n: round(v, 4) for n, v in worst.items()}
            worst_all = max(worst_all, max(worst.values()))
        rec["ratio_to_tier"] = out
        print(rec["case"], json.dumps(out), flush=True)
    doc["ratio_bar"] = ("ratio_to_tier: per case, backward blend kernel and gradient tensor the worst |GPU - oracle32| / (tier * "
                        "max(1, max|oracle32|)) over `runs_per_kernel` backward passes (tools/fuzz_exceedance_ratios.py; 1.0 = at the "
                        "north-star bar of the case's tier).  test_recorded_exceedances_are_rounding_noise fails when a tensor "
                        "lands above max(1, 1.1 x its recorded worst): a recorded case may not get more than 10 %% worse.  Worst "
                        "ratio on record: %.2f." % worst_all)
    doc["worst_ratio_to_tier"] = round(worst_all, 3)
    json.dump(doc, open(GOLDEN, "w"), indent=1)
    print("worst ratio", worst_all)


if __name__ == "__main__":
    main()
