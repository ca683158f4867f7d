"""Attribution of a gradient exceedance of tools/fuzz_parity.py: is the GPU-vs-oracle difference a defect, or the rounding
noise of binary32 on an ill-conditioned sum?  For one case (its description, or `seed:index` of the generator) this prints,
per gradient tensor, the largest absolute difference between
    the GPU and the binary32 oracle            (what the sweep flagged)
    the binary32 oracle and the SAME statements evaluated in binary64 (oracle.Frame64: gcr_oracle.c built with -DORC_F64)
    the GPU and that binary64 evaluation
next to max|gradient| and the case's tolerance tier -- for both backward blend kernels (option "bwd_wave_units"), with one
piece per list where the lists allow it (bwd_piece 223: no checkpoint reconstruction), and, when GCR_LIB_PATH names the
experiment build, with the step's two quotients as IEEE divisions (k7 debug flag 64) -- K7's two documented departures
from gcr-fp32-v2, one at a time.  Seconds per case (the dense float64 autograd formulation of round 3/4 took hours).

    python tools/fuzz_f64.py 48:469 46:368 43:95 > gpurun_out/r05_fuzz_f64.jsonl       (needs a GPU)
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


def case_of(spec):
    if spec.lstrip().startswith("{"):
        d = json.loads(spec)
        return d.get("desc", d), "given"
    seed, idx = (int(x) for x in spec.split(":"))
    rng = np.random.default_rng(seed)
    for _ in range(idx + 1):
        c = F.draw_case(rng)
    return c, "seed %d case %d" % (seed, idx)


def main():
    import gpu_util as G
    import scenes
    from gaussiancity_amd import _native as N
    from oracle import oracle as O
    O.build()
    dev = torch.device("cuda:0")
    exp_build = "exp" in os.path.basename(N.LIB_PATH)
    for spec in sys.argv[1:]:
        c, name = case_of(spec)
        rs = scenes.camera(c["W"], c["H"], pose_index=c["pose"], radius=c["radius"], altitude=c["altitude"])
        rs = rs._replace(sh_degree=c["deg"], bg=torch.tensor(c["bg"], dtype=torch.float32), scale_modifier=c["scale_modifier"])
        sc = scenes.blob_scene(c["P"], c["seed"], c["deg"], spread=c["spread"], smin=c["smin"], smax=c["smax"],
                               omin=c["omin"], omax=c["omax"])
        kw = scenes.settings_kwargs(rs)
        extra = dict(shs=sc["shs"]) if c["use_sh"] else dict(colors_precomp=sc["colors_precomp"])
        kw.update(means3D=sc["means3D"], opacities=sc["opacities"], scales=sc["scales"], rotations=sc["rotations"], **extra)
        f32, f64 = O.Frame(**kw), O.Frame64(**kw)
        dpix = np.random.default_rng(c["seed"] + 5).normal(size=(3, c["H"], c["W"])).astype(np.float32)
        g32, g64 = f32.backward(dpix), f64.backward(dpix)
        names = [n for n in F.GRADS if not (n == "dL_dsh" and not c["use_sh"]) and not (n == "dL_dcolor" and c["use_sh"])]
        tier = F.gradient_tolerance(c)
        head = {"case": name, "desc": c, "tier": tier, "longest_list": int((f32.ranges[:, 1] - f32.ranges[:, 0]).max()),
                "binary64_vs_binary32_oracle": {
                    "radii_differ": int((f32.radii != f64.radii).sum()), "n_contrib_differs_px": int((f32.n_contrib != f64.n_contrib).sum()),
                    "image_max_diff": float(np.abs(f32.out_color - f64.out_color).max())},
                "max_abs": {n: float(np.abs(g64[n]).max()) for n in names},
                "oracle32_vs_f64": {n: float(np.abs(g32[n] - g64[n]).max()) for n in names}}
        print(json.dumps(head), flush=True)
        variants = [("as the sweep ran it", dict(bwd_piece=c["piece"], bwd_wave_units=1), 0),
                    ("workgroup per item (round 5 default)", dict(bwd_piece=c["piece"], bwd_wave_units=0), 0),
                    ("one piece per list where <= 223 entries", dict(bwd_piece=223, bwd_wave_units=1), 0)]
        if exp_build:
            variants += [("IEEE division", dict(bwd_piece=c["piece"], bwd_wave_units=1), 64),
                         ("IEEE division + one piece per list", dict(bwd_piece=223, bwd_wave_units=1), 64),
                         ("IEEE division, workgroup per item", dict(bwd_piece=223, bwd_wave_units=0), 64)]
        for label, opts, flag in variants:
            opts = dict(opts, lazy_sort=c["lazy"], sort_in_blend=c["sort_in_blend"], split_preprocess=c["split_preprocess"],
                        deterministic_backward=0)
            prev = {k: N.set_option(k, v) for k, v in opts.items()}
            if exp_build:
                N.set_option("k7_skip_flush", flag)
            try:
                args, out = G.run_forward(rs, sc, dev, use_sh=c["use_sh"], for_backward=True)
                img_exact = bool(np.array_equal(out[1].cpu().numpy().view(np.uint32), f32.out_color.view(np.uint32)))
                gg = G.run_backward(args, out, dpix, dev)
            finally:
                for k, v in prev.items():
                    N.set_option(k, v)
                if exp_build:
                    N.set_option("k7_skip_flush", 0)
            row = {"case": name, "variant": label, "image_bit_exact": img_exact, "gpu_vs_oracle32": {}, "gpu_vs_f64": {}, "exceeds_tier": []}
            for n in names:
                got = gg[n].reshape(g32[n].shape)
                e32, e64 = float(np.abs(got - g32[n]).max()), float(np.abs(got - g64[n]).max())
                row["gpu_vs_oracle32"][n], row["gpu_vs_f64"][n] = e32, e64
                if e32 > tier * max(1.0, float(np.abs(g32[n]).max())):
                    row["exceeds_tier"].append(n)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
