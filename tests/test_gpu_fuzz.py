"""A fixed slice of tools/fuzz_parity.py's randomised sweep (random sizes, counts, extents, opacities, SH degrees,
poses, options) as a -m gpu test: every case must agree with the oracle -- forward bit for bit, gradients within the
THREE tolerance tiers of tools/fuzz_parity.py, which this test applies unchanged:

    1e-4 * max(1, max|ref|)   the north-star bar: every case whose largest scale stays within 8 and that is not dense
    3e-4 * max                dense frames (regimes "dense" / "*_pile": thousands of overlapping Gaussians per tile)
    5e-3 * max                scenes with scales beyond 8 (smax > 8: near-degenerate 2D covariances, 1 / det^2 in the
                              conic gradient; fp32 itself is the limit -- the float64 evidence is in the tool's docstring)

The slice = the generator's first 40 cases (8 of them in the 1e-4 tier) plus the next cases of the 1e-4 tier until 20
such cases have run, so that the looser tiers cannot swallow the bar."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_randomised_parity_sweep(oracle_mod, cuda_device):
    import fuzz_parity as F
    import gpu_util as G
    import scenes
    from gaussiancity_amd import _native as N
    rng = np.random.default_rng(20240917)
    bad = []
    regimes = set()
    strict = 0
    for i in range(160):
        c = F.draw_case(rng)
        is_strict = F.gradient_tolerance(c) == F.GRAD_TOL
        if i >= 40 and (not is_strict or strict >= 20):
            continue
        regimes.add(c["regime"])
        strict += 1 if is_strict else 0
        fails = F.run_case(c, oracle_mod, G, scenes, N, cuda_device)
        if fails:
            bad.append((i, fails, c))
    assert len(regimes) >= 4
    assert strict >= 20, "only %d cases were held to 1e-4 * max" % strict
    assert not bad, bad[:3]


def test_deterministic_records_hold_a_large_footprint(oracle_mod, cuda_device):
    """Case 635 of `tools/fuzz_parity.py --seed 41` (round 4): scales up to 11 x scale_modifier 1.7, a footprint of
    hundreds of pixels -- max |dL_dconic| is 3.5e9, beyond the Q31.32 records the deterministic mode had, and the
    gradients came back wrong by 15 x max without a word.  The records carry a per-Gaussian binary point now
    (gcr_internal.h gcr_det_frac_bits); the same scene must pass in both modes."""
    import json

    import fuzz_parity as F
    import gpu_util as G
    import scenes
    from gaussiancity_amd import _native as N
    here = os.path.dirname(os.path.abspath(__file__))
    c = json.load(open(os.path.join(here, "golden", "fuzz_case_deterministic_range.json")))
    assert c["deterministic"] == 1 and c["backward"]
    assert F.run_case(dict(c), oracle_mod, G, scenes, N, cuda_device) == []
    assert F.run_case(dict(c, deterministic=0), oracle_mod, G, scenes, N, cuda_device) == []
    assert F.run_case(dict(c, train_frame=True, piece=128), oracle_mod, G, scenes, N, cuda_device) == []


def test_recorded_exceedances_are_rounding_noise(oracle_mod, cuda_device):
    """tests/golden/fuzz_exceedances.json: the cases of the randomised sweep whose gradients ever went beyond their tier
    (three in rounds 3-4; four in round 5's own sweep).  Two bars per gradient tensor, both backward blend
    kernels, forward bit-exact:

    * attribution (round 5): the GPU is no farther from the binary32 oracle than max(its tier, 8 x the distance of that oracle
      from the SAME statements evaluated in binary64) (oracle.Frame64, recomputed here; largest ratio observed 6.4);
    * no regression against the NORTH-STAR bar (round 6): ratio = |GPU - oracle32| / (tier * max(1,
      max|oracle32|)), recorded per case and tensor as `ratio_to_tier` (tools/fuzz_exceedance_ratios.py).  The
      deterministic mode gives the same bits in every run: its ratio may not exceed max(1, 1.1 x the recorded one) -- a
      recorded case may not get more than 10 % worse.  The two float-atomic kernels' difference to the oracle jumps between
      a handful of values with the order of their atomics (up to 4x apart): the best of three runs may not exceed max(1,
      1.1 x the recorded 90th percentile of 200 runs)."""
    import json

    import fuzz_exceedance_ratios as R
    import fuzz_parity as F
    import gpu_util as G
    import scenes
    from gaussiancity_amd import ext
    doc = json.load(open(R.GOLDEN))
    assert len(doc["cases"]) >= 7
    for rec in doc["cases"]:
        c = rec["desc"]
        rs, sc, kw, dpix, names, tier = R.build_case(c, oracle_mod, scenes)
        f32, f64 = oracle_mod.Frame(**kw), oracle_mod.Frame64(**kw)
        g32, g64 = f32.backward(dpix), f64.backward(dpix)
        assert tier == rec["tier"]
        scale = {n: tier * max(1.0, float(np.abs(g32[n]).max())) for n in names}
        noise = {n: float(np.abs(g32[n] - g64[n]).max()) for n in names}
        det = R.ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, 1, G, ext, cuda_device, deterministic=1)
        for n in names:
            assert det[n] * scale[n] <= max(scale[n], 8.0 * noise[n]), (rec["case"], "deterministic", n, det[n])
            assert det[n] <= max(1.0, 1.1 * rec["ratio_to_tier"]["deterministic"][n]), (
                rec["case"], "deterministic", n, det[n], rec["ratio_to_tier"]["deterministic"][n])
        for wave_units, key in R.KERNELS:
            runs = [R.ratios_of_one_run(c, rs, sc, dpix, names, tier, g32, f32, wave_units, G, ext, cuda_device) for _ in range(3)]
            for n in names:
                best = min(r[n] for r in runs)
                assert best * scale[n] <= max(scale[n], 8.0 * noise[n]), (rec["case"], key, n, best, noise[n] / scale[n])
                assert best <= max(1.0, 1.1 * rec["ratio_to_tier"][key][n]["p90"]), (rec["case"], key, n, best, rec["ratio_to_tier"][key][n])
