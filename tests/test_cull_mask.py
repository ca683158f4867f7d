"""gcr_block_mask (gaussiancity_amd/csrc/gcr_cull.h, the 4x4-block culling of the blend kernels) is CONSERVATIVE:
brute force on the host (gcc, same header the device code includes) over random ellipses -- every block holding a
pixel the blend loop would evaluate has its bit set -- and tight (set bits ~ needed bits)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_mask_is_conservative_and_tight():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    # twice: without contraction, and with the mask function's arithmetic contracted into FMAs as the device build
    # does (`#pragma clang fp contract(fast)` inside gcr_block_mask; the pixel-exact reference side of the check
    # spells its FMAs out, so it is the contract's `power` in both builds)
    for flags in (["-ffp-contract=off"], ["-ffp-contract=fast", "-mfma"]):
        exe = os.path.join(out, "cull_mask_check")
        subprocess.check_call(["gcc", "-O2"] + flags + ["-I", os.path.join(ROOT, "gaussiancity_amd", "csrc"),
                               os.path.join(ROOT, "tests", "cull_mask_check.c"), "-lm", "-o", exe])
        r = json.loads(subprocess.check_output([exe, "1500000"]).decode())
        assert r["violations"] == 0, (flags, r)
        assert r["needed_bits"] > 1000000 and r["set_bits"] <= 1.02 * r["needed_bits"], (flags, r)
        # K6's thirty-two 2x4 blocks (and the 4x4 mask derived from them for the backward)
        r = json.loads(subprocess.check_output([exe, "1500000", "2"]).decode())
        assert r["violations"] == 0, (flags, r)
        assert r["needed_bits"] > 2000000 and r["set_bits"] <= 1.03 * r["needed_bits"], (flags, r)
