"""A fixed slice of tools/fuzz_parity.py's randomised sweep (random sizes, counts, extents, opacities, SH degrees,
poses, options) as a -m gpu test: every case must agree with the oracle -- forward bit for bit, gradients within
1e-4 * max."""
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
    for i in range(40):
        c = F.draw_case(rng)
        regimes.add(c["regime"])
        fails = F.run_case(c, oracle_mod, G, scenes, N, cuda_device)
        if fails:
            bad.append((i, fails, c))
    assert len(regimes) >= 4
    assert not bad, bad[:3]
