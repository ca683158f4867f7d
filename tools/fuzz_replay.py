"""Replay ONE case of tools/fuzz_parity.py from the description it printed (a line of its output, or the "desc" object),
optionally with some keys overridden -- against whatever library GCR_LIB_PATH names.

    python tools/fuzz_replay.py '<json>' [key=value ...]        e.g.  deterministic=0 piece=128 train_frame=1
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import fuzz_parity as F


def main():
    d = json.loads(sys.argv[1])
    c = d.get("desc", d)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        c[k] = type(c[k])(json.loads(v)) if not isinstance(c[k], bool) else bool(json.loads(v))
    import gpu_util as G
    import scenes
    from gaussiancity_amd import _native as N
    from oracle import oracle as O
    O.build()
    fails = F.run_case(c, O, G, scenes, N, torch.device("cuda:0"))
    print(json.dumps({"lib": os.environ.get("GCR_LIB_PATH", "shipped"), "overrides": sys.argv[2:], "fails": fails}), flush=True)


if __name__ == "__main__":
    main()
