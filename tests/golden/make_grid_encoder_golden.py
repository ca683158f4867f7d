#!/usr/bin/env python
"""Generates tests/golden/grid_encoder_golden.json by importing the REFERENCE's Python half
(/root/reference/extensions/grid_encoder/__init__.py) with a recording stand-in for its native module.
Build container only.  The fixture holds data: constructor results (offsets, per-level scale, sizes) and
the positional arguments / tensor shapes the reference hands to grid_encoder_ext.forward / .backward.
"""
import importlib.util
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/extensions/grid_encoder/__init__.py"

calls = []


def _desc(a):
    if isinstance(a, torch.Tensor):
        return {"tensor": list(a.shape), "dtype": str(a.dtype).replace("torch.", "")}
    if isinstance(a, bool):
        return {"bool": a}
    if isinstance(a, int):
        return {"int": a}
    if isinstance(a, float):
        return {"float": a}
    return {"other": repr(a)}


fake = types.ModuleType("grid_encoder_ext")


def _fwd(*args):
    calls.append(("forward", [_desc(a) for a in args]))
    args[3].fill_(0.25)                      # outputs [L,B,C]
    if args[10]:
        args[11].fill_(0.5)                  # dy_dx


def _bwd(*args):
    calls.append(("backward", [_desc(a) for a in args]))
    args[4].fill_(2.0)                       # grad_embeddings
    if args[11]:
        args[13].fill_(3.0)                  # grad_inputs


fake.forward, fake.backward = _fwd, _bwd
sys.modules["grid_encoder_ext"] = fake
spec = importlib.util.spec_from_file_location("ref_grid_encoder", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

CONFIGS = [
    dict(in_channels=5, n_levels=16, lvl_channels=8, desired_resolution=2048),                       # models/generator.py:37-42
    dict(in_channels=3, n_levels=16, lvl_channels=2, desired_resolution=2048, log2_hashmap_size=19),
    dict(in_channels=2, n_levels=4, lvl_channels=4, desired_resolution=128, base_resolution=8, log2_hashmap_size=10),
    dict(in_channels=3, n_levels=5, lvl_channels=1, desired_resolution=96, per_level_scale=1.5, gridtype="tiled",
         align_corners=True, log2_hashmap_size=12),
    dict(in_channels=4, n_levels=3, lvl_channels=2, desired_resolution=64, base_resolution=4, log2_hashmap_size=14),
]

out = {"configs": []}
for cfg in CONFIGS:
    torch.manual_seed(0)
    enc = ref.GridEncoder(**cfg)
    emb = enc.embeddings.detach()
    rec = dict(cfg=cfg, offsets=[int(v) for v in enc.offsets], per_level_scale=enc.per_level_scale,
               output_dim=enc.output_dim, n_params=int(enc.n_params), gridtype_id=enc.gridtype_id,
               embeddings_shape=list(emb.shape), init_abs_max_le=1e-4,
               init_ok=bool(float(emb.abs().max()) <= 1e-4 and float(emb.abs().max()) > 0),
               state_dict_keys=sorted(enc.state_dict().keys()))
    # one forward/backward through the reference's autograd.Function with the recording native module
    calls.clear()
    x = (torch.rand(7, 3, cfg["in_channels"]) * 2 - 1).requires_grad_(True)
    y = enc(x, bound=1)
    rec["forward_output_shape"] = list(y.shape)
    y.sum().backward()
    rec["calls"] = [{"fn": n, "args": a} for n, a in calls]
    rec["grad_inputs_value"] = float(x.grad.reshape(-1)[0])       # 3.0 from the stand-in, routed to `inputs`
    rec["grad_embeddings_value"] = float(enc.embeddings.grad.reshape(-1)[0])
    # and without input gradients
    calls.clear()
    y2 = enc(x.detach(), bound=2)
    y2.sum().backward()
    rec["calls_no_input_grad"] = [{"fn": n, "args": a} for n, a in calls]
    out["configs"].append(rec)

json.dump(out, open(os.path.join(ROOT, "tests", "golden", "grid_encoder_golden.json"), "w"), indent=1)
print("wrote", len(out["configs"]), "configs")
