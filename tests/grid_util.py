"""Shared helpers of the hash-grid encoder tests."""
import math

import numpy as np
import torch

PRIMES = [1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737]


def make_case(rng, B, D, C, L, base=4, desired=64, log2_hashmap=10, per_level_scale_arg=2, align_corners=False,
              oob_fraction=0.1):
    """(inputs [B,D] with some rows outside [0,1], embeddings, offsets, S, H) like GridEncoder builds them."""
    from gaussiancity_amd.grid_encoder import level_offsets
    offsets = np.array(level_offsets(D, L, per_level_scale_arg, base, log2_hashmap, align_corners), np.int32)
    pls = 2 ** (math.log2(desired / base) / max(1, L - 1))
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    n_oob = int(B * oob_fraction)
    if n_oob:
        rows = rng.choice(B, n_oob, replace=False)
        x[rows, rng.integers(0, D, n_oob)] = rng.choice(np.array([-0.25, 1.5, -1e-6, 1.0000001], np.float32), n_oob)
    x[0] = 0.0
    if B > 1:
        x[1] = 1.0  # the closed ends of the interval are in range
    return x, emb, offsets, math.log2(pls), base


def torch_reference(x, emb, offsets, scales, gridtype, align_corners):
    """Independent float64 formulation with autograd: returns outputs [B, L*C] as a function of (x, emb)."""
    B, D = x.shape
    L, C = len(offsets) - 1, emb.shape[1]
    outs = []
    inside = ((x >= 0) & (x <= 1)).all(dim=1)
    for l in range(L):
        scale = float(scales[l])
        hsize = int(offsets[l + 1] - offsets[l])
        res = int(math.ceil(scale)) + 1
        pos = x * scale + (0.0 if align_corners else 0.5)
        cell = torch.floor(pos.detach())
        frac = pos - cell
        cell = cell.long()
        acc = torch.zeros(B, C, dtype=x.dtype)
        for corner in range(1 << D):
            w = torch.ones(B, dtype=x.dtype)
            pl = []
            for d in range(D):
                if corner >> d & 1:
                    w = w * frac[:, d]
                    pl.append(cell[:, d] + 1)
                else:
                    w = w * (1 - frac[:, d])
                    pl.append(cell[:, d])
            stride, index, used_all = 1, torch.zeros(B, dtype=torch.long), True
            for d in range(D):
                if stride <= hsize:
                    index = index + pl[d] * stride
                    stride *= res if align_corners else res + 1
            if gridtype == 0 and stride > hsize:
                index = torch.zeros(B, dtype=torch.long)
                for d in range(D):
                    index = index ^ ((pl[d] * PRIMES[d]) & 0xFFFFFFFF)
            index = (index & 0xFFFFFFFF) % hsize
            acc = acc + w[:, None] * emb[int(offsets[l]) + index]
        outs.append(torch.where(inside[:, None], acc, torch.zeros_like(acc)))
    return torch.stack(outs, dim=1).reshape(B, L * C)
