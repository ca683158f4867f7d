"""numpy restatement of the hash-grid encoder for the dtypes upstream dispatches besides float32 -- TEST INFRASTRUCTURE ONLY.

extensions/grid_encoder/grid_encoder_ext.cu:97-366 with scalar_t = at::Half ("gce-f16-v1") or double ("gce-f64-v1"):
inputs, the cell fractions and the interpolation weights are float32 exactly as in oracle/gce_oracle.c; embeddings, outputs,
dy_dx, grad, grad_embeddings and grad_inputs are scalar_t and every accumulator is a scalar_t.  For at::Half that is c10's
arithmetic: float * Half -> float; `Half += float` converts the float to Half FIRST, then Half + Half is one correctly rounded
binary16 addition; Half - Half and Half * Half round once.  numpy's float16 operations round the same way (computed in
binary32, where sums, differences and products of two binary16 values are exact, then rounded once).  Vectorised over the
points; the loops run over levels, corners and channels in the reference's order.

grad_embeddings is a sum of atomics upstream (order-dependent): here the addends are accumulated in float64 and rounded
once -- a tolerance target, not a bit target (tests say which).
"""
import numpy as np

_PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint64)


def level_scales(L, S, H):
    l = np.arange(L, dtype=np.float32)
    return (np.exp2(l * np.float32(S)).astype(np.float32) * np.float32(H) - np.float32(1.0)).astype(np.float32)


def _grid_index(gridtype, align_corners, hashmap_size, resolution, pos_grid, C):
    """:71-95 without the channel term; pos_grid uint32 [B, D] -> first element of the row, uint32 [B]."""
    B, D = pos_grid.shape
    stride = np.uint64(1)
    index = np.zeros(B, np.uint64)
    grown = True
    for d in range(D):
        if stride <= np.uint64(hashmap_size):
            index = (index + pos_grid[:, d].astype(np.uint64) * stride) & np.uint64(0xFFFFFFFF)
            stride = np.uint64((int(stride) * int(resolution if align_corners else resolution + 1)) & 0xFFFFFFFF)
    if gridtype == 0 and stride > np.uint64(hashmap_size):
        h = np.zeros(B, np.uint64)
        for d in range(D):
            h ^= (pos_grid[:, d].astype(np.uint64) * _PRIMES[d]) & np.uint64(0xFFFFFFFF)
        index = h
    del grown
    return ((index % np.uint64(hashmap_size)) * np.uint64(C)).astype(np.int64)


def _locate(x, scale, align_corners):
    inside = ((x >= 0) & (x <= 1)).all(axis=1)
    pos = (x * np.float32(scale) + np.float32(0.0 if align_corners else 0.5)).astype(np.float32)
    cell = np.floor(pos).astype(np.float32)
    pg = np.where(inside[:, None], cell, 0).astype(np.int64).astype(np.uint32)
    frac = (pos - pg.astype(np.float32)).astype(np.float32)
    return inside, frac, pg


def _mulw(w, g, T):
    """float weight times a scalar_t value, as the reference's expression types it."""
    if T == np.float16:
        return (w * g.astype(np.float32)).astype(np.float32).astype(np.float16)  # float product, then Half(float)
    return w.astype(np.float64) * g                                              # float * double -> double


def forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False):
    """-> outputs [L,B,C], dy_dx [B,L,D,C] or None, in embeddings.dtype (float16 or float64)."""
    x = np.ascontiguousarray(inputs, np.float32)
    T = embeddings.dtype.type
    assert T in (np.float16, np.float64)
    B, D = x.shape
    C, L = embeddings.shape[1], len(offsets) - 1
    sc = level_scales(L, S, H)
    out = np.zeros((L, B, C), T)
    dd = np.zeros((B, L, D, C), T) if calc_grad_inputs else None
    one = np.float32(1.0)
    np.seterr(over="ignore", invalid="ignore")  # (rows outside [0, 1] compute garbage that is masked below)
    for l in range(L):
        g = embeddings[offsets[l]:offsets[l + 1]].reshape(-1)
        hs = int(offsets[l + 1] - offsets[l])
        res = int(np.ceil(sc[l])) + 1
        inside, pos, pg = _locate(x, sc[l], align_corners)
        acc = np.zeros((B, C), T)
        for idx in range(1 << D):
            w = np.ones(B, np.float32)
            pl = pg.copy()
            for d in range(D):
                if idx & (1 << d):
                    w = (w * pos[:, d]).astype(np.float32)
                    pl[:, d] = pg[:, d] + np.uint32(1)
                else:
                    w = (w * (one - pos[:, d])).astype(np.float32)
            index = _grid_index(gridtype, align_corners, hs, res, pl, C)
            for ch in range(C):
                acc[:, ch] = (acc[:, ch] + _mulw(w, g[index + ch], T)).astype(T)
        out[l] = np.where(inside[:, None], acc, T(0))
        if calc_grad_inputs:
            for gd in range(D):
                rg = np.zeros((B, C), T)
                for idx in range(1 << (D - 1)):
                    w = np.full(B, sc[l], np.float32)
                    pl = pg.copy()
                    for nd in range(D - 1):
                        d = nd + 1 if nd >= gd else nd
                        if idx & (1 << nd):
                            w = (w * pos[:, d]).astype(np.float32)
                            pl[:, d] = pg[:, d] + np.uint32(1)
                        else:
                            w = (w * (one - pos[:, d])).astype(np.float32)
                    pl[:, gd] = pg[:, gd]
                    il = _grid_index(gridtype, align_corners, hs, res, pl, C)
                    pl[:, gd] = pg[:, gd] + np.uint32(1)
                    ir = _grid_index(gridtype, align_corners, hs, res, pl, C)
                    for ch in range(C):
                        diff = (g[ir + ch] - g[il + ch]).astype(T)
                        rg[:, ch] = (rg[:, ch] + _mulw(w, diff, T)).astype(T)
                dd[:, l, gd, :] = np.where(inside[:, None], rg, T(0))
    return out, dd


def backward(grad, inputs, embeddings_shape, offsets, S, H, dy_dx=None, gridtype=0, align_corners=False):
    """grad [L,B,C] scalar_t -> grad_embeddings (float64 sum of the scalar_t addends, rounded once), grad_inputs or None."""
    x = np.ascontiguousarray(inputs, np.float32)
    T = grad.dtype.type
    B, D = x.shape
    L, _, C = grad.shape
    sc = level_scales(L, S, H)
    ge = np.zeros(int(np.prod(embeddings_shape)), np.float64)
    one = np.float32(1.0)
    for l in range(L):
        hs = int(offsets[l + 1] - offsets[l])
        res = int(np.ceil(sc[l])) + 1
        inside, pos, pg = _locate(x, sc[l], align_corners)
        for idx in range(1 << D):
            w = np.ones(B, np.float32)
            pl = pg.copy()
            for d in range(D):
                if idx & (1 << d):
                    w = (w * pos[:, d]).astype(np.float32)
                    pl[:, d] = pg[:, d] + np.uint32(1)
                else:
                    w = (w * (one - pos[:, d])).astype(np.float32)
            index = _grid_index(gridtype, align_corners, hs, res, pl, C) + int(offsets[l]) * C
            for ch in range(C):
                add = _mulw(w, grad[l, :, ch], T).astype(np.float64)
                np.add.at(ge, index[inside] + ch, add[inside])
    gi = None
    if dy_dx is not None:
        r = np.zeros((B, D), T)
        for l in range(L):
            for ch in range(C):
                prod = (grad[l, :, ch][:, None] * dy_dx[:, l, :, ch]).astype(T)   # scalar_t * scalar_t, rounded once
                r = (r + prod).astype(T)
        gi = r
    return ge.reshape(embeddings_shape).astype(T), gi
