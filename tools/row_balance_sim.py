"""How many blend steps would K6 walk with other ways of dealing a tile's pixels to the rows of its four waves?

K6 gives every 16-lane row of a wave the list of the entries whose block mask reaches the row's 4x4 block; the wave loops
to the longest of its four lists.  Today the four rows of wave w are the 4x4 blocks of quadrant w.  This tool reads what
a training-type forward leaves behind (sorted lists, 48-byte records, block masks, ranges, n_contrib) and counts
wave-steps (one step = one entry for the 64 lanes of a wave) for
  static        : quadrant grouping of 4x4 blocks, the masks the kernel computed (the shipped kernel),
  sorted        : 4x4 blocks ranked by list length, ranks 4w..4w+3 to wave w,
  ideal         : sum of the row lists / 4 (no imbalance at all),
  WxH           : rows of 64 / (number of sub-blocks) lanes owning W x H pixel blocks of the same quadrant, masks recomputed
                  here with gcr_cull.h's interval test at that granularity (4x4 recomputed too, as a check of the port).
A wave stops at the last entry any of its pixels consumed (n_contrib).  Run on the GPU box: python tools/row_balance_sim.py
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/repo')
from gaussiancity_amd import _native as N, ext, synth  # noqa: E402
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper  # noqa: E402

EPS = np.float32(0.01)


def block_bits(gx_, gy_, cx, cy, cz, op, tx0, ty0, bw, bh):
    """gcr_cull.h's gcr_block_mask at block size bw x bh: bool [n][16/bh][16/bw] (float32 like the device code)."""
    f = np.float32
    n = len(gx_)
    nbx, nby = 16 // bw, 16 // bh
    with np.errstate(all='ignore'):
        pmin = np.where(op > 0, np.maximum(f(-87.0), -np.log(f(255.0) * op) - f(1.0e-3)), f(np.inf)).astype(f)
        det = cx * cz - cy * cy
        all_ = ~((det > 0) & (cx > 0) & (cz > 0))
        rx, ry = gx_ - tx0, gy_ - ty0
        ux = np.maximum(np.abs(rx), np.abs(rx - f(15)))
        uy = np.maximum(np.abs(ry), np.abs(ry - f(15)))
        tau = -pmin * f(1.001) + f(1.0e-6) * (cx * ux * ux + cz * uy * uy) + f(1.0e-6)
        t2 = f(2) * tau
        idet = f(1) / det
        ex = np.sqrt(t2 * cz * idet) * f(1.0001)
        ey = np.sqrt(t2 * cx * idet) * f(1.0001)
        all_ |= ~(ex == ex) | ~(ey == ey)
        icx = f(1) / cx
        vr = -cy * ex / cz
        k0 = t2 * cx * f(1.0002)
        out = np.zeros((n, nby, nbx), bool)
        cols = np.arange(nbx)[None, :]
        for by in range(nby):
            a = np.maximum(f(bh * by) - ry - EPS, -ey)
            b = np.minimum(f(bh * by + bh - 1) - ry + EPS, ey)
            vh = np.minimum(np.maximum(vr, a), b)
            vl = np.minimum(np.maximum(-vr, a), b)
            sh = np.sqrt(np.maximum(k0 - det * vh * vh, 0))
            sl = np.sqrt(np.maximum(k0 - det * vl * vl, 0))
            u_hi = (sh - cy * vh) * icx
            u_lo = (-sl - cy * vl) * icx
            lo = u_lo + rx - (f(bw - 1) + EPS) - f(1.0e-4) * np.abs(u_lo)
            hi = u_hi + rx + EPS + f(1.0e-4) * np.abs(u_hi)
            flo = np.minimum(np.maximum(np.ceil(lo / f(bw)), 0), nbx)
            fhi = np.maximum(np.minimum(np.floor(hi / f(bw)), nbx - 1), -1)
            flo = np.where(np.isnan(flo), 0, flo)
            fhi = np.where(np.isnan(fhi), nbx - 1, fhi)
            ok = (a <= b)
            out[:, by, :] = ok[:, None] & (cols >= flo[:, None]) & (cols <= fhi[:, None])
        out[all_] = True
        out[~(pmin < 0)] = False
    return out


dev = torch.device('cuda:0')
for cname in sys.argv[1:] or ('C3', 'C2'):
    cfg, sc = synth.make_scene(cname)
    W, H, P = cfg['W'], cfg['H'], cfg['P']
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    e = torch.Tensor([])
    pos, quat = synth.orbit_poses()[0]
    rs = wr._get_gaussian_rasterization_settings(pos, quat)._replace(sh_degree=3)
    o = ext.rasterize_gaussians(rs.bg, t['means3D'], e, t['opacities'], t['scales'], t['rotations'], 1.0, e, rs.view_matrix,
                                rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t['shs'], 3, rs.campos, False, False)
    R, _, radii, geom, binning, img = o
    torch.cuda.synchronize()
    L = N.get_layout(P, W, H, R)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    frame = geom[L.geom_num_rendered:L.geom_num_rendered + 64].view(torch.int64).cpu().numpy()
    mask_off = int(frame[7])
    assert frame[3] != 0, 'forward left no state'
    masks = binning[mask_off:mask_off + 2 * R].view(torch.int16).cpu().numpy().view(np.uint16)
    ids = binning[0:4 * R].view(torch.int32).cpu().numpy()
    rec = geom[L.geom_rec:L.geom_rec + 64 * P].view(torch.float32).view(P, 16).cpu().numpy()
    ranges = img[L.img_ranges:L.img_ranges + 8 * T].view(torch.int32).view(T, 2).cpu().numpy()
    nc = img[L.img_n_contrib:L.img_n_contrib + 4 * W * H].view(torch.int32).view(H, W).cpu().numpy()
    ncp = np.zeros((gy * 16, gx * 16), np.int64)
    ncp[:H, :W] = nc
    tile_of = np.repeat(np.arange(T), np.maximum(ranges[:, 1] - ranges[:, 0], 0))
    assert len(tile_of) == R
    q = rec[ids]
    tx0 = ((tile_of % gx) * 16).astype(np.float32)
    ty0 = ((tile_of // gx) * 16).astype(np.float32)
    res = {'workload': cname, 'R': int(R)}
    kernel_bits = ((masks[:, None] >> np.arange(16)[None, :]) & 1).astype(bool).reshape(R, 4, 4)

    def wave_steps(bits, bw, bh, regroup=False):
        """bits [R][16/bh][16/bw]; rows of a wave = the blocks of its 8x8 quadrant (or ranked, 4x4 only)."""
        nby, nbx = 16 // bh, 16 // bw
        done_blk = ncp.reshape(gy, nby, bh, gx, nbx, bw).max(axis=(2, 5))  # [ty][by][tx][bx]
        total_steps, total_rows = 0, 0.0
        for tile in range(T):
            r0, r1 = ranges[tile]
            if r1 <= r0:
                continue
            ty, tx = divmod(tile, gx)
            b = bits[r0:r1]
            done = done_blk[ty, :, tx, :]
            if regroup:
                cnt = (b.reshape(-1, 16).sum(axis=0)) * (done.reshape(16) > 0)
                order = np.argsort(-cnt, kind='stable')
                groups = [order[4 * w:4 * w + 4] for w in range(4)]
                bf, df = b.reshape(-1, 16), done.reshape(16)
                for g in groups:
                    wd = int(df[g].max())
                    if wd:
                        total_steps += int(bf[:wd][:, g].sum(axis=0).max())
                continue
            for qy in (0, 1):
                for qx in (0, 1):
                    ys = slice(qy * nby // 2, (qy + 1) * nby // 2)
                    xs = slice(qx * nbx // 2, (qx + 1) * nbx // 2)
                    wd = int(done[ys, xs].max())
                    if wd == 0:
                        continue
                    c = b[:wd, ys, xs].sum(axis=0)
                    total_steps += int(c.max())
                    total_rows += float(c.mean())
        return total_steps, total_rows

    s, ideal = wave_steps(kernel_bits, 4, 4)
    res['static_4x4_kernel_masks'] = s
    res['ideal_4x4'] = round(ideal)
    res['sorted_4x4'] = wave_steps(kernel_bits, 4, 4, regroup=True)[0]
    for bw, bh in ((4, 4), (2, 4), (4, 2), (2, 2), (8, 1), (4, 1)):
        bits = block_bits(q[:, 0], q[:, 1], q[:, 2], q[:, 3], q[:, 4], q[:, 5], tx0, ty0, bw, bh)
        s, mean_rows = wave_steps(bits, bw, bh)
        res['rows_%dx%d' % (bw, bh)] = {'wave_steps': s, 'over_static': round(s / res['static_4x4_kernel_masks'], 4),
                                        'lists_per_wave': 64 // (bw * bh), 'mean_list': round(mean_rows)}
    print(json.dumps(res), flush=True)
