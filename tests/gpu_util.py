"""Helpers for the -m gpu parity tests: run the HIP path through the native-module surface and
decode its opaque scratch buffers (layout from gcr_get_layout) for stage-wise comparison."""
import numpy as np
import torch

from gaussiancity_amd import _native as N
from gaussiancity_amd import ext


def to_dev(a, device):
    if a is None:
        return torch.Tensor([])
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def run_forward(rs, sc, device, *, use_sh=True, use_cov3d=False, cov3D=None, for_backward=None):
    """Calls ext.rasterize_gaussians with the reference's positional signature (`for_backward`: None = the function's
    own default, i.e. the backward's state is written; False = an inference frame; True = what
    RasterizeGaussiansFunction says when an input requires a gradient)."""
    colors = torch.Tensor([]) if use_sh else to_dev(sc["colors_precomp"], device)
    sh = to_dev(sc["shs"], device) if use_sh else torch.Tensor([])
    scales = torch.Tensor([]) if use_cov3d else to_dev(sc["scales"], device)
    rots = torch.Tensor([]) if use_cov3d else to_dev(sc["rotations"], device)
    cov = to_dev(cov3D, device) if use_cov3d else torch.Tensor([])
    args = (rs.bg.to(device), to_dev(sc["means3D"], device), colors, to_dev(sc["opacities"], device),
            scales, rots, rs.scale_modifier, cov, rs.view_matrix.to(device), rs.proj_matrix.to(device),
            rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, sh, rs.sh_degree, rs.campos.to(device),
            rs.prefiltered, rs.debug)
    out = ext.rasterize_gaussians(*args, _for_backward=for_backward)
    torch.cuda.synchronize()
    return args, out


def run_backward(args, out, dL_dpix, device):
    (bg, means3D, colors, opacity, scales, rots, scale_modifier, cov, view, proj, tfx, tfy, H, W,
     sh, degree, campos, prefiltered, debug) = args
    R, color, radii, geom, binning, img = out
    g = ext.rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rots, scale_modifier,
                                         cov, view, proj, tfx, tfy, to_dev(dL_dpix, device), sh,
                                         degree, campos, geom, R, binning, img, debug)
    torch.cuda.synchronize()
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh",
             "dL_dscale", "dL_drot")
    return {n: t.cpu().numpy() for n, t in zip(names, g)}


def decode(P, W, H, out):
    """Opaque buffers -> dict of numpy arrays named like the oracle's Frame attributes."""
    R, color, radii, geom, binning, img = out
    L = N.get_layout(P, W, H, R)
    gb, bb, ib = geom.cpu().numpy(), binning.cpu().numpy(), img.cpu().numpy()
    rec = gb[L.geom_rec:L.geom_rec + P * 64].view(np.float32).reshape(P, 16)   # 64-byte records (ABI v8)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    d = dict(
        R=R, out_color=color.cpu().numpy(), radii=radii.cpu().numpy(),
        means2D=rec[:, 0:2], conic_opacity=np.concatenate([rec[:, 2:5], rec[:, 5:6]], 1),
        rgb=rec[:, 6:9], depths=rec[:, 9],
        rect=rec[:, 10:12].copy().view(np.uint32),   # (min | max << 16) in x and y, tile units
        clamped=rec[:, 12].copy().view(np.uint32).astype(np.uint8),   # the record's fourth quad carries the clamp mask
        final_T=ib[L.img_final_T:L.img_final_T + 4 * W * H].view(np.float32),
        n_contrib=ib[L.img_n_contrib:L.img_n_contrib + 4 * W * H].view(np.uint32),
        ranges=ib[L.img_ranges:L.img_ranges + 8 * T].view(np.uint32).reshape(T, 2),
    )
    if R > 0:
        s = L.bin_sorted
        d["point_list"] = bb[L.bin_vals[s]:L.bin_vals[s] + 4 * R].view(np.uint32)
        if lazy_sort_active():
            # option "lazy_sort": {n_sorted, 0, L lo, L hi} per tile -- how far each tile's list is in final order
            d["tile_sorted"] = ib[L.img_tile_lazy:L.img_tile_lazy + 16 * T].view(np.uint32).reshape(T, 4)[:, 0].copy()
    return d


def lazy_sort_active():
    """The tile sort of the default binning path writes the lazy-sort states (not the global radix path, nor the
    blend's own sort of short lists)."""
    return bool(N.get_option("lazy_sort")) and not N.get_option("force_radix") and not N.get_option("sort_in_blend")


def assert_point_list(d, fr):
    """The sorted instance list against the oracle's.  Under option "lazy_sort" a tile's list is final only as far as
    `tile_sorted` says: that prefix must be the oracle's, cover everything the tile's pixels consumed, and be the whole
    list for tiles of up to 1024 entries."""
    ref = fr.point_list[:fr.R]
    got = d["point_list"]
    ns = d.get("tile_sorted")
    if ns is None:
        np.testing.assert_array_equal(got, ref)
        return
    r0 = fr.ranges[:, 0].astype(np.int64)
    n = fr.ranges[:, 1].astype(np.int64) - r0
    ns = ns.astype(np.int64)
    assert (ns <= n).all() and (ns[n <= 1024] == n[n <= 1024]).all() and (ns[n > 1024] >= 1).all()
    gx = (fr.W + 15) // 16
    H16, W16 = ((fr.H + 15) // 16) * 16, gx * 16
    nc = np.zeros((H16, W16), np.int64)
    nc[:fr.H, :fr.W] = fr.n_contrib.reshape(fr.H, fr.W)
    consumed = nc.reshape(H16 // 16, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    assert (ns >= consumed).all(), "a tile consumed entries behind its sorted prefix"
    # one flat comparison of all the final prefixes
    pos = np.arange(fr.R, dtype=np.int64)
    tile_of = np.repeat(np.arange(len(n)), n)
    final = pos - r0[tile_of] < ns[tile_of]
    np.testing.assert_array_equal(got[final], ref[final])
