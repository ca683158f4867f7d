"""Independent float64 dense formulation of the rasterizer in PyTorch autograd.

TEST INFRASTRUCTURE ONLY (only tests/ import this).  Purpose: pin oracle/gcr_oracle.c -- the
C restatement of the CUDA sources -- against a formulation that shares no code with it:
whole-image tensor ops in float64, gradients from autograd instead of the hand-derived
backward of cr/backward.cu.  Usable on tiny scenes only (O(P*H*W) memory-light loop).

Reference semantics that plain autograd would NOT reproduce are encoded explicitly:
  * alpha = min(0.99, o*G) passes gradient even where it clamps (cr/backward.cu:526-575 never
    gates on the clamp)                          -> straight-through min;
  * the frustum clamp of t.x,t.y in computeCov2D zeroes dL/dt.x but differentiates the rest
    as if the clamped t.x were a constant (cr/backward.cu:168-171,268-273)
                                                  -> clamped value detached;
  * d(loss)/d(scale) as the reference returns it is the true derivative DIVIDED by scale_modifier:
    cr/backward.cu:342-344 takes dot(Rt[i], dL_dMt[i]) for M = (mod*S) R without the chain-rule factor
    `mod` (harmless for GaussianCity, which always passes scale_modifier = 1, dgr/__init__.py:393)
                                                  -> callers divide autograd's scale gradient by mod
                                                     (tests/test_oracle.py:_dense);
  * radius, tile rectangle, depth order, the power>0 / alpha<1/255 / T<1e-4 tests are
    discrete decisions without gradient; decisions are taken on float32-rounded values
    where the reference stores float32 (depth, pixel centre, radius).
"""
import numpy as np
import torch

D = torch.float64

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
         0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _sh_color(deg, shs, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
               + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6] + SH_C2[3] * xz * shs[:, 7]
               + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
               + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def render(*, img_h, img_w, tanfovx, tanfovy, bg, scale_modifier, view_matrix, proj_matrix,
           sh_degree, campos, means3D, means2D, opacities, shs=None, colors_precomp=None,
           scales=None, rotations=None, cov3D_precomp=None):
    """All tensor arguments are float64 torch tensors (requires_grad as desired).
    view_matrix / proj_matrix are [4,4] in the reference's row-vector convention
    (p_row @ M).  Returns (image[3,H,W], radii[P] int64)."""
    H, W = int(img_h), int(img_w)
    P = means3D.shape[0]
    ones = torch.ones((P, 1), dtype=D)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ view_matrix                      # [P,4]
    p_hom = hom @ proj_matrix
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    ndc = p_hom[:, :2] * p_w[:, None] + means2D[:, :2]
    tz = p_view[:, 2]
    in_front = tz > 0.2

    if cov3D_precomp is None:
        s = scale_modifier * scales
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        R = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
        RS = R * s[:, None, :]
        Sigma = RS @ RS.transpose(1, 2)             # R S S^T R^T
    else:
        c = cov3D_precomp
        Sigma = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1),
                             torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                             torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], 1)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(in_front, tz, torch.ones_like(tz))  # avoid 0-division on culled points
    txtz, tytz = p_view[:, 0] / tzs, p_view[:, 1] / tzs
    cx_ = (txtz < -limx) | (txtz > limx)
    cy_ = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx_, (txtz.clamp(-limx, limx) * tzs).detach(), p_view[:, 0])
    ty = torch.where(cy_, (tytz.clamp(-limy, limy) * tzs).detach(), p_view[:, 1])
    zero = torch.zeros_like(tzs)
    J = torch.stack([torch.stack([fx / tzs, zero, -fx * tx / (tzs * tzs)], -1),
                     torch.stack([zero, fy / tzs, -fy * ty / (tzs * tzs)], -1)], 1)  # [P,2,3]
    Wm = view_matrix[:3, :3].transpose(0, 1)        # column-vector world->camera rotation
    JW = J @ Wm
    cov2 = JW @ Sigma @ JW.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + 0.3
    det = a * c2 - b * b
    ok = in_front & (det != 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    conx, cony, conz = c2 / dets, -b / dets, a / dets
    mid = 0.5 * (a + c2)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], -1)
    pix32 = pix.detach().to(torch.float32).to(D)    # rect decisions on float32 pixel centres
    gx, gy = (W + 15) // 16, (H + 15) // 16
    minx = torch.clamp(torch.trunc((pix32[:, 0] - radius) / 16.0), 0, gx)
    maxx = torch.clamp(torch.trunc((pix32[:, 0] + radius + 15.0) / 16.0), 0, gx)
    miny = torch.clamp(torch.trunc((pix32[:, 1] - radius) / 16.0), 0, gy)
    maxy = torch.clamp(torch.trunc((pix32[:, 1] + radius + 15.0) / 16.0), 0, gy)
    ok = ok & ((maxx - minx) * (maxy - miny) > 0)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int64)

    if colors_precomp is None:
        dirs = means3D - campos[None, :]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        colors = _sh_color(sh_degree, shs, dirs)
    else:
        colors = colors_precomp

    # global (depth, index) order == every tile's order restricted to that tile
    depth32 = tz.detach().to(torch.float32).numpy()
    order = np.lexsort((np.arange(P), depth32))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=D), torch.arange(W, dtype=D), indexing="ij")
    tyi, txi = torch.floor(ys / 16), torch.floor(xs / 16)
    T = torch.ones((H, W), dtype=D)
    C = torch.zeros((3, H, W), dtype=D)
    done = torch.zeros((H, W), dtype=torch.bool)
    for i in order:
        i = int(i)
        if not bool(ok[i]):
            continue
        in_rect = (txi >= minx[i]) & (txi < maxx[i]) & (tyi >= miny[i]) & (tyi < maxy[i])
        if not bool(in_rect.any()):
            continue
        dx, dy = pix[i, 0] - xs, pix[i, 1] - ys
        power = -0.5 * (conx[i] * dx * dx + conz[i] * dy * dy) - cony[i] * dx * dy
        G = torch.exp(power)
        araw = opacities[i] * G
        alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()   # straight-through min
        live = in_rect & ~done & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        kill = live & (test_T < 0.0001)
        done = done | kill
        use = live & ~kill
        w = torch.where(use, alpha * T, torch.zeros_like(T))
        C = C + colors[i][:, None, None] * w[None]
        T = torch.where(use, test_T, T)
    image = C + T[None] * bg[:, None, None]
    return image, radii
