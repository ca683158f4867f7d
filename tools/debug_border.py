"""Debug helper: reproduce test_precull_never_changes_a_decision[border_huggers] and report the first
stage where the HIP path and the oracle differ."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes, gpu_util as G
from oracle import oracle as O
import test_gpu_parity as TP

P, W, H = 20000, 208, 120
rng = np.random.default_rng(32)
rs = scenes.camera(W, H, pose_index=7)._replace(sh_degree=0)
sc = scenes.blob_scene(P, 40, 0, spread=120.0)
vm = rs.view_matrix.numpy().astype(np.float64)
inv = np.linalg.inv(vm)
u = rng.uniform(-1.6, 1.6, P) * rs.tanfovx
v = rng.uniform(-1.6, 1.6, P) * rs.tanfovy
edge = rng.integers(0, 4, P)
u[edge == 0] = rs.tanfovx * rng.uniform(0.98, 1.25, (edge == 0).sum())
u[edge == 1] = -rs.tanfovx * rng.uniform(0.98, 1.25, (edge == 1).sum())
v[edge == 2] = rs.tanfovy * rng.uniform(0.98, 1.25, (edge == 2).sum())
v[edge == 3] = -rs.tanfovy * rng.uniform(0.98, 1.25, (edge == 3).sum())
z = np.exp(rng.uniform(np.log(0.15), np.log(400.0), P))
z[::11] = 0.2 + rng.uniform(-1e-4, 1e-4, z[::11].shape)
pv = np.stack([u * z, v * z, z, np.ones(P)], 1)
sc["means3D"] = (pv @ inv)[:, :3].astype(np.float32)
sc["scales"] = np.exp(rng.uniform(np.log(0.01), np.log(30.0), (P, 3))).astype(np.float32)
sc["means3D"][5] = np.nan; sc["means3D"][6, 0] = np.inf; sc["scales"][8] = np.nan; sc["rotations"][9] = np.inf
fr = TP._frame(O, rs, sc, use_sh=False)
dev = torch.device("cuda", 0)
args, out = G.run_forward(rs, sc, dev, use_sh=False)
d = G.decode(P, W, H, out)
vis = fr.radii > 0
print("R", d["R"], fr.R, "vis", vis.sum())
for name in ("means2D", "conic_opacity", "depths"):
    a, b = d[name][vis], getattr(fr, name)[:P][vis]
    bad = np.nonzero(~(a.view(np.uint32) == b.view(np.uint32)).reshape(len(a), -1).all(1))[0]
    print(name, "mismatch rows", len(bad), (np.nonzero(vis)[0][bad[:5]], a[bad[:3]], b[bad[:3]]) if len(bad) else "")
print("rgb eq", np.array_equal(d["rgb"][vis].view(np.uint32), fr.colors_precomp[vis].view(np.uint32)) if hasattr(fr, "colors_precomp") else "n/a")
print("ranges eq", np.array_equal(d["ranges"], fr.ranges))
pl_eq = np.array_equal(d["point_list"], fr.point_list[:fr.R])
print("point_list eq", pl_eq)
if not pl_eq:
    bad = np.nonzero(d["point_list"] != fr.point_list[:fr.R])[0]
    print(" first diffs at", bad[:10], d["point_list"][bad[:10]], fr.point_list[bad[:10]])
    i = bad[0]
    ids = [d["point_list"][i], fr.point_list[i]]
    print(" depths", fr.depths[ids], fr.depths[ids].view(np.uint32))
print("n_contrib eq", np.array_equal(d["n_contrib"], fr.n_contrib), "final_T eq", np.array_equal(d["final_T"].view(np.uint32), fr.final_T.view(np.uint32)))
diff = np.nonzero(d["out_color"].view(np.uint32) != fr.out_color.view(np.uint32))
print("image diff pixels", len(diff[0]), "first", [x[:5] for x in diff])
nc_bad = np.nonzero(d["n_contrib"] != fr.n_contrib)[0]
print("n_contrib diff count", len(nc_bad), nc_bad[:5], d["n_contrib"][nc_bad[:5]], fr.n_contrib[nc_bad[:5]])
if len(diff[0]):
    y, x = int(diff[1][0]), int(diff[2][0])
    tile = (y // 16) * ((W + 15) // 16) + x // 16
    r0, r1 = fr.ranges[tile]
    print("pixel", x, y, "tile", tile, "range", r0, r1, "n_contrib gpu/cpu", d["n_contrib"][y * W + x], fr.n_contrib[y * W + x])
    # walk the oracle's list for this pixel in float32 like the kernels do
    T = np.float32(1.0)
    for k in range(r0, min(r1, r0 + 4000)):
        g = fr.point_list[k]
        xy = fr.means2D[g]; co = fr.conic_opacity[g]
        dx = np.float32(xy[0] - np.float32(x)); dy = np.float32(xy[1] - np.float32(y))
        power = -0.5 * (float(co[0]) * dx * dx + float(co[2]) * dy * dy) - float(co[1]) * dx * dy
        if k - r0 < 12 or abs(power) < 1e-3:
            print("  entry", k - r0, "id", g, "xy", xy, "conic", co[:3], "op", co[3], "power", power, "radius", fr.radii[g], "depth", fr.depths[g])
