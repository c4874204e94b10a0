import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
cam = importlib.import_module("gaussian_splatting_amd.camera")
lod = importlib.import_module("gaussian_splatting_amd.lod")
tree = lod.build_octree(synth.clustered_cloud(6_000_000, seed=6))
par = tree.parent.astype(np.int64)
world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), 2.0 * np.pi / 120)] * 0)
view, proj, pos, vp = cam.camera_matrices(world, 45.0, 1920 / 1080, want_viewproj=True)
cut = tree.cut_camera(world, 0.002, 45.0, 1920 / 1080)
seen = np.zeros(tree.n, bool); c = cut
while len(c):
    c = c[~seen[c]]; seen[c] = True
    c = np.unique(par[c]); c = c[c != 0xFFFFFFFF]
ids = np.nonzero(seen)[0]
f = np.float32
m = np.asarray(vp, f).ravel()
C = tree.center3[ids].astype(f); H = tree.half3[ids].astype(f)
thr = f(0.002)
rows = {}
d = []
for k, r in enumerate((0, 1, 3)):
    m0, m1, m2, m3 = m[r], m[4 + r], m[8 + r], m[12 + r]
    base = m0 * C[:, 0] + m1 * C[:, 1] + m2 * C[:, 2] + m3
    ax, ay, az = m0 * H[:, 0], m1 * H[:, 1], m2 * H[:, 2]
    S = abs(m0) * abs(C[:, 0]) + abs(m1) * abs(C[:, 1]) + abs(m2) * abs(C[:, 2]) + abs(m3) + abs(ax) + abs(ay) + abs(az)
    d.append(S * f(2.0 ** -20))
    out = []
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                out.append(base + sx * ax + sy * ay + sz * az)
    rows[k] = np.stack(out, 1)
w = rows[2]; nx = rows[0] / w; ny = rows[1] / w
mnx, mxx, mny, mxy = nx.min(1), nx.max(1), ny.min(1), ny.max(1)
wlo = np.abs(w).min(1) - d[2]
nxm = np.maximum(abs(mnx), abs(mxx)); nym = np.maximum(abs(mny), abs(mxy))
iw = 1.0001 / wlo
ex = (nxm * d[2] + d[0]) * iw + nxm * 2.0 ** -21
ey = (nym * d[2] + d[1]) * iw + nym * 2.0 ** -21
dx, dy = mxx - mnx, mxy - mny
area = dx * dy
margin = 4 * (ex * (dy + ey) + ey * (dx + ex)) + area * 2.0 ** -21
und = ~((wlo > 1e-30) & (nxm < 1e6) & (nym < 1e6) & (abs(area - thr) > margin))
print("looked", len(ids), "undecided", int(und.sum()), und.mean(), "by wlo", int((~(wlo > 1e-30)).sum()), "by n", int((~((nxm < 1e6) & (nym < 1e6))).sum()))
print("P(wave has one)", 1 - (1 - und.mean()) ** 64)
print("median rel margin", np.median(margin / area))
