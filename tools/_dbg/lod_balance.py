import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
cam = importlib.import_module("gaussian_splatting_amd.camera")
lod = importlib.import_module("gaussian_splatting_amd.lod")
t0 = time.time()
tree = lod.build_octree(synth.clustered_cloud(6_000_000, seed=6))
print("build", time.time() - t0, tree.n)
par = tree.parent.astype(np.int64)
# levels (BFS order: parent id < child id)
lvl = np.zeros(tree.n, np.int32)
root = par == 0xFFFFFFFF
# iterate: level = level[parent]+1, parents precede children in BFS so one vectorised pass per level
order_ok = np.all(par[~root] < np.nonzero(~root)[0])
print("parents precede", order_ok)
for i0 in range(0, tree.n, 1 << 20):
    pass
# compute by levels using level starts
lv = np.full(tree.n, -1, np.int32); lv[root] = 0
cur = np.nonzero(root)[0]
starts = [0]
l = 0
while True:
    nxt = np.nonzero(lv == -1)[0]
    if len(nxt) == 0: break
    cand = nxt[lv[par[nxt]] == l]
    if len(cand) == 0: break
    lv[cand] = l + 1; l += 1
print("levels", l + 1, [int((lv == k).sum()) for k in range(l + 1)])
for f in (0, 60):
    world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), 2.0 * np.pi / 120)] * f)
    cut = tree.cut_camera(world, 0.002, 45.0, 1920 / 1080)
    seen = np.zeros(tree.n, bool)
    c = cut
    while len(c):
        c = c[~seen[c]]; seen[c] = True
        c = np.unique(par[c]); c = c[c != 0xFFFFFFFF]
    print("frame", f, "cut", len(cut), "looked", int(seen.sum()), "per level", [int((seen & (lv == k)).sum()) for k in range(l + 1)])
    # subtree looked-at sizes per root level R
    for R in (4, 5, 6, 7):
        anc = np.arange(tree.n)
        # ancestor at level R for nodes deeper than R
        a = np.arange(tree.n)
        for k in range(l, R, -1):
            m = lv[a] == k
            # move nodes at level k up
            a = np.where(lv[a] > R, par[a], a)
        deep = seen & (lv >= R)
        cnt = np.bincount(a[deep], minlength=tree.n)
        roots = np.nonzero(seen & (lv == R))[0]
        sz = cnt[roots]
        print("  R", R, "roots", len(roots), "mean", sz.mean(), "max", sz.max(), "p99", np.percentile(sz, 99))
        for W in (1024, 2048, 4096):
            load = np.zeros(W, np.int64)
            np.add.at(load, np.arange(len(roots)) % W, sz)
            print("     W", W, "mean load", load.mean(), "max load", load.max())
