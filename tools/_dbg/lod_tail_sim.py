import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
cam = importlib.import_module("gaussian_splatting_amd.camera")
lod = importlib.import_module("gaussian_splatting_amd.lod")
tree = lod.build_octree(synth.clustered_cloud(6_000_000, seed=6))
par = tree.parent.astype(np.int64); fc = tree.first_child.astype(np.int64); nc = tree.n_children.astype(np.int64)
world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), 2.0 * np.pi / 120)] * 0)
cut = tree.cut_camera(world, 0.002, 45.0, 1920 / 1080)
incut = np.zeros(tree.n, bool); incut[cut] = True
seen = np.zeros(tree.n, bool)
c = cut
while len(c):
    c = c[~seen[c]]; seen[c] = True
    c = np.unique(par[c]); c = c[c != 0xFFFFFFFF]
# level 7 = ids [start7, start8)
lv_start = [0, 1, 9, 73, 585, 3480, 23803, 174932, 1167283]
q = np.nonzero(seen[lv_start[7]:lv_start[8]])[0] + lv_start[7]
print("queue", len(q))

def run(NT, chunks, label):
    rounds = 0; wrounds = 0; maxst = 0; hist = {}
    for cur in chunks:
        stack = np.zeros(0, np.int64); r = 0
        while True:
            rounds += 1; wrounds += (len(cur) + 63) // 64; r += 1
            ns = cur[~incut[cur]]
            if len(ns):
                ch = (fc[ns][:, None] + np.arange(8)[None, :]); ch = ch[np.arange(8)[None, :] < nc[ns][:, None]]
                stack = np.concatenate([stack, ch])
            maxst = max(maxst, len(stack))
            if len(stack) == 0: break
            k = min(NT, len(stack)); cur = stack[len(stack) - k:]; stack = stack[:len(stack) - k]
        hist[r] = hist.get(r, 0) + 1
    print(label, "NT", NT, "wg-rounds", rounds, "wave-rounds", wrounds, "max stack", maxst, "rounds hist", sorted(hist.items()))
for NT in (256, 512):
    W = (len(q) + NT - 1) // NT
    run(NT, [q[b:b + NT] for b in range(0, len(q), NT)], "consecutive")
    for gs in (1, 8, 16):
        ng = (len(q) + gs - 1) // gs
        chunks = []
        for w in range(W):
            g = w + np.arange(NT // gs) * W
            idx = (g[:, None] * gs + np.arange(gs)[None, :]).ravel()
            idx = idx[idx < len(q)]
            chunks.append(q[idx])
        run(NT, chunks, "groups of %d dealt" % gs)
